# round 6, lease 14: the straight-line fast epilogue of k_conv_gemm (buffer-descriptor stores, bias / embedding tables in LDS by DMA, residual one
# block ahead) against the round-6 base library (lib/libcyclediff_r6base.so = commit 3ee904b), one box: op / model tests on the new
# library, phase timing (probe build), default / single-batch lines alternating, per-shape GEMM log at B' = 64
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_14; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_r6base.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1; echo "probe rc=$?"; grep "^==" $OUT/probe/report.txt | head -20
for i in 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err; tail -1 $OUT/bench_new_$i.json | cut -c1-200
  CYCLEDIFF_LIB=$OLD timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_old_$i.json 2> $OUT/bench_old_$i.err; tail -1 $OUT/bench_old_$i.json | cut -c1-200
done
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_new.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_new.txt
CYCLEDIFF_LIB=$OLD CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_old.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_old.txt
