# round 6, lease 15: lease 14's epilogue + table copies behind the ring's first tiles + the packed-fp32 GELU of the GEGLU epilogues, against the
# round-6 base library (lib/libcyclediff_r6base.so = commit 3ee904b), one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_15; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_r6base.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err; tail -1 $OUT/bench_new_$i.json | cut -c1-200
  CYCLEDIFF_LIB=$OLD timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_old_$i.json 2> $OUT/bench_old_$i.err; tail -1 $OUT/bench_old_$i.json | cut -c1-200
done
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_new.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_new.txt
CYCLEDIFF_LIB=$OLD CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_old.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_old.txt
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1; echo "probe rc=$?"; grep "^==" $OUT/probe/report.txt | head -20
