# round 6, lease 4: the 12-operation pack8 (v_med3_f32 + v_cvt_pk_f16_f32) against the f2bf form (-DCD_PACK8_F2BF build), one box:
# op tests on the new library, then default / single-batch lines alternating, then per-shape GEMM logs at B' = 64
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_04; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_pack8old.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err; tail -1 $OUT/bench_new_$i.json | cut -c1-200
  CYCLEDIFF_LIB=$OLD timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_old_$i.json 2> $OUT/bench_old_$i.err; tail -1 $OUT/bench_old_$i.json | cut -c1-200
done
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_new.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_new.txt
CYCLEDIFF_LIB=$OLD CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_old.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_old.txt
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 12 10 gemmlog > $OUT/unet_b12_gemmlog_new.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b12_gemmlog_new.txt
