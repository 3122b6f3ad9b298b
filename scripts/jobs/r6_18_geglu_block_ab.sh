# round 6, lease 18: the GEGLU block epilogue of k_conv_gemm with its bias vectors from the LDS table and row-bounded buffer stores, against
# the previous commit's library (lib/libcyclediff_v2.so = a20d6ca), one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_18; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
V2=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_v2.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_new_$i.json 2> $OUT/bench_new_$i.err; echo "new $(tail -1 $OUT/bench_new_$i.json | cut -c1-140)"
  CYCLEDIFF_LIB=$V2 timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_v2_$i.json 2> $OUT/bench_v2_$i.err; echo "v2  $(tail -1 $OUT/bench_v2_$i.json | cut -c1-140)"
done
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_new.txt 2>&1; grep "ms/forward\|launches\|act3" $OUT/unet_b64_gemmlog_new.txt
CYCLEDIFF_LIB=$V2 CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_v2.txt 2>&1; grep "ms/forward\|launches\|act3" $OUT/unet_b64_gemmlog_v2.txt
