# round 4, lease 24: channel-major K order as a compile-time specialisation on top of the previous commit's kernel: op + CLIP
# tests, one B' = 32 forward per variant on one box (previous library / this one tap-major / channel-major), default line
OUT=$PWD/gpurun_out/r4_24; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_clip_text.py -q > $OUT/pytest_k1.log 2>&1; echo "tests korder1 rc=$?"; tail -2 $OUT/pytest_k1.log
CYCLEDIFF_KORDER=0 timeout 600 python -m pytest tests/test_gpu_ops.py -q -k conv > $OUT/pytest_k0.log 2>&1; echo "tests korder0 rc=$?"; tail -2 $OUT/pytest_k0.log
for v in prev new_k0 new_k1 prev new_k1; do
  L=cycle-diffusion_amd/lib/libcyclediff.so; K=1
  [ $v = prev ] && L=cycle-diffusion_amd/lib/libcyclediff_prev.so
  [ $v = new_k0 ] && K=0
  CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_KORDER=$K CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_$v.txt 2>&1
  echo "$v: $(grep 'B=32' $OUT/unet_b32_$v.txt) $(grep '\[conv_gemm\]' $OUT/unet_b32_$v.txt)"
done
for v in new_k1 prev; do
  L=cycle-diffusion_amd/lib/libcyclediff.so
  [ $v = prev ] && L=cycle-diffusion_amd/lib/libcyclediff_prev.so
  CYCLEDIFF_LIB=$PWD/$L timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "$v: $(tail -1 $OUT/bench_$v.json | cut -c1-140)"
done
