# round 4, lease 23: same-box A/B of the 16-bit path: the library of the commit before the channel-major / fp32-epilogue work
# against the current one in both K orders (one B' = 32 forward with the per-shape table; then the default line)
OUT=$PWD/gpurun_out/r4_23; mkdir -p $OUT
for v in prev new_k0 new_k1 prev new_k1; do
  L=cycle-diffusion_amd/lib/libcyclediff.so; K=1
  [ $v = prev ] && L=cycle-diffusion_amd/lib/libcyclediff_prev.so
  [ $v = new_k0 ] && K=0
  CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_KORDER=$K CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_$v.txt 2>&1
  echo "$v: $(grep 'B=32' $OUT/unet_b32_$v.txt) $(grep '\[conv_gemm\]' $OUT/unet_b32_$v.txt)"
done
for v in prev new_k1; do
  L=cycle-diffusion_amd/lib/libcyclediff.so
  [ $v = prev ] && L=cycle-diffusion_amd/lib/libcyclediff_prev.so
  CYCLEDIFF_LIB=$PWD/$L timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "$v: $(tail -1 $OUT/bench_$v.json | cut -c1-140)"
done
