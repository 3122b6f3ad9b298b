# round 4, lease 4: in-situ effect of the block epilogue and of the channel-major K order (one B' = 32 U-Net forward, three
# libraries / orders on one box), the default bench line of the new tree
OUT=gpurun_out/r4_04; mkdir -p $OUT
for v in base new new_k1; do
  L=cycle-diffusion_amd/lib/libcyclediff.so; K=0
  [ $v = base ] && L=cycle-diffusion_amd/lib/libcyclediff_r4base.so
  [ $v = new_k1 ] && K=1
  CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_KORDER=$K CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_$v.txt 2>&1
  grep "B=32\|\[conv_gemm\]" $OUT/unet_b32_$v.txt
done
timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-300
CYCLEDIFF_KORDER=1 timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_korder1.json 2> $OUT/bench_korder1.err; tail -1 $OUT/bench_korder1.json | cut -c1-200
