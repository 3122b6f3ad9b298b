# round 4, lease 12: throughput of the text U-Net in the reference's arithmetic (one batch of 4 per launch set)
OUT=$PWD/gpurun_out/r4_12; mkdir -p $OUT
for P in fp32x3 fp32; do
  timeout 900 python bench.py --precision $P --coalesce 1 --steps 1 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_$P.json 2> $OUT/bench_c2_$P.err; tail -1 $OUT/bench_c2_$P.json | cut -c1-260; tail -2 $OUT/bench_c2_$P.err | cut -c1-200
done
timeout 900 python bench.py --precision fp32x3 --coalesce 2 --steps 2 --warmup 2 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3_c2.json 2> $OUT/bench_c2_fp32x3_c2.err; tail -1 $OUT/bench_c2_fp32x3_c2.json | cut -c1-260
