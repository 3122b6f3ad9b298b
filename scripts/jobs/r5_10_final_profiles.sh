# round 5, lease 10: final tree - default line + the same under rocprofv3 (kernel stats), PMC traffic at B' = 64 / 128 (the two forward
# types of the default launch set), per-shape GEMM logs, the driver's command line, C3 and the fp32x3 lines
bash scripts/profile_bench.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_10; mkdir -p $OUT; cp -r $GRAFT_REPO_ROOT/gpurun_out/prof_bench $OUT/
cd $GRAFT_REPO_ROOT
bash scripts/profile_unet_pmc.sh 64 128
cp -r gpurun_out/prof_pmc $OUT/
for B in 32 64 128; do CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 2 gemmlog > $OUT/unet_b${B}_gemmlog.txt 2>&1; tail -4 $OUT/unet_b${B}_gemmlog.txt | head -2; done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; tail -1 $OUT/bench_driver_cmd.json | cut -c1-200
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-160
timeout 900 python bench.py --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3.json 2> $OUT/bench_c2_fp32x3.err; tail -1 $OUT/bench_c2_fp32x3.json | cut -c1-160
timeout 900 python bench.py --precision fp32x3 --coalesce 2 --steps 2 --warmup 2 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3_c2.json 2> $OUT/bench_c2_fp32x3_c2.err; tail -1 $OUT/bench_c2_fp32x3_c2.json | cut -c1-160
