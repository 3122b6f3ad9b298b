# round 4, lease 22: final tree - default line, the same under rocprofv3, PMC traffic at B' = 32 / 64, C3 / C5 / split-mode lines
bash scripts/profile_bench.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_22; mkdir -p $OUT; cp -r $GRAFT_REPO_ROOT/gpurun_out/prof_bench $OUT/
cd $GRAFT_REPO_ROOT
bash scripts/profile_unet_pmc.sh 32 64
cp -r gpurun_out/prof_pmc $OUT/
timeout 900 python bench.py --steps 8 --warmup 8 > $OUT/bench_default_full.json 2> $OUT/bench_default_full.err; tail -1 $OUT/bench_default_full.json | cut -c1-200
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-160
timeout 600 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-160
timeout 900 python bench.py --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3.json 2> $OUT/bench_c2_fp32x3.err; tail -1 $OUT/bench_c2_fp32x3.json | cut -c1-160
timeout 900 python bench.py --precision fp32x3 --coalesce 2 --steps 2 --warmup 2 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3_c2.json 2> $OUT/bench_c2_fp32x3_c2.err; tail -1 $OUT/bench_c2_fp32x3_c2.json | cut -c1-160
