# round 5, lease 9: the driver's command line (--steps 20 --warmup 5) with balanced launch sets (10 + 10: B' = 40 / 80); tile choices
# for those batch sizes into a tune cache (split-K candidates on), then the same command with the cache warm
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_09; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CYCLEDIFF_TUNE_SPLITK=1
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-batch > $OUT/bench_a.json 2> $OUT/bench_a.err; tail -1 $OUT/bench_a.json | cut -c1-260
wc -l $OUT/tune_new.txt
timeout 1200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_b.json 2> $OUT/bench_b.err; tail -1 $OUT/bench_b.json | cut -c1-260
wc -l $OUT/tune_new.txt
