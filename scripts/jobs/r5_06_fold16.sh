# round 5, lease 6: the C2 line at 16 steps per launch set (64 images through the DPM-Encoder, 128 rows through the guided decode)
# against the default 8, same box; the tile choices for the new batch sizes go to a tune cache (split-K candidates on)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_06; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CYCLEDIFF_TUNE_SPLITK=1
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 900 python bench.py --coalesce 8 --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c8.json 2> $OUT/bench_c8.err; tail -1 $OUT/bench_c8.json | cut -c1-300
timeout 1200 python bench.py --coalesce 16 --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c16.json 2> $OUT/bench_c16.err; tail -1 $OUT/bench_c16.json | cut -c1-300
tail -3 $OUT/bench_c16.err
wc -l $OUT/tune_new.txt
# second pass with the cache warm (no tuning launches at all in the process)
timeout 900 python bench.py --coalesce 16 --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c16_b.json 2> $OUT/bench_c16_b.err; tail -1 $OUT/bench_c16_b.json | cut -c1-300
