# round 5, lease 7: config 5 on the current tree - reduced chain under rocprofv3 (kernel breakdown), then the full 1000 / 850 / 100
# chain with the workload's default folding (B = 16 per forward) and at one batch per launch set
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_07; mkdir -p $OUT
cd /tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c -- python $GRAFT_REPO_ROOT/bench.py --workload c5r --steps 4 --warmup 0 --no-cpu-baseline > $OUT/c5r_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py $OUT/stats > $OUT/c5r_kernel_breakdown.txt 2>&1; head -25 $OUT/c5r_kernel_breakdown.txt
find $OUT -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-200
timeout 900 python bench.py --workload c5 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -1 $OUT/bench_c5.json | cut -c1-200
