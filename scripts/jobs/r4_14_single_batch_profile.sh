# round 4, lease 14: kernel statistics of the single-batch operating point (one batch of 4 per launch set)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_14; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1stats -o b -- python $GRAFT_REPO_ROOT/bench.py --coalesce 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/c1.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/c1stats > $OUT/bench_coalesce1_kernel_breakdown.txt 2>&1
head -45 $OUT/bench_coalesce1_kernel_breakdown.txt
tail -1 $OUT/c1.log | cut -c1-200
