# round 6, lease 12: config 5 (reduced chain) by steps per launch set, one box; kernel breakdown of the default
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_12; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for c in 4 8 2 4; do
  timeout 900 python bench.py --workload c5r --coalesce $c --steps 8 --warmup 0 --no-cpu-baseline > $OUT/bench_c5r_c$c.json 2> $OUT/err_c$c.txt; echo "c5r coalesce $c: $(tail -1 $OUT/bench_c5r_c$c.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])" 2>&1 | tail -1) $(tail -1 $OUT/err_c$c.txt | cut -c1-150)"
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5stats -o c -- python $GRAFT_REPO_ROOT/bench.py --workload c5r --steps 4 --warmup 0 --no-cpu-baseline > $OUT/c5r_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/c5stats > $OUT/c5r_kernel_breakdown.txt 2>&1; head -24 $OUT/c5r_kernel_breakdown.txt
