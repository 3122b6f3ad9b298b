# round 5, lease 19: images/s against steps per launch set (main.py --fold N / bench.py --coalesce N) on one box: 1, 2, 4 (8 and 16: lease 18)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_19; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for c in 1 2 4; do
  timeout 900 python bench.py --coalesce $c --steps 8 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/c$c.json 2> $OUT/c$c.err; tail -1 $OUT/c$c.json | cut -c1-170
done
