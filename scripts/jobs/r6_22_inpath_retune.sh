# round 6, lease 22: in-path tile choice for the 1 x 1 / linear shapes (K <= 1280, M >= 16384) of the B' = 64 / 128 forwards on the tree with the
# straight-line epilogues (the table's choices date from rounds 2-5), then the default line with the old and the new table on the same box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_22; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BK=65536
export INPATH_BATCHES=64,128 INPATH_MIN_M=16384 INPATH_MAX_K=1280 INPATH_ITERS=3
export INPATH_CANDIDATES="20,22,$((BK+22)),23,5,$((BK+5)),6,$((BK+6)),30"
timeout 2400 python scripts/inpath_tune.py $OUT/tune_inpath.txt $OUT/inpath_report.txt > $OUT/inpath.log 2>&1; tail -14 $OUT/inpath.log | cut -c1-200
head -1 $OUT/inpath_report.txt
for t in new old new old; do
  if [ $t = new ]; then export CYCLEDIFF_TUNE_DEFAULT=$OUT/tune_inpath.txt; else unset CYCLEDIFF_TUNE_DEFAULT; fi
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --no-single-batch > $OUT/bench_$t.json 2> $OUT/err.txt; echo "table $t: $(tail -1 $OUT/bench_$t.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])")"
done
