# round 6, lease 6: in-path tile choice for the 1 x 1 / linear shapes of the coupled single-batch forward (B' = 12: 49152 / 12288 /
# 3072 / 768 rows), then the single-batch line with the old and the new table on the same box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_06; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
BK=65536
export INPATH_BATCHES=12 INPATH_MIN_M=768 INPATH_MAX_K=5120 INPATH_ONLY_M=49152,12288,3072,768
export INPATH_CANDIDATES="1,4,5,6,10,11,14,17,18,19,20,21,22,23,24,25,$((BK+5)),$((BK+14)),$((BK+22)),$((BK+4)),$((BK+6)),30"
timeout 2400 python scripts/inpath_tune.py $OUT/tune_inpath_b12.txt $OUT/inpath_b12_report.txt > $OUT/inpath.log 2>&1; tail -25 $OUT/inpath.log | cut -c1-200
head -1 $OUT/inpath_b12_report.txt
for t in new old new old; do
  if [ $t = new ]; then export CYCLEDIFF_TUNE_DEFAULT=$OUT/tune_inpath_b12.txt; else unset CYCLEDIFF_TUNE_DEFAULT; fi
  timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-bf16 --no-single-batch > $OUT/bench_c1_$t.json 2> $OUT/err.txt; echo "table $t: $(tail -1 $OUT/bench_c1_$t.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])")"
done
