# round 6, lease 5: depth-first chunks of the 64 x 64 transformer blocks (CYCLEDIFF_ST_CHUNK) and the LayerNorm-fold row threshold,
# A/B on one box; then the whole GPU suite on this tree
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for ch in 16 0 32 16 0; do
  CYCLEDIFF_ST_CHUNK=$ch timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --no-single-batch > $OUT/bench_chunk${ch}_$RANDOM.json 2> $OUT/err.txt; echo "chunk $ch: $(tail -1 $OUT/err.txt | cut -c1-100)"; ls -t $OUT/bench_chunk${ch}_*.json | head -1 | xargs tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])"
done
for mr in 65536 32768 65536 32768; do
  CYCLEDIFF_LN_FOLD_MIN_ROWS=$mr timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-bf16 --no-single-batch > $OUT/bench_lnfold${mr}_$RANDOM.json 2> $OUT/err.txt; echo "ln fold min rows $mr:"; ls -t $OUT/bench_lnfold${mr}_*.json | head -1 | xargs tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])"
done
for B in 64 128; do for ch in 16 0; do CYCLEDIFF_ST_CHUNK=$ch timeout 300 python scripts/bench_unet.py $B 3 > $OUT/unet_b${B}_chunk$ch.txt 2>&1; echo "B $B chunk $ch: $(grep ms/forward $OUT/unet_b${B}_chunk$ch.txt)"; done; done
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_all.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
