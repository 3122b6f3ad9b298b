# round 6, lease 19: the whole GPU suite on the tree with the straight-line GEMM epilogues, slowest tests listed; smoke()
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_19; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_all.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
