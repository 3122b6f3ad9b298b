# round 6, lease 10: the whole GPU suite on the final tree (with the B = 4 fixture), slowest tests listed
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_10; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_all.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
