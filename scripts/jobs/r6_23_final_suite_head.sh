# round 6, lease 23: the whole GPU suite + smoke() + the driver's command on the tree with the table-copy fix (140fa8d)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_23; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_all.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; tail -1 $OUT/bench_driver_cmd.json | cut -c1-300
