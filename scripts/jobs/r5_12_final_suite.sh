# round 5, lease 12: the whole GPU suite on the final tree (parity report kept), then smoke()
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_12; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
