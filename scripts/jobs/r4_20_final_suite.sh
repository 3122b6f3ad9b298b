# round 4, lease 20: whole GPU suite on the final tree (channel-major K order default)
OUT=$PWD/gpurun_out/r4_20; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/parity_report_full_suite.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-250
