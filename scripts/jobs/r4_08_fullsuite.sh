# round 4, lease 8: the whole GPU suite on the round-4 kernel + merged table
OUT=$PWD/gpurun_out/r4_08; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
