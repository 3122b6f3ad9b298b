# round 4, lease 31: white_box_steps below the chain / -1 through the text wrapper vs the oracle; the wrapper tests; the diagnostic
OUT=$PWD/gpurun_out/r4_31; mkdir -p $OUT
timeout 400 python -m pytest -q -x tests/test_gpu_wrappers.py tests/test_gpu_ops.py::test_sustained_mfma_rate_diagnostic_is_consistent > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
