# round 4, lease 21: CLIP towers after restricting the channel-major order to <= 8 x 8 taps; op tests; the remaining tests that failed
OUT=$PWD/gpurun_out/r4_21; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_clip_text.py tests/test_gpu_ops.py tests/test_gpu_wrappers.py -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log
