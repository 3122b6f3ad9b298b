# round 5, lease 5: main.py --fold against batch-by-batch on C2, the fp32 feed-forward row chunks, the unconditional-LDM wrapper
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_05; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_model_api.py tests/test_gpu_ldm_uncond.py tests/test_gpu_fullsize.py -q -m gpu -k "main_driver or ldm or chunks or wrapper" --durations=8 2>&1 | tail -30 | tee $OUT/pytest.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
