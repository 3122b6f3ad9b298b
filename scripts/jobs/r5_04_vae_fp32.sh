# round 5, lease 4: first stage in fp32 / split mode (oracle parity at 512 x 512, C2 / C3 end to end through the wrappers),
# and the tightened end-to-end gates (50 dB floors, every slot of the folded batches) with the new d = 40 attention kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_04; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "vae" 2>&1 | tail -15 | tee $OUT/pytest_vae.txt
timeout 1500 python -m pytest tests/test_gpu_e2e_fullsize.py -q -m gpu -k "not c5" --durations=12 2>&1 | tail -40 | tee $OUT/pytest_e2e.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
