# round 4, lease 32: white_box_steps below the chain through the unconditional-LDM wrapper (bit-exact consistency checks)
OUT=$PWD/gpurun_out/r4_32; mkdir -p $OUT
timeout 300 python -m pytest -q -x tests/test_gpu_ldm_uncond.py -k "white_box or wrapper_vs_reference" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
