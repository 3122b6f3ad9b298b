# round 4, lease 29: the tree after the per-device attribute flags and the folded ranker scoring: smoke + the tests that touch them
OUT=$PWD/gpurun_out/r4_29; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python -m pytest -q -x tests/test_gpu_clip_text.py::test_openai_clip_vit_b32_and_directional_scores tests/test_gpu_ldm_uncond.py::test_vq_first_stage_vs_oracle tests/test_gpu_wrappers.py::test_ensemble_folding_equals_member_by_member tests/test_gpu_wrappers.py::test_text_wrapper_api_vs_oracle "tests/test_gpu_ops.py" -k "not fullsize" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
