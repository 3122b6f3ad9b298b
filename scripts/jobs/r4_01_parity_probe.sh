# round 4, lease 1: the three still-unpinned operating points + refusals + self-launch; phase timing of the wide tiles
OUT=gpurun_out/r4_01; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py tests/test_gpu_models.py tests/test_gpu_dist_bench.py -q -x \
  -k "folded or ensemble_decode_call or inside_a_batch or guided or self_launch" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1
echo "probe rc=$?"; head -60 $OUT/probe/report.txt
