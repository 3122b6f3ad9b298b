# round 4, lease 11: the text U-Nets in the reference's arithmetic (fp32 / fp32x3 with SpatialTransformer blocks)
OUT=$PWD/gpurun_out/r4_11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -q -k "fp32_modes or reference_arithmetic" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log | cut -c1-220
cp gpurun_out/parity_report.json $OUT/
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')): print(json.dumps(row)[:400])
PY
