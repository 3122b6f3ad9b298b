# round 4, lease 16: split mode with the attention / GEGLU outputs and the residual stream as fp16 pairs: parity + throughput
OUT=$PWD/gpurun_out/r4_16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -q -k "fp32_modes or reference_arithmetic" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest.log | cut -c1-220
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')): print(json.dumps(row)[:330])
PY
timeout 900 python bench.py --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3.json 2> $OUT/bench_c2_fp32x3.err; tail -1 $OUT/bench_c2_fp32x3.json | cut -c1-200
timeout 900 python bench.py --precision fp32x3 --coalesce 2 --steps 2 --warmup 2 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3_c2.json 2> $OUT/bench_c2_fp32x3_c2.err; tail -1 $OUT/bench_c2_fp32x3_c2.json | cut -c1-200
