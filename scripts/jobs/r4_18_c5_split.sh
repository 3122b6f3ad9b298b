# round 4, lease 18: split mode - skip projections / attention proj_out as three-term products, fp32 flavour of the block
# epilogue: parity of the fp32-path tests, C5 reduced + C2 split-mode lines
OUT=$PWD/gpurun_out/r4_18; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_f32_path.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py tests/test_gpu_ldm_uncond.py -q -k "f32 or fp32 or x3 or split or c5 or reference_arithmetic or ops or uncond or wrapper" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest.log | cut -c1-220
cp gpurun_out/parity_report.json $OUT/
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')):
    if any(k in row['name'] for k in ('c5','x3','fp32','uncond')): print(json.dumps(row)[:300])
PY
timeout 900 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-200
timeout 900 python bench.py --workload c5r --coalesce 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5r_c1.json 2> $OUT/bench_c5r_c1.err; tail -1 $OUT/bench_c5r_c1.json | cut -c1-200
timeout 900 python bench.py --precision fp32x3 --coalesce 2 --steps 2 --warmup 2 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3_c2.json 2> $OUT/bench_c2_fp32x3_c2.err; tail -1 $OUT/bench_c2_fp32x3_c2.json | cut -c1-200
