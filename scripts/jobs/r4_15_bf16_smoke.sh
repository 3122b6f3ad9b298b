# round 4, lease 15: smoke(); the bf16 build of the final tree: op tests + the C2 / C3 / folded end-to-end fixtures (floor 32 dB)
OUT=$PWD/gpurun_out/r4_15; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log | cut -c1-300
export CYCLEDIFF_LIB=$PWD/cycle-diffusion_amd/lib/libcyclediff_bf16.so
timeout 900 python -m pytest tests/test_gpu_ops.py -q > $OUT/pytest_ops_bf16.log 2>&1; echo "bf16 ops rc=$?"; tail -2 $OUT/pytest_ops_bf16.log
timeout 1200 python -m pytest tests/test_gpu_e2e_fullsize.py -q -k "end_to_end or folded" > $OUT/pytest_e2e_bf16.log 2>&1; echo "bf16 e2e rc=$?"; tail -3 $OUT/pytest_e2e_bf16.log
cp gpurun_out/parity_report.json $OUT/parity_e2e_bf16_build.json
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')):
    print(row['name'], {k: round(v,3) for k,v in row.items() if isinstance(v,float) and 'psnr' in k})
PY
timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; tail -1 $OUT/bench_bf16.json | cut -c1-200
