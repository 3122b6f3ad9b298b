# round 5, lease 11: the bf16 build (the storage BASELINE.json's C2 line names) of the final tree - op tests, the C2 / C3 / folded
# end-to-end fixtures at the round-5 floors (34 dB), and its bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_11; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CYCLEDIFF_LIB=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_bf16.so
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $OUT/pytest_ops_bf16.log 2>&1; echo "bf16 ops rc=$?"; tail -2 $OUT/pytest_ops_bf16.log
timeout 1500 python -m pytest tests/test_gpu_e2e_fullsize.py -q -m gpu -k "end_to_end_vs_reference or folded or ensemble" > $OUT/pytest_e2e_bf16.log 2>&1; echo "bf16 e2e rc=$?"; tail -3 $OUT/pytest_e2e_bf16.log
cp gpurun_out/parity_report.json $OUT/parity_e2e_bf16_build.json
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')):
    print(row['name'], {k: round(v,3) for k,v in row.items() if isinstance(v,float) and 'psnr' in k})
PY
timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; tail -1 $OUT/bench_bf16.json | cut -c1-200
