#!/bin/bash
# round 3, GPU call 25: the bf16 build of the final tree: op tests, end-to-end fixtures, default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call25
mkdir -p $OUT
cd $ROOT
export CYCLEDIFF_LIB=$ROOT/cycle-diffusion_amd/lib/libcyclediff_bf16.so
timeout 900 python -m pytest tests/test_gpu_ops.py -q > $OUT/t_ops.log 2>&1
tail -4 $OUT/t_ops.log
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -q -k "not c5_afhq" > $OUT/t_e2e.log 2>&1
tail -4 $OUT/t_e2e.log
cp gpurun_out/parity_report.json $OUT/parity_report_bf16.json 2>/dev/null
timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
tail -1 $OUT/bench_bf16.json | cut -c 1-260
