#!/bin/bash
# round 3, GPU call 24: split mode: U-Net input as fp16 pairs (conv_in on the matrix cores) and GroupNorm statistics from
# the split convs' epilogues: parity tests, C5 lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call24
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py tests/test_gpu_models.py tests/test_gpu_wrappers.py -q -x > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -q -x -k "c5_afhq" > $OUT/t_c5r.log 2>&1
tail -3 $OUT/t_c5r.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3.json | cut -c 1-260
timeout 900 python bench.py --workload c5 --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-single-batch > $OUT/bench_c5_x3.json 2> $OUT/bench_c5_x3.err
tail -1 $OUT/bench_c5_x3.json | cut -c 1-260
