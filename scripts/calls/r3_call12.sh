#!/bin/bash
# round 3, GPU call 12: the split-fp16 mode of the fp32 path (CD_PREC_F32X3): parity tests, then C5 bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call12
mkdir -p $OUT
cd $ROOT
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 900 python -m pytest tests/test_gpu_f32_path.py -q -x > $OUT/t_f32.log 2>&1
tail -15 $OUT/t_f32.log
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -q -x -k "c5_afhq" > $OUT/t_c5r.log 2>&1
tail -15 $OUT/t_c5r.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
CYCLEDIFF_GEMM_LOG=1 timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3.json | cut -c 1-400
grep -v "^\[" $OUT/bench_c5r_x3.err | tail -5
grep "^  \|conv_gemm\]" $OUT/bench_c5r_x3.err | sort -t= -k2 | tail -40
timeout 600 python bench.py --workload c5r --precision fp32 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_f32.json 2> $OUT/bench_c5r_f32.err
tail -1 $OUT/bench_c5r_f32.json | cut -c 1-400
timeout 900 python bench.py --workload c5 --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-single-batch > $OUT/bench_c5_x3.json 2> $OUT/bench_c5_x3.err
tail -1 $OUT/bench_c5_x3.json | cut -c 1-400
tail -3 $OUT/bench_c5_x3.err
