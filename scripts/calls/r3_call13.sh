#!/bin/bash
# round 3, GPU call 13: split mode (CD_PREC_F32X3): range guard test, tile choices with split-K for the deep-K / few-row
# layers, C5 lines with those choices, kernel trace of one reduced chain
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call13
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py -q -x > $OUT/t_f32.log 2>&1
tail -5 $OUT/t_f32.log
cp gpurun_out/parity_report.json $OUT/parity_f32.json 2>/dev/null
# tile + split choices for the x3 shapes at B = 4 and B = 16
export CYCLEDIFF_TUNE_SPLITK=1
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_x3.txt
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-single-batch > $OUT/tune_run1.json 2> $OUT/tune_run1.err
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 4 --steps 4 --warmup 4 --no-single-batch > $OUT/tune_run4.json 2> $OUT/tune_run4.err
wc -l $OUT/tune_x3.txt
unset CYCLEDIFF_TUNE_SPLITK
CYCLEDIFF_GEMM_LOG=1 timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3.json | cut -c 1-300
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 4 --steps 8 --warmup 4 --no-single-batch > $OUT/bench_c5r_x3_c4.json 2> $OUT/bench_c5r_x3_c4.err
tail -1 $OUT/bench_c5r_x3_c4.json | cut -c 1-300
timeout 900 python bench.py --workload c5 --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-single-batch > $OUT/bench_c5_x3.json 2> $OUT/bench_c5_x3.err
tail -1 $OUT/bench_c5_x3.json | cut -c 1-300
# kernel trace of the reduced chain (1 warm-up + 1 timed + 1 instrumented step)
cd /tmp
export PYTHONPATH=$ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 1 --warmup 0 --no-single-batch > $OUT/stats.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/stats > $OUT/c5r_x3_kernel_breakdown.txt 2>&1
head -40 $OUT/c5r_x3_kernel_breakdown.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
