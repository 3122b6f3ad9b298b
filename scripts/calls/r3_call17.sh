#!/bin/bash
# round 3, GPU call 17: kernel trace of the single-batch operating point (--coalesce 1: B' = 4 encode, 8 decode)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call17
mkdir -p $OUT
cd /tmp
export PYTHONPATH=$ROOT
CYCLEDIFF_GEMM_LOG=1 timeout 600 python $ROOT/bench.py --coalesce 1 --steps 2 --warmup 1 --no-single-batch --no-cpu-baseline > $OUT/bench_c1.json 2> $OUT/bench_c1.err
tail -1 $OUT/bench_c1.json | cut -c 1-200
grep "^  M" $OUT/bench_c1.err | sort -t= -k2 -n | awk '{print}' > $OUT/gemmlog_c1.txt
sort -k8 -n -r $OUT/gemmlog_c1.txt | head -5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --coalesce 1 --steps 2 --warmup 0 --no-single-batch --no-cpu-baseline > $OUT/stats.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/stats > $OUT/c2_coalesce1_kernel_breakdown.txt 2>&1
head -45 $OUT/c2_coalesce1_kernel_breakdown.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
