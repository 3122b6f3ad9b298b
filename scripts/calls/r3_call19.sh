#!/bin/bash
# round 3, GPU call 19: the whole GPU suite on the current tree, then the kernel-stats profile of the default bench
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call19
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
bash scripts/profile_bench.sh > $OUT/profile_bench.log 2>&1
head -30 gpurun_out/prof_bench/kernel_breakdown.txt
