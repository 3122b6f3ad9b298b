#!/bin/bash
# round 3, GPU call 22: final evidence on the tree with classifier-free-guidance prefix sharing: whole GPU suite, default
# bench line (reference CPU baseline leg included), kernel-stats profile of the same command
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call22
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c 1-300
bash scripts/profile_bench.sh > $OUT/profile_bench.log 2>&1
head -12 gpurun_out/prof_bench/kernel_breakdown.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
