#!/bin/bash
# round 3, GPU call 30: whole GPU suite on the final tree; C3 and C5-reduced (4 batches per launch set) bench lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call30
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -q -m gpu --durations=8 > $OUT/tests.log 2>&1
tail -16 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err
tail -1 $OUT/bench_c3.json | cut -c 1-240
timeout 600 python bench.py --workload c5r --coalesce 4 --steps 8 --warmup 4 > $OUT/bench_c5r_c4.json 2> $OUT/bench_c5r_c4.err
tail -1 $OUT/bench_c5r_c4.json | cut -c 1-240
wc -l $OUT/tune_new.txt
