#!/bin/bash
# round 3, GPU call 34: final tree: whole GPU suite, smoke, default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call34
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c 1-300
