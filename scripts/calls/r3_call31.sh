#!/bin/bash
# round 3, GPU call 31: C5 full chain with four queued batches folded into one launch set (the workload's default)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call31
mkdir -p $OUT
cd $ROOT
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 1200 python bench.py --workload c5 --steps 4 --warmup 4 > $OUT/bench_c5_c4.json 2> $OUT/bench_c5_c4.err
tail -1 $OUT/bench_c5_c4.json | cut -c 1-260
tail -2 $OUT/bench_c5_c4.err | cut -c 1-200
wc -l $OUT/tune_new.txt
