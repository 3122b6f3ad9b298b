#!/bin/bash
# round 3, GPU call 33: ensemble bench with per-sample guidance scales and even engine calls; wrapper tests again
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call33
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_wrappers.py -q > $OUT/t_wrap.log 2>&1
tail -3 $OUT/t_wrap.log
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_ens.txt
timeout 900 python bench.py --workload c2e --steps 1 --warmup 0 > $OUT/bench_c2e_folded.json 2> $OUT/bench_c2e_folded.err
tail -1 $OUT/bench_c2e_folded.json | cut -c 1-300
wc -l $OUT/tune_ens.txt
