#!/bin/bash
# round 3, GPU call 32: per-sample guidance scales (cd_ddim_decode_v): wrapper tests, ensemble fixture, ensemble bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call32
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_wrappers.py tests/test_abi.py -q > $OUT/t_wrap.log 2>&1
tail -5 $OUT/t_wrap.log
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -q -k "ensemble or sd_v14" > $OUT/t_e2e.log 2>&1
tail -3 $OUT/t_e2e.log
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_ens.txt
timeout 900 python bench.py --workload c2e --steps 1 --warmup 0 > $OUT/bench_c2e_folded.json 2> $OUT/bench_c2e_folded.err
tail -1 $OUT/bench_c2e_folded.json | cut -c 1-300
tail -2 $OUT/bench_c2e_folded.err | cut -c 1-200
wc -l $OUT/tune_ens.txt
