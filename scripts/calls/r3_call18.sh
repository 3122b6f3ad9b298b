#!/bin/bash
# round 3, GPU call 18: the round's evidence: default bench line (with the reference CPU baseline), rocprofv3 kernel
# stats of the same command, PMC traffic of the GEMM family over one forward at B' = 32 / 64
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call18
mkdir -p $OUT
cd $ROOT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c 1-300
bash scripts/profile_bench.sh > $OUT/profile_bench.log 2>&1
tail -40 $OUT/profile_bench.log
bash scripts/profile_unet_pmc.sh 32 64 > $OUT/profile_pmc.log 2>&1
tail -4 $OUT/profile_pmc.log
