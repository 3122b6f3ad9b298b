#!/bin/bash
# round 3, GPU call 36: unconditional-LDM wrapper with precision = fp32x3 at full size (model API), its small-net tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call36
mkdir -p $OUT
cd $ROOT
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 600 python -m pytest tests/test_gpu_model_api.py tests/test_gpu_ldm_uncond.py -q --durations=3 > $OUT/t.log 2>&1
tail -12 $OUT/t.log
wc -l $OUT/tune_new.txt 2>/dev/null
