#!/bin/bash
# round 3, GPU call 29: the reference's ffhq256 -> celeba256 unconditional-LDM config at full size through the model API
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call29
mkdir -p $OUT
cd $ROOT
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_ldm_uncond.txt
timeout 900 python -m pytest tests/test_gpu_model_api.py -q -x --durations=3 > $OUT/t_api.log 2>&1
tail -25 $OUT/t_api.log
wc -l $OUT/tune_ldm_uncond.txt
