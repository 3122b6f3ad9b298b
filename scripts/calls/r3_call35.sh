#!/bin/bash
# round 3, GPU call 35: the unconditional-LDM U-Net at full size through encode / decode / refine vs the reference fixture
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call35
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ldm_uncond.py -q --durations=3 > $OUT/t.log 2>&1
tail -20 $OUT/t.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
