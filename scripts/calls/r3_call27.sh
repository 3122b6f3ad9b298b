#!/bin/bash
# round 3, GPU call 27: BASELINE config 5 with the reference's full 1000 / 850 / 100 chain against the reference fixture
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call27
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_e2e_fullsize.py -q -x -k "c5_afhq" --durations=5 > $OUT/t_c5.log 2>&1
tail -12 $OUT/t_c5.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
