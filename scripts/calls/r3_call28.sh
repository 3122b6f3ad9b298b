#!/bin/bash
# round 3, GPU call 28: the SD wrapper's ensemble loops (2 trials x 2 skips x 2 decoder scales, SD-sized nets at 256 px)
# against the reference fixture
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call28
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -q -x -k "ensemble" --durations=3 > $OUT/t_ens.log 2>&1
tail -25 $OUT/t_ens.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
