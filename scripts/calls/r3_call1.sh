#!/bin/bash
# round 3, GPU call 1: the lin_areg prototype on hardware, the full GPU test suite (new: C5 at 256^2, the B' = 32
# folded-batch parity, EMA weights), the bf16 build's end-to-end PSNR
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call1
mkdir -p $OUT
cd $ROOT
for shape in "131072 320 320" "131072 960 320" "131072 2560 320" "32768 640 640" "32768 5120 640"; do
  timeout 120 scripts/ubench/lin_areg $shape 20 >> $OUT/lin_areg.txt 2>&1
done
cat $OUT/lin_areg.txt
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report_fp16.json 2>/dev/null
CYCLEDIFF_LIB=$ROOT/cycle-diffusion_amd/lib/libcyclediff_bf16.so timeout 600 python -m pytest tests/test_gpu_e2e_fullsize.py -q > $OUT/tests_bf16_e2e.log 2>&1
tail -5 $OUT/tests_bf16_e2e.log
cp gpurun_out/parity_report.json $OUT/parity_report_bf16_e2e.json 2>/dev/null
