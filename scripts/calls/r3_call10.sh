#!/bin/bash
# round 3, GPU call 10: the ping-pong 256x320 tile (conv_pp.hip, id 31): op tests (bit-identical to tile 20), isolated
# timings on the big convolution shapes of a B' = 32 forward against tile 20 and the table's choice
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call10
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "t31 or pingpong" > $OUT/t_ops.log 2>&1
tail -6 $OUT/t_ops.log
export CYCLEDIFF_TUNE_DEFAULT=$ROOT/cycle-diffusion_amd/tune_gfx950.txt
AB=scripts/ubench/abi_bench
{
for a in "32 64 320 0 320 3 1 0 0" "32 64 320 320 320 3 1 0 0" "32 64 640 320 320 3 1 0 0" "32 32 640 0 640 3 1 0 0" "32 32 640 640 640 3 1 0 0" "32 16 1280 0 1280 3 1 0 0" "32 16 1280 1280 1280 3 1 0 0" "32 64 1280 0 320 1 1 0 0" "32 32 640 0 640 3 1 1 0" "32 32 2560 0 640 1 1 0 0" "64 64 320 0 320 3 1 0 0" "4 64 320 0 320 3 1 0 0"; do
  for tile in 31 20 0; do
    timeout 60 $AB conv $a $tile 10 | tail -1
  done
done
} > $OUT/conv_pp_isolated.txt 2>&1
cat $OUT/conv_pp_isolated.txt
