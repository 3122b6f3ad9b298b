#!/bin/bash
# round 3, GPU call 11: schedule variants of the ping-pong tile (ids 31-35) against tile 20 on three big shapes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call11
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "pingpong" > $OUT/t_ops.log 2>&1
tail -4 $OUT/t_ops.log
AB=scripts/ubench/abi_bench
{
for a in "32 64 320 0 320 3 1 0 0" "32 32 640 640 640 3 1 0 0" "32 64 640 320 320 3 1 0 0"; do
  for tile in 20 31 32 33 34 35 20 32 35; do
    timeout 60 $AB conv $a $tile 10 | tail -1
  done
done
} > $OUT/conv_pp_variants.txt 2>&1
cat $OUT/conv_pp_variants.txt
