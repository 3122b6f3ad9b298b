#!/bin/bash
# round 3, GPU call 9: LDS-DMA fill-rate microbenchmark; LayerNorm fold A/B on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call9
mkdir -p $OUT
cd $ROOT
timeout 120 scripts/ubench/dma_fill > $OUT/dma_fill.txt 2>&1
cat $OUT/dma_fill.txt
for rep in 1 2; do
for f in 0 1; do
  CYCLEDIFF_LN_FOLD=$f timeout 600 python scripts/bench_unet.py 32 6 > $OUT/unet_b32_lnfold_${f}_$rep.txt 2>&1
  echo "LN fold $f: $(grep ms/forward $OUT/unet_b32_lnfold_${f}_$rep.txt)"
done
done
