#!/bin/bash
# HBM-side traffic of one SD-v1 U-Net forward PER GEMM SHAPE (rocprofv3 PMC, one counter per pass, as profile_unet_pmc.sh)
# at the batch sizes given as arguments; the FETCH pass also records the launch order (CYCLEDIFF_GEMM_TRACE=1).
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pmc
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp
for B in ${@:-64}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcs_$c
    CYCLEDIFF_GEMM_TRACE=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$c -o p -- python $ROOT/scripts/bench_unet.py $B 1 > $OUT/pmcs_${c}_b$B.log 2>&1
  done
  f=$(find /tmp/pmcs_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  w=$(find /tmp/pmcs_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python $ROOT/scripts/pmc_traffic.py $f $w $OUT/conv_gemm_traffic_b$B.json
    python $ROOT/scripts/pmc_traffic_by_shape.py $f $w $OUT/pmcs_FETCH_SIZE_b$B.log $OUT/conv_gemm_traffic_by_shape_b$B.json
  else tail -3 $OUT/pmcs_FETCH_SIZE_b$B.log; fi
done
