#!/bin/bash
# HBM-side traffic (rocprofv3 PMC, one counter per pass) of one SD-v1 U-Net forward at the batch sizes given as
# arguments (default 32 64 - the two forward types of the default coalesced C2 launch set: 32 images through the
# DPM-Encoder, 64 rows through the CFG decode).
# (The full bench.py process crashes inside rocprofv3's counter collection; one forward has the same kernel
# population as the launch set: 99 B'=16 + 99 B'=32 forwards.)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pmc
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp
for B in ${@:-32 64}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $ROOT/scripts/bench_unet.py $B 1 > $OUT/pmc_${c}_b$B.log 2>&1
  done
  f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then python $ROOT/scripts/pmc_traffic.py $f $w $OUT/conv_gemm_traffic_b$B.json; else tail -3 $OUT/pmc_FETCH_SIZE_b$B.log; fi
done
