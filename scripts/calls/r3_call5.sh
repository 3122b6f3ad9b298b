#!/bin/bash
# round 3, GPU call 5: bisect the statistics-variant corruption of lin_stream on hardware
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call5
mkdir -p $OUT
cd $ROOT
for d in 0 1 2 3 4; do
  echo "== CYCLEDIFF_LIN_DBG=$d" >> $OUT/stats_diag.txt
  CYCLEDIFF_LIN_DBG=$d timeout 300 python scripts/diag/lin_stats_diag.py 2>&1 | grep "tile=30 stats=1" >> $OUT/stats_diag.txt
done
cat $OUT/stats_diag.txt
