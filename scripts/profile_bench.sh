#!/bin/bash
# Round profile of bench.py on the GPU box: the bench line (tunes, writes the tile-table cache), then kernel
# stats of an already-tuned single-stream process. Writes under gpurun_out/prof_bench/; copy the summaries into profiles/.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench
mkdir -p $OUT
export CYCLEDIFF_TUNE_CACHE=/tmp/cd_tune.txt
export PYTHONPATH=$ROOT
cd /tmp
timeout 600 python $ROOT/bench.py --steps 8 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 1 --warmup 0 --in-flight 1 --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete
# (HBM-traffic PMC passes: scripts/profile_unet_pmc.sh - rocprofv3 counter collection crashes on the full bench)
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -30
