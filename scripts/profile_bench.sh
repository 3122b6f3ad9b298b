#!/bin/bash
# Round profile of bench.py on the GPU box: tune once (cache file), then kernel stats and the two PMC passes
# on already-tuned processes. Writes under gpurun_out/prof_bench/; copy the summaries into profiles/.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench
mkdir -p $OUT
export CYCLEDIFF_TUNE_CACHE=/tmp/cd_tune.txt
export PYTHONPATH=$ROOT
cd /tmp
timeout 600 python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
f=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then python $ROOT/scripts/pmc_traffic.py $f $w $OUT/conv_gemm_traffic.json; fi
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -30
