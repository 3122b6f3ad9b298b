#!/bin/bash
# Round profile of bench.py on the GPU box: the bench line, then rocprofv3 kernel stats of the same single-stream
# process (shipped tile table preloaded: no autotuning launches). Writes under gpurun_out/prof_bench/; the summaries
# are then copied into profiles/.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_bench
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp
timeout 900 python $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
# 16 steps = one launch set timed + one warm-up set + the event-instrumented set: 3 identical launch sets in the trace
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $ROOT/bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/stats.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/stats > $OUT/kernel_breakdown.txt 2>&1
python $ROOT/scripts/kernel_by_grid.py $OUT/stats k_gn_apply k_gn_fold k_layernorm k_attention k_copy_strided > $OUT/stream_kernels_by_grid.txt 2>&1
head -30 $OUT/kernel_breakdown.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -30
