#!/bin/bash
# GPU call: round-2b profile set - bench line + rocprofv3 kernel stats of the default bench, PMC traffic of the two
# forward types of the default launch set, the default bench with its CPU baseline leg, comparison lines.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call6
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd $ROOT
echo "== new / changed op tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "saturate or layernorm or attention or probe" 2>&1 | tail -3
bash scripts/profile_bench.sh 2>&1 | tail -45
bash scripts/profile_unet_pmc.sh 32 64 2>&1 | tail -6
cd /tmp
echo "== default bench incl. CPU baseline"
timeout 1200 python $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json | cut -c1-300
echo "== comparison lines"
timeout 900 python $ROOT/bench.py --coalesce 4 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -1 $OUT/bench_c4.json | cut -c1-160
timeout 900 python $ROOT/bench.py --coalesce 1 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_c1.json 2> $OUT/bench_c1.err; tail -1 $OUT/bench_c1.json | cut -c1-160
timeout 900 python $ROOT/bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-160
