#!/bin/bash
# round 3, GPU call 16: confirm the three-launch fp32 GroupNorm is back at the call-13 timing; C5 lines for profiles/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call16
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py -q -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3.json | cut -c 1-260
timeout 900 python bench.py --workload c5 --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-single-batch > $OUT/bench_c5_x3.json 2> $OUT/bench_c5_x3.err
tail -1 $OUT/bench_c5_x3.json | cut -c 1-260
