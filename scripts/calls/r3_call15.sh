#!/bin/bash
# round 3, GPU call 15: fp32 GroupNorm statistics exchange through sc1 stores / loads (no fences): tests + C5 lines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call15
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py tests/test_gpu_models.py -q -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
for i in 1 2; do
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3_$i.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3_$i.json | cut -c 1-260
done
timeout 600 python bench.py --workload c5r --precision fp32 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_f32.json 2> $OUT/bench_c5r_f32.err
tail -1 $OUT/bench_c5r_f32.json | cut -c 1-260
