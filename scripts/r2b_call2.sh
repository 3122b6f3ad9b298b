#!/bin/bash
# GPU call: token-major V (LDS transpose reads) + fused q|k|v projection; default launch set of 8 steps.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call2
mkdir -p $OUT
export PYTHONPATH=$ROOT
export CYCLEDIFF_SYNTHETIC_WEIGHTS=1
cd $ROOT
echo "== ops tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -8 | tee $OUT/t_ops.log
echo "== model / fullsize / e2e tests"; timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -x -q 2>&1 | tail -8 | tee $OUT/t_models.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
cd /tmp
echo "== unet fwd B=32"; timeout 300 python $ROOT/scripts/bench_unet.py 32 5 2>&1 | grep "ms/forward" | tee -a $OUT/ab.log
echo "== kernel breakdown of one B=32 forward"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace32 -o t -- python $ROOT/scripts/bench_unet.py 32 3 > $OUT/trace32.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/trace32 @k_timestep_embedding 2>&1 | head -24 | tee $OUT/breakdown_b32.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
echo "== bench default (tunes the unseen shapes into tune_new.txt)"
timeout 1200 python $ROOT/bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/bench_tuning.json 2> $OUT/bench_tuning.err
tail -1 $OUT/bench_tuning.json | cut -c1-200
echo "== bench default (tuned)"
timeout 900 python $ROOT/bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
echo "== bench coalesce 4"
timeout 900 python $ROOT/bench.py --steps 8 --warmup 4 --coalesce 4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
tail -1 $OUT/bench_c4.json | cut -c1-200
timeout 900 python $ROOT/bench.py --steps 8 --warmup 4 --coalesce 4 --no-cpu-baseline > $OUT/bench_c4b.json 2> $OUT/bench_c4b.err
tail -1 $OUT/bench_c4b.json | cut -c1-200
wc -l $OUT/tune_new.txt
