#!/bin/bash
# GPU call: deferred-maximum attention + multi-row LayerNorm. Parity first, then A/B against the previous build
# (lib/libcyclediff_prev.so, same ABI), then the bench line and one forward's kernel breakdown.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call1
mkdir -p $OUT
export PYTHONPATH=$ROOT
export CYCLEDIFF_SYNTHETIC_WEIGHTS=1
cd $ROOT
echo "== ops tests"; timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attention or layernorm" 2>&1 | tail -5 | tee $OUT/t_ops.log
echo "== model / fullsize tests"; timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5 | tee $OUT/t_models.log
cd /tmp
for B in 32; do
  echo "== unet fwd B=$B prev"; CYCLEDIFF_LIB=$ROOT/cycle-diffusion_amd/lib/libcyclediff_prev.so timeout 300 python $ROOT/scripts/bench_unet.py $B 5 2>&1 | grep "ms/forward" | tee -a $OUT/ab.log
  echo "== unet fwd B=$B new";  timeout 300 python $ROOT/scripts/bench_unet.py $B 5 2>&1 | grep "ms/forward" | tee -a $OUT/ab.log
done
echo "== kernel breakdown of one B=32 forward (new)"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace32 -o t -- python $ROOT/scripts/bench_unet.py 32 3 > $OUT/trace32.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/trace32 @k_timestep_embedding 2>&1 | head -24 | tee $OUT/breakdown_b32_new.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== bench default"
timeout 900 python $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-300
echo "== bench coalesce 8"
timeout 900 python $ROOT/bench.py --steps 8 --warmup 8 --coalesce 8 --no-cpu-baseline > $OUT/bench_c8.json 2> $OUT/bench_c8.err
tail -1 $OUT/bench_c8.json | cut -c1-300
