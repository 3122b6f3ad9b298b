#!/bin/bash
# GPU call: same-box A/B of the V^T build (lib/libcyclediff_prev.so = commit de6dd96) against the token-major V build:
# attention kernel in both layouts, in-situ per-shape GEMM table of one B=32 forward for both builds.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call3
mkdir -p $OUT
export PYTHONPATH=$ROOT
export CYCLEDIFF_SYNTHETIC_WEIGHTS=1
PREV=$ROOT/cycle-diffusion_amd/lib/libcyclediff_prev.so
cd $ROOT
echo "== ops tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -5 | tee $OUT/t_ops.log
cd /tmp
echo "== attention kernel, both layouts (B=32 T=4096 H=8 d=40; then d=80 T=1024)"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/attn -o t -- python $ROOT/scripts/bench_attn.py 32 4096 8 40 3 > $OUT/attn.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/attn 2>&1 | grep -i "attention\|transpose" | tee $OUT/attn_breakdown.txt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/attn80 -o t -- python $ROOT/scripts/bench_attn.py 32 1024 8 80 3 > $OUT/attn80.log 2>&1
python $ROOT/scripts/kernel_breakdown.py $OUT/attn80 2>&1 | grep -i "attention\|transpose" | tee -a $OUT/attn_breakdown.txt
for i in 1 2; do
echo "== unet fwd B=32 prev"; CYCLEDIFF_LIB=$PREV timeout 300 python $ROOT/scripts/bench_unet.py 32 5 2>&1 | grep "ms/forward" | tee -a $OUT/ab.log
echo "== unet fwd B=32 new";  timeout 300 python $ROOT/scripts/bench_unet.py 32 5 2>&1 | grep "ms/forward" | tee -a $OUT/ab.log
done
echo "== in-situ GEMM table, prev"; CYCLEDIFF_LIB=$PREV CYCLEDIFF_GEMM_LOG=1 timeout 300 python $ROOT/scripts/bench_unet.py 32 2 gemmlog > $OUT/gemmlog_prev.txt 2>&1; tail -3 $OUT/gemmlog_prev.txt
echo "== in-situ GEMM table, new"; CYCLEDIFF_GEMM_LOG=1 timeout 300 python $ROOT/scripts/bench_unet.py 32 2 gemmlog > $OUT/gemmlog_new.txt 2>&1; tail -3 $OUT/gemmlog_new.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
