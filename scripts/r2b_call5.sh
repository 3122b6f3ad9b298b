#!/bin/bash
# GPU call: whole -m gpu suite + smoke on the fixed build; in-path trial of the three short-K tile configurations.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call5
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd $ROOT
echo "== full gpu suite"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $OUT/t_all.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
cd /tmp
export CYCLEDIFF_SYNTHETIC_WEIGHTS=1
echo "== isolated sweep, short-K shapes, B=32"
timeout 600 python $ROOT/scripts/bench_gemm.py 32 10 lin 20,22,23,65560,65561,26 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_lin_b32.txt
echo "== in-path trial B'=32,64"
INPATH_CANDIDATES=65560,65561,26 INPATH_BATCHES=32,64 timeout 900 python $ROOT/scripts/inpath_tune.py $OUT/tune_inpath.txt $OUT/inpath_report.txt 2>&1 | tail -8
head -3 $OUT/inpath_report.txt
