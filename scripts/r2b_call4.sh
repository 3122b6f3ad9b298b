#!/bin/bash
# GPU call: the whole -m gpu suite on the final build, then the default bench with the tile choices of the new shapes
# (fused q|k|v, B' = 64 launch sets) captured into tune_new.txt for the shipped table.
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2b_call4
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd $ROOT
echo "== full gpu suite"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/t_all.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
for wlc in "c2 8" "c2 4" "c2 1" "c3 4"; do
  set -- $wlc
  echo "== tune capture: workload $1 coalesce $2"
  timeout 900 python $ROOT/bench.py --workload $1 --steps $2 --warmup $2 --coalesce $2 --no-cpu-baseline > $OUT/bench_$1_c$2_tuning.json 2> $OUT/bench_$1_c$2_tuning.err
  tail -1 $OUT/bench_$1_c$2_tuning.json | cut -c1-160
done
echo "== bench default (tuned)"
timeout 900 python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-200
wc -l $OUT/tune_new.txt
