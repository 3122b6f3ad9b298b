#!/bin/bash
# round 3, GPU call 21: classifier-free-guidance prefix sharing (conv_in .. first self-attention once for both halves):
# parity tests, then same-box A/B of the default bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call21
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_e2e_fullsize.py tests/test_gpu_models.py tests/test_gpu_wrappers.py tests/test_gpu_fullsize.py tests/test_gpu_model_api.py -q -x > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
for share in 0 1 0 1; do
  CYCLEDIFF_CFG_SHARE=$share timeout 600 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_share${share}.json 2> $OUT/bench.err
  echo "CYCLEDIFF_CFG_SHARE=$share $(tail -1 $OUT/bench_share${share}.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"])')" | tee -a $OUT/cfg_share_ab.txt
done
