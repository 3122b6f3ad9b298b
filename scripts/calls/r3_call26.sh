#!/bin/bash
# round 3, GPU call 26: two launch sets in flight (two engine replicas / streams) against the default, same box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call26
mkdir -p $OUT
cd $ROOT
for rep in 1 2 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 8 --in-flight $rep --no-cpu-baseline --no-single-batch > $OUT/bench_if$rep.json 2> $OUT/bench.err
  echo "--in-flight $rep $(tail -1 $OUT/bench_if$rep.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["host_cpu_cores_used"], d["config"]["host_busiest_threads"])')" | tee -a $OUT/in_flight_ab.txt
done
