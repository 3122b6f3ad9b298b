#!/bin/bash
# round 3, GPU call 4: diagnostics - lin_stream statistics variants vs conv_gemm vs torch; SQ counters of the streaming
# kernel (GEGLU and residual shapes); which host threads burn CPU during the bench
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call4
mkdir -p $OUT
cd $ROOT
timeout 300 python scripts/diag/lin_stats_diag.py > $OUT/stats_diag.txt 2>&1
cat $OUT/stats_diag.txt | tail -14
AB=$ROOT/scripts/ubench/abi_bench
cd /tmp
SETA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
SETB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA"
i=0
for args in "32 64 320 0 2560 1 1 0 3 30 3" "32 64 320 0 320 1 1 0 256 30 3" "32 64 320 0 320 1 1 0 0 30 3"; do
  i=$((i+1))
  for set in A B; do
    if [ $set = A ]; then C="$SETA"; else C="$SETB"; fi
    rm -rf /tmp/pmc_$i$set
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$i$set -o p -- $AB conv $args > $OUT/pmc_$i$set.log 2>&1
    f=$(find /tmp/pmc_$i$set -name "*counter_collection.csv" | head -1)
    echo "== conv $args (set $set)" >> $OUT/lin_stream_sq_counters.txt
    if [ -n "$f" ]; then python $ROOT/scripts/pmc_summary.py $f lin_stream >> $OUT/lin_stream_sq_counters.txt; else tail -3 $OUT/pmc_$i$set.log >> $OUT/lin_stream_sq_counters.txt; fi
  done
done
cat $OUT/lin_stream_sq_counters.txt
cd $ROOT
awk '!($3==320 && $4==1)' cycle-diffusion_amd/tune_gfx950.txt > /tmp/tune_nolin.txt
cat scripts/calls/tune_r3_call3.txt >> /tmp/tune_nolin.txt
export CYCLEDIFF_TUNE_DEFAULT=/tmp/tune_nolin.txt
timeout 900 python bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['host_cpu_cores_used'], d['config'].get('host_busiest_threads'))"
