#!/bin/bash
# round 3, GPU call 23: SQ counters of the d = 40 self-attention kernel (B' = 32, 64 x 64 level), three passes
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call23
mkdir -p $OUT
export PYTHONPATH=$ROOT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python $ROOT/scripts/bench_attn.py 32 4096 8 40 3 > $OUT/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $ROOT/scripts/pmc_summary.py $f k_attention >> $OUT/attention_d40_sq_counters.txt; else tail -3 $OUT/pmc_$i.log; fi
done
cat $OUT/attention_d40_sq_counters.txt
