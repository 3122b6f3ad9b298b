# round 5, lease 13: SQ counters of the two d = 40 attention kernels (CD_ATTN_D40 = 0 / 1), B' = 32, 4096 tokens, 8 heads
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_13; mkdir -p $OUT
AB=$GRAFT_REPO_ROOT/scripts/ubench/abi_bench
SETA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
SETB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"
cd /tmp
for v in 0 1; do
  for s in A B; do
    if [ $s = A ]; then C="$SETA"; else C="$SETB"; fi
    rm -rf /tmp/pmc_at$v$s
    CD_ATTN_D40=$v timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_at$v$s -o p -- $AB attn 32 4096 8 40 1 3 > $OUT/attn_${v}_pmc$s.log 2>&1
    f=$(find /tmp/pmc_at$v$s -name "*counter_collection.csv" | head -1)
    echo "== CD_ATTN_D40=$v (set $s)" >> $OUT/attn_sq_counters.txt
    if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $f k_attention >> $OUT/attn_sq_counters.txt; else tail -3 $OUT/attn_${v}_pmc$s.log >> $OUT/attn_sq_counters.txt; fi
  done
done
cat $OUT/attn_sq_counters.txt
