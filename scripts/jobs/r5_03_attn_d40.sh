# round 5, lease 3: the LDS-DMA / 16-row-block d = 40 attention kernel (k_attention_d40) - tests, then same-box A/B against the
# round-3 kernel (CD_ATTN_D40=0) by rocprofv3 kernel durations at the C2 shape (B' = 32 and 64, 4096 tokens, 8 heads)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_03; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -5 | tee $OUT/pytest_attention.txt
AB=$GRAFT_REPO_ROOT/scripts/ubench/abi_bench
cd /tmp
for B in 32 64; do
for v in 0 1; do
  rm -rf /tmp/at_$v
  CD_ATTN_D40=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$v -o a -- $AB attn $B 4096 8 40 1 6 > $OUT/attn_b${B}_d40_$v.txt 2>&1
  f=$(find /tmp/at_$v -name "*kernel_stats.csv" | head -1)
  echo "== B' = $B CD_ATTN_D40=$v" | tee -a $OUT/attn_ab.txt
  grep "fingerprint" $OUT/attn_b${B}_d40_$v.txt | tee -a $OUT/attn_ab.txt
  grep -i "k_attention" $f | tee -a $OUT/attn_ab.txt
done
done
