# round 4, lease 26: which part of a 64-key tile costs the d = 40 self-attention what - timing experiments of the probe
# build (CD_ATTN_DBG bits, attn.hip; results are wrong by construction): B' = 32, 4096 tokens, 8 heads
OUT=$PWD/gpurun_out/r4_26; mkdir -p $OUT
for d in ${@:-0 1 2 3 8 16 32 48 64 72 75 51 0}; do
  CD_ATTN_TIME=1 CD_ATTN_DBG=$d timeout 60 scripts/ubench/abi_bench_probe attn 32 4096 8 40 1 6 > $OUT/dbg_$d.txt 2>&1
  echo "dbg $d: $(grep 'attn d40' $OUT/dbg_$d.txt | tail -4 | awk '{print $6}' | tr '\n' ' ')"
done
