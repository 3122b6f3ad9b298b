# round 4, lease 28: shader clock and package power while the product d = 40 self-attention runs back to back (B' = 32)
OUT=$PWD/gpurun_out/r4_28; mkdir -p $OUT
timeout 60 scripts/ubench/abi_bench attn 32 4096 8 40 1 4000 > $OUT/load_attn.txt 2>&1 & BG=$!
sleep 2.5
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr -s ' ' | tr '\n' ';' >> $OUT/smi_attn.txt; echo >> $OUT/smi_attn.txt; done
wait $BG
cat $OUT/smi_attn.txt; tail -1 $OUT/load_attn.txt
