# round 4, lease 27: power and clocks under load (rocm-smi sampled beside a steady run): the bare MFMA loop, the product conv
# tile (320 -> 320, 3 x 3, 64 x 64, B' = 32) and the d = 40 self-attention. Is the chip at its power cap?
OUT=$PWD/gpurun_out/r4_27; mkdir -p $OUT
rocm-smi --showmaxpower --showpower --showclocks --showperflevel > $OUT/smi_idle.txt 2>&1
sample() {  # $1 = tag; samples while the background job $! runs
  local n=0
  sleep 1.5
  while kill -0 $BG 2>/dev/null && [ $n -lt 6 ]; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';' >> $OUT/smi_$1.txt; echo >> $OUT/smi_$1.txt
    n=$((n+1))
  done
  wait $BG
}
PIPE_REPS=250 timeout 60 scripts/ubench/pipe_ubench k_t8_2x5_nord > $OUT/load_mfma.txt 2>&1 & BG=$!; sample mfma
PIPE_REPS=250 timeout 60 scripts/ubench/pipe_ubench k_kl8_2x5_late_barrier_swz > $OUT/load_kloop.txt 2>&1 & BG=$!; sample kloop
CYCLEDIFF_TUNE_DEFAULT=cycle-diffusion_amd/tune_gfx950.txt timeout 60 scripts/ubench/abi_bench conv 32 64 320 0 320 3 1 0 0 20 20000 > $OUT/load_conv.txt 2>&1 & BG=$!; sample conv
timeout 60 scripts/ubench/abi_bench attn 32 4096 8 40 1 60 > $OUT/load_attn.txt 2>&1 & BG=$!; sample attn
PIPE_REPS=1500 timeout 60 scripts/ubench/pipe_ubench k_att_compiled > $OUT/load_attskel.txt 2>&1 & BG=$!; sample attskel
head -40 $OUT/smi_idle.txt
for t in mfma kloop conv attn attskel; do echo "== $t"; cat $OUT/smi_$t.txt | cut -c1-400; tail -2 $OUT/load_$t.txt; done
