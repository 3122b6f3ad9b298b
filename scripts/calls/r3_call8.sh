#!/bin/bash
# round 3, GPU call 8: LayerNorm-folded lin_stream variants (op + network tests), attention 256-query workgroups A/B,
# in-situ forward timing
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call8
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -k "layernorm_fold or lin_stream or batch16 or sd_unet" > $OUT/t_ln.log 2>&1
tail -6 $OUT/t_ln.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
cd /tmp
AB=$ROOT/scripts/ubench/abi_bench
for w8 in 0 1; do
  rm -rf /tmp/at_$w8
  CD_ATTN_W8=$w8 timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/at_$w8 -o a -- $AB attn 32 4096 8 40 1 4 > $OUT/attn_w8_$w8.log 2>&1
  echo "== CD_ATTN_W8=$w8 (B 32, 4096 tokens, 8 heads, d 40)" >> $OUT/attention_wg256_ab.txt
  python $ROOT/scripts/kernel_breakdown.py /tmp/at_$w8 | grep -i "attention\|transpose" >> $OUT/attention_wg256_ab.txt
done
cat $OUT/attention_wg256_ab.txt
cd $ROOT
for w8 in 0 1; do
  CD_ATTN_W8=$w8 CYCLEDIFF_GEMM_LOG=1 timeout 600 python scripts/bench_unet.py 32 4 gemmlog > $OUT/unet_b32_w8_$w8.txt 2>&1
  grep -E "ms/forward|conv_gemm\]|N2560 K320|N320 K320" $OUT/unet_b32_w8_$w8.txt | head -6
done
