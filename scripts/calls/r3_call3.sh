#!/bin/bash
# round 3, GPU call 3: lin_stream with loads-only wait counts: all op cases, the in-situ forward, end-to-end parity with
# the K = 320 linears re-tuned online, then the bench line with host pacing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call3
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "lin_stream or geglu" > $OUT/t_ops.log 2>&1
tail -12 $OUT/t_ops.log
awk '!($3==320 && $4==1)' cycle-diffusion_amd/tune_gfx950.txt > /tmp/tune_nolin.txt
export CYCLEDIFF_TUNE_DEFAULT=/tmp/tune_nolin.txt
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
CYCLEDIFF_GEMM_LOG=1 timeout 600 python scripts/bench_unet.py 32 3 gemmlog > $OUT/unet_b32_gemmlog.txt 2>&1
grep -E "K320 |ms/forward|conv_gemm\]" $OUT/unet_b32_gemmlog.txt | head -12
CYCLEDIFF_GEMM_LOG=1 timeout 600 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog.txt 2>&1
grep -E "K320 |ms/forward|conv_gemm\]" $OUT/unet_b64_gemmlog.txt | head -12
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -q -k "sd or c2_sd or folded" > $OUT/t_full.log 2>&1
tail -5 $OUT/t_full.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 900 python bench.py --steps 8 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-1500
tail -3 $OUT/bench.err
