#!/bin/bash
# round 3, GPU call 6: lin_stream after the statistics fix + scalar-bookkeeping diet: op tests, isolated timings, in-situ
# forwards with the K = 320 linears re-tuned, end-to-end parity, bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call6
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "lin_stream or geglu or saturate" > $OUT/t_ops.log 2>&1
tail -4 $OUT/t_ops.log
export CYCLEDIFF_TUNE_DEFAULT=$ROOT/cycle-diffusion_amd/tune_gfx950.txt
AB=scripts/ubench/abi_bench
{
for a in "32 64 320 0 320 1 1 0 0" "32 64 320 0 320 1 1 0 256" "32 64 320 0 320 1 1 0 768" "32 64 320 0 640 1 1 0 0" "64 64 320 0 320 1 1 0 256" "32 64 320 0 2560 1 1 0 3" "64 64 320 0 2560 1 1 0 3"; do
  timeout 60 $AB conv $a 30 20 | tail -1
done
} > $OUT/lin_stream_isolated.txt 2>&1
cat $OUT/lin_stream_isolated.txt
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
tail -1 $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('single_batch_value'), d['roofline']['achieved'], d['config']['host_cpu_cores_used'], d['config'].get('host_busiest_threads'))"
