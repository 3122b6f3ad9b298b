#!/bin/bash
# round 3, GPU call 20: software-pipelined d = 40 attention (k_attention_pipe40): op tests, then same-box A/B of the
# kernel inside a B' = 32 U-Net forward (CD_ATTN_PIPE = 0 / 8 / 4)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call20
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" > $OUT/t_attn.log 2>&1
tail -6 $OUT/t_attn.log
cd /tmp
export PYTHONPATH=$ROOT
for pipe in 0 8 4; do
  rm -rf /tmp/tr_$pipe
  CD_ATTN_PIPE=$pipe timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$pipe -o t -- python $ROOT/scripts/bench_unet.py 32 3 > $OUT/unet_pipe$pipe.log 2>&1
  echo "== CD_ATTN_PIPE=$pipe" >> $OUT/attention_pipe_ab.txt
  grep "ms/forward" $OUT/unet_pipe$pipe.log >> $OUT/attention_pipe_ab.txt
  python $ROOT/scripts/kernel_breakdown.py /tmp/tr_$pipe @k_timestep_embedding 2>/dev/null | grep "k_attention\|kernels " >> $OUT/attention_pipe_ab.txt
done
cat $OUT/attention_pipe_ab.txt
