#!/bin/bash
# round 3, GPU call 7: attention with 256-query workgroups (A/B), the whole GPU suite on the new tile table, the bench
# line with the sleeping host pacer
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call7
mkdir -p $OUT
cd /tmp
AB=$ROOT/scripts/ubench/abi_bench
for w8 in 0 1; do
  for B in 32 64; do
    rm -rf /tmp/at_$w8$B
    CD_ATTN_W8=$w8 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/at_$w8$B -o a -- $AB attn $B 4096 8 40 1 4 > $OUT/attn_w8_${w8}_b$B.log 2>&1
    f=$(find /tmp/at_$w8$B -name "*kernel_stats.csv" | head -1)
    echo "== CD_ATTN_W8=$w8 B=$B" >> $OUT/attention_wg256_ab.txt
    grep k_attention $f | cut -d, -f1-4 | cut -c1-160 >> $OUT/attention_wg256_ab.txt
  done
done
cat $OUT/attention_wg256_ab.txt
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 900 python bench.py --steps 8 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['config']['host_cpu_cores_used'], d['config'].get('host_busiest_threads'))"
