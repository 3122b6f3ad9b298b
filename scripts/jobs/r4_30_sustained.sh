# round 4, lease 30: the sustained-MFMA-rate diagnostic (csrc/diag.hip) alone and inside a short bench line
OUT=$PWD/gpurun_out/r4_30; mkdir -p $OUT
timeout 60 scripts/ubench/abi_bench peak 300 > $OUT/peak.txt 2>&1; echo "peak rc=$?"; cat $OUT/peak.txt
timeout 400 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-single-batch > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -1 $OUT/bench.json | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps({k: d['roofline'].get(k) for k in ('achieved', 'frac', 'sustained_peak', 'frac_of_sustained')}))"
tail -3 $OUT/bench.err
