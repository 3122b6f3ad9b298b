# round 4, lease 17: channel-major K order with the per-step offset from masks + a scalar displacement: op tests in both
# orders, isolated 3 x 3 sweep, one B' = 32 forward, the default bench line, fabric traffic of one forward - orders 0 / 1
OUT=$PWD/gpurun_out/r4_17; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x > $OUT/pytest_ops_k1.log 2>&1; echo "ops korder1 rc=$?"; tail -2 $OUT/pytest_ops_k1.log
CYCLEDIFF_KORDER=0 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or lin_stream" > $OUT/pytest_ops_k0.log 2>&1; echo "ops korder0 rc=$?"; tail -2 $OUT/pytest_ops_k0.log
for K in 0 1 0 1; do
  echo "== korder $K" >> $OUT/gemm_ab.log
  CYCLEDIFF_KORDER=$K GEMM_ACT_OR=0x200 timeout 300 python scripts/bench_gemm.py 32 20 "conv3" 20,23 2>&1 | grep -v "^shapes" >> $OUT/gemm_ab.log
done
grep "==\|weighted" $OUT/gemm_ab.log
for K in 0 1 0 1; do
  CYCLEDIFF_KORDER=$K CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_k$K.txt 2>&1
  echo "korder $K: $(grep 'B=32' $OUT/unet_b32_k$K.txt) $(grep '\[conv_gemm\]' $OUT/unet_b32_k$K.txt)"
done
for K in 0 1; do
  CYCLEDIFF_KORDER=$K timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-single-batch > $OUT/bench_k$K.json 2> $OUT/bench_k$K.err; echo "korder $K: $(tail -1 $OUT/bench_k$K.json | cut -c1-150)"
done
cd /tmp
for K in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    CYCLEDIFF_KORDER=$K timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/scripts/bench_unet.py 32 1 > /dev/null 2>&1
  done
  f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py $f $w $OUT/traffic_b32_k$K.json | cut -c1-400
done
