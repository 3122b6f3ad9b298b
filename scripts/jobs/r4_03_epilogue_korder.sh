# round 4, lease 3: the 2 GiB fix + block epilogue (op tests, VAE batches, folded slots >= 16), A/B of the block epilogue and of
# the channel-major K order against the round-3 library, L2 hit counters of the 320 -> 320 conv in both orders
OUT=gpurun_out/r4_03; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.log
CYCLEDIFF_KORDER=1 timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or lin_stream" > $OUT/pytest_ops_korder1.log 2>&1; echo "ops korder1 rc=$?"; tail -3 $OUT/pytest_ops_korder1.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -q -x -k "2_gib or folded_into_a_batch_of_32 or ensemble_decode_call" > $OUT/pytest_big.log 2>&1; echo "big rc=$?"; tail -3 $OUT/pytest_big.log
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
SH="conv3"
for v in base new new_k1; do
  for actor in 0 0x200 0x300; do
    L=cycle-diffusion_amd/lib/libcyclediff.so; K=0
    [ $v = base ] && L=cycle-diffusion_amd/lib/libcyclediff_r4base.so
    [ $v = new_k1 ] && K=1
    echo "== $v act|$actor" >> $OUT/gemm_ab.log
    CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_KORDER=$K GEMM_ACT_OR=$actor timeout 300 python scripts/bench_gemm.py 32 20 "$SH" 20,23 2>&1 | grep -v "^shapes" >> $OUT/gemm_ab.log
  done
done
grep "==\|weighted" $OUT/gemm_ab.log
for K in 0 1; do
  CYCLEDIFF_KORDER=$K timeout 300 python scripts/probe_report.py run $OUT/probe_k$K > $OUT/probe_k$K.log 2>&1
  grep "^==\|per K step\|row passes\|whole wave\|shader clock" $OUT/probe_k$K/report.txt | head -60 > $OUT/probe_k${K}_summary.txt
done
cd /tmp
for K in 0 1; do
  for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pmc_$n
    CYCLEDIFF_KORDER=$K timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$n -o p -- python $GRAFT_REPO_ROOT/scripts/bench_gemm.py 32 3 "conv3 320>320" 20 > /dev/null 2>&1
    f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
    echo "== korder $K : $c" >> $GRAFT_REPO_ROOT/$OUT/pmc.log
    [ -n "$f" ] && python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $f k_conv_gemm >> $GRAFT_REPO_ROOT/$OUT/pmc.log
  done
done
cat $GRAFT_REPO_ROOT/$OUT/pmc.log
