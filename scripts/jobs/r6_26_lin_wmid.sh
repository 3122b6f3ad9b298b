# round 6, lease 26: the streaming kernel's W pieces issued between the MFMA slices (CYCLEDIFF_LIN_WMID=1) instead of at the head of an iteration (=0),
# on the straight-from-the-accumulators GEGLU epilogue (CYCLEDIFF_GEGLU_DIRECT=1): op tests, isolated launches, U-Net forwards at B' = 64 / 128
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_26; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export CYCLEDIFF_GEGLU_DIRECT=1
for m in 1; do
  CYCLEDIFF_LIN_WMID=$m timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lin_stream" > $OUT/pytest_mode$m.log 2>&1; echo "mode $m pytest rc=$?"; tail -2 $OUT/pytest_mode$m.log
done
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1; do
      echo "== rep $rep B=$B wmid $m"
      CYCLEDIFF_LIN_WMID=$m timeout 300 python scripts/bench_gemm.py $B 30 "320>" 30 2>&1 | grep "geglu 320\|lin 320"
    done
  done
done > $OUT/lin_wmid_isolated.txt 2>&1
cat $OUT/lin_wmid_isolated.txt | cut -c1-150
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1; do
      echo "== rep $rep B=$B wmid $m"
      CYCLEDIFF_LIN_WMID=$m CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 5 gemmlog 2>&1 | grep "lin_stream\|ms/forward\|launches"
    done
  done
done > $OUT/unet_by_mode.txt 2>&1
cat $OUT/unet_by_mode.txt | cut -c1-170
