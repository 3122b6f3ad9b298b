# round 6, lease 28: the streaming kernel waits for a tile's NEXT pieces ahead of the tile's stores (CYCLEDIFF_LIN_PREWAIT=1) instead of behind them (=0:
# the wait, written in loads, then also sits out the stores' acknowledgements): op tests in mode 1, U-Net forwards at B' = 64 / 128 per mode
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_28; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CYCLEDIFF_LIN_PREWAIT=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lin_stream" > $OUT/pytest_mode1.log 2>&1; echo "mode 1 pytest rc=$?"; tail -2 $OUT/pytest_mode1.log
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1; do
      echo "== rep $rep B=$B prewait $m"
      CYCLEDIFF_LIN_PREWAIT=$m CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 5 gemmlog 2>&1 | grep "lin_stream\|ms/forward\|launches"
    done
  done
done > $OUT/unet_by_mode.txt 2>&1
cat $OUT/unet_by_mode.txt | cut -c1-170
