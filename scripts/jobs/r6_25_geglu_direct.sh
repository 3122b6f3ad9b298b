# round 6, lease 25: the GEGLU epilogue of the streaming kernel straight from the accumulators (CYCLEDIFF_GEGLU_DIRECT=1: no LDS transpose, 8-byte
# stores) against the staged form (=0): op tests in both modes (bit for bit against a conv_gemm tile), isolated launches at B' = 64 / 128, U-Net
# forwards at B' = 64 / 128
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_25; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in 0 1; do
  CYCLEDIFF_GEGLU_DIRECT=$m timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lin_stream" > $OUT/pytest_mode$m.log 2>&1; echo "mode $m pytest rc=$?"; tail -2 $OUT/pytest_mode$m.log
done
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1; do
      echo "== rep $rep B=$B mode $m"
      CYCLEDIFF_GEGLU_DIRECT=$m timeout 300 python scripts/bench_gemm.py $B 30 "geglu 320" 30 2>&1 | grep -v "^shapes\|weighted\|amdgpu.ids"
    done
  done
done > $OUT/geglu_direct_isolated.txt 2>&1
cat $OUT/geglu_direct_isolated.txt | cut -c1-150
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1; do
      echo "== rep $rep B=$B mode $m"
      CYCLEDIFF_GEGLU_DIRECT=$m CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 5 gemmlog 2>&1 | grep "N2560 K320\|ms/forward\|launches"
    done
  done
done > $OUT/unet_by_mode.txt 2>&1
cat $OUT/unet_by_mode.txt | cut -c1-170
