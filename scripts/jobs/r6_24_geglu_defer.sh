# round 6, lease 24: the GEGLU epilogue of the streaming kernel run out of phase on each SIMD's two waves (CYCLEDIFF_GEGLU_DEFER: bit 0 = waves 4 .. 7
# run a tile's epilogue behind the next barrier, bit 1 = waves 0 .. 3 raise their priority for the second half of a tile's MFMAs): op tests in every
# mode (bit for bit against a conv_gemm tile), isolated launches at B' = 64 / 128, U-Net forwards at B' = 64
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_24; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for m in 1 3 2; do
  CYCLEDIFF_GEGLU_DEFER=$m timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lin_stream" > $OUT/pytest_mode$m.log 2>&1; echo "mode $m pytest rc=$?"; tail -2 $OUT/pytest_mode$m.log
done
for rep in 1 2; do
  for B in 64 128; do
    for m in 0 1 3 2; do
      echo "== rep $rep B=$B mode $m"
      CYCLEDIFF_GEGLU_DEFER=$m timeout 300 python scripts/bench_gemm.py $B 30 "geglu 320" 30 2>&1 | grep -v "^shapes\|weighted\|amdgpu.ids"
    done
  done
done > $OUT/geglu_defer_isolated.txt 2>&1
cat $OUT/geglu_defer_isolated.txt | cut -c1-150
for rep in 1 2; do
  for m in 0 3 1; do
    echo "== rep $rep mode $m"
    CYCLEDIFF_GEGLU_DEFER=$m CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 5 gemmlog 2>&1 | grep "N2560 K320\|ms/forward\|launches"
  done
done > $OUT/unet_b64_by_mode.txt 2>&1
cat $OUT/unet_b64_by_mode.txt | cut -c1-170
