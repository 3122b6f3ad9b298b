# round 6, lease 7: HBM-side traffic per GEMM shape of one U-Net forward at B' = 64, for the three K orders of the 3 x 3 convs
# (CYCLEDIFF_KORDER=1 product default: channel-major on large activations; 0 tap-major everywhere; 2 channel-major everywhere)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_07; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for ko in 1 0 2; do
  export CYCLEDIFF_KORDER=$ko
  bash scripts/profile_unet_pmc_by_shape.sh 64 > $OUT/pmc_k$ko.log 2>&1; tail -16 $OUT/pmc_k$ko.log
  cp gpurun_out/prof_pmc/conv_gemm_traffic_by_shape_b64.json $OUT/conv_gemm_traffic_by_shape_b64_korder$ko.json
  cp gpurun_out/prof_pmc/conv_gemm_traffic_b64.json $OUT/conv_gemm_traffic_b64_korder$ko.json
  CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 3 gemmlog > $OUT/unet_b64_gemmlog_korder$ko.txt 2>&1; grep "ms/forward\|launches" $OUT/unet_b64_gemmlog_korder$ko.txt
done
