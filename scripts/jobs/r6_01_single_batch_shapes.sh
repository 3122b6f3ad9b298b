# round 6, lease 1: where the single-batch operating point (B' = 4 encode / 8 guided decode) spends its time - per-shape GEMM
# logs in situ and one-forward kernel breakdowns by rocprofv3
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_01; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for B in 4 8; do CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 20 gemmlog > $OUT/unet_b${B}_gemmlog.txt 2>&1; grep "ms/forward" $OUT/unet_b${B}_gemmlog.txt; done
cd /tmp
for B in 4 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ub$B -o b -- python $GRAFT_REPO_ROOT/scripts/bench_unet.py $B 10 > $OUT/rocprof_b$B.log 2>&1
  python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/ub$B @k_timestep_embedding > $OUT/unet_b${B}_kernel_breakdown.txt 2>&1
  head -30 $OUT/unet_b${B}_kernel_breakdown.txt
done
