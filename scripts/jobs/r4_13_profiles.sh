# round 4, lease 13: rocprofv3 kernel statistics of the default bench line, PMC traffic of the GEMM family, kernel breakdown of
# the C2 line in the split mode of the reference's arithmetic
bash scripts/profile_bench.sh
mkdir -p gpurun_out/r4_13; cp -r gpurun_out/prof_bench gpurun_out/r4_13/
bash scripts/profile_unet_pmc.sh 32 64
cp -r gpurun_out/prof_pmc gpurun_out/r4_13/
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3stats -o b -- python $GRAFT_REPO_ROOT/bench.py --precision fp32x3 --coalesce 1 --steps 1 --warmup 0 --no-cpu-baseline --no-single-batch > $GRAFT_REPO_ROOT/gpurun_out/r4_13/x3.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/x3stats > $GRAFT_REPO_ROOT/gpurun_out/r4_13/c2_fp32x3_kernel_breakdown.txt 2>&1
head -25 $GRAFT_REPO_ROOT/gpurun_out/r4_13/c2_fp32x3_kernel_breakdown.txt
