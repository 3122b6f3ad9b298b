# round 6, lease 20: final tree (straight-line GEMM epilogues) - the driver's command (all legs), the folded line under rocprofv3 (kernel stats), the coupled
# single batch under rocprofv3, PMC traffic at B' = 64 / 128 and per shape, per-shape GEMM logs, C3 and the reference-arithmetic lines, phase timing
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_20; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; tail -1 $OUT/bench_driver_cmd.json | cut -c1-200
bash scripts/profile_bench.sh > $OUT/profile_bench.log 2>&1; tail -40 $OUT/profile_bench.log | head -34
cp -r gpurun_out/prof_bench $OUT/
bash scripts/profile_unet_pmc_by_shape.sh 64 128 > $OUT/pmc.log 2>&1; grep "launches/forward\|bytes_per_launch" $OUT/pmc.log | cut -c1-200
cp -r gpurun_out/prof_pmc $OUT/
for B in 12 64 128; do CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 3 gemmlog > $OUT/unet_b${B}_gemmlog.txt 2>&1; grep "launches\|ms/forward" $OUT/unet_b${B}_gemmlog.txt; done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1stats -o b -- python $GRAFT_REPO_ROOT/bench.py --coalesce 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/c1_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/c1stats > $OUT/bench_coalesce1_kernel_breakdown.txt 2>&1; head -12 $OUT/bench_coalesce1_kernel_breakdown.txt
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/bench_c1.json 2> $OUT/bench_c1.err; tail -1 $OUT/bench_c1.json | cut -c1-160
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-160
timeout 900 python bench.py --precision fp32x3 --coalesce 1 --steps 1 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c2_fp32x3.json 2> $OUT/bench_c2_fp32x3.err; tail -1 $OUT/bench_c2_fp32x3.json | cut -c1-160
timeout 900 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-160
timeout 1200 python bench.py --workload c5 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -1 $OUT/bench_c5.json | cut -c1-160
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1; echo "probe rc=$?"
