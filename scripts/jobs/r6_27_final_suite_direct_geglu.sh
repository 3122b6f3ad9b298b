# round 6, lease 27: final tree (GEGLU epilogue of the streaming kernel straight from the accumulators): the whole GPU suite, smoke(), the driver's
# command, the folded line under rocprofv3 (kernel stats), per-shape GEMM logs at B' = 64 / 128
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_27; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_all.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; tail -1 $OUT/bench_driver_cmd.json | cut -c1-300
bash scripts/profile_bench.sh > $OUT/profile_bench.log 2>&1; tail -40 $OUT/profile_bench.log | head -34
cp -r gpurun_out/prof_bench $OUT/
for B in 64 128; do CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py $B 3 gemmlog > $OUT/unet_b${B}_gemmlog.txt 2>&1; grep "launches\|ms/forward" $OUT/unet_b${B}_gemmlog.txt; done
