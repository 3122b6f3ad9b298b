# round 5, lease 14: the improved-DDPM AttentionBlocks of the fp32 path on the fp32 flash kernel - fp32-path tests, config-5 fixtures,
# then the reduced-chain and full-chain lines
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_14; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_e2e_fullsize.py -q -m gpu -k "f32 or fp32 or c5 or iddpm or afhq or toy or c1" 2>&1 | tail -6 | tee $OUT/pytest.txt
cp gpurun_out/parity_report.json $OUT/
timeout 600 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-200
timeout 900 python bench.py --workload c5 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -1 $OUT/bench_c5.json | cut -c1-200
