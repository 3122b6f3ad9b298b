# round 6, lease 2: the coupled encode / decode loop (cd_cycle_translate) - parity tests, then the single-batch and the folded
# C2 lines with and without it on one box; tile choices of the new batch sizes (B' = 12, 192) go to a tune cache
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_02; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_coupled.py tests/test_gpu_wrappers.py tests/test_gpu_model_api.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
export CYCLEDIFF_TUNE_SPLITK=1
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c1_coupled.json 2> $OUT/bench_c1_coupled.err; tail -1 $OUT/bench_c1_coupled.json | cut -c1-200
CYCLEDIFF_COUPLE=0 timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c1_two_loops.json 2> $OUT/bench_c1_two_loops.err; tail -1 $OUT/bench_c1_two_loops.json | cut -c1-200
timeout 900 python bench.py --coalesce 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-batch > $OUT/bench_c1_coupled_b.json 2> $OUT/bench_c1_coupled_b.err; tail -1 $OUT/bench_c1_coupled_b.json | cut -c1-200
timeout 1500 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c16_coupled.json 2> $OUT/bench_c16_coupled.err; tail -1 $OUT/bench_c16_coupled.json | cut -c1-200; tail -3 $OUT/bench_c16_coupled.err
timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c16_coupled_b.json 2> $OUT/bench_c16_coupled_b.err; tail -1 $OUT/bench_c16_coupled_b.json | cut -c1-200
CYCLEDIFF_COUPLE=0 timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch > $OUT/bench_c16_two_loops.json 2> $OUT/bench_c16_two_loops.err; tail -1 $OUT/bench_c16_two_loops.json | cut -c1-200
wc -l $OUT/tune_new.txt
