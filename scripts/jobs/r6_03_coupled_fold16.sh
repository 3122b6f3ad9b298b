# round 6, lease 3: inner-seam tests (the reference's DDIMSampler over the HIP U-Net), the bf16-library child test, the coupled
# loop at 16 steps per launch set (B' = 192) and on C3, tile choices to a tune cache; kernel statistics of the coupled single batch
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_03; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_compat.py tests/test_gpu_coupled.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 1500 python -m pytest tests/test_gpu_e2e_fullsize.py -x -q -m gpu -k "bf16_library" > $OUT/pytest_bf16.log 2>&1; echo "pytest bf16 rc=$?"; tail -5 $OUT/pytest_bf16.log
cp gpurun_out/parity_report*.json $OUT/ 2>/dev/null
export CYCLEDIFF_TUNE_SPLITK=1
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_new.txt
timeout 1500 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/bench_c16_coupled.json 2> $OUT/bench_c16_coupled.err; tail -1 $OUT/bench_c16_coupled.json | cut -c1-200; tail -3 $OUT/bench_c16_coupled.err
timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/bench_c16_coupled_b.json 2> $OUT/bench_c16_coupled_b.err; tail -1 $OUT/bench_c16_coupled_b.json | cut -c1-200
CYCLEDIFF_COUPLE=0 timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/bench_c16_two_loops.json 2> $OUT/bench_c16_two_loops.err; tail -1 $OUT/bench_c16_two_loops.json | cut -c1-200
timeout 900 python bench.py --coalesce 8 --steps 16 --warmup 0 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/bench_c8_coupled.json 2> $OUT/bench_c8_coupled.err; tail -1 $OUT/bench_c8_coupled.json | cut -c1-200
timeout 900 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3_coupled.json 2> $OUT/bench_c3_coupled.err; tail -1 $OUT/bench_c3_coupled.json | cut -c1-200
timeout 900 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3_coupled_b.json 2> $OUT/bench_c3_coupled_b.err; tail -1 $OUT/bench_c3_coupled_b.json | cut -c1-200
wc -l $OUT/tune_new.txt
unset CYCLEDIFF_TUNE_SPLITK
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1stats -o b -- python $GRAFT_REPO_ROOT/bench.py --coalesce 1 --steps 2 --warmup 1 --no-cpu-baseline --no-single-batch --no-bf16 > $OUT/c1_rocprof.log 2>&1
python $GRAFT_REPO_ROOT/scripts/kernel_breakdown.py /tmp/c1stats > $OUT/bench_coalesce1_coupled_kernel_breakdown.txt 2>&1
head -30 $OUT/bench_coalesce1_coupled_kernel_breakdown.txt
tail -1 $OUT/c1_rocprof.log | cut -c1-200
