# round 4, lease 7: GEGLU block epilogue check, then re-measure the tile table with the round-4 kernel (C2 launch sets of 8 / 4 / 1
# steps, C3, C5 reduced at B = 4 / 16 in the split mode), merge, and run the default bench line on the merged table
OUT=$PWD/gpurun_out/r4_07; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -2 $OUT/pytest_ops.log
export CYCLEDIFF_TUNE_DEFAULT=/dev/null CYCLEDIFF_TUNE_SPLITK=1
T0=$(date +%s)
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c2_8.txt timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/tune_c2_8.log 2>&1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c2_4.txt timeout 900 python bench.py --steps 4 --warmup 4 --coalesce 4 --no-cpu-baseline --no-single-batch > $OUT/tune_c2_4.log 2>&1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c3.txt timeout 900 python bench.py --workload c3 --steps 4 --warmup 4 --no-cpu-baseline > $OUT/tune_c3.log 2>&1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c5r_16.txt timeout 900 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/tune_c5r_16.log 2>&1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c5r_4.txt timeout 900 python bench.py --workload c5r --steps 1 --warmup 1 --coalesce 1 --no-cpu-baseline > $OUT/tune_c5r_4.log 2>&1
echo "tuning took $(( $(date +%s) - T0 )) s"; wc -l $OUT/tune_*.txt
python scripts/merge_tune.py cycle-diffusion_amd/tune_gfx950.txt $OUT/tune_c2_8.txt $OUT/tune_c2_4.txt $OUT/tune_c3.txt $OUT/tune_c5r_16.txt $OUT/tune_c5r_4.txt -o $OUT/tune_merged.txt
unset CYCLEDIFF_TUNE_SPLITK
export CYCLEDIFF_TUNE_DEFAULT=$OUT/tune_merged.txt
timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-200
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_retuned.txt 2>&1; grep "B=32\|\[conv_gemm\]" $OUT/unet_b32_retuned.txt
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-200
timeout 600 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r.json 2> $OUT/bench_c5r.err; tail -1 $OUT/bench_c5r.json | cut -c1-200
