# round 4, lease 10: tiny unconditional-LDM test in both precisions; tuner with split-K where a tile configuration leaves CUs
# idle and the two-per-CU configurations (ids 24, 25): C2 launch sets re-measured, in-situ table
OUT=$PWD/gpurun_out/r4_10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ldm_uncond.py tests/test_gpu_ops.py -q > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log
python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')):
    if row['name'].startswith('ldm_uncond/wrapper'): print(row['name'], {k: row[k] for k in ('eps_rel','latent_rel_to_max','psnr_norefine_db','psnr_refined_db','flipped_cells','psnr_norefine_away_from_flipped_cells_db')})
PY
echo "== isolated, tiles 20 23 22 24 25" > $OUT/gemm_new_tiles.log
GEMM_ACT_OR=0 timeout 300 python scripts/bench_gemm.py 32 20 "" 20,23,22,24,25 2>&1 | grep -v "^shapes" >> $OUT/gemm_new_tiles.log
grep "lin\|geglu\|weighted" $OUT/gemm_new_tiles.log | cut -c1-150
export CYCLEDIFF_TUNE_DEFAULT=/dev/null CYCLEDIFF_TUNE_SPLITK=1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c2_8.txt timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/tune_c2_8.log 2>&1
CYCLEDIFF_TUNE_CACHE=$OUT/tune_c3.txt timeout 900 python bench.py --workload c3 --steps 4 --warmup 4 --no-cpu-baseline > $OUT/tune_c3.log 2>&1
python scripts/merge_tune.py cycle-diffusion_amd/tune_gfx950.txt $OUT/tune_c2_8.txt $OUT/tune_c3.txt -o $OUT/tune_merged.txt
unset CYCLEDIFF_TUNE_SPLITK
export CYCLEDIFF_TUNE_DEFAULT=$OUT/tune_merged.txt
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_retuned.txt 2>&1; grep "B=32\|\[conv_gemm\]" $OUT/unet_b32_retuned.txt
CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 6 gemmlog > $OUT/unet_b64_retuned.txt 2>&1; grep "B=64\|\[conv_gemm\]" $OUT/unet_b64_retuned.txt
timeout 900 python bench.py --steps 8 --warmup 8 --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.json | cut -c1-200
timeout 600 python bench.py --workload c3 --steps 8 --warmup 4 --no-cpu-baseline > $OUT/bench_c3.json 2> $OUT/bench_c3.err; tail -1 $OUT/bench_c3.json | cut -c1-200
