# round 6, lease 11: the 13-operation gelu_fast (max(x, 0) - 0.5 |x| q) against the select form (-DCD_GELU_SELECT_FORM build), one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_11; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_geluold.so
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_clip_text.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 4 gemmlog > $OUT/unet_b64_new_$i.txt 2>&1; echo "new: $(grep 'launches' $OUT/unet_b64_new_$i.txt | head -1)"; grep "act3" $OUT/unet_b64_new_$i.txt | cut -c1-130
  CYCLEDIFF_LIB=$OLD CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 4 gemmlog > $OUT/unet_b64_old_$i.txt 2>&1; echo "old: $(grep 'launches' $OUT/unet_b64_old_$i.txt | head -1)"; grep "act3" $OUT/unet_b64_old_$i.txt | cut -c1-130
done
for i in 1 2; do
  timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_new_$i.json 2> $OUT/err.txt; echo "new $(tail -1 $OUT/bench_new_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['single_batch_value'])")"
  CYCLEDIFF_LIB=$OLD timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_old_$i.json 2> $OUT/err.txt; echo "old $(tail -1 $OUT/bench_old_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['single_batch_value'])")"
done
timeout 900 python -m pytest tests/test_gpu_e2e_fullsize.py -x -q -m gpu -k "end_to_end_vs_reference" > $OUT/pytest_e2e.log 2>&1; echo "e2e rc=$?"; tail -3 $OUT/pytest_e2e.log
