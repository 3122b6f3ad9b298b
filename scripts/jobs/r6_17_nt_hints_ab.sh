# round 6, lease 17: op tests of the 16-bit epilogue (tests/test_gpu_ops.py::test_conv2d_16bit_epilogue); streaming (nt) cache policy on the
# epilogue's output stores (build.py --ntst) / residual loads (--ntld) against the product library, one box, alternating
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_17; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
  for v in prod ntst ntld; do
    if [ $v = prod ]; then unset CYCLEDIFF_LIB; else export CYCLEDIFF_LIB=$L/libcyclediff_$v.so; fi
    timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --single-steps 3 > $OUT/bench_${v}_$i.json 2> $OUT/bench_${v}_$i.err; echo "$v $(tail -1 $OUT/bench_${v}_$i.json | cut -c1-140)"
  done
done
unset CYCLEDIFF_LIB
