# round 6, lease 21: the two-per-CU tiles (24 = 128x256, 25 = 256x128; 4 waves, BK 32, 3-deep ring) with their co-resident workgroups started half a
# tile apart (CYCLEDIFF_DEPHASE_TICKS shader cycles of delay for the workgroup in the odd wave slot, first wave front only), against the table's
# choices, isolated launches of the GEGLU / ff2 / qkv shapes at B' = 64 and 128 (scripts/bench_gemm.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_21; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv2d_16bit or lin_stream or geglu" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
for B in 64 128; do
  for T in 0 6000 12000 20000 30000 45000; do
    echo "== B=$B dephase $T"
    for s in geglu "(ff2)" "(qk)" "lin 640>640"; do
      CYCLEDIFF_DEPHASE_TICKS=$T timeout 300 python scripts/bench_gemm.py $B 20 "$s" 20,22,24,25 2>&1 | grep -v "^shapes\|weighted\|amdgpu.ids"
    done
  done
done > $OUT/dephase_sweep.txt 2>&1
grep "==\|geglu" $OUT/dephase_sweep.txt | cut -c1-170
