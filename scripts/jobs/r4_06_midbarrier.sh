# round 4, lease 6: K loop with the barrier inside the segment that holds register-resident MFMAs (cross-tile fragment prefetch)
OUT=gpurun_out/r4_06; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > $OUT/pytest_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/pytest_ops.log
for v in base new; do
  L=cycle-diffusion_amd/lib/libcyclediff.so
  [ $v = base ] && L=cycle-diffusion_amd/lib/libcyclediff_r4base.so
  echo "== $v" >> $OUT/gemm_ab.log
  CYCLEDIFF_LIB=$PWD/$L GEMM_ACT_OR=0x200 timeout 300 python scripts/bench_gemm.py 32 20 "" 20,23,22,5,1 2>&1 | grep -v "^shapes" >> $OUT/gemm_ab.log
done
grep "==\|weighted" $OUT/gemm_ab.log
for v in base new; do
  L=cycle-diffusion_amd/lib/libcyclediff.so
  [ $v = base ] && L=cycle-diffusion_amd/lib/libcyclediff_r4base.so
  CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 32 10 gemmlog > $OUT/unet_b32_$v.txt 2>&1
  grep "B=32\|\[conv_gemm\]" $OUT/unet_b32_$v.txt
done
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1
awk '/^==/{print} /shader clock/{print "   " $0} /per K step|row passes|whole wave|MFMA pipe/{c[$0]++; if (c[$0]==1) print}' $OUT/probe/report.txt | cut -c1-175 | awk '/^==/{n=0} {n++; if(n<=8) print}' | head -40
