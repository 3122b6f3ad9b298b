# round 4, lease 2: where does a folded batch stop agreeing with its samples run alone; phase timing with a calibrated clock
OUT=gpurun_out/r4_02; mkdir -p $OUT
timeout 600 python scripts/diag/batch_consistency.py > $OUT/diag.log 2>&1; echo "diag rc=$?"; cat $OUT/diag.log | tail -40
CYCLEDIFF_CFG_SHARE=0 DIAG_B=25 timeout 300 python scripts/diag/batch_consistency.py decode decode_v > $OUT/diag_noshare.log 2>&1; tail -5 $OUT/diag_noshare.log
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1
echo "probe rc=$?"; tail -3 $OUT/probe.log; grep -v "^    \|^  wave\|^$" $OUT/probe/report.txt | head -60
# clocks and power while the 320>320 conv runs back to back
(for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head -6; echo --; done) > $OUT/smi.log 2>&1 &
SMI=$!
timeout 60 python scripts/bench_gemm.py 32 400 "conv3 320>320" 20 > $OUT/gemm_loop.log 2>&1
wait $SMI; tail -20 $OUT/smi.log; tail -3 $OUT/gemm_loop.log
