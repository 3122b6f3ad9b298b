# round 4, lease 5: how much of the K step is memory latency? The probe build with every tile's A operand L2-resident
OUT=gpurun_out/r4_05; mkdir -p $OUT
timeout 600 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1
awk '/^==/{print} /shader clock/{print "   " $0} /per K step|row passes|whole wave|MFMA pipe/{c[$0]++; if (c[$0]==1) print}' $OUT/probe/report.txt | cut -c1-175 | awk '/^==/{n=0} {n++; if(n<=8) print}'
