# round 6, lease 13: phase timing of the conv tile (probe build) on the round-6 tree: what the row passes cost after the pack8 change
OUT=gpurun_out/r6_13; mkdir -p $OUT
timeout 900 python scripts/probe_report.py run $OUT/probe > $OUT/probe.log 2>&1
echo "probe rc=$?"; tail -3 $OUT/probe.log; grep -v "^    \|^  wave\|^$" $OUT/probe/report.txt | head -80
