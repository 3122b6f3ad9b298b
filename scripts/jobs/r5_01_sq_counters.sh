# round 5, lease 1: SQ counters of the short-K half of the GEMM family on the tree round 4 ended with:
# k_lin_stream N = K = 320 (plain / GEGLU 320 -> 2560) and the 256 x 320 tile on N = K = 640 / 1280, isolated launches.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_01; mkdir -p $OUT
export CYCLEDIFF_TUNE_DEFAULT=$GRAFT_REPO_ROOT/cycle-diffusion_amd/tune_gfx950.txt
AB=$GRAFT_REPO_ROOT/scripts/ubench/abi_bench
SETA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
SETB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
cd /tmp
run() {  # name, then the abi_bench conv arguments
  name=$1; shift
  timeout 60 $AB conv "$@" > $OUT/${name}_time.txt 2>&1; tail -1 $OUT/${name}_time.txt
  for s in A B; do
    if [ $s = A ]; then C="$SETA"; else C="$SETB"; fi
    rm -rf /tmp/pmc_$name$s
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$name$s -o p -- $AB conv "$@" > $OUT/${name}_pmc$s.log 2>&1
    f=$(find /tmp/pmc_$name$s -name "*counter_collection.csv" | head -1)
    echo "== conv $* (set $s)" >> $OUT/sq_counters.txt
    if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $f k_ >> $OUT/sq_counters.txt; else tail -3 $OUT/${name}_pmc$s.log >> $OUT/sq_counters.txt; fi
  done
}
#          B  H  C0 C1 N    k s up act tile iters
run lin320   32 64 320 0 320  1 1 0 0 30 4
run geglu320 32 64 320 0 2560 1 1 0 3 30 4
run t640     32 32 640 0 640  1 1 0 0 20 4
run t1280    32 16 1280 0 1280 1 1 0 0 23 4
run geglu640 32 32 640 0 5120 1 1 0 3 65558 4
cat $OUT/sq_counters.txt | head -150
