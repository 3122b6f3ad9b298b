# round 5, lease 18: one box, back to back - the round-4 final library (commit e837bcf, built into lib/libcyclediff_r4final.so) at its
# operating point (8 steps per launch set), the round-5 library at 8 and at 16 (default): what of the round's gain is kernels, what is fold
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_18; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
F="--steps 16 --warmup 0 --no-cpu-baseline --no-single-batch"
CYCLEDIFF_LIB=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_r4final.so timeout 900 python bench.py --coalesce 8 $F > $OUT/r4lib_c8.json 2> $OUT/r4lib_c8.err; tail -1 $OUT/r4lib_c8.json | cut -c1-170
timeout 900 python bench.py --coalesce 8 $F > $OUT/r5lib_c8.json 2> $OUT/r5lib_c8.err; tail -1 $OUT/r5lib_c8.json | cut -c1-170
timeout 900 python bench.py --coalesce 16 $F > $OUT/r5lib_c16.json 2> $OUT/r5lib_c16.err; tail -1 $OUT/r5lib_c16.json | cut -c1-170
CYCLEDIFF_LIB=$GRAFT_REPO_ROOT/cycle-diffusion_amd/lib/libcyclediff_r4final.so timeout 900 python bench.py --coalesce 8 $F > $OUT/r4lib_c8_again.json 2> $OUT/r4lib_c8_again.err; tail -1 $OUT/r4lib_c8_again.json | cut -c1-170
