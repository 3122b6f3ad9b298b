# round 4, lease 25 (run three times as the generator grew): hardware questions for the next tile design
# (scripts/ubench/gen_pipe_ubench.py): what the LDS fragment reads, the refill stream, the barrier and hipcc's read placement
# cost the power-limited MFMA loop; whether the attention kernel's VALU and MFMA work overlap; the attention skeleton
OUT=$PWD/gpurun_out/r4_25; mkdir -p $OUT
timeout 120 scripts/ubench/pipe_ubench ${1:-} > $OUT/pipe_ubench.txt 2>&1; echo "rc=$?"
cat $OUT/pipe_ubench.txt
