# (needs the candidate configuration 26 = launch_cfg<128, 320, 32, 2, 2, 2>, which was measured and removed: see
# profiles/r5_tile_128x320_two_per_cu_isolated.txt)
# round 5, lease 2: the two-per-CU 128 x 320 tile (id 26) against the table's choice on the short-K shapes, isolated launches
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_02; mkdir -p $OUT
export CYCLEDIFF_TUNE_DEFAULT=$GRAFT_REPO_ROOT/cycle-diffusion_amd/tune_gfx950.txt
AB=$GRAFT_REPO_ROOT/scripts/ubench/abi_bench
run() { timeout 60 $AB conv "$@" 2>&1 | tail -1 | tee -a $OUT/tile26_isolated.txt; }
for rep in 1 2; do
#   B  H  C0 C1 N    k s up act tile iters
for t in 20 26 24 25; do run 32 32 640 0 640 1 1 0 0 $t 50; done
for t in 23 26 21; do run 32 16 1280 0 1280 1 1 0 0 $t 50; done
for t in 20 26; do run 32 32 640 0 1920 1 1 0 0 $t 50; done
for t in 20 26; do run 32 32 2560 0 640 1 1 0 0 $t 50; done
for t in 23 26; do run 32 16 5120 0 1280 1 1 0 0 $t 50; done
for t in 20 26; do run 32 64 1280 0 320 1 1 0 0 $t 50; done
for t in 65558 26; do run 32 32 640 0 5120 1 1 0 3 $t 20; done
for t in 22 26; do run 32 16 1280 0 10240 1 1 0 3 $t 20; done
for t in 30 26 20; do run 32 64 320 0 320 1 1 0 0 $t 50; done
for t in 20 26; do run 32 64 320 0 320 3 1 0 0 $t 20; done
for t in 20 26; do run 32 32 640 0 640 3 1 0 0 $t 20; done
done
