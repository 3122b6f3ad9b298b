# round 6, lease 8: grouped tile walk (CYCLEDIFF_TILE_GROUP) for the wide-N contractions: per-shape in-situ times and traffic at B' = 64,
# then the folded line, one box
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_08; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cfg in "0 2048" "4 2048" "8 2048" "16 2048" "4 1280" "8 640"; do
  set -- $cfg
  export CYCLEDIFF_TILE_GROUP=$1 CYCLEDIFF_TILE_GROUP_MIN_N=$2
  CYCLEDIFF_GEMM_LOG=1 timeout 300 python scripts/bench_unet.py 64 4 gemmlog > $OUT/unet_b64_g$1_n$2.txt 2>&1
  echo "group $1 min_n $2: $(grep 'launches' $OUT/unet_b64_g$1_n$2.txt | head -1) | $(grep 'ms/forward' $OUT/unet_b64_g$1_n$2.txt)"
  grep "N10240 K1280\|N5120 K640\|N2560 K320 \|N3840 K1280\|N1920 K640\|M16384 N1280 K1280 \|M65536 N640 K5760 \|M16384 N1280 K11520 k3 s1 z" $OUT/unet_b64_g$1_n$2.txt | cut -c1-130
done
export CYCLEDIFF_TILE_GROUP=8 CYCLEDIFF_TILE_GROUP_MIN_N=2048
bash scripts/profile_unet_pmc_by_shape.sh 64 > $OUT/pmc_g8.log 2>&1; tail -14 $OUT/pmc_g8.log | cut -c1-140
cp gpurun_out/prof_pmc/conv_gemm_traffic_by_shape_b64.json $OUT/conv_gemm_traffic_by_shape_b64_group8.json
for g in 8 0 8 0; do
  CYCLEDIFF_TILE_GROUP=$g timeout 900 python bench.py --steps 16 --warmup 0 --no-cpu-baseline --no-bf16 --no-single-batch > $OUT/bench_g${g}_$RANDOM.json 2> $OUT/err.txt
  echo "bench group $g: $(ls -t $OUT/bench_g${g}_*.json | head -1 | xargs tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])")"
done
