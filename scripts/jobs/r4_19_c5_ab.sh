# round 4, lease 19: same-box A/B of the C5 reduced line: previous commit's library, this tree in both K orders
OUT=$PWD/gpurun_out/r4_19; mkdir -p $OUT
for v in prev new_k1 new_k0 prev new_k1; do
  L=cycle-diffusion_amd/lib/libcyclediff.so; K=1
  [ $v = prev ] && L=cycle-diffusion_amd/lib/libcyclediff_prev.so
  [ $v = new_k0 ] && K=0
  CYCLEDIFF_LIB=$PWD/$L CYCLEDIFF_KORDER=$K timeout 900 python bench.py --workload c5r --steps 4 --warmup 4 --no-cpu-baseline > $OUT/bench_c5r_$v.json 2> $OUT/bench_c5r_$v.err
  echo "$v: $(tail -1 $OUT/bench_c5r_$v.json | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(round(r["value"],4), round(r["roofline"]["achieved"],1))')"
done
