# round 4, lease 9: the tiny unconditional-LDM chain (16-bit engine) on the round-3 library vs the round-4 one; remaining GPU tests
OUT=$PWD/gpurun_out/r4_09; mkdir -p $OUT
for v in base new; do
  L=cycle-diffusion_amd/lib/libcyclediff.so
  [ $v = base ] && L=cycle-diffusion_amd/lib/libcyclediff_r4base.so
  CYCLEDIFF_LIB=$PWD/$L timeout 600 python -m pytest tests/test_gpu_ldm_uncond.py -q -k "wrapper_vs_reference" > $OUT/pytest_$v.log 2>&1; tail -2 $OUT/pytest_$v.log
  python - <<PY
import json
for row in json.load(open('gpurun_out/parity_report.json')):
    if row['name']=='ldm_uncond/wrapper': print("$v", {k: row[k] for k in ('eps_rel','latent_rel_to_max','psnr_norefine_db','psnr_refined_db','flipped_cells')})
PY
done
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_ldm_uncond.py::test_latentdiff_stochastic_wrapper_vs_reference > $OUT/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -5 $OUT/pytest_rest.log
cp gpurun_out/parity_report.json $OUT/
