#!/bin/bash
# round 3, GPU call 14: fp32 GroupNorm with the last-slab fold; the reference's SD ensemble experiment (540 candidates
# per image) through bench.py and main.py, folded and (1-trial subset) unfolded
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_call14
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_f32_path.py tests/test_gpu_wrappers.py tests/test_gpu_models.py -q -x > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
timeout 600 python bench.py --workload c5r --precision fp32x3 --coalesce 1 --steps 2 --warmup 1 --no-single-batch > $OUT/bench_c5r_x3.json 2> $OUT/bench_c5r_x3.err
tail -1 $OUT/bench_c5r_x3.json | cut -c 1-300
export CYCLEDIFF_TUNE_CACHE=$OUT/tune_ens.txt
timeout 900 python bench.py --workload c2e --steps 1 --warmup 0 > $OUT/bench_c2e_folded.json 2> $OUT/bench_c2e_folded.err
tail -1 $OUT/bench_c2e_folded.json | cut -c 1-300
tail -3 $OUT/bench_c2e_folded.err
timeout 900 python bench.py --workload c2e --steps 1 --warmup 0 --no-fold --trials 1 > $OUT/bench_c2e_unfolded_1trial.json 2> $OUT/bench_c2e_unfolded_1trial.err
tail -1 $OUT/bench_c2e_unfolded_1trial.json | cut -c 1-300
tail -3 $OUT/bench_c2e_unfolded_1trial.err
# the same experiment through the evaluation driver on one synthetic 512 x 512 image
mkdir -p $OUT/data
python - <<PY
import json, numpy as np
from PIL import Image
rng = np.random.default_rng(5)
base = rng.random((16, 16, 3))
img = np.kron(base, np.ones((32, 32, 1)))  # blocky synthetic picture, 512 x 512
Image.fromarray((img * 255).astype("uint8")).save("$OUT/data/img0.png")
json.dump([{"img_path": "img0.png", "encode_text": "a photo of a cat", "decode_text": "a photo of a dog"}],
          open("$OUT/data/triplets.json", "w"))
PY
timeout 900 python main.py --cfg experiments/translate_text2img256_stable_diffusion_stochastic_1.cfg --data $OUT/data/triplets.json --output_dir $OUT/main_out --per_device_eval_batch_size 1 --synthetic-weights > $OUT/main_ensemble.json 2> $OUT/main_ensemble.err
tail -1 $OUT/main_ensemble.json
tail -3 $OUT/main_ensemble.err
rm -rf $OUT/data $OUT/main_out/*.png
