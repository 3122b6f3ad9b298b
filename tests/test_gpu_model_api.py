"""The model API the reference's main.py drives (model/unsupervised_translation.py:9-62,
model/text_unsupervised_translation.py:24-40): config file -> get_model(args.model.name)(args) -> forward(**batch)
returns ((original_image, img), zeros[B], {}). BASELINE config 1 (toy DDPM, two wrappers from one [gan] section) runs
unchanged from config/experiments/toy_ddpm_c1.cfg."""
import os
import warnings

import pytest
import torch

from cycle_diffusion_amd import _ffi
from cycle_diffusion_amd.utils.config_utils import get_config
from cycle_diffusion_amd.utils.program_utils import get_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unsupervised_translation_from_the_c1_config():
    args = get_config("experiments/toy_ddpm_c1.cfg", config_root=os.path.join(ROOT, "config"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = get_model(args.model.name)(args).eval()
    assert model.source_gan_wrapper.resolution == model.target_gan_wrapper.resolution == 32
    assert model.source_gan_wrapper.latent_dim == 32 * 32 * 3 * 50
    img = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    sid = torch.arange(2).cuda()
    with torch.no_grad():
        (orig, out), loss, extra = model(sample_id=sid, original_image=img)
    assert orig is img and out.shape == img.shape and torch.isfinite(out).all()
    assert loss.shape == (2,) and float(loss.abs().sum()) == 0.0 and extra == {}
    assert next(model.parameters()).is_cuda  # the trainer reads the device from the parameters
    # source and target are the same toy network (same model type -> same seeded weights): translating is a cycle
    mse = ((out.clamp(0, 1) - img) ** 2).mean().item()
    assert mse < 0.2, mse


def test_main_driver_on_the_c1_config(tmp_path):
    """main.py: config + triplet JSON -> images + metrics.json (unpaired entries carry only img_path)."""
    import json
    import sys
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(0)
    meta = []
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (40, 32, 3), dtype=np.uint8)).save(tmp_path / ("im%d.png" % i))
        meta.append({"img_path": "im%d.png" % i})
    (tmp_path / "data.json").write_text(json.dumps(meta))
    sys.path.insert(0, ROOT)
    import main as driver
    out = tmp_path / "out"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert driver.main(["--cfg", "experiments/toy_ddpm_c1.cfg", "--data", str(tmp_path / "data.json"),
                            "--output_dir", str(out), "--per_device_eval_batch_size", "2"]) == 0
    res = json.loads((out / "metrics.json").read_text())
    assert len(res["samples"]) == 3 and [r["sample_id"] for r in res["samples"]] == [0, 1, 2]
    assert all(np.isfinite(r["psnr"]) and 0 <= r["ssim"] <= 1 for r in res["samples"])
    assert sorted(p.name for p in out.glob("*.png")) == ["000000.png", "000001.png", "000002.png"]


def test_main_driver_fold_look_ahead_matches_batch_by_batch_on_c2(tmp_path):
    """`main.py --fold N` (N dataloader batches in one model() call - the operating point of bench.py's headline) against the
    reference's batch-by-batch loop (trainer/trainer.py:788-833, README.md:153 `--per_device_eval_batch_size 4`) on the C2
    configuration at its real size (SD-v1.4-shaped U-Net + KL-f8 VAE, 512 x 512, 99 + 99 steps, CFG 3; synthetic weights):
    6 triplets as 3 batches of 2, once with --fold 1 and once with --fold 3 (one call of 6 images: other tiles, split-K
    factors, streaming kernels). Every image of the folded run must match its batch-by-batch twin >= 45 dB (the PNGs
    themselves, 8-bit), be a different image from its neighbours, and carry the same sample ids / texts in metrics.json."""
    import json
    import sys
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(3)
    meta = []
    for i in range(6):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)).resize((512, 512), Image.BICUBIC).save(
            tmp_path / ("im%d.png" % i))
        meta.append({"img_path": "im%d.png" % i, "encode_text": "source %d" % i, "decode_text": "target %d" % i})
    (tmp_path / "data.json").write_text(json.dumps(meta))
    sys.path.insert(0, ROOT)
    import main as driver
    outs = {}
    for fold in (1, 3):
        out = tmp_path / ("out_fold%d" % fold)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert driver.main(["--cfg", "experiments/bench_sd_c2.cfg", "--data", str(tmp_path / "data.json"),
                                "--output_dir", str(out), "--per_device_eval_batch_size", "2", "--fold", str(fold),
                                "--synthetic-weights"]) == 0
        res = json.loads((out / "metrics.json").read_text())
        assert [r["sample_id"] for r in res["samples"]] == list(range(6))
        assert [r["decode_text"] for r in res["samples"]] == ["target %d" % i for i in range(6)]
        outs[fold] = [np.asarray(Image.open(out / ("%06d.png" % i)), dtype=np.float64) / 255.0 for i in range(6)]
    ps = []
    for a_, b_ in zip(outs[1], outs[3]):
        mse = float(((a_ - b_) ** 2).mean())
        ps.append(99.0 if mse == 0 else -10.0 * np.log10(mse))
    spread = min(float(np.abs(outs[3][i] - outs[3][i + 1]).mean()) for i in range(5))
    fmt = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0
    assert min(ps) >= (45.0 if fmt == 1.0 else 25.0), ps
    assert spread > 1e-3, spread


def test_unconditional_ldm_translation_config_at_full_size(tmp_path):
    """The reference's ffhq256 -> celeba256 experiment (config/experiments/
    translate_ffhq256_to_celeba256_latentdiff_ddim_eta01.cfg): gan_type LatentDiffStochastic, two full-size
    unconditional LDMs (224-channel U-Net on 64 x 64 latents, VQ-f4 first stage with its 8192-row codebook),
    custom_steps 999, white_box_steps 1000, eta 0.1, refine_steps 400 - 2398 U-Net forwards for one 256 x 256 image,
    through get_config -> get_model -> forward, with `precision = fp32x3` (this repo's key: the U-Nets on the split mode of
    the fp32 path, which is what follows the reference on eta-0.1 chains of this length). Seeded synthetic weights (no checkpoints in this tree): what is checked
    is that the whole configuration runs at its real size and returns a finite image of the right shape; parity of this
    wrapper is pinned on the small networks of test_gpu_ldm_uncond.py."""
    cfg = tmp_path / "ffhq_to_celeba.cfg"
    cfg.write_text("[model]\nname = unsupervised_translation\n\n[gan]\ngan_type = LatentDiffStochastic\n"
                   "source_model_type = ffhq256\ntarget_model_type = celeba256\ncustom_steps = 999\n"
                   "white_box_steps = 1000\neta = 0.1\nrefine_steps = 400\nprecision = fp32x3\n")
    os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"
    args = get_config(str(cfg))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = get_model(args.model.name)(args).eval()
    src, tgt = model.source_gan_wrapper, model.target_gan_wrapper
    assert src.resolution == tgt.resolution == 256 and src.latent_dim == 64 * 64 * 3 * 1000
    assert (src.custom_steps, src.white_box_steps, tgt.refine_steps) == (999, 1000, 400)
    assert src.precision == tgt.precision == "fp32x3"
    img = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        (orig, out), loss, extra = model(sample_id=torch.zeros(1, dtype=torch.int64).cuda(), original_image=img)
    assert orig is img and out.shape == img.shape and torch.isfinite(out).all()
    assert float(loss.abs().sum()) == 0.0 and extra == {}
