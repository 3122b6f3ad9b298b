"""End-to-end parity at BASELINE.json's full sizes against the REFERENCE's own CPU run (north_star: "outputs match
the reference CPU path on identical triplets and fixed seeds within a stated PSNR tolerance").

tests/golden/c2_sd512_e2e.npz and c3_ldm256_e2e.npz were produced by oracle/gen_golden_full.py from the reference's
UNetModel / Encoder / Decoder / DDIMSampler (CPU fp32): one (image, source text, target text) triplet through
VAE encode -> posterior sample (SD) / mean (LDM) -> 99-step DPM-Encoder (eta 0.1, encoder scale 1) -> 99-step decode
towards the target text with classifier-free guidance 3 -> VAE decode -> (x + 1) / 2. Here the drop-in wrappers run
the same triplet through their reference API (encode / __call__) on the same weights (rebuilt from the (name, shape)
lists in the fixture), the same contexts and the same CPU-drawn noise.

Stated tolerance: image PSNR >= 35 dB vs the reference image (images in [0, 1],
evaluation/utils.py:60-67); the latent, x_T and the extracted eps slots are compared as well and reported."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

import golden_util as gu
from cycle_diffusion_amd import _ffi
from cycle_diffusion_amd.gan_wrapper.latent_text_wrapper import (LatentDiffStochasticTextWrapper,
                                                                 SDStochasticTextWrapper)
from oracle import nets

pytestmark = pytest.mark.gpu

FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0
PSNR_FLOOR = 35.0 if FMT == 1.0 else 22.0


class SeededEmbedder:
    """cond_stage stand-in: the contexts of the fixture (N(0,1) tensors by seed) keyed by prompt."""

    def __init__(self, dim, seeds):
        self.dim, self.by_text = dim, {"source": seeds["c_src"], "target": seeds["c_tgt"], "": seeds["uc"]}

    def __call__(self, texts):
        return torch.cat([gu.rnd((1, 77, self.dim), self.by_text[t]) for t in texts], 0)


def _run(cls, fx_name, res, ctx_dim, report):
    path = os.path.join(gu.GOLD, fx_name + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated" % fx_name)
    fx = np.load(path, allow_pickle=False)
    seeds = json.loads(str(fx["seeds"]))
    os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = cls(source_model_type="sd-v1-4.ckpt" if ctx_dim == 768 else "text2img-large", custom_steps=int(fx["steps"]),
                eta=float(fx["eta"]), white_box_steps=int(fx["steps"]) + 1, skip_steps=[0],
                encoder_unconditional_guidance_scales=[1.0],
                decoder_unconditional_guidance_scales=[float(fx["dec_scale"])], n_trials=1,
                cond_stage=SeededEmbedder(ctx_dim, seeds), noise_on_cpu=True)
    for net, key, seed in ((w.unet, "unet_names", seeds["unet"]), (w.vae, "vae_names", seeds["vae"])):
        sd = nets.synth_state_dict(json.loads(str(fx[key])), seed)
        n, first = w.engine.load_state_dict(net, sd)
        assert n == 0, first
        assert set(k for k, _ in w.engine.net_params(net)) == set(sd.keys())
        del sd
    image = torch.rand((1, 3, res, res), generator=torch.Generator().manual_seed(seeds["image"]))
    torch.manual_seed(seeds["noise"])  # posterior draw, randn_like(x0), then one draw per sample_xt_next
    with torch.no_grad():
        z_ens = w.encode(image.cuda(), ["source"])
        img = w(z_ens, image.cuda(), ["source"], ["target"])
        lat = res // 8
        z = z_ens[0].view(1, int(fx["steps"]) + 1, 4, lat, lat)
        x_tgt = w.engine.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, z.contiguous(), w._schedule().coef_decode(0),
                                     ctx_c=w.cond_stage(["target"]).cuda(), ctx_uc=w.cond_stage([""]).cuda(),
                                     guidance=float(fx["dec_scale"]))
        # same-text decode (encoder scale 1): the 99-step full-size cycle must return the encoder's own x0
        x_same = w.engine.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, z.contiguous(), w._schedule().coef_decode(0),
                                      ctx_c=w.cond_stage(["source"]).cuda(), guidance=1.0)
    x0_ref = torch.as_tensor(fx["x0"])
    cyc = (x_same.cpu() - x0_ref).abs()
    zc = z.cpu()
    slots = [int(s) for s in fx["z_sub_slots"]]
    zref = torch.as_tensor(fx["z_sub"])
    # x_T = sqrt(a) x0 + sqrt(1-a) n: carries only the VAE-encode error of x0
    xT_err = (zc[:, 0] - zref[:, 0]).abs().max().item()
    eps_rel = [((zc[:, s] - zref[:, i]).abs().max() / zref[:, i].abs().max()).item()
               for i, s in enumerate(slots) if s > 0]
    zn_ref = torch.as_tensor(fx["z_norms"])
    zn_rel = ((zc.flatten(2).norm(dim=2) - zn_ref).abs() / zn_ref).max().item()
    lat_ref = torch.as_tensor(fx["x_tgt"])
    lat_err = (x_tgt.cpu() - lat_ref).abs()
    img_ref = torch.as_tensor(fx["img"])
    p = gu.psnr(img.cpu(), img_ref)
    report.add("e2e/" + fx_name, psnr_db=p, img_maxabs=(img.cpu() - img_ref).abs().max().item(),
               latent_maxabs=lat_err.max().item(), latent_rms=lat_err.pow(2).mean().sqrt().item(),
               latent_ref_rms=lat_ref.pow(2).mean().sqrt().item(), xT_maxabs=xT_err,
               eps_rel_slots=dict(zip([str(s) for s in slots if s > 0], eps_rel)), z_norm_rel=zn_rel,
               cycle99_maxabs=cyc.max().item(), cycle99_rms=cyc.pow(2).mean().sqrt().item(),
               reference_cpu_seconds=float(fx["cpu_seconds"]))
    assert img.shape == (1, 3, res, res) and torch.isfinite(img).all()
    assert p >= PSNR_FLOOR, p
    assert zn_rel < 2e-3 * FMT, zn_rel
    assert max(eps_rel) < 5e-2 * FMT, eps_rel
    # 99-step self-cycle at full size on a random-init 860 M-parameter network (the fp32 reference closes it to 1.6e-5,
    # SURVEY.md 8c; a 16-bit engine re-quantises x_t every step). The chain is chaotic in its low bits: over six
    # builds of round 2 the rms error was 0.017-0.020 (C2) / 0.029-0.032 (C3) of a unit-variance latent, the maximum
    # over 16 K / 4 K elements 0.11-0.16. The rms is the stable figure and carries the bound (2x); the maximum gets
    # the same factor.
    assert cyc.pow(2).mean().sqrt().item() < 0.065 * FMT, cyc.pow(2).mean().sqrt().item()
    assert cyc.max().item() < 0.32 * FMT, cyc.max().item()
    return p


def test_c2_sd_v14_512_end_to_end_vs_reference(report):
    """BASELINE config 2 (headline): SD-v1.4 shapes at 512 x 512 through SDStochasticTextWrapper
    (stable_diffusion_stochastic_text_wrapper.py:169-249)."""
    _run(SDStochasticTextWrapper, "c2_sd512_e2e", 512, 768, report)


def test_c3_ldm_text2img_256_end_to_end_vs_reference(report):
    """BASELINE config 3: LDM text2img-large shapes at 256 x 256 through LatentDiffStochasticTextWrapper
    (latentdiff_stochastic_text_wrapper.py:168-201; posterior mean)."""
    _run(LatentDiffStochasticTextWrapper, "c3_ldm256_e2e", 256, 1280, report)
