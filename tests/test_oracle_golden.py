"""CPU: the oracle restatement replays the committed reference fixtures (tests/golden, produced by
oracle/gen_golden.py from the reference's own modules). fp32 on both sides: tolerance covers only
summation-order differences between functional and module code paths."""
import numpy as np
import torch

import golden_util as gu
from oracle import nets, samplers

ATOL = 2e-5


def _close(a, b, atol=ATOL, rtol=1e-4):
    a = torch.as_tensor(np.asarray(a)).float()
    b = torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), err


def test_schedule_tables_exact():
    fx = gu.load("schedule_sd_s99_eta0p1")
    ac = samplers.sd_alphas_cumprod()
    assert np.array_equal(ac.numpy(), fx["alphas_cumprod"])
    ts, a, a_prev, sig, r = samplers.ddim_tables(ac, 99, 0.1)
    assert np.array_equal(ts, fx["timesteps"])
    assert np.array_equal(a.numpy(), fx["a"])
    assert np.array_equal(a_prev, fx["a_prev"])
    # sigma is consumed as fp32 (torch.full at use, ddim.py:572): exact there, 1 ulp slack in fp64
    assert np.array_equal(sig.astype(np.float32), fx["sigma"].astype(np.float32))
    assert np.allclose(sig, fx["sigma"], rtol=1e-14, atol=0)
    assert np.array_equal(r.numpy(), fx["r"])
    # the constants SURVEY.md §8(c) quotes
    assert ts[0] == 1 and ts[-1] == 981 and len(ts) == 99
    assert abs(float(sig[0]) - 2.06484e-3) < 1e-7 and abs(float(sig[-1]) - 3.31827e-2) < 1e-6


def test_unet_tiny_sd():
    fx = gu.load("unet_tiny_sd")
    x, t, ctx = gu.tiny_sd_inputs()
    with torch.no_grad():
        y = nets.openai_unet(gu.weights(fx), gu.TINY_SD_CFG, x, t, ctx)
    _close(y, fx["y"])


def test_unet_tiny_iddpm():
    fx = gu.load("unet_tiny_iddpm")
    with torch.no_grad():
        y = nets.openai_unet(gu.weights(fx), gu.TINY_IDDPM_CFG, gu.rnd((2, 3, 32, 32), 3), torch.tensor([3.0, 700.0]))
    _close(y, fx["y"])


def test_vae_tiny():
    fx = gu.load("vae_tiny")
    sd = gu.weights(fx)
    img = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(4)) * 2 - 1
    with torch.no_grad():
        _close(nets.vae_encode_moments(sd, gu.TINY_VAE_CFG, img), fx["moments"])
        _close(nets.vae_decode(sd, gu.TINY_VAE_CFG, gu.rnd((2, 4, 16, 16), 5, 0.5)), fx["dec"])


def test_unet_toy_ho():
    fx = gu.load("unet_toy_ho")
    with torch.no_grad():
        y = nets.ho_unet(gu.weights(fx), gu.TOY_HO_CFG, gu.rnd((1, 3, 32, 32), 6), torch.tensor([490.0]))
    _close(y, fx["y"])


def test_latent_cycle_tiny():
    """99-step DPM-Encoder + coupled decode (same condition -> cycle closes; target condition with CFG 3)."""
    fx = gu.load("latent_cycle_tiny")
    sd = gu.weights(fx)
    x0, c, uc, c2 = gu.latent_cycle_inputs()
    unet = lambda x, t, cc: nets.openai_unet(sd, gu.TINY_SD_CFG, x, t, cc)
    noises = gu.latent_noise(int(fx["noise_seed"]), x0.shape, 99)
    with torch.no_grad():
        z = torch.stack(samplers.latent_encode(samplers.cfg_model(unet, c, uc, 1.0), x0, 99, 0.1, noises), dim=1)
        assert z.shape == (2, 100, 4, 16, 16)
        # eps extraction divides by sigma in [2e-3, 3e-2]: fp32 noise of eps_hat is amplified up to 500x
        _close(z[:, [0, 1, 50, 99]], fx["z_sub"], atol=5e-3, rtol=1e-3)
        _close(z.flatten(2).norm(dim=2), fx["z_norms"], atol=2e-2, rtol=1e-3)
        x_same = samplers.latent_decode(samplers.cfg_model(unet, c, uc, 1.0), z[:, 0], z[:, 1:], 99, 0.1)
        x_tgt = samplers.latent_decode(samplers.cfg_model(unet, c2, uc, 3.0), z[:, 0], z[:, 1:], 99, 0.1)
    assert (x_same - x0).abs().max().item() < 1e-3  # cycle consistency (reference: fx["cycle_err"])
    assert float(fx["cycle_err"]) < 1e-3
    _close(x_same, fx["x_same"], atol=1e-3)
    _close(x_tgt, fx["x_tgt"], atol=5e-3, rtol=1e-3)


def _c1(fx_name, steps, eta, sample_type):
    fx = gu.load(fx_name)
    sd = gu.weights(fx)
    img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(11))
    x0 = (img - 0.5) * 2.0
    net = lambda x, t: nets.ho_unet(sd, gu.TOY_HO_CFG, x, t)
    enc_noise, last = gu.pixel_noise(int(fx["noise_seed"]), x0.shape, steps)
    betas = samplers.pixel_betas()
    with torch.no_grad():
        z = torch.stack(samplers.pixel_encode(net, x0, betas, steps, steps, eta, enc_noise, sample_type), dim=1)
        _close(z[:, [0, 1, steps // 2, steps - 1]], fx["z_sub"], atol=5e-3, rtol=1e-3)
        _close(z.flatten(2).norm(dim=2), fx["z_norms"], atol=2e-2, rtol=1e-3)
        x = samplers.pixel_decode(net, z, betas, steps, steps, eta, last, sample_type)
    out = (x + 1.0) / 2.0  # post_process Normalize(mean=-1, std=2)
    _close(out, fx["img"], atol=2e-3)
    return out, img


def test_c1_toy_ddpm_ddim_eta():
    out, img = _c1("c1_toy_ddpm", 50, 0.1, "ddim")
    # BASELINE config 1: same-model encode+decode nearly reproduces the image (last step is unconstrained)
    assert (out - img).abs().max().item() < 0.1


def test_c1_toy_ddpm_ddpm_type():
    _c1("c1_toy_ddpm_ddpmtype", 20, None, "ddpm")


def test_ldm_uncond_chain_with_refine():
    """gan_type LatentDiffStochastic on small networks (oracle/gen_golden.py:gen_ldm_uncond): VQ first stage, 49-step
    DPM-Encoder, decode, 10-step eta-1 refinement, VQ decode - the oracle restatement against the reference's
    DDIMSampler / Encoder / Decoder run (the codebook lookup itself is the oracle's on both sides: taming-transformers
    is not installed, see oracle/nets.py:vq_quantize)."""
    import json
    fx = gu.load("ldm_uncond_tiny")
    usd = nets.synth_state_dict(json.loads(str(fx["unet_names"])), int(fx["useed"]))
    vsd = nets.synth_state_dict(json.loads(str(fx["vae_names"])), int(fx["vseed"]))
    S, R = int(fx["steps"]), int(fx["refine_steps"])
    ac = samplers.sd_alphas_cumprod(1000, 0.0015, 0.0195)
    image = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(int(fx["img_seed"])))
    unet = lambda x, t: nets.openai_unet(usd, gu.TINY_LDM_UNCOND_CFG, x, t)
    with torch.no_grad():
        x0 = nets.vae_encode_moments(vsd, gu.TINY_VQ_CFG, (image - 0.5) * 2.0)
        _close(x0, fx["x0"])
        nz = gu.latent_noise(int(fx["noise_seed"]), x0.shape, S)
        z = samplers.latent_encode(unet, x0, S, 0.1, nz, white_box_steps=S + 1, alphas_cumprod=ac)
        zs = torch.stack(z, 1)
        slots = [int(s) for s in fx["z_sub_slots"]]
        _close(zs[:, slots], fx["z_sub"], atol=2e-3, rtol=2e-3)  # eps = (...)/sigma amplifies fp32 order differences
        x_dec = samplers.latent_decode(unet, zs[:, 0], zs[:, 1:], S, 0.1, alphas_cumprod=ac)
        _close(x_dec, fx["x_dec"], atol=2e-4)
        torch.manual_seed(int(fx["refine_seed"]))
        rn = [torch.randn(x0.shape) for _ in range(R + 1)]
        x_ref = samplers.latent_refine(unet, torch.as_tensor(fx["x_dec"]), S, R, rn, alphas_cumprod=ac)
        _close(x_ref, fx["x_ref"], atol=2e-4)
        img = (nets.vae_decode(vsd, gu.TINY_VQ_CFG, nets.vq_quantize(torch.as_tensor(fx["x_ref"]),
                                                                     vsd["quantize.embedding.weight"])) + 1.0) / 2.0
        _close(img, fx["img"], atol=1e-4)
