"""gan_type = LatentDiffStochastic through the drop-in LatentDiffStochasticWrapper (model/gan_wrapper/
latentdiff_stochastic_wrapper.py:175-316) on small networks, against the reference's own CPU run of the same chain
(tests/golden/ldm_uncond_tiny.npz, oracle/gen_golden.py:gen_ldm_uncond): VQ first stage -> DPM-Encoder (49 steps,
eta 0.1, the 0.0015..0.0195 schedule of the celeba256 / ffhq256 LDMs) -> decode -> DDIMSampler.refine (10 steps,
eta 1) -> nearest-codebook decode. Identical weights, image and CPU-drawn noise."""
import json
import warnings

import pytest
import torch

import cycle_diffusion_amd as cda
import golden_util as gu
from cycle_diffusion_amd import _ffi
from cycle_diffusion_amd.gan_wrapper.latent_wrapper import LatentDiffStochasticWrapper
from oracle import nets

pytestmark = pytest.mark.gpu
FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0


def tiny_uncond_unet_desc(precision=_ffi.CD_PREC_16):
    d = cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=16, in_channels=3, out_channels=3, model_channels=32,
                      num_res_blocks=1, channel_mult=(1, 2, 3), attn=(2, 4), num_head_channels=32)
    d.precision = precision
    return d


def tiny_vq_desc():
    return cda.make_desc(_ffi.CD_NET_VAE_KL, image_size=0, in_channels=3, out_channels=3, model_channels=32,
                         num_res_blocks=1, channel_mult=(1, 2, 4), z_channels=3, embed_dim=3, double_z=False, n_embed=256)


def _wrapper(fx, refine_steps, precision=_ffi.CD_PREC_16):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = LatentDiffStochasticWrapper("celeba256", custom_steps=int(fx["steps"]), eta=0.1,
                                        white_box_steps=int(fx["steps"]) + 1, refine_steps=refine_steps, noise_on_cpu=True,
                                        unet_desc=tiny_uncond_unet_desc(precision), vae_desc=tiny_vq_desc(),
                                        allow_lossy_16bit=True)
    usd = nets.synth_state_dict(json.loads(str(fx["unet_names"])), int(fx["useed"]))
    vsd = nets.synth_state_dict(json.loads(str(fx["vae_names"])), int(fx["vseed"]))
    for net, sd in ((w.unet, usd), (w.vae, vsd)):
        n, first = w.engine.load_state_dict(net, sd)
        assert n == 0, first
        assert set(k for k, _ in w.engine.net_params(net)) == set(sd.keys())
    return w, vsd


def test_vq_first_stage_vs_oracle(engine, report):
    """VQModelInterface.encode / decode (autoencoder.py:264-282): quant_conv output without sampling; decode snaps
    every latent vector to its nearest codebook row first."""
    fx = gu.load("ldm_uncond_tiny")
    w, vsd = _wrapper(fx, 0)
    image = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(int(fx["img_seed"])))
    x0 = w.engine.vae_encode(w.vae, ((image - 0.5) * 2.0).cuda(), sample=False, scale=1.0).cpu()
    ref = torch.as_tensor(fx["x0"])
    r_enc = ((x0 - ref).abs().max() / ref.abs().max()).item()
    # decode of the REFERENCE latent: same codebook rows must be chosen (ties aside), so images agree to 16-bit rounding
    xr = torch.as_tensor(fx["x_ref"])
    img = (w.engine.vae_decode(w.vae, xr.cuda(), scale=1.0).cpu() + 1.0) / 2.0
    p = gu.psnr(img, torch.as_tensor(fx["img"]))
    report.add("ldm_uncond/vq_first_stage", enc_rel_to_max=r_enc, dec_psnr_db=p)
    assert r_enc < 8e-3 * FMT, r_enc
    assert p >= 40.0, p


@pytest.mark.parametrize("prec", [_ffi.CD_PREC_F32X3, _ffi.CD_PREC_16], ids=["fp32x3", "16bit"])
def test_latentdiff_stochastic_wrapper_vs_reference(report, prec):
    """`fp32x3` is the wrapper's default for these eta-0.1 chains (latent_wrapper.py); the 16-bit engine is the lossy opt-in:
    its latent error (5e-3 .. 2e-2 of the range, chaotic from build to build) flips a handful of codebook cells."""
    if prec == _ffi.CD_PREC_F32X3 and FMT != 1.0:
        pytest.skip("the split mode needs the fp16 build")
    x3 = prec == _ffi.CD_PREC_F32X3
    fx = gu.load("ldm_uncond_tiny")
    S, R = int(fx["steps"]), int(fx["refine_steps"])
    image = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(int(fx["img_seed"])))
    w, vsd = _wrapper(fx, 0, prec)
    assert w.precision == ("fp32x3" if x3 else "fp16")
    assert w.resolution == 64 and w.latent_dim == 16 * 16 * 3 * (S + 1)
    torch.manual_seed(int(fx["noise_seed"]))
    with torch.no_grad():
        z = w.encode(image.cuda())
        img0 = w(z)
    assert z.shape == (1, w.latent_dim)
    z5 = z.view(1, S + 1, 3, 16, 16).cpu()
    slots = [int(s) for s in fx["z_sub_slots"]]
    zref = torch.as_tensor(fx["z_sub"])
    xT = (z5[:, 0] - zref[:, 0]).abs().max().item()
    eps_rel = [((z5[:, s] - zref[:, i]).abs().max() / zref[:, i].abs().max()).item() for i, s in enumerate(slots) if s]
    p0 = gu.psnr(img0, torch.as_tensor(fx["img_norefine"]))
    # the decoded latent itself (before the codebook lookup), against the reference's
    from cycle_diffusion_amd import schedule
    sch = schedule.DDIMSchedule(w.alphas_cumprod, S, 0.1)
    x_dec = w.engine.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, z.view(1, S + 1, 3, 16, 16).contiguous(), sch.coef_decode()).cpu()
    xd_ref = torch.as_tensor(fx["x_dec"])
    lat_rel = ((x_dec - xd_ref).abs().max() / xd_ref.abs().max()).item()
    # refinement on the same z: re-noise + 10 random eta-1 steps with the reference's draws
    w.refine_steps = R
    torch.manual_seed(int(fx["refine_seed"]))
    with torch.no_grad():
        img1 = w(z)
    p1 = gu.psnr(img1, torch.as_tensor(fx["img"]))
    # ---- where the unrefined image differs, cell by cell: which latent vectors picked another codebook row, how close
    # to a cell boundary the reference's own latent was there, and the PSNR away from those 4 x 4 pixel patches
    cb = vsd["quantize.embedding.weight"]

    def cells(x):
        zf = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
        return (zf ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1) - 2 * zf @ cb.t()

    d_ref, d_eng = cells(xd_ref), cells(x_dec)
    i_ref, i_eng = d_ref.argmin(1), d_eng.argmin(1)
    flipped = i_ref != i_eng
    srt = d_ref.sort(1).values
    margin = srt[:, 1] - srt[:, 0]  # distance gap between the reference latent's best and second-best rows
    delta = (x_dec - xd_ref).permute(0, 2, 3, 1).reshape(-1, 3).norm(dim=1)
    for c in flipped.nonzero().flatten().tolist():
        a, b = int(i_ref[c]), int(i_eng[c])
        # a flip needs d(z, b) - d(z, a) <= 2 |delta| |e_b - e_a| (Cauchy-Schwarz): the engine's row is the true nearest
        # row of ITS latent, and the reference's latent sat within the latent error of that boundary
        assert float(d_ref[c, b] - d_ref[c, a]) <= 2.0 * float(delta[c]) * float((cb[b] - cb[a]).norm()) * 1.01 + 1e-7
    keep = ~flipped.view(1, 1, 16, 16)
    keep = ~(torch.nn.functional.max_pool2d((~keep).float(), 3, 1, 1) > 0)  # also drop the neighbours of a flipped cell
    mask = keep.float().repeat_interleave(4, 2).repeat_interleave(4, 3).expand(1, 3, 64, 64).bool()
    ref0 = torch.as_tensor(fx["img_norefine"]).clamp(0, 1)
    p0_away = float(-10 * torch.log10(((img0.cpu().clamp(0, 1) - ref0)[mask] ** 2).mean()))
    report.add("ldm_uncond/wrapper" + ("_fp32x3" if x3 else ""), xT_maxabs=xT, eps_rel=eps_rel, latent_rel_to_max=lat_rel, psnr_norefine_db=p0,
               psnr_refined_db=p1, flipped_cells=int(flipped.sum()), cells=int(flipped.numel()),
               flipped_margin_max=float(margin[flipped].max()) if flipped.any() else 0.0,
               margin_median=float(margin.median()), psnr_norefine_away_from_flipped_cells_db=p0_away)
    # a few cells flip at most, each of them one whose reference latent was (far) closer to a boundary than typical. The
    # 16-bit count is chaotic: the same chain gave 5 flips (latent 1e-3) on the round-3 build, 5 (5e-3) and 13 (1.6e-2) on two
    # round-4 builds whose kernels differ only in fp32 summation order - the eta-0.1 decode amplifies the last bits of eps_hat
    assert int(flipped.sum()) <= (4 if x3 else 24 * FMT), int(flipped.sum())  # split mode: 2, margins 8e-5 of a 6e-3 median
    assert (not flipped.any()) or float(margin[flipped].max()) < 0.5 * float(margin.median())
    # the decoder's mid-block attention spreads a flipped cell's change thinly over the whole image: measured 30-32.5 dB
    # away from the flipped patches on the 16-bit engine
    # (split mode: 45.8 dB with 2 flipped cells on the builds of rounds 4-6; 38.2 dB with 3 on the build whose GEMM epilogue forms
    # its GroupNorm sums of squares by fma - latent error 1.9e-3 instead of 2.1e-3, the third cell's margin 1.5e-4 of a 6e-3 median)
    nflip = int(flipped.sum())
    assert p0_away >= ((42.0 if nflip <= 2 else 36.0) if x3 else 26.0), (p0_away, nflip)
    # (the 16-bit VQ encoder's x0 differs by 2e-3 in both modes; the split-mode eps slots go 2e-6 -> 3e-4 along the chain)
    assert xT < (1e-4 if x3 else 2e-2 * FMT) and max(eps_rel) < (2e-3 if x3 else 2e-2 * FMT), (xT, eps_rel)
    assert lat_rel < (1e-2 if x3 else 4e-2 * FMT), lat_rel  # split mode: 2e-3
    # Images: a latent vector that lands on the other side of a codebook cell boundary swaps its codebook row and changes a
    # 4 x 4 pixel patch, so the whole-image floor of the 16-bit engine is looser than for the KL first stage (measured 28.8
    # - 31.8 dB without / 29.6 - 52.2 dB with refinement on this 256-row codebook over three builds)
    # split mode: 43.1 / 66.8 dB with 2 flipped cells, 35.4 / 66.9 dB with 3 (each flipped cell is a visibly different 4 x 4 patch)
    assert p0 >= ((38.0 if nflip <= 2 else 33.0) if x3 else 22.0) and p1 >= (45.0 if x3 else 22.0), (p0, p1, nflip)


@pytest.mark.parametrize("prec", [_ffi.CD_PREC_16, _ffi.CD_PREC_F32, _ffi.CD_PREC_F32X3], ids=["16bit", "fp32", "fp32x3"])
def test_ldm_uncond_unet_at_full_size_encode_decode_refine_vs_reference(engine, report, prec):
    """The celeba256 / ffhq256 U-Net at its real size (224 channels, mult (1, 2, 3, 4), legacy-order AttentionBlocks at
    three levels) through the three sampler stages of LatentDiffStochasticWrapper - DPM-Encoder, decode with the
    injected eps, DDIMSampler.refine (eta 1) - on a 3 x 64 x 64 latent, against tests/golden/ldm_uncond_full_latent.npz
    (oracle/gen_golden_full.py:gen_ldm_uncond_full: the reference's DDIMSampler, 99 / 100 / 40 steps = the cfg's 999 /
    1000 / 400 divided by ~10). Same engine calls, schedules and draw order as the wrapper's encode() / generate();
    the VQ first stage is not part of this fixture (its lookup is unpinned)."""
    import os
    import numpy as np
    from cycle_diffusion_amd import schedule
    from cycle_diffusion_amd.engine import ldm_uncond_unet_desc
    path = os.path.join(gu.GOLD, "ldm_uncond_full_latent.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    fx = np.load(path, allow_pickle=False)
    S, R = int(fx["steps"]), int(fx["refine_steps"])
    if prec == _ffi.CD_PREC_F32X3 and FMT != 1.0:
        pytest.skip("the split mode needs the fp16 build")
    d = ldm_uncond_unet_desc()
    d.precision = prec
    net = engine.create_net(d)
    sd = nets.synth_state_dict(json.loads(str(fx["unet_names"])), int(fx["useed"]))
    n, first = engine.load_state_dict(net, sd)
    assert n == 0, first
    assert set(k for k, _ in engine.net_params(net)) == set(sd.keys())
    ac = schedule.latent_alphas_cumprod(1000, 0.0015, 0.0195)
    sch = schedule.DDIMSchedule(ac, S, 0.1)
    x0 = torch.as_tensor(fx["x0"])
    torch.manual_seed(int(fx["noise_seed"]))  # randn_like(x0), then one draw per sample_xt_next (latent_wrapper.encode)
    nz = torch.stack([torch.randn(x0.shape) for _ in range(len(sch))], 0)
    z = engine.dpm_encode(net, _ffi.CD_SCHED_DDIM, x0.cuda(), sch.coef_encode(), noise=nz.cuda(), last_uses_x0=True)
    z5 = z.view(1, S + 1, 3, 64, 64)
    x_dec = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z5.contiguous(), sch.coef_decode())
    rs = schedule.DDIMSchedule(ac, S, 1.0)  # convsample_ddim refines with eta = 1 (latentdiff_stochastic_wrapper.py:70-78)
    torch.manual_seed(int(fx["refine_seed"]))
    nz2 = torch.stack([torch.randn(x0.shape) for _ in range(R + 1)], 0)
    x_ref = engine.pix_refine(net, _ffi.CD_SCHED_DDIM, x_dec, rs.coef_refine(R), noise=nz2.cuda())
    zc = z5.cpu()
    slots = [int(s) for s in fx["z_sub_slots"]]
    zref = torch.as_tensor(fx["z_sub"])
    xT = (zc[:, 0] - zref[:, 0]).abs().max().item()
    eps_rel = [((zc[:, s] - zref[:, i]).abs().max() / zref[:, i].abs().max()).item() for i, s in enumerate(slots) if s]
    zn_ref = torch.as_tensor(fx["z_norms"])
    zn_rel = ((zc.flatten(2).norm(dim=2) - zn_ref).abs() / zn_ref).max().item()

    def rel(a, b):
        b = torch.as_tensor(b)
        return ((a.cpu() - b).abs().max() / b.abs().max()).item(), float(-10 * torch.log10(((a.cpu() - b) ** 2).mean() / (b ** 2).mean()))

    (d_rel, d_snr), (r_rel, r_snr) = rel(x_dec, fx["x_dec"]), rel(x_ref, fx["x_ref"])
    tag = {_ffi.CD_PREC_16: "", _ffi.CD_PREC_F32: "_fp32", _ffi.CD_PREC_F32X3: "_fp32x3"}[prec]
    report.add("ldm_uncond/full_size_latent" + tag, xT_maxabs=xT, eps_rel=eps_rel, z_norm_rel=zn_rel, x_dec_rel_to_max=d_rel,
               x_dec_snr_db=d_snr, x_refined_rel_to_max=r_rel, x_refined_snr_db=r_snr,
               reference_cpu_seconds=float(fx["cpu_seconds"]))
    assert xT < 1e-4 and zn_rel < 2e-3 * FMT and max(eps_rel) < 5e-2 * FMT, (xT, zn_rel, eps_rel)
    # latents: signal-to-error ratio (the latent is not an image in [0, 1]). The eta-0.1 decode is nearly deterministic
    # and amplifies a perturbation of eps_hat on this random-init network: the 16-bit engine, whose encode agrees to
    # 3-6e-4 per slot, ends 99 steps later at 25 dB (5 % rms); the fp32 path and its split mode follow the reference
    floor = 20.0 if prec == _ffi.CD_PREC_16 else 60.0
    if FMT != 1.0 and prec == _ffi.CD_PREC_16:
        floor = 8.0
    assert d_snr >= floor and r_snr >= floor, (d_snr, r_snr)


def test_ema_shadow_weights_are_what_the_unet_runs_on():
    """A pl-style checkpoint whose EMA shadow differs from the raw U-Net weights: the wrapper must evaluate the
    shadow (use_ema defaults to True for celeba256 / ffhq256; latentdiff_stochastic_wrapper.py:116-164)."""
    fx = gu.load("ldm_uncond_tiny")
    usd = nets.synth_state_dict(json.loads(str(fx["unet_names"])), int(fx["useed"]))
    vsd = nets.synth_state_dict(json.loads(str(fx["vae_names"])), int(fx["vseed"]))
    ckpt = {"first_stage_model." + k: v for k, v in vsd.items()}
    for k, v in usd.items():
        ckpt["model.diffusion_model." + k] = torch.zeros_like(v)  # raw weights: all zero (eps would be 0)
        ckpt["model_ema." + ("diffusion_model." + k).replace(".", "")] = v
    ckpt["model_ema.decay"], ckpt["model_ema.num_updates"] = torch.tensor(0.9999), torch.tensor(1)
    kw = dict(custom_steps=int(fx["steps"]), eta=0.1, white_box_steps=int(fx["steps"]) + 1, noise_on_cpu=True,
              unet_desc=tiny_uncond_unet_desc(), vae_desc=tiny_vq_desc(), allow_lossy_16bit=True)
    w_ema = LatentDiffStochasticWrapper("celeba256", state_dict=ckpt, **kw)
    w_ref, _ = _wrapper(fx, 0)
    x = gu.rnd((1, 3, 16, 16), 5).cuda()
    t = torch.tensor([500], dtype=torch.int64).cuda()
    a, b = w_ema.engine.unet_forward(w_ema.unet, x, t), w_ref.engine.unet_forward(w_ref.unet, x, t)
    assert torch.equal(a, b) and float(a.abs().max()) > 0
    with pytest.raises(KeyError):
        LatentDiffStochasticWrapper("celeba256", state_dict={k: v for k, v in ckpt.items()
                                                            if not k.startswith("model_ema.")}, **kw)


def test_latentdiff_wrapper_white_box_prefix_is_the_truncated_chain():
    """`white_box_steps` below the chain + 1 (latentdiff ddim.py:484: the DPM-Encoder loop breaks after white_box_steps - 1
    steps; :436: decode steps beyond the list draw fresh noise). The engine runs it on a truncated coefficient table, so with
    the same draws (a) the short z is, bit for bit, the head of the full chain's z and (b) decoding it equals decoding the
    full-length list whose missing slots hold the fresh-noise tensors (an injected eps and a drawn noise enter a step the
    same way: sigma_t * eps, ddim.py:640-643)."""
    fx = gu.load("ldm_uncond_tiny")
    S, wb = int(fx["steps"]), 21
    image = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(int(fx["img_seed"])))
    w_full, _ = _wrapper(fx, 0)
    w_short, _ = _wrapper(fx, 0)
    w_short.white_box_steps = wb
    w_short.latent_dim = 16 * 16 * 3 * wb
    with torch.no_grad():
        torch.manual_seed(11)
        z_full = w_full.encode(image.cuda()).view(1, S + 1, 3, 16, 16)
        torch.manual_seed(11)
        z_short = w_short.encode(image.cuda())
        assert z_short.shape == (1, w_short.latent_dim)
        assert torch.equal(z_short.view(1, wb, 3, 16, 16), z_full[:, :wb])
        torch.manual_seed(12)
        img_short = w_short(z_short)
        torch.manual_seed(12)
        tail = torch.stack([torch.randn(1, 3, 16, 16) for _ in range(S + 1 - wb)], 1).cuda()
        img_cat = w_full(torch.cat([z_short.view(1, wb, 3, 16, 16), tail], dim=1).reshape(1, -1))
    assert torch.isfinite(img_short).all() and torch.equal(img_short, img_cat)
