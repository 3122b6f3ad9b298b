"""Network- and sampler-level parity on the GPU: the HIP engine (through the C ABI) against the CPU
oracle and against the committed reference fixtures, on identical weights / inputs / noise.

Tolerance model: the 16-bit engine stores activations in fp16 (10-bit mantissa; bf16 is a build switch) and
accumulates in fp32, so a network output differs from the fp32 reference by a few 1e-3 of its scale; the fp32 path
(CD_PREC_F32, pixel-space DDPMs) differs by fp32 round-off only. Bounds below are REL (max|err|/max|ref|) and PSNR
in image space (evaluation/utils.py:60-67 convention)."""
import numpy as np
import pytest
import torch

import cycle_diffusion_amd as cda
import golden_util as gu
from cycle_diffusion_amd import _ffi, schedule
from oracle import nets, samplers

pytestmark = pytest.mark.gpu

# Bounds are ~3x the errors measured on MI355X with the default fp16 storage (gpurun_out/parity_report.json:
# networks 1.1e-3..2.7e-3 of the output range); the bf16 build (CD_ACT_FP16=0) has 8x the rounding step.
FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0
NET_REL = 8e-3 * FMT
NET_MEAN = 8e-3 * FMT


def _stats(got, ref):
    got, ref = got.detach().float().cpu(), torch.as_tensor(np.asarray(ref)).float()
    d = (got - ref).abs()
    return dict(rel_to_max=d.max().item() / (ref.abs().max().item() + 1e-12),
                mean_rel=d.mean().item() / (ref.abs().mean().item() + 1e-12), finite=bool(torch.isfinite(got).all()))


def _check(report, name, got, ref, rel=NET_REL, mean=NET_MEAN):
    st = _stats(got, ref)
    report.add(name, **st)
    assert st["finite"] and st["rel_to_max"] < rel and st["mean_rel"] < mean, (name, st)


def tiny_sd_desc():
    return cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=16, in_channels=4, out_channels=4, model_channels=64,
                         num_res_blocks=1, channel_mult=(1, 2), attn=(1, 2), num_heads=2,
                         use_spatial_transformer=True, context_dim=64)


def tiny_iddpm_desc():
    return cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=32, in_channels=3, out_channels=6, model_channels=32,
                         num_res_blocks=1, channel_mult=(1, 2, 2), attn=(2,), num_heads=4, num_head_channels=32,
                         use_scale_shift_norm=True, resblock_updown=True)


def tiny_vae_desc():
    return cda.make_desc(_ffi.CD_NET_VAE_KL, image_size=0, in_channels=3, out_channels=3, model_channels=32,
                         num_res_blocks=1, channel_mult=(1, 2, 4), z_channels=4, embed_dim=4, double_z=True)


def _load(engine, desc, fx):
    net = engine.create_net(desc)
    sd = gu.weights(fx)
    n, first = engine.load_state_dict(net, sd)
    assert n == 0, first
    # the reference's names are the loading contract: nothing in the state_dict may be left unused
    assert set(k for k, _ in engine.net_params(net)) == set(sd.keys())
    return net, sd


def test_unet_tiny_sd_vs_reference_fixture(engine, report):
    fx = gu.load("unet_tiny_sd")
    net, _ = _load(engine, tiny_sd_desc(), fx)
    x, t, ctx = gu.tiny_sd_inputs()
    y = engine.unet_forward(net, x.cuda(), t.float().cuda(), ctx.cuda())
    _check(report, "net/unet_tiny_sd", y, fx["y"])


@pytest.mark.parametrize("prec", [_ffi.CD_PREC_F32, _ffi.CD_PREC_F32X3], ids=["fp32", "fp32x3"])
def test_unet_tiny_sd_fp32_modes_vs_reference_fixture(engine, report, prec):
    """The text-conditioned U-Net (ResBlocks + SpatialTransformer blocks, attention.py:152-261) in the reference's own
    arithmetic - `precision = "full"`, stable_diffusion_stochastic_text_wrapper.py:117: fp32 storage, fp32 LayerNorm, fp32
    flash attention (st_f32.hip), exact-erf GEGLU - and in its split-fp16 mode, against the same reference fixture as the
    16-bit engine (whose bound is 8e-3)."""
    if prec == _ffi.CD_PREC_F32X3 and FMT != 1.0:
        pytest.skip("the split mode needs the fp16 build")
    fx = gu.load("unet_tiny_sd")
    d = tiny_sd_desc()
    d.precision = prec
    net, _ = _load(engine, d, fx)
    x, t, ctx = gu.tiny_sd_inputs()
    y = engine.unet_forward(net, x.cuda(), t.float().cuda(), ctx.cuda())
    _check(report, "net/unet_tiny_sd_" + ("fp32" if prec == _ffi.CD_PREC_F32 else "fp32x3"), y, fx["y"], rel=5e-5, mean=5e-5)
    y2 = engine.unet_forward(net, x.cuda(), t.float().cuda(), ctx.cuda())
    assert torch.equal(y, y2)  # no tuner, no split-K on this path: bit-reproducible


def test_unet_tiny_iddpm_vs_reference_fixture(engine, report):
    fx = gu.load("unet_tiny_iddpm")
    net, _ = _load(engine, tiny_iddpm_desc(), fx)
    y = engine.unet_forward(net, gu.rnd((2, 3, 32, 32), 3).cuda(), torch.tensor([3.0, 700.0]).cuda())
    _check(report, "net/unet_tiny_iddpm", y, fx["y"])


def test_unet_toy_ho_vs_reference_fixture(engine, report):
    fx = gu.load("unet_toy_ho")
    net, _ = _load(engine, cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,)), fx)
    y = engine.unet_forward(net, gu.rnd((1, 3, 32, 32), 6).cuda(), torch.tensor([490.0]).cuda())
    _check(report, "net/unet_toy_ho", y, fx["y"])


def test_vae_tiny_vs_reference_fixture(engine, report):
    fx = gu.load("vae_tiny")
    net, sd = _load(engine, tiny_vae_desc(), fx)
    img = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(4)) * 2 - 1
    # encode: mode() * scale == mean half of the moments
    z = engine.vae_encode(net, img.cuda(), sample=False, scale=1.0)
    _check(report, "net/vae_tiny_mean", z, fx["moments"][:, :4])
    # encode with injected posterior noise (SD samples the posterior, ddpm.py:538)
    nz = gu.rnd((2, 4, 16, 16), 77)
    zs = engine.vae_encode(net, img.cuda(), noise=nz.cuda(), sample=True, scale=0.18215)
    ref = nets.posterior_sample(torch.as_tensor(fx["moments"]), nz) * 0.18215
    _check(report, "net/vae_tiny_sample", zs, ref)
    zz = gu.rnd((2, 4, 16, 16), 5, 0.5)
    dec = engine.vae_decode(net, zz.cuda(), scale=1.0)
    _check(report, "net/vae_tiny_decode", dec, fx["dec"])


def test_latent_cycle_tiny(engine, report):
    """DPM-Encoder (99 steps, eta 0.1, enc scale 1) + decode under the same and under a target condition
    with CFG 3 — the C2 [gan] settings on a small network, identical noise as the reference run."""
    fx = gu.load("latent_cycle_tiny")
    net, sd = _load(engine, tiny_sd_desc(), fx)
    x0, c, uc, c2 = gu.latent_cycle_inputs()
    K = 99
    noises = torch.stack(gu.latent_noise(int(fx["noise_seed"]), x0.shape, K), 0)  # [K, B, C, H, W]
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), 99, 0.1)
    z = engine.dpm_encode(net, _ffi.CD_SCHED_DDIM, x0.cuda(), sch.coef_encode(), ctx_c=c.cuda(), ctx_uc=uc.cuda(),
                          guidance=1.0, noise=noises.cuda())
    assert z.shape == (2, 100, 4, 16, 16)
    zc = z.cpu()
    # x_T is pure scheduler math on identical noise: exact
    assert torch.equal(zc[:, 0], torch.as_tensor(fx["z_sub"][:, 0]))
    # extracted eps, value by value, on the slots the fixture carries (1, 50, 99): eps = (...)/sigma divides the
    # 16-bit eps_hat error by sigma_t, so the bound is relative to each slot's own scale
    zr = torch.as_tensor(fx["z_sub"][:, 1:])
    zs = zc[:, [1, 50, 99]]
    eps_rel = ((zs - zr).flatten(2).abs().max(dim=2).values / zr.flatten(2).abs().max(dim=2).values).max(dim=0).values
    report.add("sampler/latent_eps_slots", rel_slot1=float(eps_rel[0]), rel_slot50=float(eps_rel[1]),
               rel_slot99=float(eps_rel[2]))
    assert (eps_rel < 6e-3 * FMT).all(), eps_rel  # measured 0.4e-3 .. 0.9e-3 at full size (test_gpu_e2e_fullsize.py)
    zn = zc.flatten(2).norm(dim=2)
    rel = ((zn - torch.as_tensor(fx["z_norms"])).abs() / torch.as_tensor(fx["z_norms"])).max().item()
    report.add("sampler/latent_z_norm_rel", rel=rel)
    assert rel < 1e-3 * FMT  # measured 2.8e-5
    coef_d = sch.coef_decode()
    x_same = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z, coef_d, ctx_c=c.cuda(), ctx_uc=uc.cuda(), guidance=1.0)
    cyc = (x_same.cpu() - x0).abs().max().item()
    report.add("sampler/latent_cycle_maxabs", err=cyc, reference=float(fx["cycle_err"]))
    # the engine's U-Net is deterministic, so its own cycle closes to fp32 round-off of the scheduler math
    assert cyc < 8e-3 * FMT, cyc  # measured 2.2e-3 .. 3.0e-3 across runs (tile choices vary the rounding)
    x_tgt = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z, coef_d, ctx_c=c2.cuda(), ctx_uc=uc.cuda(), guidance=3.0)
    _check(report, "sampler/latent_x_tgt", x_tgt, fx["x_tgt"], rel=3e-2 * FMT, mean=1.5e-2 * FMT)  # measured 6e-3 / 3e-3


def _c1(engine, report, fx_name, steps, eta, sample_type, precision=_ffi.CD_PREC_16):
    fx = gu.load(fx_name)
    net, sd = _load(engine, cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=precision), fx)
    f32 = precision == _ffi.CD_PREC_F32
    tag = fx_name + ("_f32" if f32 else "")
    img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(11))
    x0 = (img - 0.5) * 2.0
    enc_noise, last = gu.pixel_noise(int(fx["noise_seed"]), x0.shape, steps)
    sch = schedule.PixelSchedule(steps, steps, sample_type=sample_type, eta=eta)
    z = engine.dpm_encode(net, sch.kind, x0.cuda(), sch.coef_encode(), noise=torch.stack(enc_noise, 0).cuda(),
                          last_uses_x0=False)
    assert z.shape == (1, steps, 3, 32, 32)
    assert torch.allclose(z[:, 0].cpu(), torch.as_tensor(fx["z_sub"][:, 0]), atol=1e-6)
    # extracted eps vs the reference's: the fp16 eps_hat error is divided by c1 (sigma_t), so the
    # bound is relative to the eps scale; the last slot has the smallest sigma (largest gain)
    zs = z.cpu()[:, [1, steps // 2, steps - 1]]
    zr = torch.as_tensor(fx["z_sub"][:, 1:])
    zerr = ((zs - zr).flatten(2).abs().max(dim=2).values / zr.flatten(2).abs().max(dim=2).values)[0]
    report.add("sampler/" + tag + "_z", rel_first=float(zerr[0]), rel_mid=float(zerr[1]), rel_last=float(zerr[2]))
    if f32:  # fp32 round-off, amplified by the chain towards the last slots
        assert zerr[0] < 1e-4 and zerr[1] < 1e-3 and zerr[2] < 1e-2, zerr
    else:
        assert zerr[0] < 6e-3 * FMT and zerr[1] < 6e-3 * FMT and zerr[2] < 2e-2 * FMT, zerr  # measured ~1e-3
    x = engine.ddim_decode(net, sch.kind, z, sch.coef_decode(), n_eps=steps - 1, noise_tail=last[None].cuda())
    out = (x.cpu() + 1.0) / 2.0
    p_ref = gu.psnr(out, torch.as_tensor(fx["img"]))
    p_img = gu.psnr(out, img)
    report.add("sampler/" + tag, psnr_vs_reference=p_ref, psnr_vs_input=p_img,
               ref_psnr_vs_input=gu.psnr(torch.as_tensor(fx["img"]), img))
    return p_ref, p_img


def test_guided_call_with_the_ddpm_posterior_form_is_refused(engine):
    """The 'ddpm' posterior step kernels have no classifier-free-guidance combine (the pixel DDPMs are unconditional,
    ddpm_ddim_wrapper.py:230-238): a guided call - one scale or one per sample - must raise, not run unguided."""
    fx = gu.load("latent_cycle_tiny")
    net, _sd = _load(engine, tiny_sd_desc(), fx)
    x0, c, uc, _c2 = gu.latent_cycle_inputs()
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), 4, 0.1)
    z = torch.zeros((2, 5, 4, 16, 16)).cuda()
    for g in (3.0, [2.0, 3.0]):
        with pytest.raises(RuntimeError, match="CD_SCHED_DDIM"):
            engine.ddim_decode(net, _ffi.CD_SCHED_DDPM, z, sch.coef_decode(), ctx_c=c.cuda(), ctx_uc=uc.cuda(), guidance=g)
    with pytest.raises(RuntimeError, match="CD_SCHED_DDIM"):
        engine.dpm_encode(net, _ffi.CD_SCHED_DDPM, x0.cuda(), sch.coef_encode(), ctx_c=c.cuda(), ctx_uc=uc.cuda(),
                          guidance=3.0)
    x = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z, sch.coef_decode(), ctx_c=c.cuda(), ctx_uc=uc.cuda(), guidance=[2.0, 3.0])
    assert torch.isfinite(x).all()  # the engine is usable after the refusals


def test_c1_toy_ddpm_ddim_eta_fp32(engine, report):
    """BASELINE config 1 ('ddim', eta 0.1, 50 + 50 steps) on the engine's fp32 path vs the reference's CPU run:
    image PSNR >= 40 dB against the reference image (whose own PSNR against the input is 48 dB), and the result is
    bit-identical from run to run (fixed tiles, fixed accumulation order)."""
    p_ref, p_img = _c1(engine, report, "c1_toy_ddpm", 50, 0.1, "ddim", precision=_ffi.CD_PREC_F32)
    assert p_ref >= 40.0, p_ref
    assert p_img >= 40.0, p_img


def test_c1_toy_ddpm_ddim_eta_16bit(engine, report):
    """The same chain on the 16-bit engine: reported, and held only to a liveness floor (see the comment)."""
    p_ref, p_img = _c1(engine, report, "c1_toy_ddpm", 50, 0.1, "ddim")
    # The 'ddim' chain of the DDPM linear schedule rescales x by sqrt(abar_{t-1}/abar_t) every step
    # (x130 end to end) and, on a RANDOM-INIT network, nothing damps it: the fp32 reference closes the
    # cycle only through exact cancellation, a 16-bit engine cannot (DESIGN.md "Numerics"). What is
    # pinned here: x_T exact, every extracted eps within the bound above, finite output; the image-space
    # PSNR is reported, and held to a floor that catches gross breakage only. The well-conditioned
    # variants (sample_type='ddpm' below, the latent SD chain above) are held to tight bounds.
    # measured 14-21 dB from run to run (the tile / split-K choices of the autotuner change the fp32 summation
    # order, and this chain amplifies any difference ~130x): a floor for gross breakage, not a parity bound
    assert p_ref > 9.0, p_ref
    assert p_img > 9.0, p_img


def test_c1_toy_ddpm_ddpm_type(engine, report):
    p_ref, _ = _c1(engine, report, "c1_toy_ddpm_ddpmtype", 20, None, "ddpm")
    assert p_ref > (60.0 if FMT == 1.0 else 30.0), p_ref  # measured 99.5 dB (fp16)
