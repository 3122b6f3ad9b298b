"""The engine's fp32 path (cd_net_desc.precision = CD_PREC_F32: fp32 NHWC activations, fp32 weights,
v_mfma_f32_32x32x2_f32) - what the pixel-space wrapper runs by default, because the reference's 'ddim' chain
(ddpm_ddim_wrapper.py:283-307,114-227) is only reproducible with eps_hat at fp32 resolution.

Networks against the reference fixtures and the oracle (fp32 round-off level), then the chains the reference
runs with them: BASELINE config 1 with refinement, and a reduced C5-shaped chain (improved-DDPM architecture,
custom_steps 100 / es_steps 85 / refine_steps 10 = the reference AFHQ cfg divided by 10) through the drop-in
DDPMDDIMWrapper API, all on identical weights / images / CPU-drawn noise as the reference's own CPU run."""
import warnings

import pytest
import torch

import cycle_diffusion_amd as cda
import golden_util as gu
from cycle_diffusion_amd import _ffi
from cycle_diffusion_amd.gan_wrapper.ddpm_ddim_wrapper import DDPMDDIMWrapper
from oracle import nets

pytestmark = pytest.mark.gpu
F32 = _ffi.CD_PREC_F32
F32X3 = _ffi.CD_PREC_F32X3  # the same network, GroupNorm-fed convolutions as three-term split-fp16 products
BOTH = pytest.mark.parametrize("prec", [F32, F32X3], ids=["fp32", "fp32x3"])


def _rel(got, ref):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    d = (got - ref).abs()
    return d.max().item() / (ref.abs().max().item() + 1e-12), d.mean().item() / (ref.abs().mean().item() + 1e-12)


def _load(engine, desc, fx):
    net = engine.create_net(desc)
    sd = gu.weights(fx)
    n, first = engine.load_state_dict(net, sd)
    assert n == 0, first
    return net, sd


def tiny_iddpm_desc(precision=F32):
    return cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=32, in_channels=3, out_channels=6, model_channels=32,
                         num_res_blocks=1, channel_mult=(1, 2, 2), attn=(2,), num_heads=4, num_head_channels=32,
                         use_scale_shift_norm=True, resblock_updown=True, precision=precision)


@BOTH
def test_f32_networks_vs_reference_fixtures(engine, report, prec):
    """improved-DDPM (FiLM, resblock up/down, legacy attention) and Ho-DDPM (asymmetric-pad downsample, concat
    skips, single-head attention) forwards against the reference modules' outputs: fp32 round-off only."""
    fx = gu.load("unet_tiny_iddpm")
    net, _ = _load(engine, tiny_iddpm_desc(prec), fx)
    x, t = gu.rnd((2, 3, 32, 32), 3).cuda(), torch.tensor([3.0, 700.0]).cuda()
    y = engine.unet_forward(net, x, t)
    r1 = _rel(y, fx["y"])
    assert torch.equal(y, engine.unet_forward(net, x, t))  # deterministic
    fx2 = gu.load("unet_toy_ho")
    net2, _ = _load(engine, cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=prec), fx2)
    y2 = engine.unet_forward(net2, gu.rnd((1, 3, 32, 32), 6).cuda(), torch.tensor([490.0]).cuda())
    r2 = _rel(y2, fx2["y"])
    report.add("f32/nets_vs_fixture" + ("_x3" if prec == F32X3 else ""), iddpm_rel_to_max=r1[0], iddpm_mean_rel=r1[1], ho_rel_to_max=r2[0], ho_mean_rel=r2[1])
    assert r1[0] < 2e-5 and r1[1] < 2e-5, r1
    assert r2[0] < 2e-5 and r2[1] < 2e-5, r2


@BOTH
def test_f32_afhq_iddpm_full_size_vs_oracle(engine, report, prec):
    """BASELINE config 5's network at 256 x 256 on the fp32 path vs the CPU oracle."""
    net = engine.create_net(cda.afhq_iddpm_desc(256, precision=prec))
    sd = nets.synth_state_dict(engine.net_params(net), 2)
    assert engine.load_state_dict(net, sd)[0] == 0
    cfg = nets.OpenAIUNetCfg(in_channels=3, out_channels=6, model_channels=128, num_res_blocks=1,
                             channel_mult=(1, 1, 2, 2, 4, 4), attn_ds=(16,), num_heads=4, num_head_channels=64,
                             use_scale_shift_norm=True, resblock_updown=True)
    x = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(8))
    t = torch.tensor([640.0])
    with torch.no_grad():
        ref = nets.openai_unet(sd, cfg, x, t)
    y = engine.unet_forward(net, x.cuda(), t.cuda())
    r = _rel(y, ref)
    report.add("f32/afhq_iddpm_fullsize" + ("_x3" if prec == F32X3 else ""), rel_to_max=r[0], mean_rel=r[1])
    assert r[0] < 5e-5 and r[1] < 5e-5, r


def _wrapper(fx, desc, sample_type="ddim", eta=0.1, refine_steps=0):
    import os
    os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = DDPMDDIMWrapper(source_model_type="toy32", sample_type=sample_type, custom_steps=int(fx["custom_steps"]),
                            es_steps=int(fx["es_steps"]), eta=eta, refine_steps=refine_steps, noise_on_cpu=True,
                            net_desc=desc)
    n, first = w.engine.load_state_dict(w.net, gu.weights(fx))
    assert n == 0, first
    return w


def _chain(report, fx_name, desc):
    """wrapper.encode -> wrapper(z) without refinement -> wrapper(z) with refinement, seeds as in
    oracle/gen_golden.py:_pixel_chain"""
    fx = gu.load(fx_name)
    es = int(fx["es_steps"])
    img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(int(fx["img_seed"])))
    w = _wrapper(fx, desc)
    torch.manual_seed(int(fx["noise_seed"]))
    with torch.no_grad():
        z = w.encode(img.cuda())
        out0 = w(z)
    assert z.shape == (1, es * 3 * 32 * 32)
    z5 = z.view(1, es, 3, 32, 32).cpu()
    slots = [int(s) for s in fx["z_sub_slots"]]
    zref = torch.as_tensor(fx["z_sub"])
    assert torch.allclose(z5[:, 0], zref[:, 0], atol=1e-6)
    zerr = [((z5[:, s] - zref[:, i]).abs().max() / zref[:, i].abs().max()).item() for i, s in enumerate(slots) if s]
    w.refine_steps = int(fx["refine_steps"])
    w.sched = type(w.sched)(w.custom_steps, w.es_steps, sample_type=w.sample_type, eta=w.eta, t_0=w.t_0,
                            refine_steps=w.refine_steps)
    torch.manual_seed(int(fx["refine_seed"]))
    with torch.no_grad():
        out1 = w(z)
    p0 = gu.psnr(out0, torch.as_tensor(fx["img"]))
    p1 = gu.psnr(out1, torch.as_tensor(fx["img_refined"]))
    report.add("f32/" + fx_name + ("_x3" if desc.precision == F32X3 else ""), psnr_vs_reference=p0, psnr_refined_vs_reference=p1, psnr_vs_input=gu.psnr(out0, img),
               eps_rel=zerr)
    return p0, p1, zerr


@BOTH
def test_c1_chain_with_refinement_vs_reference(report, prec):
    """BASELINE config 1 network, 'ddim' eta 0.1, 50 + 50 steps, then the refinement loop (refine_steps = 10:
    re-noise to t = 9, ten random eta-1 steps; ddpm_ddim_wrapper.py:431-453) - cd_pix_refine."""
    p0, p1, zerr = _chain(report, "c1_toy_ddpm_refine", cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=prec))
    assert p0 >= 40.0 and p1 >= 40.0, (p0, p1)
    assert max(zerr) < 1e-2, zerr


@BOTH
def test_c5_reduced_chain_iddpm_vs_reference(report, prec):
    """C5-shaped chain on the improved-DDPM architecture (6 -> 3 channel drop, ddpm_ddim_wrapper.py:237-238):
    custom_steps 100, es_steps 85, refine_steps 10."""
    p0, p1, zerr = _chain(report, "c5_tiny_iddpm_chain", tiny_iddpm_desc(prec))
    assert p0 >= 40.0 and p1 >= 40.0, (p0, p1)
    assert max(zerr) < 1e-2, zerr


@BOTH
def test_f32_chain_is_bit_reproducible(engine, prec):
    """two identical encodes on the fp32 path give identical bits (no autotuner, no split-K on this path; the split
    mode's 16-bit GEMMs accumulate k-ascending in every tile configuration)"""
    fx = gu.load("c1_toy_ddpm")
    net, _ = _load(engine, cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=prec), fx)
    from cycle_diffusion_amd import schedule
    sch = schedule.PixelSchedule(20, 20, sample_type="ddim", eta=0.1)
    x0 = gu.rnd((2, 3, 32, 32), 21, 0.5).cuda()
    nz = gu.rnd((20, 2, 3, 32, 32), 22).cuda()
    z1 = engine.dpm_encode(net, sch.kind, x0, sch.coef_encode(), noise=nz, last_uses_x0=False)
    z2 = engine.dpm_encode(net, sch.kind, x0, sch.coef_encode(), noise=nz, last_uses_x0=False)
    assert torch.equal(z1, z2)
    # sample 0 alone == sample 0 in the batch of 2: the fp32 kernels' accumulation order does not depend on M
    z3 = engine.dpm_encode(net, sch.kind, x0[:1], sch.coef_encode(), noise=nz[:, :1].contiguous(), last_uses_x0=False)
    assert torch.equal(z3, z1[:1])


def test_split_mode_range_guard_raises_instead_of_saturating(engine):
    """CD_PREC_F32X3 keeps GroupNorm outputs as fp16 pairs scaled by 16: a value beyond +-4094 must surface as an error
    at the next synchronisation point, and the engine must stay usable afterwards."""
    fx = gu.load("unet_toy_ho")
    net = engine.create_net(cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=F32X3))
    sd = dict(gu.weights(fx))
    name = next(k for k in sd if k.endswith("norm1.bias"))
    sd[name] = torch.full_like(torch.as_tensor(sd[name]), 5000.0)
    assert engine.load_state_dict(net, sd)[0] == 0
    x, t = gu.rnd((1, 3, 32, 32), 6).cuda(), torch.tensor([490.0]).cuda()
    with pytest.raises(RuntimeError, match="fp16 range"):
        engine.unet_forward(net, x, t)
        engine.synchronize()
    sd[name] = torch.as_tensor(gu.weights(fx)[name])
    assert engine.load_state_dict(net, sd)[0] == 0
    y = engine.unet_forward(net, x, t)
    engine.synchronize()
    assert _rel(y, fx["y"])[0] < 1e-4
