"""The inner seam (cycle_diffusion_amd/compat.py): the REFERENCE's own, unmodified `DDIMSampler`
(model/lib/stable_diffusion/ldm/models/diffusion/ddim.py - imported from /root/reference where it is mounted, on the GPU box
from the copy oracle/stage_reference.py stages next to the built library; test side only) drives the HIP U-Net through
`LatentDiffusionHIP.apply_model`, and its z / latents are compared with the engine's fused loops on the same network, contexts
and noise. Two independent implementations of the loop around ONE network: the reference's torch scheduler arithmetic with a
C-ABI call per forward, against cd_dpm_encode / cd_ddim_decode / cd_cycle_translate (one fused step kernel per step)."""
import pytest
import torch

import golden_util as gu
from cycle_diffusion_amd import _ffi, schedule
from cycle_diffusion_amd.compat import LatentDiffusionHIP
from oracle import ref_import
from test_gpu_models import _load, tiny_sd_desc

pytestmark = pytest.mark.gpu

FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0


def _draws(seed, shape, K):
    """the device draws of DDIMSampler._ddpm_ddim_encoding in its own order: randn_like(x0) (ddim.py:479), then one
    noise_like(shape, device) per sample_xt_next with index > 0 (ddim.py:599)"""
    torch.cuda.manual_seed(seed)
    first = torch.randn(shape, device="cuda")
    return torch.stack([first] + [torch.randn(shape, device="cuda") for _ in range(K - 1)], 0)


@pytest.mark.parametrize("dec_scale", [1.0, 3.0])
def test_reference_ddim_sampler_over_the_hip_unet_matches_the_fused_loops(engine, report, dec_scale):
    if not ref_import.available():
        pytest.skip("no reference tree (neither /root/reference nor oracle/_ref)")
    fx = gu.load("latent_cycle_tiny")
    net, _sd = _load(engine, tiny_sd_desc(), fx)
    x0, c, uc, c2 = (t.cuda() for t in gu.latent_cycle_inputs())
    S, skip = 99, 91
    K = S - skip  # 8 steps of the real 99-step schedule
    model = LatentDiffusionHIP(engine, net)
    with ref_import.session():
        from ldm.models.diffusion.ddim import DDIMSampler  # the reference's class, as it is
        torch.cuda.manual_seed(4242)
        with ref_import.quiet(), torch.no_grad():
            z_list = DDIMSampler(model).ddpm_ddim_encoding(S, batch_size=x0.shape[0], shape=(4, 16, 16), conditioning=c, eta=0.1,
                                                           white_box_steps=S + 1, skip_steps=skip, verbose=False, x0=x0,
                                                           unconditional_guidance_scale=1, unconditional_conditioning=uc)
            z_ref = torch.stack(z_list, dim=1)
            x_ref, _ = DDIMSampler(model).sample_with_eps(S, z_ref[:, 1:], conditioning=c2, batch_size=x0.shape[0],
                                                          shape=(4, 16, 16), eta=0.1, verbose=False, x_T=z_ref[:, 0],
                                                          skip_steps=skip, unconditional_guidance_scale=dec_scale,
                                                          unconditional_conditioning=uc)
    assert z_ref.shape == (2, K + 1, 4, 16, 16) and z_ref.is_cuda
    noise = _draws(4242, tuple(x0.shape), K)
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), S, 0.1)
    z = engine.dpm_encode(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(skip), ctx_c=c, ctx_uc=uc, guidance=1.0, noise=noise)
    x = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z, sch.coef_decode(skip), ctx_c=c2, ctx_uc=uc, guidance=dec_scale)
    zc, xc = engine.cycle_translate(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(skip), sch.coef_decode(skip), enc_ctx_c=c,
                                    enc_ctx_uc=uc, dec_ctx_c=c2, dec_ctx_uc=uc, dec_guidance=dec_scale, noise=noise)
    engine.synchronize()
    # x_T is scheduler arithmetic on identical draws; every later slot has been through the same 16-bit network, fed with an
    # x_t that two fp32 implementations of the step (torch's op-by-op kernels here, one fused -ffp-contract=off kernel there)
    # may round differently in the last bit before the network's 16-bit input rounding
    xT = (z[:, 0] - z_ref[:, 0]).abs().max().item()
    eps_rel = ((z[:, 1:] - z_ref[:, 1:]).flatten(2).abs().max(dim=2).values /
               z_ref[:, 1:].flatten(2).abs().max(dim=2).values).max().item()
    lat_rel = ((x - x_ref).abs().max() / x_ref.abs().max()).item()
    report.add("compat/reference_sampler_over_hip_unet_scale%g" % dec_scale, xT_maxabs=xT, eps_rel=eps_rel, latent_rel=lat_rel)
    assert xT < 1e-6, xT
    assert eps_rel < 5e-3 * FMT, eps_rel
    assert lat_rel < 5e-3 * FMT, lat_rel
    assert torch.equal(zc, z) and torch.equal(xc, x)  # the coupled loop is the two fused loops, bit for bit


def test_first_stage_seam_round_trip(engine):
    """encode_first_stage / get_first_stage_encoding / decode_first_stage with the reference's call pattern
    (sd_wrapper:185-187, 135-137) against the engine calls the wrappers make"""
    from test_gpu_models import tiny_vae_desc
    fxv = gu.load("vae_tiny")
    vae, _sd = _load(engine, tiny_vae_desc(), fxv)
    fx = gu.load("latent_cycle_tiny")
    net, _sd2 = _load(engine, tiny_sd_desc(), fx)
    model = LatentDiffusionHIP(engine, net, vae=vae, vae_factor=4)
    img = (torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    torch.manual_seed(6)
    z = model.get_first_stage_encoding(model.encode_first_stage(img))
    torch.manual_seed(6)
    nz = torch.randn(2, 4, 16, 16).cuda()
    z2 = engine.vae_encode(vae, img, noise=nz, sample=True, scale=0.18215)
    assert torch.allclose(z, z2, rtol=0, atol=1e-6 * float(z2.abs().max()))
    zm = model.get_first_stage_encoding(model.encode_first_stage(img).mode())
    assert torch.allclose(zm, engine.vae_encode(vae, img, sample=False, scale=0.18215), rtol=0, atol=1e-6 * float(zm.abs().max()))
    out = model.decode_first_stage(z)
    assert torch.equal(out, engine.vae_decode(vae, z, scale=0.18215)) and out.shape == (2, 3, 64, 64)
