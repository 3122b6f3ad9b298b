"""Which engine call stops agreeing with its one-sample-at-a-time result, and from which sample on?
Compares, per sample: VAE decode, VAE encode, a 2-step CFG decode (prefix sharing on / off via CYCLEDIFF_CFG_SHARE),
a 2-step encode - at batch sizes 16, 17, 25, 32 on the SD-v1.4-shaped networks (synthetic weights)."""
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"
import golden_util as gu  # noqa: E402
from cycle_diffusion_amd import _ffi  # noqa: E402
from cycle_diffusion_amd.gan_wrapper.latent_text_wrapper import SDStochasticTextWrapper  # noqa: E402

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    w = SDStochasticTextWrapper(source_model_type="sd-v1-4.ckpt", custom_steps=99, eta=0.1, white_box_steps=100,
                                skip_steps=[0], encoder_unconditional_guidance_scales=[1.0],
                                decoder_unconditional_guidance_scales=[3.0], n_trials=1, noise_on_cpu=True)
eng = w.engine
which = sys.argv[1:] or ["vae_dec", "vae_enc", "decode", "decode_v", "encode"]
sch = w._schedule()


def per_sample(a, b):
    d = (a - b).flatten(1).abs().max(dim=1).values
    s = b.flatten(1).abs().max(dim=1).values
    return (d / s).cpu()


def show(tag, rel):
    bad = [i for i, v in enumerate(rel.tolist()) if v > 2e-2]
    print("%-34s B=%2d  max rel %.2e  median %.2e  first bad sample %s  n_bad %d" % (
        tag, len(rel), rel.max().item(), rel.median().item(), bad[0] if bad else "-", len(bad)), flush=True)


for B in [int(x) for x in os.environ.get("DIAG_B", "16,17,25,32").split(",")]:
    g = torch.Generator().manual_seed(B)
    if "vae_dec" in which:
        z = torch.randn(B, 4, 64, 64, generator=g).cuda()
        full = eng.vae_decode(w.vae, z, scale=0.18215, out_mul=0.5, out_add=0.5)
        one = torch.cat([eng.vae_decode(w.vae, z[i:i + 1], scale=0.18215, out_mul=0.5, out_add=0.5) for i in range(B)], 0)
        show("vae_decode 512", per_sample(full, one))
    if "vae_enc" in which:
        img = (torch.rand(B, 3, 512, 512, generator=g) * 2 - 1).cuda()
        full = eng.vae_encode(w.vae, img, sample=False, scale=0.18215)
        one = torch.cat([eng.vae_encode(w.vae, img[i:i + 1], sample=False, scale=0.18215) for i in range(B)], 0)
        show("vae_encode 512", per_sample(full, one))
    c = torch.randn(B, 77, 768, generator=g).cuda()
    uc = torch.randn(1, 77, 768, generator=g).cuda().repeat(B, 1, 1)
    K = 2
    zz = torch.randn(B, K + 1, 4, 64, 64, generator=g).cuda()
    coef = sch.coef_decode(97)
    assert len(coef) == K, len(coef)
    if "decode" in which:
        full = eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz, coef, ctx_c=c, ctx_uc=uc, guidance=3.0)
        one = torch.cat([eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz[i:i + 1].contiguous(), coef, ctx_c=c[i:i + 1],
                                         ctx_uc=uc[i:i + 1], guidance=3.0) for i in range(B)], 0)
        show("cfg decode 2 steps (share=%s)" % os.environ.get("CYCLEDIFF_CFG_SHARE", "1"), per_sample(full, one))
        full1 = eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz, coef, ctx_c=c, ctx_uc=uc, guidance=1.0)
        one1 = torch.cat([eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz[i:i + 1].contiguous(), coef, ctx_c=c[i:i + 1],
                                          ctx_uc=uc[i:i + 1], guidance=1.0) for i in range(B)], 0)
        show("scale-1 decode 2 steps", per_sample(full1, one1))
    if "decode_v" in which:
        gv = [1.5 + (i % 5) for i in range(B)]
        full = eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz, coef, ctx_c=c, ctx_uc=uc, guidance=gv)
        one = torch.cat([eng.ddim_decode(w.unet, _ffi.CD_SCHED_DDIM, zz[i:i + 1].contiguous(), coef, ctx_c=c[i:i + 1],
                                         ctx_uc=uc[i:i + 1], guidance=float(gv[i])) for i in range(B)], 0)
        show("per-sample-scale decode 2 steps", per_sample(full, one))
    if "encode" in which:
        x0 = torch.randn(B, 4, 64, 64, generator=g).cuda()
        nz = torch.randn(K, B, 4, 64, 64, generator=g).cuda()
        ce = sch.coef_encode(97)
        full = eng.dpm_encode(w.unet, _ffi.CD_SCHED_DDIM, x0, ce, ctx_c=c, ctx_uc=uc, guidance=1.0, noise=nz, last_uses_x0=True)
        one = torch.cat([eng.dpm_encode(w.unet, _ffi.CD_SCHED_DDIM, x0[i:i + 1], ce, ctx_c=c[i:i + 1], ctx_uc=uc[i:i + 1],
                                        guidance=1.0, noise=nz[:, i:i + 1].contiguous(), last_uses_x0=True) for i in range(B)], 0)
        show("encode 2 steps (z)", per_sample(full.flatten(1), one.flatten(1)))
