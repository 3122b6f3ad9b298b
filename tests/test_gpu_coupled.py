"""The coupled source -> target loop (include/cyclediff.h cd_cycle_translate; wrapper.translate): one U-Net forward per step
over [encoder rows | decoder rows] must reproduce what the two loops - cd_dpm_encode, then cd_ddim_decode(_v) on its z - give.

The reference composes the two loops in Model.forward (model/text_unsupervised_translation.py:24-40); the encoder step and the
decoder step of index k evaluate the network at the same timestep (ddim.py:482-500 and :423-447 walk the same `time_range`),
and the decoder needs eps_k only after its forward (ddim.py:634-645). On the small networks below no GEMM shape has a split-K
entry in the tile table, tiles never change a result bit and GroupNorm / attention are per-sample: the coupled call is
compared with the two calls BIT FOR BIT. At full size (other split-K factors for other batch sizes) the check is the stated
tolerance against the reference's own image."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

import golden_util as gu
from cycle_diffusion_amd import _ffi, schedule
from test_gpu_models import _load, tiny_sd_desc

pytestmark = pytest.mark.gpu

FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0


def _two_loops(engine, net, x0, sch, skip, c_src, uc, enc_g, c_tgt, dec_g, n_dec, noise):
    """cd_dpm_encode, then one cd_ddim_decode(_v) over the n_dec * B decoder rows (row j * B + b decodes sample b)"""
    B = x0.shape[0]
    z = engine.dpm_encode(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(skip), ctx_c=c_src, ctx_uc=uc, guidance=enc_g,
                          noise=noise)
    x = engine.ddim_decode(net, _ffi.CD_SCHED_DDIM, z.repeat(n_dec, 1, 1, 1, 1), sch.coef_decode(skip), ctx_c=c_tgt,
                           ctx_uc=uc.repeat(n_dec, 1, 1), guidance=dec_g)
    return z, x


@pytest.mark.parametrize("case", ["enc1_cfg3", "enc1_two_scales", "enc1_cond_only", "enc_cfg2_cfg3", "skip30_cfg3"])
def test_coupled_loop_equals_the_two_loops_bit_for_bit(engine, case):
    fx = gu.load("latent_cycle_tiny")
    net, _sd = _load(engine, tiny_sd_desc(), fx)
    x0, c, uc, c2 = (t.cuda() for t in gu.latent_cycle_inputs())
    S, B = 99, x0.shape[0]
    skip = 30 if case.startswith("skip30") else 0
    K = S - skip
    noise = torch.stack(gu.latent_noise(77, x0.shape, K), 0).cuda()
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), S, 0.1)
    enc_g = 2.0 if case.startswith("enc_cfg2") else 1.0
    n_dec, dec_g = 1, 3.0
    if case == "enc1_two_scales":
        n_dec, dec_g = 2, [1.5] * B + [4.0] * B
    elif case == "enc1_cond_only":
        dec_g = 1.0
    c_tgt = c2.repeat(n_dec, 1, 1)
    z_ref, x_ref = _two_loops(engine, net, x0, sch, skip, c, uc, enc_g, c_tgt, dec_g, n_dec, noise)
    z, x = engine.cycle_translate(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(skip), sch.coef_decode(skip), enc_ctx_c=c,
                                  enc_ctx_uc=uc, enc_guidance=enc_g, dec_ctx_c=c_tgt, dec_ctx_uc=uc.repeat(n_dec, 1, 1),
                                  dec_guidance=dec_g, n_dec=n_dec, noise=noise)
    engine.synchronize()
    assert z.shape == (B, K + 1, 4, 16, 16) and x.shape == (n_dec * B, 4, 16, 16)
    assert torch.isfinite(x).all() and x.abs().max() > 0.1
    assert torch.equal(z, z_ref), (z - z_ref).abs().max().item()
    assert torch.equal(x, x_ref), (x - x_ref).abs().max().item()
    if n_dec == 2:  # the two scales really decode to different latents
        assert (x[:B] - x[B:]).abs().max() > 1e-3


@pytest.mark.parametrize("prec", [_ffi.CD_PREC_F32, _ffi.CD_PREC_F32X3], ids=["fp32", "fp32x3"])
def test_coupled_loop_in_the_fp32_modes(engine, prec):
    """the fp32 / split-fp16 networks rebuild their NHWC input from x_t before every forward: both halves of the coupled batch"""
    if prec == _ffi.CD_PREC_F32X3 and FMT != 1.0:
        pytest.skip("the split mode needs the fp16 build")
    fx = gu.load("latent_cycle_tiny")
    d = tiny_sd_desc()
    d.precision = prec
    net, _sd = _load(engine, d, fx)
    x0, c, uc, c2 = (t.cuda() for t in gu.latent_cycle_inputs())
    S, B = 20, x0.shape[0]
    noise = torch.stack(gu.latent_noise(78, x0.shape, S), 0).cuda()
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), S, 0.1)
    z_ref, x_ref = _two_loops(engine, net, x0, sch, 0, c, uc, 1.0, c2, 3.0, 1, noise)
    z, x = engine.cycle_translate(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(0), sch.coef_decode(0), enc_ctx_c=c,
                                  enc_ctx_uc=uc, enc_guidance=1.0, dec_ctx_c=c2, dec_ctx_uc=uc, dec_guidance=3.0, noise=noise)
    engine.synchronize()
    assert torch.equal(z, z_ref) and torch.equal(x, x_ref), ((z - z_ref).abs().max().item(), (x - x_ref).abs().max().item())


def test_coupled_loop_refuses_what_it_cannot_do(engine):
    fx = gu.load("latent_cycle_tiny")
    net, _sd = _load(engine, tiny_sd_desc(), fx)
    x0, c, uc, c2 = (t.cuda() for t in gu.latent_cycle_inputs())
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), 4, 0.1)
    with pytest.raises(RuntimeError, match="CD_SCHED_DDIM"):  # the 'ddpm' posterior kernels carry no guidance combine
        engine.cycle_translate(net, _ffi.CD_SCHED_DDPM, x0, sch.coef_encode(0), sch.coef_decode(0), enc_ctx_c=c, enc_ctx_uc=uc,
                               dec_ctx_c=c2, dec_ctx_uc=uc, dec_guidance=3.0)
    with pytest.raises(RuntimeError, match="contexts"):
        engine.cycle_translate(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(0), sch.coef_decode(0), enc_ctx_c=c, enc_ctx_uc=uc)
    z, x = engine.cycle_translate(net, _ffi.CD_SCHED_DDIM, x0, sch.coef_encode(0), sch.coef_decode(0), enc_ctx_c=c, enc_ctx_uc=uc,
                                  dec_ctx_c=c2, dec_ctx_uc=uc, dec_guidance=3.0)
    assert torch.isfinite(x).all()  # the engine is usable after the refusals


def _tiny_wrapper(**kw):
    from test_gpu_wrappers import _make
    return _make(True, **kw)


def test_wrapper_translate_equals_encode_then_forward(report):
    """wrapper.translate(image, src, tgt) against wrapper(wrapper.encode(image, src), image, src, tgt) on the small networks:
    the whole ensemble machinery (2 trials x skips [3, 5] x decoder scales [1, 2, 3] -> 12 candidates, one cond-only scale that
    decodes from the returned z, ranking) with identical draws - bit-identical images."""
    w, _emb, _usd, _vsd = _tiny_wrapper(n_trials=2, skip_steps=[3, 5], decoder_unconditional_guidance_scales=[1.0, 2.0, 3.0],
                                        ranker=lambda img, orig, s, t: img.flatten(1).mean(1))
    image = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(5)).cuda()
    src, tgt = ["a photo", "a cat"], ["a drawing", "a dog"]
    calls = []
    real = w.engine.cycle_translate
    w.engine.cycle_translate = lambda *a, **k: (calls.append(k.get("n_dec")), real(*a, **k))[1]
    with torch.no_grad():
        torch.manual_seed(9)
        a = w(w.encode(image, src), image, src, tgt)
        assert not calls
        torch.manual_seed(9)
        b = w.translate(image, src, tgt)
        assert calls and all(n == 2 for n in calls), calls  # the two guided scales ride with the encoder
        w.couple = False
        torch.manual_seed(9)
        c = w.translate(image, src, tgt)
    assert a.shape == (2, 3, 64, 64) and torch.isfinite(a).all()
    assert torch.equal(a, b), (a - b).abs().max().item()
    assert torch.equal(a, c)


def test_c2_full_size_through_translate_vs_reference(report):
    """BASELINE config 2 end to end through the COUPLED loop (SDStochasticTextWrapper.translate: what the model API runs) against
    the reference's own image: same 50 dB floor as the two-call path (tests/test_gpu_e2e_fullsize.py; bf16 build 34 dB)."""
    from test_gpu_e2e_fullsize import PSNR_FLOOR, SeededEmbedder
    from cycle_diffusion_amd.gan_wrapper.latent_text_wrapper import SDStochasticTextWrapper
    from oracle import nets
    path = os.path.join(gu.GOLD, "c2_sd512_e2e.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    fx = np.load(path, allow_pickle=False)
    seeds = json.loads(str(fx["seeds"]))
    os.environ["CYCLEDIFF_SYNTHETIC_WEIGHTS"] = "1"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = SDStochasticTextWrapper(source_model_type="sd-v1-4.ckpt", custom_steps=int(fx["steps"]), eta=float(fx["eta"]),
                                    white_box_steps=int(fx["steps"]) + 1, skip_steps=[0],
                                    encoder_unconditional_guidance_scales=[1.0],
                                    decoder_unconditional_guidance_scales=[float(fx["dec_scale"])], n_trials=1,
                                    cond_stage=SeededEmbedder(768, seeds), noise_on_cpu=True)
    for net, key, seed in ((w.unet, "unet_names", seeds["unet"]), (w.vae, "vae_names", seeds["vae"])):
        sd = nets.synth_state_dict(json.loads(str(fx[key])), seed)
        assert w.engine.load_state_dict(net, sd)[0] == 0
        del sd
    image = torch.rand((1, 3, 512, 512), generator=torch.Generator().manual_seed(seeds["image"]))
    torch.manual_seed(seeds["noise"])
    with torch.no_grad():
        img = w.translate(image.cuda(), ["source"], ["target"])
    p = gu.psnr(img.cpu(), torch.as_tensor(fx["img"]))
    report.add("e2e/c2_sd512_coupled_loop", psnr_db=p)
    assert p >= PSNR_FLOOR, p
