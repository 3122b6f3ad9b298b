"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from
/root/reference, CPU fp32) on seeded synthetic weights and inputs.

Run here (the reference does not exist on the GPU box):   python -m oracle.gen_golden
The fixtures pin the oracle restatement (tests/test_oracle_golden.py) and carry everything needed to
rebuild the exact inputs without the reference: the (name, shape) list of every weight tensor plus
the seeds (weights come from oracle.nets.synth_state_dict, inputs from torch.Generator seeds).
TEST INFRASTRUCTURE ONLY.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets, ref_import, samplers  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# ---- the tiny architectures used by fixtures and by the GPU parity tests -------------------------
TINY_SD = dict(image_size=16, in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1,
               attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True,
               transformer_depth=1, context_dim=64, legacy=False)
TINY_IDDPM = dict(image_size=32, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
                  attention_resolutions=(2,), channel_mult=(1, 2, 2), num_heads=4, num_head_channels=32,
                  use_scale_shift_norm=True, resblock_updown=True)
TINY_VAE = dict(ch=32, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=[], in_channels=3,
                resolution=64, z_channels=4, double_z=True, dropout=0.0)
TOY_HO = dict(ch=32, out_ch=3, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=[16], dropout=0.0,
              in_channels=3, resamp_with_conv=True, image_size=32)


def named_shapes(module):
    return [(k, list(v.shape)) for k, v in module.state_dict().items()]


def load_synth(module, seed):
    ns = named_shapes(module)
    sd = nets.synth_state_dict(ns, seed)
    module.load_state_dict(sd)
    module.eval()
    return ns, sd


def build_ref_sd_unet(cfg=TINY_SD):
    ref_import.setup()
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    return UNetModel(**cfg)


def build_ref_iddpm(cfg=TINY_IDDPM):
    ref_import.setup()
    from model.lib.ddpm_ddim.models.improved_ddpm.unet import UNetModel
    return UNetModel(**cfg)


class RefVAE(torch.nn.Module):
    """AutoencoderKL.encode/decode glue (autoencoder.py:324-333) around the reference Encoder/Decoder."""

    def __init__(self, cfg=TINY_VAE, embed_dim=4):
        super().__init__()
        ref_import.setup()
        from ldm.modules.diffusionmodules.model import Decoder, Encoder
        with ref_import.quiet():
            self.encoder = Encoder(**cfg)
            self.decoder = Decoder(**cfg)
        self.quant_conv = torch.nn.Conv2d(2 * cfg["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = torch.nn.Conv2d(embed_dim, cfg["z_channels"], 1)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def build_ref_ho(cfg=TOY_HO):
    ref_import.setup()
    from types import SimpleNamespace as NS
    from model.lib.ddpm_ddim.models.ddpm.diffusion import DDPM
    c = NS(model=NS(ch=cfg["ch"], out_ch=cfg["out_ch"], ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"],
                    attn_resolutions=cfg["attn_resolutions"], dropout=cfg["dropout"], in_channels=cfg["in_channels"],
                    resamp_with_conv=cfg["resamp_with_conv"]),
           data=NS(image_size=cfg["image_size"]))
    return DDPM(c)


def build_ref_pixel_wrapper(generator, custom_steps, es_steps, eta, sample_type="ddim", refine_steps=0, resolution=32):
    """DDPMDDIMWrapper without checkpoint loading (mirrors __init__, ddpm_ddim_wrapper.py:319-390)."""
    ref_import.setup()
    import model.gan_wrapper.ddpm_ddim_wrapper as W
    w = W.DDPMDDIMWrapper.__new__(W.DDPMDDIMWrapper)
    torch.nn.Module.__init__(w)
    betas = W.get_beta_schedule(beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000)
    w.register_buffer("betas", torch.from_numpy(betas).float())
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    acp = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - acp) / (1.0 - ac)
    w.logvar = np.log(np.maximum(pv, 1e-20))
    w.generator = generator
    w.learn_sigma = False
    w.enforce_class_input = None
    w.custom_steps, w.es_steps, w.eta, w.sample_type = custom_steps, es_steps, eta, sample_type
    w.refine_steps, w.refine_iterations, w.t_0 = refine_steps, 1, 999
    w.resolution, w.channels = resolution, 3
    w.latent_dim = resolution ** 2 * 3 * es_steps
    w.post_process = W.transforms.Compose([W.transforms.Normalize(mean=[-1.0, -1.0, -1.0], std=[2.0, 2.0, 2.0])])
    return w


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


def gen_schedule():
    Sampler = ref_import.ddim_sampler_cls()
    shim = ref_import.LatentShim(None)
    s = Sampler(shim)
    with ref_import.quiet():
        s.make_schedule(ddim_num_steps=99, ddim_eta=0.1, verbose=False)
    save("schedule_sd_s99_eta0p1", timesteps=np.asarray(s.ddim_timesteps),
         a=np.asarray(s.ddim_alphas, dtype=np.float32), a_prev=np.asarray(s.ddim_alphas_prev, dtype=np.float64),
         sigma=np.asarray(s.ddim_sigmas, dtype=np.float64),
         r=np.asarray(s.ddim_sqrt_one_minus_alphas, dtype=np.float32),
         alphas_cumprod=shim.alphas_cumprod.numpy())


def gen_nets():
    with torch.no_grad():
        u = build_ref_sd_unet()
        ns, _ = load_synth(u, 101)
        x, t, ctx = rnd((2, 4, 16, 16), 1), torch.tensor([11, 981]), rnd((2, 77, 64), 2)
        save("unet_tiny_sd", names=json.dumps(ns), y=u(x, t, context=ctx), wseed=101)

        u = build_ref_iddpm()
        ns, _ = load_synth(u, 102)
        x, t = rnd((2, 3, 32, 32), 3), torch.tensor([3.0, 700.0])
        save("unet_tiny_iddpm", names=json.dumps(ns), y=u(x, t), wseed=102)

        v = RefVAE()
        ns, _ = load_synth(v, 103)
        img = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(4)) * 2 - 1
        mom = v.moments(img)
        zz = rnd((2, 4, 16, 16), 5, 0.5)
        save("vae_tiny", names=json.dumps(ns), moments=mom, dec=v.decode(zz), wseed=103)

        h = build_ref_ho()
        ns, _ = load_synth(h, 104)
        x, t = rnd((1, 3, 32, 32), 6), torch.tensor([490.0])
        save("unet_toy_ho", names=json.dumps(ns), y=h(x, t), wseed=104)


def gen_latent_cycle():
    """Reference DDIMSampler encode -> decode with a tiny cross-attention U-Net (S=99, eta=0.1,
    encoder scale 1, decoder scale 3 = the C2 [gan] settings on a small network)."""
    Sampler = ref_import.ddim_sampler_cls()
    with torch.no_grad():
        u = build_ref_sd_unet()
        ns, _ = load_synth(u, 105)
        shim = ref_import.LatentShim(u)
        x0 = rnd((2, 4, 16, 16), 7, 0.8)
        c, uc, c2 = rnd((2, 77, 64), 8), rnd((2, 77, 64), 9), rnd((2, 77, 64), 10)
        torch.manual_seed(1234)  # the sampler draws from the global generator (ddim.py:479,599)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(99, batch_size=2, shape=(4, 16, 16), conditioning=c, eta=0.1,
                                                      white_box_steps=100, skip_steps=0, verbose=False, x0=x0,
                                                      unconditional_guidance_scale=1, unconditional_conditioning=uc)
            z = torch.stack(z_list, dim=1)
            x_same, _ = Sampler(shim).sample_with_eps(99, z[:, 1:], conditioning=c, batch_size=2, shape=(4, 16, 16),
                                                      eta=0.1, verbose=False, x_T=z[:, 0], skip_steps=0,
                                                      unconditional_guidance_scale=1, unconditional_conditioning=uc)
            x_tgt, _ = Sampler(shim).sample_with_eps(99, z[:, 1:], conditioning=c2, batch_size=2, shape=(4, 16, 16),
                                                     eta=0.1, verbose=False, x_T=z[:, 0], skip_steps=0,
                                                     unconditional_guidance_scale=3.0, unconditional_conditioning=uc)
        save("latent_cycle_tiny", names=json.dumps(ns), wseed=105, noise_seed=1234, z_sub=z[:, [0, 1, 50, 99]],
             z_norms=z.flatten(2).norm(dim=2), x_same=x_same, x_tgt=x_tgt, cycle_err=(x_same - x0).abs().max())


def gen_c1():
    """BASELINE config 1: toy 32x32 Ho-DDPM, 50-step DPM-Encoder invert + decode, batch 1, 'ddim' eta 0.1."""
    with torch.no_grad():
        h = build_ref_ho()
        ns, _ = load_synth(h, 106)
        w = build_ref_pixel_wrapper(h, custom_steps=50, es_steps=50, eta=0.1)
        img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(11))
        torch.manual_seed(4321)
        with ref_import.quiet():
            z = w.encode(image=img)
            out = w(z=z)
        z5 = z.view(1, 50, 3, 32, 32)
        save("c1_toy_ddpm", names=json.dumps(ns), wseed=106, noise_seed=4321, z_sub=z5[:, [0, 1, 25, 49]],
             z_norms=z5.flatten(2).norm(dim=2), img=out)
        # 'ddpm' sample type on a shorter chain
        w2 = build_ref_pixel_wrapper(h, custom_steps=20, es_steps=20, eta=None, sample_type="ddpm")
        torch.manual_seed(999)
        with ref_import.quiet():
            z2 = w2.encode(image=img)
            out2 = w2(z=z2)
        z25 = z2.view(1, 20, 3, 32, 32)
        save("c1_toy_ddpm_ddpmtype", names=json.dumps(ns), wseed=106, noise_seed=999, z_sub=z25[:, [0, 1, 10, 19]],
             z_norms=z25.flatten(2).norm(dim=2), img=out2)


def _pixel_chain(name, net, wseed, custom_steps, es_steps, refine_steps, img_seed, seeds):
    """DDPMDDIMWrapper.encode -> forward without refinement -> forward with refinement (fresh seed each), 'ddim'
    eta 0.1 (ddpm_ddim_wrapper.py:455-534, 392-453)."""
    with torch.no_grad():
        ns, _ = load_synth(net, wseed)
        w = build_ref_pixel_wrapper(net, custom_steps=custom_steps, es_steps=es_steps, eta=0.1, refine_steps=0)
        img = torch.rand((1, 3, 32, 32), generator=torch.Generator().manual_seed(img_seed))
        torch.manual_seed(seeds[0])
        with ref_import.quiet():
            z = w.encode(image=img)
            out0 = w(z=z)
        w.refine_steps = refine_steps
        torch.manual_seed(seeds[1])
        with ref_import.quiet():
            out1 = w(z=z)
        z5 = z.view(1, es_steps, 3, 32, 32)
        slots = [0, 1, es_steps // 2, es_steps - 1]
        save(name, names=json.dumps(ns), wseed=wseed, noise_seed=seeds[0], refine_seed=seeds[1], img_seed=img_seed,
             custom_steps=custom_steps, es_steps=es_steps, refine_steps=refine_steps, z_sub=z5[:, slots],
             z_sub_slots=np.asarray(slots), z_norms=z5.flatten(2).norm(dim=2), img=out0, img_refined=out1)


def gen_pixel_refine():
    """a11 / C5: the refinement loop (refine_steps > 0) on the C1 toy network, and a reduced C5-shaped chain
    (custom_steps 100, es_steps 85, refine_steps 10 = the reference cfg's 1000 / 850 / 100 divided by 10) on the
    improved-DDPM architecture (6 output channels of which 3 are dropped, ddpm_ddim_wrapper.py:237-238)."""
    _pixel_chain("c1_toy_ddpm_refine", build_ref_ho(), 106, 50, 50, 10, 11, (4321, 777))
    _pixel_chain("c5_tiny_iddpm_chain", build_ref_iddpm(), 107, 100, 85, 10, 12, (2468, 1357))


TINY_LDM_UNCOND = dict(image_size=16, in_channels=3, out_channels=3, model_channels=32, num_res_blocks=1,
                       attention_resolutions=[2, 4], channel_mult=[1, 2, 3], num_head_channels=32)
TINY_VQ = dict(ch=32, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=[], in_channels=3,
               resolution=64, z_channels=3, double_z=False, dropout=0.0)


class RefVQ(torch.nn.Module):
    """VQModelInterface glue (model/lib/latentdiff/ldm/models/autoencoder.py:264-282) around the reference Encoder /
    Decoder; the quantiser itself lives in taming-transformers (absent): oracle.nets.vq_quantize restates it."""

    def __init__(self, cfg=TINY_VQ, embed_dim=3, n_embed=256):
        super().__init__()
        ref_import.setup()
        from ldm.modules.diffusionmodules.model import Decoder, Encoder
        with ref_import.quiet():
            self.encoder = Encoder(**cfg)
            self.decoder = Decoder(**cfg)
        self.quantize = torch.nn.Module()
        self.quantize.embedding = torch.nn.Embedding(n_embed, embed_dim)
        self.quant_conv = torch.nn.Conv2d(cfg["z_channels"], embed_dim, 1)
        self.post_quant_conv = torch.nn.Conv2d(embed_dim, cfg["z_channels"], 1)

    def encode(self, x):
        return self.quant_conv(self.encoder(x))

    def decode(self, h):
        return self.decoder(self.post_quant_conv(nets.vq_quantize(h, self.quantize.embedding.weight)))


def gen_ldm_uncond():
    """gan_type LatentDiffStochastic on small networks: VQ first stage -> reference DDIMSampler encode (49 steps, eta
    0.1, linear 0.0015..0.0195 schedule of the celeba256 / ffhq256 LDMs) -> sample_with_eps -> refine (10 steps,
    eta 1; ddim.py:114-168,339-393) -> VQ decode -> (x + 1) / 2 (latentdiff_stochastic_wrapper.py:59-80,262-305)."""
    Sampler = ref_import.ddim_sampler_cls()
    S, R = 49, 10
    with torch.no_grad():
        u = build_ref_sd_unet(TINY_LDM_UNCOND)
        uns, _ = load_synth(u, 108)
        v = RefVQ()
        vns, _ = load_synth(v, 109)
        shim = ref_import.LatentShim(u, linear_start=0.0015, linear_end=0.0195)
        image = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(13))
        x0 = v.encode((image - 0.5) * 2.0)
        torch.manual_seed(5151)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(S, batch_size=1, shape=(3, 16, 16), eta=0.1, white_box_steps=S + 1,
                                                      verbose=False, x0=x0)
            z = torch.stack(z_list, dim=1)
            x_dec, _ = Sampler(shim).sample_with_eps(S, z[:, 1:], batch_size=1, shape=(3, 16, 16), eta=0.1,
                                                     verbose=False, x_T=z[:, 0])
        torch.manual_seed(6262)
        with ref_import.quiet():
            x_ref, _ = Sampler(shim).refine(S, refine_steps=R, batch_size=1, shape=(3, 16, 16), eta=1, verbose=False,
                                            x0=x_dec)
        img0 = (v.decode(x_dec) + 1.0) / 2.0
        img = (v.decode(x_ref) + 1.0) / 2.0
        slots = [0, 1, 25, 49]
        save("ldm_uncond_tiny", unet_names=json.dumps(uns), vae_names=json.dumps(vns), useed=108, vseed=109, img_seed=13,
             noise_seed=5151, refine_seed=6262, steps=S, refine_steps=R, x0=x0, z_sub=z[:, slots],
             z_sub_slots=np.asarray(slots), z_norms=z.flatten(2).norm(dim=2), x_dec=x_dec, x_ref=x_ref, img_norefine=img0,
             img=img)


def gen_xtr_text():
    """LDM text encoder: the reference's vendored x-transformers TransformerWrapper on seeded weights / ids."""
    ref_import.setup()
    sys.path.insert(0, os.path.join(ref_import.REF, "model", "lib", "latentdiff"))
    from ldm.modules.x_transformer import Encoder, TransformerWrapper
    from oracle import xtr_text
    cfg = xtr_text.XtrTextCfg(width=64, layers=2, vocab=300, positions=77)
    m = TransformerWrapper(num_tokens=cfg.vocab, max_seq_len=cfg.positions,
                           attn_layers=Encoder(dim=cfg.width, depth=cfg.layers), emb_dropout=0.0).eval()
    sd = xtr_text.synth_state_dict(cfg, 17)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("to_logits") for k in missing), (missing, unexpected)
    ids = torch.randint(0, cfg.vocab, (3, 77), generator=torch.Generator().manual_seed(18))
    with torch.no_grad():
        y = m(ids, return_embeddings=True)
    save("xtr_text_tiny", y=y, ids=ids, wseed=17, width=cfg.width, layers=cfg.layers, vocab=cfg.vocab)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = dict(schedule=gen_schedule, nets=gen_nets, latent=gen_latent_cycle, c1=gen_c1, xtr=gen_xtr_text,
                pixel_refine=gen_pixel_refine, ldm_uncond=gen_ldm_uncond)
    for k, fn in todo.items():
        if not a.only or a.only == k:
            fn()
