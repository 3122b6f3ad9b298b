"""Full-size end-to-end fixtures from the REFERENCE's own modules (CPU fp32, imported from /root/reference).

  python -m oracle.gen_golden_full --only c2      (~25 min on 8 cores)
  python -m oracle.gen_golden_full --only c3      (~6 min)
  python -m oracle.gen_golden_full --only c5r     (~8 min; BASELINE config 5 at its real size, reduced chain)
  python -m oracle.gen_golden_full --only ldm_uncond  (~6 min; the unconditional-LDM U-Net at full size, latent side)
  python -m oracle.gen_golden_full --only c2ens   (~25 min; the SD wrapper's ensemble loops, SD-sized nets at 256 x 256)
  python -m oracle.gen_golden_full --only c5      (~30 min; the same with the reference's full 1000 / 850 / 100 chain)
  python -m oracle.gen_golden_full --only c2b4    (~2 h; config 2 with FOUR triplets per reference call, skip 20, scales [1, 3])
  python -m oracle.gen_golden_full --only c3b16   (~70 min; config 3 with SIXTEEN triplets per reference call, skip 30, scale 2)

One (image, source-text, target-text) triplet per BASELINE configuration, run the way the reference's
text wrapper composes it (stable_diffusion_stochastic_text_wrapper.py:169-249): VAE encode -> posterior
sample (SD, ddpm.py:538) / mean (LDM, latentdiff ddpm.py:535-538) x 0.18215 -> DDIMSampler.ddpm_ddim_encoding
(99 steps, eta 0.1, encoder scale 1; ddim.py:230-286,450-501) -> DDIMSampler.sample_with_eps towards the
target text with classifier-free guidance 3 (ddim.py:170-228,395-448) -> VAE decode -> (x + 1) / 2.

Weights: oracle.nets.synth_state_dict over the reference modules' own (name, shape) lists (stored in the
fixture so the GPU box rebuilds them without the reference); inputs and contexts from torch.Generator seeds;
every noise draw comes from the global generator after torch.manual_seed(noise_seed), in the reference's
own draw order (posterior sample first, then randn_like(x0), then one randn per sample_xt_next).
TEST INFRASTRUCTURE ONLY.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from oracle.gen_golden import (RefVAE, build_ref_pixel_wrapper, build_ref_sd_unet, load_synth, rnd,  # noqa: E402
                               save)

FULL_VAE = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], in_channels=3,
                resolution=256, z_channels=4, double_z=True, dropout=0.0)
SD_UNET = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
               num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
               transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False)  # v1-inference.yaml:29-44
LDM_UNET = dict(SD_UNET, context_dim=1280)  # txt2img-1p4B-eval.yaml

SEEDS = dict(unet=0, vae=1, image=1, c_src=2, uc=3, c_tgt=5, noise=4)


def run_text_triplet(name, unet_cfg, res, ctx_dim, sample_posterior, steps=99, eta=0.1, dec_scale=3.0):
    ref_import.setup()
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    Sampler = ref_import.ddim_sampler_cls()
    t0 = time.time()
    with torch.no_grad():
        u = build_ref_sd_unet(unet_cfg)
        uns, _ = load_synth(u, SEEDS["unet"])
        v = RefVAE(FULL_VAE)
        vns, _ = load_synth(v, SEEDS["vae"])
        shim = ref_import.LatentShim(u)
        lat = res // 8
        image = torch.rand((1, 3, res, res), generator=torch.Generator().manual_seed(SEEDS["image"]))
        c_src, uc, c_tgt = (rnd((1, 77, ctx_dim), SEEDS[k]) for k in ("c_src", "uc", "c_tgt"))
        torch.manual_seed(SEEDS["noise"])
        # encode_first_stage + get_first_stage_encoding (ddpm.py:536-543)
        post = DiagonalGaussianDistribution(v.moments((image - 0.5) * 2.0))
        x0 = (post.sample() if sample_posterior else post.mode()) * 0.18215
        print(name, "vae encode done", time.time() - t0, flush=True)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(steps, batch_size=1, shape=(4, lat, lat), conditioning=c_src,
                                                      eta=eta, white_box_steps=steps + 1, skip_steps=0, verbose=False,
                                                      x0=x0, unconditional_guidance_scale=1,
                                                      unconditional_conditioning=uc)
        z = torch.stack(z_list, dim=1)
        print(name, "encode done", time.time() - t0, flush=True)
        with ref_import.quiet():
            x_tgt, _ = Sampler(shim).sample_with_eps(steps, z[:, 1:], conditioning=c_tgt, batch_size=1,
                                                     shape=(4, lat, lat), eta=eta, verbose=False, x_T=z[:, 0],
                                                     skip_steps=0, unconditional_guidance_scale=dec_scale,
                                                     unconditional_conditioning=uc)
        print(name, "decode done", time.time() - t0, flush=True)
        img = (v.decode(x_tgt / 0.18215) + 1.0) / 2.0  # decode_first_stage (ddpm.py:705) + post_process
    save(name, unet_names=json.dumps(uns), vae_names=json.dumps(vns), seeds=json.dumps(SEEDS), steps=steps, eta=eta,
         dec_scale=dec_scale, sample_posterior=int(sample_posterior), x0=x0,
         z_sub=z[:, [0, 1, 50, 99]], z_sub_slots=np.asarray([0, 1, 50, 99]), z_norms=z.flatten(2).norm(dim=2),
         x_tgt=x_tgt, img=img, cpu_seconds=time.time() - t0, cpu_threads=torch.get_num_threads())


def gen_c2():
    """BASELINE config 2 (headline): Stable-Diffusion-v1.4 shapes, 512 x 512."""
    run_text_triplet("c2_sd512_e2e", SD_UNET, 512, 768, sample_posterior=True)


def gen_c3():
    """BASELINE config 3: LDM text2img-large shapes, 256 x 256 (posterior mean)."""
    run_text_triplet("c3_ldm256_e2e", LDM_UNET, 256, 1280, sample_posterior=False)


def gen_c2_ensemble():
    """The ensemble loops of the SD wrapper (stable_diffusion_stochastic_text_wrapper.py:189-204 encode: trial -> encoder
    scale -> skip_steps; :142-167 generate: member -> decoder scale) with the SD-v1.4-sized U-Net and the KL-f8 VAE on a
    256 x 256 image (32 x 32 latent: the networks are fully convolutional; a quarter of the CPU time of 512 x 512):
    n_trials 2, skip_steps [40, 50], encoder scale 1, decoder scales [1, 3] -> 4 members, 8 candidates, each produced by
    the reference's own DDIMSampler calls with the wrapper's arguments, one member at a time. Pins skip_steps > 0, the
    scale-1 and classifier-free-guidance decodes and the member order at the real network size; the engine folds the
    members that share (skip, scale) into one batch."""
    ref_import.setup()
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    Sampler = ref_import.ddim_sampler_cls()
    res, steps, eta, wb = 256, 99, 0.1, 100
    skips, dec_scales, n_trials = [40, 50], [1.0, 3.0], 2
    seeds = dict(SEEDS, image=7, noise=44)
    t0 = time.time()
    with torch.no_grad():
        u = build_ref_sd_unet(SD_UNET)
        uns, _ = load_synth(u, seeds["unet"])
        v = RefVAE(FULL_VAE)
        vns, _ = load_synth(v, seeds["vae"])
        shim = ref_import.LatentShim(u)
        lat = res // 8
        image = torch.rand((1, 3, res, res), generator=torch.Generator().manual_seed(seeds["image"]))
        c_src, uc, c_tgt = (rnd((1, 77, 768), seeds[k]) for k in ("c_src", "uc", "c_tgt"))
        torch.manual_seed(seeds["noise"])
        post = DiagonalGaussianDistribution(v.moments((image - 0.5) * 2.0))
        x0 = post.sample() * 0.18215  # one posterior draw per encode() call (:185-187)
        z_ens = []
        for _trial in range(n_trials):
            for enc_scale in (1,):
                for skip in skips:
                    with ref_import.quiet():
                        z_list = Sampler(shim).ddpm_ddim_encoding(steps, conditioning=c_src, batch_size=1,
                                                                  shape=(4, lat, lat), eta=eta, white_box_steps=wb,
                                                                  skip_steps=skip, verbose=False, x0=x0,
                                                                  unconditional_guidance_scale=enc_scale,
                                                                  unconditional_conditioning=uc)
                    z_ens.append(torch.stack(z_list, dim=1))
                    print("c2 ensemble: member", len(z_ens), "encoded", time.time() - t0, flush=True)
        imgs, lats = [], []
        for i, z in enumerate(z_ens):
            skip = skips[i % len(skips)]
            assert z.shape[1] == wb - skip
            for sc in dec_scales:
                with ref_import.quiet():
                    x, _ = Sampler(shim).sample_with_eps(steps, z[:, 1:], conditioning=c_tgt, batch_size=1,
                                                         shape=(4, lat, lat), eta=eta, verbose=False, x_T=z[:, 0],
                                                         skip_steps=skip, unconditional_guidance_scale=sc,
                                                         unconditional_conditioning=uc)
                lats.append(x)
                imgs.append((v.decode(x / 0.18215) + 1.0) / 2.0)
                print("c2 ensemble: candidate", len(imgs), "decoded", time.time() - t0, flush=True)
    save("c2_sd_ensemble256_e2e", unet_names=json.dumps(uns), vae_names=json.dumps(vns), seeds=json.dumps(seeds),
         steps=steps, eta=eta, white_box_steps=wb, skip_steps=np.asarray(skips), dec_scales=np.asarray(dec_scales),
         n_trials=n_trials, x0=x0, x_T=torch.cat([z[:, 0] for z in z_ens], 0),
         z_norms=np.concatenate([z.flatten(2).norm(dim=2).numpy().ravel() for z in z_ens]),
         lat=torch.cat(lats, 0), img=torch.cat(imgs, 0).to(torch.float16), cpu_seconds=time.time() - t0,
         cpu_threads=torch.get_num_threads())


def gen_c2_b4():
    """A second, independent pin of BASELINE config 2 at the reference harness's own batch size: FOUR (image, source text,
    target text) triplets in ONE call of every reference function (README.md:153 `--per_device_eval_batch_size 4`;
    trainer/trainer.py:788-789), the way SDStochasticTextWrapper composes them (stable_diffusion_stochastic_text_wrapper.py:
    169-206 encode, :142-167 generate) with `skip_steps = [20]`, encoder scale 1 and decoder scales [1, 3] -> 2 candidates per
    triplet. Other seeds than c2_sd512_e2e for the images, the twelve contexts and the noise; same synthetic networks. Draws:
    one posterior sample [4, 4, 64, 64], randn_like(x0), one randn [4, 4, 64, 64] per sample_xt_next (ddim.py:479, 599).
    ~2 h on 6 threads (79 encoder forwards at B = 4, 79 decoder forwards at B = 4 and 79 at B = 8)."""
    ref_import.setup()
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    Sampler = ref_import.ddim_sampler_cls()
    res, steps, eta, wb, B = 512, 99, 0.1, 100, 4
    skip, dec_scales = 20, [1.0, 3.0]
    seeds = dict(unet=SEEDS["unet"], vae=SEEDS["vae"], image=[101, 102, 103, 104], c_src=[111, 112, 113, 114],
                 c_tgt=[121, 122, 123, 124], uc=131, noise=2024)
    t0 = time.time()
    with torch.no_grad():
        u = build_ref_sd_unet(SD_UNET)
        uns, _ = load_synth(u, seeds["unet"])
        v = RefVAE(FULL_VAE)
        vns, _ = load_synth(v, seeds["vae"])
        shim = ref_import.LatentShim(u)
        lat = res // 8
        image = torch.cat([torch.rand((1, 3, res, res), generator=torch.Generator().manual_seed(s))
                           for s in seeds["image"]], 0)
        c_src = torch.cat([rnd((1, 77, 768), s) for s in seeds["c_src"]], 0)
        c_tgt = torch.cat([rnd((1, 77, 768), s) for s in seeds["c_tgt"]], 0)
        uc = rnd((1, 77, 768), seeds["uc"]).repeat(B, 1, 1)  # get_condition: bs * [""] (sd_wrapper:28-36)
        torch.manual_seed(seeds["noise"])
        post = DiagonalGaussianDistribution(v.moments((image - 0.5) * 2.0))
        x0 = post.sample() * 0.18215
        print("c2_b4: vae encode done", time.time() - t0, flush=True)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(steps, conditioning=c_src, batch_size=B, shape=(4, lat, lat),
                                                      eta=eta, white_box_steps=wb, skip_steps=skip, verbose=False, x0=x0,
                                                      unconditional_guidance_scale=1, unconditional_conditioning=uc)
        z = torch.stack(z_list, dim=1)
        assert z.shape[1] == wb - skip
        print("c2_b4: encode done", time.time() - t0, flush=True)
        lats, imgs = [], []
        for sc in dec_scales:
            with ref_import.quiet():
                x, _ = Sampler(shim).sample_with_eps(steps, z[:, 1:], conditioning=c_tgt, batch_size=B,
                                                     shape=(4, lat, lat), eta=eta, verbose=False, x_T=z[:, 0],
                                                     skip_steps=skip, unconditional_guidance_scale=sc,
                                                     unconditional_conditioning=uc)
            lats.append(x)
            imgs.append(torch.cat([(v.decode(x[i:i + 1] / 0.18215) + 1.0) / 2.0 for i in range(B)], 0))
            print("c2_b4: scale", sc, "decoded", time.time() - t0, flush=True)
    slots = [0, 1, 40, wb - skip - 1]
    save("c2_sd512_b4_e2e", unet_names=json.dumps(uns), vae_names=json.dumps(vns), seeds=json.dumps(seeds), steps=steps,
         eta=eta, white_box_steps=wb, skip_steps=np.asarray([skip]), dec_scales=np.asarray(dec_scales), x0=x0,
         z_sub=z[:, slots], z_sub_slots=np.asarray(slots), z_norms=z.flatten(2).norm(dim=2),
         lat=torch.stack(lats, 0), img=torch.stack(imgs, 0).to(torch.float16), cpu_seconds=time.time() - t0,
         cpu_threads=torch.get_num_threads())


def gen_c3_b16():
    """BASELINE config 3 at ITS batch size in one reference call: LDM text2img-large shapes, 256 x 256, SIXTEEN triplets per
    call of every reference function (README.md:195 `--per_device_eval_batch_size 16`), posterior mean
    (latentdiff ddpm.py:535-538), `skip_steps = [30]`, encoder scale 1, decoder scale 2; other seeds than c3_ldm256_e2e.
    ~70 min on 8 threads (69 encoder forwards at B = 16, 69 decoder forwards at B = 32)."""
    ref_import.setup()
    from ldm.modules.distributions.distributions import DiagonalGaussianDistribution
    Sampler = ref_import.ddim_sampler_cls()
    res, steps, eta, wb, B = 256, 99, 0.1, 100, 16
    skip, dec_scale = 30, 2.0
    seeds = dict(unet=SEEDS["unet"], vae=SEEDS["vae"], image=list(range(301, 301 + B)), c_src=list(range(331, 331 + B)),
                 c_tgt=list(range(361, 361 + B)), uc=391, noise=4048)
    t0 = time.time()
    with torch.no_grad():
        u = build_ref_sd_unet(LDM_UNET)
        uns, _ = load_synth(u, seeds["unet"])
        v = RefVAE(FULL_VAE)
        vns, _ = load_synth(v, seeds["vae"])
        shim = ref_import.LatentShim(u)
        lat = res // 8
        image = torch.cat([torch.rand((1, 3, res, res), generator=torch.Generator().manual_seed(s))
                           for s in seeds["image"]], 0)
        c_src = torch.cat([rnd((1, 77, 1280), s) for s in seeds["c_src"]], 0)
        c_tgt = torch.cat([rnd((1, 77, 1280), s) for s in seeds["c_tgt"]], 0)
        uc = rnd((1, 77, 1280), seeds["uc"]).repeat(B, 1, 1)
        torch.manual_seed(seeds["noise"])
        x0 = torch.cat([DiagonalGaussianDistribution(v.moments((image[i:i + 4] - 0.5) * 2.0)).mode() for i in range(0, B, 4)],
                       0) * 0.18215
        print("c3_b16: vae encode done", time.time() - t0, flush=True)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(steps, conditioning=c_src, batch_size=B, shape=(4, lat, lat), eta=eta,
                                                      white_box_steps=wb, skip_steps=skip, verbose=False, x0=x0,
                                                      unconditional_guidance_scale=1, unconditional_conditioning=uc)
        z = torch.stack(z_list, dim=1)
        print("c3_b16: encode done", time.time() - t0, flush=True)
        with ref_import.quiet():
            x, _ = Sampler(shim).sample_with_eps(steps, z[:, 1:], conditioning=c_tgt, batch_size=B, shape=(4, lat, lat),
                                                 eta=eta, verbose=False, x_T=z[:, 0], skip_steps=skip,
                                                 unconditional_guidance_scale=dec_scale, unconditional_conditioning=uc)
        img = torch.cat([(v.decode(x[i:i + 2] / 0.18215) + 1.0) / 2.0 for i in range(0, B, 2)], 0)
        print("c3_b16: decoded", time.time() - t0, flush=True)
    slots = [0, 1, 35, wb - skip - 1]
    save("c3_ldm256_b16_e2e", unet_names=json.dumps(uns), vae_names=json.dumps(vns), seeds=json.dumps(seeds), steps=steps,
         eta=eta, white_box_steps=wb, skip_steps=np.asarray([skip]), dec_scales=np.asarray([dec_scale]), x0=x0,
         z_sub=z[:, slots], z_sub_slots=np.asarray(slots), z_norms=z.flatten(2).norm(dim=2), lat=x,
         img=img.to(torch.float16), cpu_seconds=time.time() - t0, cpu_threads=torch.get_num_threads())


LDM_UNCOND_UNET = dict(image_size=64, in_channels=3, out_channels=3, model_channels=224, attention_resolutions=[8, 4, 2],
                       num_res_blocks=2, channel_mult=[1, 2, 3, 4], num_head_channels=32)  # celeba256 / ffhq256 config.yaml:17-34


def gen_ldm_uncond_full():
    """gan_type LatentDiffStochastic at the real network size, latent side: the celeba256 / ffhq256 U-Net (224 channels,
    AttentionBlocks with the legacy QKV order) on a 3 x 64 x 64 latent through the reference's DDIMSampler -
    ddpm_ddim_encoding, sample_with_eps and refine (ddim.py:114-168, 339-393; eta 1) - with the chain of the reference
    cfg (999 / 1000 / 400) divided by ~10: 99 steps, white_box_steps 100, refine_steps 40. The VQ first stage is left out
    (its codebook lookup lives in taming-transformers, absent here): x0 is a seeded latent, outputs are latents."""
    Sampler = ref_import.ddim_sampler_cls()
    S, R = 99, 40
    t0 = time.time()
    with torch.no_grad():
        u = build_ref_sd_unet(LDM_UNCOND_UNET)
        uns, _ = load_synth(u, 308)
        shim = ref_import.LatentShim(u, linear_start=0.0015, linear_end=0.0195)
        x0 = rnd((1, 3, 64, 64), 313)
        torch.manual_seed(7171)
        with ref_import.quiet():
            z_list = Sampler(shim).ddpm_ddim_encoding(S, batch_size=1, shape=(3, 64, 64), eta=0.1, white_box_steps=S + 1,
                                                      verbose=False, x0=x0)
            z = torch.stack(z_list, dim=1)
            print("ldm_uncond_full: encode done", time.time() - t0, flush=True)
            x_dec, _ = Sampler(shim).sample_with_eps(S, z[:, 1:], batch_size=1, shape=(3, 64, 64), eta=0.1,
                                                     verbose=False, x_T=z[:, 0])
        torch.manual_seed(8282)
        with ref_import.quiet():
            x_ref, _ = Sampler(shim).refine(S, refine_steps=R, batch_size=1, shape=(3, 64, 64), eta=1, verbose=False,
                                            x0=x_dec)
        slots = [0, 1, 50, 99]
    save("ldm_uncond_full_latent", unet_names=json.dumps(uns), useed=308, x0_seed=313, noise_seed=7171, refine_seed=8282,
         steps=S, refine_steps=R, x0=x0, z_sub=z[:, slots], z_sub_slots=np.asarray(slots),
         z_norms=z.flatten(2).norm(dim=2), x_dec=x_dec, x_ref=x_ref, cpu_seconds=time.time() - t0)


def gen_c5r(name="c5r_afhq256_e2e", custom_steps=100, es_steps=85, refine_steps=10, unrefined=True):
    """BASELINE config 5 at its real size: two `i_DDPM('AFHQ')` networks (improved_ddpm/script_util.py:5-22,102-104)
    at 256 x 256, source encodes and target decodes exactly as UnsupervisedTranslation.forward composes them
    (model/unsupervised_translation.py:49-50): z = source.encode(image) (ddpm_ddim_wrapper.py:455-534), img =
    target(z) (:392-453, 536-542), sample_type 'ddim', eta 0.1, REDUCED chain custom_steps 100 / es_steps 85 /
    refine_steps 10 (translate_afhqcat256_to_afhqdog256_ddim_eta01.cfg:10-14 divided by 10), batch 1. One global
    generator seed before encode (the draws of encode, generate's last step and the refinement loop follow in the
    reference's own order). The unrefined image (refine_steps 0, fresh seed) is stored beside it."""
    ref_import.setup()
    from model.lib.ddpm_ddim.models.improved_ddpm.script_util import i_DDPM
    seeds = dict(source=201, target=202, image=13, noise=8642, noise_unrefined=97531)
    t0 = time.time()
    with torch.no_grad():
        src, tgt = i_DDPM("AFHQ"), i_DDPM("AFHQ")
        ns, _ = load_synth(src, seeds["source"])
        nt, _ = load_synth(tgt, seeds["target"])
        assert ns == nt
        ws = build_ref_pixel_wrapper(src, custom_steps=custom_steps, es_steps=es_steps, eta=0.1,
                                     refine_steps=refine_steps, resolution=256)
        wt = build_ref_pixel_wrapper(tgt, custom_steps=custom_steps, es_steps=es_steps, eta=0.1,
                                     refine_steps=refine_steps, resolution=256)
        # learn_sigma stays False as the reference sets it for AFHQ (ddpm_ddim_wrapper.py:370-374): the 6-channel
        # output is split by the shape test (:236-238, :131-134) and the variance half is dropped
        img = torch.rand((1, 3, 256, 256), generator=torch.Generator().manual_seed(seeds["image"]))
        torch.manual_seed(seeds["noise"])
        with ref_import.quiet():
            z = ws.encode(image=img)
            print(name, "encode done", time.time() - t0, flush=True)
            out = wt(z=z)
            print(name, "decode + refine done", time.time() - t0, flush=True)
            out0 = torch.zeros(0)
            if unrefined:
                wt.refine_steps = 0
                torch.manual_seed(seeds["noise_unrefined"])
                out0 = wt(z=z)
        z5 = z.view(1, es_steps, 3, 256, 256)
        slots = [0, 1, es_steps // 2, es_steps - 1]
    save(name, names=json.dumps(ns), seeds=json.dumps(seeds), custom_steps=custom_steps, es_steps=es_steps,
         refine_steps=refine_steps, eta=0.1, z_sub=z5[:, slots], z_sub_slots=np.asarray(slots),
         z_norms=z5.flatten(2).norm(dim=2), img=out, img_unrefined=out0, cpu_seconds=time.time() - t0,
         cpu_threads=torch.get_num_threads())


def gen_c5():
    """BASELINE config 5 with the reference's FULL chain (translate_afhqcat256_to_afhqdog256_ddim_eta01.cfg:10-14:
    custom_steps 1000, es_steps 850, refine_steps 100): 1000 encoder forwards on the source net, 850 + 100 on the
    target net, batch 1. Same networks, image and seeds as the reduced fixture; no separate unrefined pass."""
    gen_c5r("c5_afhq256_full_chain_e2e", 1000, 850, 100, unrefined=False)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    for k, fn in dict(c3=gen_c3, c2=gen_c2, c5r=gen_c5r).items():
        if not a.only or a.only == k:
            fn()
    if a.only == "c5":  # ~30 min: only on request
        gen_c5()
    if a.only == "c2ens":  # ~25 min: only on request
        gen_c2_ensemble()
    if a.only == "ldm_uncond":  # ~6 min
        gen_ldm_uncond_full()
    if a.only == "c2b4":  # ~2 h: only on request
        gen_c2_b4()
    if a.only == "c3b16":  # ~70 min: only on request
        gen_c3_b16()
