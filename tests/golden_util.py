"""Shared helpers for tests that replay the committed reference fixtures (tests/golden/*.npz)."""
import json
import os

import numpy as np
import torch

from oracle import nets, samplers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_SD_CFG = nets.OpenAIUNetCfg(in_channels=4, out_channels=4, model_channels=64, num_res_blocks=1,
                                 channel_mult=(1, 2), attn_ds=(1, 2), num_heads=2, use_spatial_transformer=True,
                                 context_dim=64)
TINY_IDDPM_CFG = nets.OpenAIUNetCfg(in_channels=3, out_channels=6, model_channels=32, num_res_blocks=1,
                                    channel_mult=(1, 2, 2), attn_ds=(2,), num_heads=4, num_head_channels=32,
                                    use_scale_shift_norm=True, resblock_updown=True)
TINY_VAE_CFG = nets.VAECfg(ch=32, ch_mult=(1, 2, 4), num_res_blocks=1)
TINY_LDM_UNCOND_CFG = nets.OpenAIUNetCfg(in_channels=3, out_channels=3, model_channels=32, num_res_blocks=1,
                                         channel_mult=(1, 2, 3), attn_ds=(2, 4), num_head_channels=32)
TINY_VQ_CFG = nets.VAECfg(ch=32, ch_mult=(1, 2, 4), num_res_blocks=1, z_channels=3, embed_dim=3)
TOY_HO_CFG = nets.HoCfg(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(16,), resolution=32)


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def weights(fx):
    return nets.synth_state_dict(json.loads(str(fx["names"])), int(fx["wseed"]))


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def latent_noise(seed, shape, K):
    """Draw order of DDIMSampler._ddpm_ddim_encoding: randn_like(x0), then K-1 x randn(shape)."""
    torch.manual_seed(seed)
    return [torch.randn(shape) for _ in range(K)]


def pixel_noise(seed, shape, es_steps):
    """Draw order of DDPMDDIMWrapper.encode (x_T + es-1 posterior draws) then generate's last step."""
    torch.manual_seed(seed)
    enc = [torch.randn(shape) for _ in range(es_steps)]
    last = torch.randn(shape)
    return enc, last


def tiny_sd_inputs():
    return rnd((2, 4, 16, 16), 1), torch.tensor([11, 981]), rnd((2, 77, 64), 2)


def latent_cycle_inputs():
    x0 = rnd((2, 4, 16, 16), 7, 0.8)
    return x0, rnd((2, 77, 64), 8), rnd((2, 77, 64), 9), rnd((2, 77, 64), 10)


def psnr(a, b, peak=1.0):
    """evaluation/utils.py:60-67 convention: images in [0,1] (clamped), 10*log10(peak^2 / mse)."""
    a = a.detach().float().cpu().clamp(0, 1)
    b = b.detach().float().cpu().clamp(0, 1)
    mse = ((a - b) ** 2).mean().item()
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)
