"""cycle-diffusion_amd: MI355X-native CycleDiffusion hot path (DPM-Encoder inversion + coupled
DDIM/DDPM decode, U-Net and VAE forward) behind the reference's gan_wrapper / model API.

The directory name carries a hyphen (it is fixed by the project layout); import it as
`cycle_diffusion_amd` (the top-level shim package redirects here).
"""
from . import _ffi  # noqa: F401
from .engine import (Engine, afhq_iddpm_desc, bert_xtransformer_desc, clip_text_desc, ho_ddpm_desc,  # noqa: F401
                     kl_f8_vae_desc, oclip_text_desc, oclip_vision_desc,
                     ldm_text_unet_desc, make_desc, sd_v1_unet_desc)

__all__ = ["Engine", "make_desc", "sd_v1_unet_desc", "ldm_text_unet_desc", "kl_f8_vae_desc",
           "afhq_iddpm_desc", "ho_ddpm_desc", "clip_text_desc", "bert_xtransformer_desc", "oclip_text_desc",
           "oclip_vision_desc"]
