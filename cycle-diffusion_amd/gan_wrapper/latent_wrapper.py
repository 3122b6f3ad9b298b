"""Unconditional latent-diffusion wrapper on the HIP engine: drop-in for LatentDiffStochasticWrapper
(model/gan_wrapper/latentdiff_stochastic_wrapper.py:175-316, `gan_type = LatentDiffStochastic`; the reference's
FFHQ -> CelebA-HQ experiment, config/experiments/translate_ffhq256_to_celeba256_latentdiff_ddim_eta01.cfg).

Same constructor kwargs, encode(image, class_label=None) -> z [B, white_box_steps * C * h * w], forward(z,
class_label=None) -> img in [0, 1], attributes .resolution .latent_dim .enforce_class_input.

Path: (image - 0.5) * 2 -> VQ-f4 encoder + quant_conv (VQModelInterface.encode: no sampling, scale factor 1) ->
DPM-Encoder with the unconditional U-Net -> [x_T, eps...]; forward: decode with the injected eps ->
DDIMSampler.refine (eta 1: re-noise to the DDIM level refine_steps - 1, then refine_steps random steps;
ddim.py:114-168, 339-393) -> nearest-codebook quantisation + post_quant_conv + decoder -> (x + 1) / 2.
Class-conditional models (`enforce_class_input`, cin256) are not used by any reference config and raise.
"""
import os

import numpy as np
import torch

from .. import _ffi, schedule
from ..engine import ldm_uncond_unet_desc, vq_f4_vae_desc
from ..runtime import apply_ema_shadow, get_engine, load_or_init_weights, read_checkpoint

# source_model_type -> (U-Net descriptor, first-stage descriptor, linear_start, linear_end, scale_factor, use_ema)
# from model/lib/latentdiff/models/ldm/<type>/config.yaml. Neither config sets `use_ema`, so DDPM's default True
# holds (latentdiff ddpm.py:55,88-91) and every sampler call of the reference wrapper runs inside
# `model.ema_scope("Plotting")` (latentdiff_stochastic_wrapper.py:116,135,155,164): the U-Net is evaluated on the
# EMA shadow weights `model_ema.*`, not on `model.diffusion_model.*`.
MODEL_TYPES = {
    "celeba256": (ldm_uncond_unet_desc, vq_f4_vae_desc, 0.0015, 0.0195, 1.0, True),
    "ffhq256": (ldm_uncond_unet_desc, vq_f4_vae_desc, 0.0015, 0.0195, 1.0, True),
}


LDM_PRECISIONS = {"fp16": _ffi.CD_PREC_16, "16": _ffi.CD_PREC_16, "fp32": _ffi.CD_PREC_F32, "fp32x3": _ffi.CD_PREC_F32X3}


class LatentDiffStochasticWrapper(torch.nn.Module):

    def __init__(self, source_model_type, custom_steps, eta, white_box_steps, refine_steps=0,
                 enforce_class_input=None, unconditional_guidance_scale=None, device=None, noise_on_cpu=False,
                 unet_desc=None, vae_desc=None, state_dict=None, precision=None, allow_lossy_16bit=False):
        super().__init__()
        if enforce_class_input:
            raise NotImplementedError("class-conditional LDMs (cin256) are not used by the reference configs")
        self.enforce_class_input = enforce_class_input
        self.unconditional_guidance_scale = unconditional_guidance_scale
        self.refine_steps = int(refine_steps)
        self.custom_steps, self.eta, self.white_box_steps = int(custom_steps), float(eta), int(white_box_steps)
        assert self.eta > 0
        self.noise_on_cpu = bool(noise_on_cpu)
        self.noise_source = None  # callable(shape) -> tensor: one stream per dataloader batch (main.py --fold) or per sample
        if source_model_type not in MODEL_TYPES:
            raise NotImplementedError(source_model_type)
        udesc_fn, vdesc_fn, ls, le, self.scale_factor, self.use_ema = MODEL_TYPES[source_model_type]
        # precision of the U-Net (the VQ first stage is 16-bit either way): 'fp32x3' (default: the fp32 network with its
        # GroupNorm-fed convolutions as split-fp16 products), 'fp32' (the reference's own arithmetic), or 'fp16'. These
        # LDMs are sampled with eta 0.1 over up to 999 steps: the nearly deterministic decode amplifies the 16-bit
        # round-off of eps_hat (full-size fixture, 99 steps: 25 dB against the reference's latent; 'fp32' 73 dB, 'fp32x3'
        # 76 dB - tests/test_gpu_ldm_uncond.py). The 16-bit engine is therefore refused unless asked for by name AND
        # acknowledged (`allow_lossy_16bit`), as DDPMDDIMWrapper refuses 16-bit 'ddim' chains (allow_lossy_ddim).
        fp16_build = _ffi.load_library().cd_act_format() == 1  # a property of the library: no engine (no GPU) needed yet
        if precision is None:
            # the split mode packs its weights as fp16 pairs: it exists in the fp16 build only (launch_pack_w3); the bf16
            # build (CD_ACT_FP16=0) falls back to the plain fp32 path, which holds the same parity at a third of the speed
            precision = "fp32x3" if fp16_build else "fp32"
        if str(precision) not in LDM_PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(LDM_PRECISIONS))
        self.precision = str(precision)
        udesc = unet_desc if unet_desc is not None else udesc_fn()
        if unet_desc is None:
            udesc.precision = LDM_PRECISIONS[self.precision]
        else:  # an explicit descriptor carries its own precision
            self.precision = {v: k for k, v in LDM_PRECISIONS.items() if k != "16"}[int(udesc.precision)]
        if LDM_PRECISIONS[self.precision] == _ffi.CD_PREC_F32X3 and not fp16_build:
            raise ValueError("precision='fp32x3' needs the fp16 build of the library (this is the bf16 build): use 'fp32'")
        # the refusal covers an explicit descriptor too: a CD_PREC_16 `unet_desc` is the same lossy engine
        if LDM_PRECISIONS[self.precision] == _ffi.CD_PREC_16 and not allow_lossy_16bit:
            raise ValueError("precision=%r does not reproduce the reference on the eta-0.1 chains of the unconditional LDMs "
                             "(about 25 dB latent signal-to-error after 99 steps against 73-76 dB); use 'fp32x3' (default) "
                             "or 'fp32', or pass allow_lossy_16bit=True" % self.precision)
        self.engine = get_engine(device)
        vdesc = vae_desc if vae_desc is not None else vdesc_fn()
        self.channels, self.image_size = udesc.in_channels, udesc.image_size
        self.unet = self.engine.create_net(udesc)
        self.vae = self.engine.create_net(vdesc)
        self.vae_factor = 2 ** (vdesc.n_mult - 1)
        ckpt = os.path.join("ckpts", "ldm_models", "ldm", source_model_type, "model.ckpt")  # prepare_latentdiff (:16)
        sd = state_dict if state_dict is not None else read_checkpoint(ckpt)
        if sd is not None and self.use_ema:
            sd = apply_ema_shadow(sd)
        self.weights_origin = load_or_init_weights(self.engine, ckpt, {
            self.unet: "model.diffusion_model.", self.vae: "first_stage_model."}, state_dict=sd)
        self.resolution = self.image_size * self.vae_factor
        self.latent_dim = self.image_size ** 2 * self.channels * self.white_box_steps
        self.alphas_cumprod = schedule.latent_alphas_cumprod(1000, ls, le)
        self._anchor = torch.nn.Parameter(torch.zeros(1, device=self.engine.device), requires_grad=True)

    def _randn(self, n, shape):
        if self.noise_source is not None:
            return torch.stack([self.noise_source(tuple(shape)) for _ in range(n)], 0).to(self.device, torch.float32)
        if self.noise_on_cpu:  # one tensor per reference draw, in the reference's order
            return torch.stack([torch.randn(shape) for _ in range(n)], 0).to(self.device)
        return torch.randn((n,) + tuple(shape), device=self.device)

    def encode(self, image, class_label=None):
        image = (image - 0.5) * 2.0
        assert image.shape[2] == image.shape[3] == self.resolution
        x0 = self.engine.vae_encode(self.vae, image.to(self.device, torch.float32), sample=False,
                                    scale=self.scale_factor)
        sch = schedule.DDIMSchedule(self.alphas_cumprod, self.custom_steps, self.eta)
        K = len(sch)
        # the DPM-Encoder loop breaks after white_box_steps - 1 steps (latentdiff ddim.py:484, skip_steps = 0): the reference's
        # configs set white_box_steps = the chain + 1 (n_loop = K); a shorter prefix is the same call on the table rows
        # K-n_loop .. K-1 + the x_T row (every row carries its timestep) without the `index 0 returns x0` special case
        n_loop = max(0, min(K, self.white_box_steps - 1))
        assert self.white_box_steps >= 1 and self.white_box_steps <= K + 1, \
            "white_box_steps counts x_T and the encoded steps: 1 .. custom chain + 1 (latent_dim is sized by it)"
        coef = sch.coef_encode()
        if n_loop < K:
            coef = np.concatenate([coef[K - n_loop:K], coef[K:K + 1]])
        # draw order of _ddpm_ddim_encoding: randn_like(x0), then one randn per sample_xt_next except index 0
        nz = self._randn(K if n_loop == K else n_loop + 1, tuple(x0.shape))
        z = self.engine.dpm_encode(self.unet, _ffi.CD_SCHED_DDIM, x0, coef, noise=nz, last_uses_x0=n_loop == K)
        z = z.view(x0.shape[0], -1)
        assert z.shape[1] == self.latent_dim
        return z

    def generate(self, z, class_label=None):
        bsz = z.shape[0]
        zz = z.view(bsz, self.white_box_steps, self.channels, self.image_size, self.image_size).contiguous()
        sch = schedule.DDIMSchedule(self.alphas_cumprod, self.custom_steps, self.eta)
        n_tail = len(sch) - (self.white_box_steps - 1)  # decode steps beyond the list draw fresh noise (ddim.py:436)
        tail = self._randn(n_tail, (bsz, self.channels, self.image_size, self.image_size)) if n_tail > 0 else None
        x = self.engine.ddim_decode(self.unet, _ffi.CD_SCHED_DDIM, zz, sch.coef_decode(), noise_tail=tail)
        if self.refine_steps > 0:  # convsample_ddim: refine with eta = 1 (latentdiff_stochastic_wrapper.py:70-78)
            rs = schedule.DDIMSchedule(self.alphas_cumprod, self.custom_steps, 1.0)
            nz = self._randn(self.refine_steps + 1, tuple(x.shape))
            x = self.engine.pix_refine(self.unet, _ffi.CD_SCHED_DDIM, x, rs.coef_refine(self.refine_steps), noise=nz)
        if self.precision == "fp32x3":
            self.engine.synchronize()  # surfaces the split mode's range guard before the result is used
        return self.engine.vae_decode(self.vae, x, scale=self.scale_factor)

    def forward(self, z, class_label=None):
        img = self.generate(z.to(self.device, torch.float32), class_label)
        return (img + 1.0) / 2.0  # post_process Normalize(mean=-1, std=2)

    @property
    def device(self):
        return next(self.parameters()).device
