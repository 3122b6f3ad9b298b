"""Pixel-space DDPM wrapper on the HIP engine: drop-in for DDPMDDIMWrapper
(model/gan_wrapper/ddpm_ddim_wrapper.py:317-542): same constructor kwargs, encode(image,
class_label=None) -> z [B, es_steps*C*R*R], forward(z, class_label=None) -> img in [0,1],
attributes .resolution .latent_dim .enforce_class_input.

Precision: `[gan] precision = fp32` (default) runs the network in fp32 end to end on the engine's fp32 path - what the
reference computes in (use_fp16=False) and what the 'ddim' chain needs to be reproducible;
`precision = fp32x3` is the same fp32 network with its GroupNorm-fed convolutions (97 % of the FLOPs) evaluated as
three-term split-fp16 products on the 16-bit matrix cores (2^-22 per product; 2.3x the fp32 throughput on C5, same
parity floors; a value outside the fp16 range of the split raises instead of saturating);
`precision = fp16` selects the 16-bit engine (about 6x the throughput; fine for sample_type 'ddpm').

Unlike the reference, decode is batched: its 'ddim' branch compares [B,1,1,1] tensors and only runs
at batch 1 (ddpm_ddim_wrapper.py:216; README.md:254); per-sample results are identical.
"""
import os

import torch

from .. import _ffi, schedule
from ..engine import afhq_iddpm_desc, ho_ddpm_desc
from ..runtime import get_engine, load_or_init_weights

# source_model_type -> (architecture, default checkpoint path); the reference reads these from
# ckpts/ddpm/configs/*.yml (absent from the tree); values are the DiffusionCLIP configs it was built on
MODEL_TYPES = {
    "celeba256": ("ho256", "ckpts/ddpm/celeba_hq.ckpt"),
    "bedroom256": ("ho256", "ckpts/ddpm/bedroom.ckpt"),
    "church_outdoor256": ("ho256", "ckpts/ddpm/church_outdoor.ckpt"),
    "afhqdog256": ("iddpm256", None),
    "afhqcat256": ("iddpm256", None),
    "afhqwild256": ("iddpm256", None),
    "ffhq256": ("iddpm256", "ckpts/ddpm/ffhq_10m.pt"),
    "toy32": ("toy32", None),  # BASELINE config 1: ch 32, mult (1,2,2), 1 res block, attention at 16
}


# "fp32x3": the fp32 network with its large convolutions as three-term split-fp16 products (include/cyclediff.h)
PRECISIONS = {"fp32": _ffi.CD_PREC_F32, "fp32x3": _ffi.CD_PREC_F32X3, "fp16": _ffi.CD_PREC_16, "16": _ffi.CD_PREC_16}


def _desc(arch, precision=_ffi.CD_PREC_F32):
    if arch == "ho256":
        return ho_ddpm_desc(256, 128, (1, 1, 2, 2, 4, 4), 2, (16,), precision=precision)
    if arch == "iddpm256":
        return afhq_iddpm_desc(256, precision=precision)
    if arch == "toy32":
        return ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,), precision=precision)
    raise NotImplementedError(arch)


class DDPMDDIMWrapper(torch.nn.Module):

    def __init__(self, source_model_type, sample_type, custom_steps, es_steps, source_model_path=None,
                 refine_steps=0, refine_iterations=1, eta=None, t_0=None, enforce_class_input=None, device=None,
                 noise_on_cpu=False, precision="fp32", net_desc=None, allow_lossy_ddim=False):
        super().__init__()
        if str(precision) not in PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
        self.precision = str(precision)
        if PRECISIONS[self.precision] == _ffi.CD_PREC_16 and sample_type == "ddim" and not allow_lossy_ddim:
            # the 'ddim' chain rescales x by up to x130 end to end; a 16-bit quantiser inside the network
            # turns it into a ~15 dB reconstruction. Known-broken, so refuse unless asked for by name (bench.py's
            # throughput-only `--precision fp16` line does).
            raise ValueError("precision=%r with sample_type='ddim' does not reproduce the reference (about 15 dB); use "
                             "precision='fp32' (default) or pass allow_lossy_ddim=True" % self.precision)
        # parity runs draw every noise tensor on the CPU, one tensor per reference draw, in the reference's order
        self.noise_on_cpu = bool(noise_on_cpu)
        self.noise_source = None  # callable(shape) -> CPU tensor: one stream per sample of a batch (parity tests)
        self.enforce_class_input = enforce_class_input
        self.custom_steps, self.es_steps = custom_steps, es_steps
        self.refine_steps, self.refine_iterations = refine_steps, refine_iterations
        self.sample_type, self.eta = sample_type, eta
        self.t_0 = t_0 if t_0 is not None else 999
        if sample_type == "ddim":
            assert eta > 0
        elif sample_type == "ddpm":
            assert eta is None
        else:
            raise ValueError()
        if source_model_type not in MODEL_TYPES:
            raise NotImplementedError(source_model_type)
        arch, default_path = MODEL_TYPES[source_model_type]
        if default_path is not None and source_model_type != "ffhq256":
            assert source_model_path is None
        if arch == "iddpm256" and default_path is None and net_desc is None:
            # afhq*: the reference asserts a path is given (ddpm_ddim_wrapper.py:60-74)
            from ..runtime import synthetic_allowed
            assert source_model_path is not None or synthetic_allowed(), \
                "%s needs source_model_path (or CYCLEDIFF_SYNTHETIC_WEIGHTS=1)" % source_model_type
        path = source_model_path or default_path
        self.engine = get_engine(device)
        d = net_desc if net_desc is not None else _desc(arch, PRECISIONS[self.precision])
        self.net = self.engine.create_net(d)
        self.weights_origin = load_or_init_weights(self.engine, path, {self.net: ""},
                                                   seed=abs(hash_str(source_model_type)) % 1000)
        self.resolution, self.channels = d.image_size, d.in_channels
        self.latent_dim = self.resolution ** 2 * self.channels * self.es_steps
        self.sched = schedule.PixelSchedule(custom_steps, es_steps, sample_type=sample_type, eta=eta, t_0=self.t_0,
                                            refine_steps=refine_steps)
        self._anchor = torch.nn.Parameter(torch.zeros(1, device=self.engine.device), requires_grad=True)

    def _randn(self, n, shape):
        if self.noise_source is not None:
            return torch.stack([self.noise_source(tuple(shape)) for _ in range(n)], 0).to(self.device, torch.float32)
        if self.noise_on_cpu:
            return torch.stack([torch.randn(shape) for _ in range(n)], 0).to(self.device)
        return torch.randn((n,) + tuple(shape), device=self.device)

    def encode(self, image, class_label=None):
        image = (image - 0.5) * 2.0
        assert image.shape[2] == image.shape[3] == self.resolution
        if self.enforce_class_input:
            assert class_label is not None
            raise NotImplementedError()
        x0 = image.to(self.device, torch.float32)
        bsz = x0.shape[0]
        # draw order: sample_xt's randn_like, then one randn_like per posterior step (:313, :298/303)
        nz = self._randn(self.es_steps, tuple(x0.shape))
        z = self.engine.dpm_encode(self.net, self.sched.kind, x0, self.sched.coef_encode(), noise=nz,
                                   last_uses_x0=False)
        z = z.view(bsz, -1)
        assert z.shape[1] == self.latent_dim
        self._range_guard()
        return z

    def _range_guard(self):
        # 'fp32x3' keeps GroupNorm outputs as scaled fp16 pairs; a value outside that range raises at the engine's next
        # synchronisation point rather than returning a saturated chain. One sync per chain of >= custom_steps forwards.
        if self.precision == "fp32x3":
            self.engine.synchronize()

    def generate(self, z, class_label):
        if self.enforce_class_input:
            assert class_label is not None
            raise NotImplementedError()
        bsz = z.shape[0]
        zz = z.view(bsz, self.es_steps, self.channels, self.resolution, self.resolution).contiguous()
        last = self._randn(1, tuple(zz[:, 0].shape))  # denoising_step's randn_like
        x = self.engine.ddim_decode(self.net, self.sched.kind, zz, self.sched.coef_decode(),
                                    n_eps=self.es_steps - 1, noise_tail=last)
        if self.refine_steps:
            assert self.refine_steps < self.custom_steps
            for _ in range(self.refine_iterations):
                nz = self._randn(self.refine_steps + 1, tuple(x.shape))
                x = self.engine.pix_refine(self.net, self.sched.kind, x, self.sched.coef_refine(), noise=nz)
        self._range_guard()
        return x

    def forward(self, z, class_label=None):
        img = self.generate(z.to(self.device, torch.float32), class_label)
        return (img + 1.0) / 2.0  # post_process Normalize(mean=-1, std=2) (:386-389)

    @property
    def device(self):
        return next(self.parameters()).device


def hash_str(s):
    h = 0
    for ch in s:
        h = (h * 131 + ord(ch)) % (2 ** 31)
    return h
