"""Host-side schedules: the per-step scalar coefficients the fused scheduler kernels consume.

Mirrors, in the reference's own dtypes and operation order:
  LatentDiffusion.register_schedule            ldm/models/diffusion/ddpm.py:117-169
  make_beta_schedule / make_ddim_timesteps / make_ddim_sampling_parameters
                                               ldm/modules/diffusionmodules/util.py:21-75
  DDIMSampler.make_schedule                    ldm/models/diffusion/ddim.py:25-55
  per-step scalars                             ddim.py:570-579, 592-600, 634-645
  DDPMDDIMWrapper step schedule and coefficients
                                               model/gan_wrapper/ddpm_ddim_wrapper.py:283-314, 392-523
Every scalar is evaluated exactly as the reference's [B,1,1,1] fp32 tensors would be, so the HIP
kernels only multiply / add / divide per element.
"""
import numpy as np
import torch

from ._ffi import STEP_COEF_DTYPE

f32 = np.float32
ONE = np.float32(1.0)


# ------------------------------------------------------------------------------------ latent (SD / LDM)
def latent_alphas_cumprod(timesteps=1000, linear_start=0.00085, linear_end=0.0120):
    """'linear' schedule of ddpm.py register_schedule: fp64 betas -> cumprod -> fp32 buffer."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    return torch.tensor(np.cumprod(1. - betas, axis=0), dtype=torch.float32)


class DDIMSchedule:
    """make_schedule(ddim_num_steps=S, ddim_eta=eta): tables indexed by `index` in the sampler loops."""

    def __init__(self, alphas_cumprod, S, eta):
        T = alphas_cumprod.shape[0]
        c = T // S
        self.timesteps = np.asarray(list(range(0, T, c)))[:S] + 1
        ts = self.timesteps
        self.a = alphas_cumprod[ts].numpy().astype(f32)                       # torch fp32 in the reference
        a_prev = np.asarray([float(alphas_cumprod[0])] + alphas_cumprod[ts[:-1]].tolist())  # numpy fp64
        a_t = torch.from_numpy(self.a)
        # fp64 array / fp32 tensor -> Tensor.__rtruediv__ = reciprocal()*other: fp32 reciprocal, fp64 product
        rec = (1 - a_t).reciprocal().double()
        sig = eta * torch.sqrt(rec * torch.from_numpy(1 - a_prev) * (1 - a_t.double() / torch.from_numpy(a_prev)))
        self.a_prev = a_prev.astype(f32)        # rounded by torch.full at use (ddim.py:571)
        self.sigma = sig.numpy().astype(f32)    # ddim.py:572
        self.r = np.sqrt(ONE - self.a)          # ddim_sqrt_one_minus_alphas (ddim.py:50), fp32
        self.eta = eta

    def __len__(self):
        return len(self.timesteps)

    def coef_encode(self, skip_steps=0):
        """K+1 rows: rows 0..K-1 = loop steps by `index`, row K = x_T initialisation (ddim.py:477-479)."""
        K = len(self) - skip_steps
        rows = self._rows(K)
        aK = self.a[K - 1]
        rows.append((np.sqrt(aK), np.sqrt(ONE - aK), 0.0, 0.0, 0.0, 0.0, 1.0, 0))
        return _pack(rows)

    def coef_decode(self, skip_steps=0):
        return _pack(self._rows(len(self) - skip_steps))

    def coef_refine(self, refine_steps):
        """DDIMSampler.refine (ddim.py:114-168, 339-393) on a schedule built with eta = 1: rows 0..R-1 = the R random
        p_sample_ddim steps by `index`, row R = the re-noising to the DDIM level R - 1 (ddim.py:349-351)."""
        assert 0 < refine_steps < len(self)
        rows = self._rows(refine_steps)
        aR = self.a[refine_steps - 1]
        rows.append((np.sqrt(aR), np.sqrt(ONE - aR), 0.0, 0.0, 0.0, 0.0, 1.0, 0))
        return _pack(rows)

    def _rows(self, K):
        rows = []
        for k in range(K):
            a, ap, s = self.a[k], self.a_prev[k], self.sigma[k]
            rows.append((np.sqrt(a), np.sqrt(ONE - a), np.sqrt(ap), np.sqrt(ONE - ap - s * s), s, self.r[k], 1.0,
                         int(self.timesteps[k])))
        return rows


def _pack(rows):
    arr = np.zeros(len(rows), dtype=STEP_COEF_DTYPE)
    for i, r in enumerate(rows):
        arr[i] = tuple(r)
    return arr


# ------------------------------------------------------------------------------------ pixel-space DDPMs
class PixelSchedule:
    """DDPMDDIMWrapper's schedule: betas = linspace(1e-4, 2e-2, 1000) fp64 -> fp32 buffer; alpha-bar is
    re-derived with a fp32 cumprod on every call (ddpm_ddim_wrapper.py:139); logvar = log(max(posterior
    variance, 1e-20)) from the fp64 betas (:373)."""

    def __init__(self, custom_steps, es_steps, sample_type="ddim", eta=0.1, t_0=999, refine_steps=0,
                 beta_start=0.0001, beta_end=0.02, T=1000):
        if sample_type == "ddim":
            assert eta is not None and eta > 0
        elif sample_type == "ddpm":
            assert eta is None
        else:
            raise ValueError(sample_type)
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)
        self.b = torch.from_numpy(betas).float()
        self.acp = (1.0 - self.b).cumprod(dim=0).numpy().astype(f32)
        self.b = self.b.numpy().astype(f32)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        acp = np.append(1.0, ac[:-1])
        self.logvar = np.log(np.maximum(betas * (1.0 - acp) / (1.0 - ac), 1e-20)).astype(f32)
        self.custom_steps, self.es_steps, self.sample_type, self.eta = custom_steps, es_steps, sample_type, eta
        self.refine_steps = refine_steps
        if (t_0 + 1) % custom_steps == 0:
            seq = range(0, t_0 + 1, (t_0 + 1) // custom_steps)
            assert len(seq) == custom_steps
        else:
            seq = np.linspace(0, 1, custom_steps) * t_0
        self.seq = [int(s) for s in list(seq)][:es_steps]
        self.seq_next = ([-1] + list(self.seq[:-1]))[:es_steps]

    @property
    def kind(self):
        from . import _ffi
        return _ffi.CD_SCHED_DDIM if self.sample_type == "ddim" else _ffi.CD_SCHED_DDPM

    def _row(self, i, j, eta):
        at, bt = self.acp[i], self.b[i]
        atn = ONE if j == -1 else self.acp[j]
        if self.sample_type == "ddim":
            c1 = f32(eta) * np.sqrt((ONE - at / atn) * (ONE - atn) / (ONE - at))
            c2 = np.sqrt((ONE - atn) - c1 * c1)
            return (np.sqrt(at), np.sqrt(ONE - at), np.sqrt(atn), c2, c1, np.sqrt(ONE - at), 1.0, i)
        w0 = np.sqrt(atn) * bt / (ONE - at)
        wt = np.sqrt(ONE - bt) * (ONE - atn) / (ONE - at)
        sd = np.sqrt(bt * (ONE - atn) / (ONE - at))
        weight = bt / np.sqrt(ONE - at)
        inv = ONE / np.sqrt(ONE - bt)
        e = torch.exp(0.5 * torch.tensor(self.logvar[i])).numpy().astype(f32)  # torch.exp(0.5*logvar), :205,269
        return (w0, wt, sd, weight, e, inv, 0.0 if i == 0 else 1.0, i)

    def _init_row(self, index):
        at = self.acp[index]
        return (np.sqrt(at), np.sqrt(ONE - at), 0.0, 0.0, 0.0, 0.0, 1.0, 0)

    def coef_encode(self):
        """K = es_steps-1 loop steps (rows 0..K-1 by K-1-it) + the x_T row (raw index es_steps-1, :483-484)."""
        pairs = list(zip(reversed(self.seq), reversed(self.seq_next)))[: self.es_steps - 1]
        K = len(pairs)
        rows = [None] * K
        for it, (i, j) in enumerate(pairs):
            rows[K - 1 - it] = self._row(i, j, self.eta)
        rows.append(self._init_row(self.es_steps - 1))
        return _pack(rows)

    def coef_decode(self):
        """K = es_steps rows; the last loop step has t_next = -1 (alpha-bar_next := 1, :196-199)."""
        pairs = list(zip(reversed(self.seq), reversed(self.seq_next)))
        K = len(pairs)
        rows = [None] * K
        for it, (i, j) in enumerate(pairs):
            rows[K - 1 - it] = self._row(i, j, self.eta)
        return _pack(rows)

    def coef_refine(self):
        """R random DDIM(eta=1) steps after re-noising to t = refine_steps-1 (:431-453)."""
        R = self.refine_steps
        pairs = list(zip(reversed(self.seq[:R]), reversed(self.seq_next[:R])))
        rows = [None] * R
        for it, (i, j) in enumerate(pairs):
            rows[R - 1 - it] = self._row(i, j, 1)
        rows.append(self._init_row(R - 1))
        return _pack(rows)
