"""CPU oracle: schedules and the CycleDiffusion sampler loops, restated in torch fp32 with the
reference's operation order. TEST INFRASTRUCTURE (see oracle/nets.py header for the parity pin).

Follows:
  make_beta_schedule / make_ddim_timesteps / make_ddim_sampling_parameters
      model/lib/stable_diffusion/ldm/modules/diffusionmodules/util.py:21-75
  LatentDiffusion.register_schedule          ldm/models/diffusion/ddpm.py:117-169
  DDIMSampler.make_schedule / _ddpm_ddim_encoding / sample_xt_next / compute_eps /
      ddim_sampling_with_eps / p_sample_ddim_with_eps
      ldm/models/diffusion/ddim.py:25-55, 450-501, 582-601, 545-580, 395-448, 603-646
  DDPMDDIMWrapper.encode / generate and its free functions
      model/gan_wrapper/ddpm_ddim_wrapper.py:114-314, 392-523
  denoising_step / get_beta_schedule / extract
      model/lib/ddpm_ddim/utils/diffusion_utils.py:5-136
All noise is an explicit input, drawn by the caller in the reference's draw order.
"""
import numpy as np
import torch


# ----------------------------------------------------------------------------- latent schedule
def sd_alphas_cumprod(timesteps=1000, linear_start=0.00085, linear_end=0.0120):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1. - betas, axis=0)
    return torch.tensor(ac, dtype=torch.float32)


def ddim_timesteps(S, T=1000):
    c = T // S
    return np.asarray(list(range(0, T, c)))[:S] + 1


def ddim_tables(alphas_cumprod, S, eta):
    """Returns (timesteps int64[S], a_t f32[S], a_prev f64[S], sigma f64[S], sqrt_one_minus_a f32[S])."""
    ts = ddim_timesteps(S, alphas_cumprod.shape[0])
    a = alphas_cumprod[ts]  # fp32
    a_prev = np.asarray([float(alphas_cumprod[0])] + alphas_cumprod[ts[:-1]].tolist())  # fp64 holding fp32 values
    # numpy_f64 / torch_f32 dispatches to Tensor.__rtruediv__ = reciprocal()*other: the reciprocal of
    # (1 - a_t) is rounded to fp32 BEFORE the fp64 product (util.py:69 evaluated with these operand types)
    rec = (1 - a).reciprocal().double()
    sig = (eta * torch.sqrt(rec * torch.from_numpy(1 - a_prev) * (1 - a.double() / torch.from_numpy(a_prev)))).numpy()
    r = torch.sqrt(1. - a)
    return ts, a, a_prev, sig, r


def _full(B, v):
    return torch.full((B, 1, 1, 1), float(v), dtype=torch.float32)


def latent_encode(eps_model, x0, S, eta, noises, skip_steps=0, white_box_steps=100, alphas_cumprod=None):
    """DPM-Encoder. eps_model(x, t_long[B]) -> eps_hat (CFG already folded in by the caller).
    noises: list [n_T, n_{K-1}, ..., n_1] (K-1 step draws; index 0 draws nothing). Returns the z list
    [x_T, eps_{K-1}, ..., eps_0]."""
    ac = sd_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
    ts, a, a_prev, sig, r = ddim_tables(ac, S, eta)
    K = len(ts) - skip_steps
    B = x0.shape[0]
    at = a[K - 1]
    xt = at.sqrt() * x0 + (1 - at).sqrt() * noises[0]
    z = [xt]
    it = 1
    for i in range(K):
        if not i < white_box_steps - skip_steps - 1:
            break
        k = K - i - 1
        t = torch.full((B,), int(ts[k]), dtype=torch.long)
        a_t, a_p, s_t, r_t = _full(B, a[k]), _full(B, a_prev[k]), _full(B, sig[k]), _full(B, r[k])
        if k == 0:
            x_next = x0
        else:
            e_post = (xt - a_t.sqrt() * x0) / (1 - a_t).sqrt()
            x_next = a_p.sqrt() * x0 + (1. - a_p - s_t ** 2).sqrt() * e_post + s_t * noises[it]
            it += 1
        e = eps_model(xt, t)
        pred_x0 = (xt - r_t * e) / a_t.sqrt()
        eps = (x_next - a_p.sqrt() * pred_x0 - (1. - a_p - s_t ** 2).sqrt() * e) / s_t / 1.0
        z.append(eps)
        xt = x_next
    return z


def latent_decode(eps_model, x_T, eps_list, S, eta, skip_steps=0, alphas_cumprod=None, tail_noises=None):
    """Decode with injected eps; eps_list [B, n, ...], n <= K: steps beyond the list use fresh noise (`eps=None` in
    p_sample_ddim_with_eps, ddim.py:437, 640-643), here `tail_noises[i - n]`."""
    ac = sd_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
    ts, a, a_prev, sig, r = ddim_tables(ac, S, eta)
    K = len(ts) - skip_steps
    B = x_T.shape[0]
    x = x_T
    for i in range(K):
        k = K - i - 1
        t = torch.full((B,), int(ts[k]), dtype=torch.long)
        a_t, a_p, s_t, r_t = _full(B, a[k]), _full(B, a_prev[k]), _full(B, sig[k]), _full(B, r[k])
        e = eps_model(x, t)
        pred_x0 = (x - r_t * e) / a_t.sqrt()
        n = eps_list.shape[1] if eps_list is not None else 0
        inj = eps_list[:, i] if i < n else tail_noises[i - n]
        x = a_p.sqrt() * pred_x0 + (1. - a_p - s_t ** 2).sqrt() * e + s_t * inj * 1.0
    return x


def latent_refine(eps_model, x0, S, refine_steps, noises, alphas_cumprod=None):
    """DDIMSampler.refine / _refine (ddim.py:114-168, 339-393): make_schedule(S, eta=1); re-noise x0 to the DDIM
    level refine_steps - 1 with noises[0], then refine_steps random p_sample_ddim steps (ddim.py:503-543) with
    noises[1:] (one draw per step, index 0 included)."""
    ac = sd_alphas_cumprod() if alphas_cumprod is None else alphas_cumprod
    ts, a, a_prev, sig, r = ddim_tables(ac, S, 1.0)
    B = x0.shape[0]
    at = a[refine_steps - 1]
    x = at.sqrt() * x0 + (1 - at).sqrt() * noises[0]
    for i in range(refine_steps):
        k = refine_steps - i - 1
        t = torch.full((B,), int(ts[k]), dtype=torch.long)
        a_t, a_p, s_t, r_t = _full(B, a[k]), _full(B, a_prev[k]), _full(B, sig[k]), _full(B, r[k])
        e = eps_model(x, t)
        pred_x0 = (x - r_t * e) / a_t.sqrt()
        x = a_p.sqrt() * pred_x0 + (1. - a_p - s_t ** 2).sqrt() * e + s_t * noises[1 + i] * 1.0
    return x


def cfg_model(unet_fn, c, uc, scale):
    """ddim.py:550-559: scale==1 -> cond only, 0 -> uncond only, else 2B batch + combine."""
    def f(x, t):
        if uc is None or scale == 1.:
            return unet_fn(x, t, c)
        if scale == 0:
            return unet_fn(x, t, uc)
        e_u, e_c = unet_fn(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uc, c])).chunk(2)
        return e_u + scale * (e_c - e_u)
    return f


# ----------------------------------------------------------------------------- pixel-space wrapper
def pixel_betas(beta_start=0.0001, beta_end=0.02, T=1000):
    return np.linspace(beta_start, beta_end, T, dtype=np.float64)


def pixel_seq(custom_steps, es_steps, t_0=999):
    if (t_0 + 1) % custom_steps == 0:
        seq = range(0, t_0 + 1, (t_0 + 1) // custom_steps)
    else:
        seq = np.linspace(0, 1, custom_steps) * t_0
    seq = [int(s) for s in list(seq)][:es_steps]
    seq_next = ([-1] + list(seq[:-1]))[:es_steps]
    return seq, seq_next


def pixel_logvar(betas64):
    alphas = 1.0 - betas64
    ac = np.cumprod(alphas, axis=0)
    acp = np.append(1.0, ac[:-1])
    pv = betas64 * (1.0 - acp) / (1.0 - ac)
    return np.log(np.maximum(pv, 1e-20))


def _extract(a, t, shape):
    out = torch.gather(torch.as_tensor(a, dtype=torch.float), 0, t.long())
    return out.reshape((t.shape[0],) + (1,) * (len(shape) - 1))


def pixel_encode(net, x0, betas64, custom_steps, es_steps, eta, noises, sample_type="ddim", t_0=999):
    """DDPMDDIMWrapper.encode (ddpm_ddim_wrapper.py:457-523). net(x, t_float[B]) -> eps_hat (first C
    channels already selected). noises: [n_T, n_step...] one per loop step. Returns z list."""
    b = torch.from_numpy(betas64).float()
    logvar = pixel_logvar(betas64)
    seq, seq_next = pixel_seq(custom_steps, es_steps, t_0)
    B = x0.shape[0]
    acp = (1.0 - b).cumprod(dim=0)
    T = torch.ones(B) * (es_steps - 1)
    at = _extract(acp, T, x0.shape)
    xt = at.sqrt() * x0 + (1 - at).sqrt() * noises[0]
    z = [xt]
    for it, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        if not it < es_steps - 1:
            break
        t, tn = torch.ones(B) * i, torch.ones(B) * j
        bt, at, atn = _extract(b, t, x0.shape), _extract(acp, t, x0.shape), _extract(acp, tn, x0.shape)
        nz = noises[1 + it]
        et = net(xt, t)
        if sample_type == "ddpm":
            w0 = atn.sqrt() * bt / (1 - at)
            wt = (1 - bt).sqrt() * (1 - atn) / (1 - at)
            x_next = (w0 * x0 + wt * xt) + (bt * (1 - atn) / (1 - at)).sqrt() * nz
            lv = _extract(logvar, t, x0.shape)
            mean = 1 / torch.sqrt(1.0 - bt) * (xt - bt / torch.sqrt(1 - at) * et)
            eps = (x_next - mean) / torch.exp(0.5 * lv)
        else:
            e_post = (xt - at.sqrt() * x0) / (1 - at).sqrt()
            c1 = eta * ((1 - at / atn) * (1 - atn) / (1 - at)).sqrt()
            c2 = ((1 - atn) - c1 ** 2).sqrt()
            x_next = atn.sqrt() * x0 + c2 * e_post + c1 * nz
            x0_t = (xt - et * (1 - at).sqrt()) / at.sqrt()
            eps = (x_next - atn.sqrt() * x0_t - c2 * et) / c1
        z.append(eps)
        xt = x_next
    return z


def _pixel_step(net, x, t, tn, b, acp, logvar, sample_type, eta, eps):
    """denoising_step_with_eps / denoising_step (noise given explicitly either way)."""
    et = net(x, t)
    bt, at = _extract(b, t, x.shape), _extract(acp, t, x.shape)
    atn = torch.ones_like(at) if tn.sum() == -tn.shape[0] else _extract(acp, tn, x.shape)
    if sample_type == "ddpm":
        lv = _extract(logvar, t, x.shape)
        mean = 1 / torch.sqrt(1.0 - bt) * (x - bt / torch.sqrt(1 - at) * et)
        mask = (1 - (t == 0).float()).reshape((x.shape[0],) + (1,) * (x.dim() - 1))
        return (mean + mask * torch.exp(0.5 * lv) * eps).float()
    x0_t = (x - et * (1 - at).sqrt()) / at.sqrt()
    c1 = eta * ((1 - at / atn) * (1 - atn) / (1 - at)).sqrt()
    c2 = ((1 - atn) - c1 ** 2).sqrt()
    return atn.sqrt() * x0_t + c2 * et + c1 * eps


def pixel_decode(net, z, betas64, custom_steps, es_steps, eta, last_noise, sample_type="ddim", t_0=999,
                 refine_steps=0, refine_noises=None):
    """DDPMDDIMWrapper.generate (ddpm_ddim_wrapper.py:392-455), one sample per call like the reference
    (its 'ddim' branch compares [B,1,1,1] tensors). z: [B, es_steps, C, H, W]."""
    b = torch.from_numpy(betas64).float()
    logvar = pixel_logvar(betas64)
    acp = (1.0 - b).cumprod(dim=0)
    seq, seq_next = pixel_seq(custom_steps, es_steps, t_0)
    B = z.shape[0]
    x = z[:, 0]
    eps_list = z[:, 1:]
    for it, (i, j) in enumerate(zip(reversed(seq), reversed(seq_next))):
        t, tn = torch.ones(B) * i, torch.ones(B) * j
        e = eps_list[:, it] if it < es_steps - 1 else last_noise
        x = _pixel_step(net, x, t, tn, b, acp, logvar, sample_type, eta, e)
    if refine_steps:
        t = torch.ones(B) * refine_steps - 1
        at = _extract(acp, t, x.shape)
        x = at.sqrt() * x + (1 - at).sqrt() * refine_noises[0]
        for n, (i, j) in enumerate(zip(reversed(seq[:refine_steps]), reversed(seq_next[:refine_steps]))):
            t, tn = torch.ones(B) * i, torch.ones(B) * j
            x = _pixel_step(net, x, t, tn, b, acp, logvar, sample_type, 1, refine_noises[1 + n])
    return x
