"""Thin helpers that call the single-kernel C-ABI entry points (cd_op_*) with torch tensors."""
import ctypes as C

import numpy as np
import torch

from cycle_diffusion_amd import _ffi
from cycle_diffusion_amd._ffi import check, ptr


def bf16_round(t):
    """Round an fp32 tensor to the engine's 16-bit storage format (fp16 by default, cd_act_format())."""
    dt = torch.float16 if _ffi.load_library().cd_act_format() == 1 else torch.bfloat16
    return t.to(dt).to(torch.float32)


def dev(t):
    return t.detach().to(torch.float32).contiguous().cuda()


def err_stats(got, ref):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    d = (got - ref).abs()
    scale = ref.abs().max().item() + 1e-12
    return dict(max_abs=d.max().item(), rel_to_max=d.max().item() / scale,
                mean_rel=d.mean().item() / (ref.abs().mean().item() + 1e-12),
                finite=bool(torch.isfinite(got).all().item()))


def pack_conv(eng, w, geglu=False):
    w = w.detach().float().cpu().contiguous()
    if w.dim() == 2:
        w = w[:, :, None, None].contiguous()
    N, Cin, KH, KW = w.shape
    out = C.c_void_p()
    npad, cpad = C.c_int(), C.c_int()
    check(eng.lib.cd_op_pack_conv_weight(eng.h, C.c_void_p(w.data_ptr()), N, Cin, KH, KW, int(geglu),
                                         C.byref(out), C.byref(npad), C.byref(cpad)))
    return out, (N, Cin, KH, KW)


def geglu_pack_vec(v):
    """bias in the packed row order used for GEGLU weights: blocks of 64 = [32 value | 32 gate]."""
    n = v.shape[0]
    half = n // 2
    out = torch.empty_like(v)
    for j in range(n):
        blk, within = divmod(j, 64)
        src = blk * 32 + within if within < 32 else half + blk * 32 + (within - 32)
        out[j] = v[src]
    return out


def conv2d(eng, x0, w, x1=None, stride=1, pad=1, asym=False, up=False, bias=None, rowvec=None, resid=None,
           act=0, tile=0, geglu=False):
    handle, (N, Cin, KH, KW) = pack_conv(eng, w, geglu)
    B, C0, H, W = x0.shape
    C1 = x1.shape[1] if x1 is not None else 0
    Hin, Win = (2 * H, 2 * W) if up else (H, W)
    if asym:
        Ho, Wo = (Hin + 1 - KH) // stride + 1, (Win + 1 - KW) // stride + 1
    else:
        Ho, Wo = (Hin + 2 * pad - KH) // stride + 1, (Win + 2 * pad - KW) // stride + 1
    Nout = N // 2 if geglu else N
    y = torch.empty((B, Nout, Ho, Wo), device="cuda", dtype=torch.float32)
    b = None
    if bias is not None:
        b = dev(geglu_pack_vec(bias) if geglu else bias)
    xs0, xs1 = dev(x0), dev(x1) if x1 is not None else None
    rv, rs = dev(rowvec) if rowvec is not None else None, dev(resid) if resid is not None else None
    check(eng.lib.cd_op_conv2d(eng.h, ptr(xs0), C0, ptr(xs1), C1, B, H, W, handle, N, KH, KW, stride, pad,
                               int(asym), int(up), ptr(b), ptr(rv), ptr(rs), act, tile, ptr(y)))
    torch.cuda.synchronize()
    return y.cpu()


def conv2d16(eng, x0, w, stride=1, pad=1, bias=None, resid=None, act=0, tile=0, geglu=False, want_stats=False, rowvec=None):
    """cd_op_conv2d_16: the convolution with the engine's 16-bit output (+ the fused GroupNorm statistics)"""
    handle, (N, Cin, KH, KW) = pack_conv(eng, w, geglu)
    B, C0, H, W = x0.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    Nout = N // 2 if geglu else N
    y = torch.empty((B, Nout, Ho, Wo), device="cuda", dtype=torch.float32)
    st = torch.zeros((B * Ho * Wo // 32, 2, Nout), device="cuda", dtype=torch.float32) if want_stats else None
    b = None
    if bias is not None:
        b = dev(geglu_pack_vec(bias) if geglu else bias)
    xs0 = dev(x0)
    rs = dev(resid) if resid is not None else None
    rv = dev(rowvec) if rowvec is not None else None
    check(eng.lib.cd_op_conv2d_16(eng.h, ptr(xs0), C0, None, 0, B, H, W, handle, N, KH, KW, stride, pad, 0, 0, ptr(b),
                                  ptr(rv), ptr(rs), act, tile, ptr(y), ptr(st)))
    torch.cuda.synchronize()
    return (y.cpu(), st.cpu()) if want_stats else y.cpu()


def groupnorm(eng, x, gamma, beta, eps, silu=False, film=None):
    B, Cc, H, W = x.shape
    y = torch.empty_like(x, device="cuda")
    xs, g, b = dev(x), dev(gamma), dev(beta)
    f = dev(film) if film is not None else None
    check(eng.lib.cd_op_groupnorm(eng.h, ptr(xs), B, Cc, H, W, 32, C.c_float(eps), ptr(g), ptr(b), ptr(f),
                                  int(silu), ptr(y)))
    torch.cuda.synchronize()
    return y.cpu()


def layernorm(eng, x, gamma, beta, eps=1e-5):
    rows, Cc = x.shape
    y = torch.empty_like(x, device="cuda")
    xs, g, b = dev(x), dev(gamma), dev(beta)
    check(eng.lib.cd_op_layernorm(eng.h, ptr(xs), rows, Cc, ptr(g), ptr(b), C.c_float(eps), ptr(y)))
    torch.cuda.synchronize()
    return y.cpu()


def attention(eng, q, k, v, heads, scale, v_transposed=False):
    B, Tq, Cc = q.shape
    Tk = k.shape[1]
    o = torch.empty_like(q, device="cuda")
    qs, ks, vs = dev(q), dev(k), dev(v)
    check(eng.lib.cd_op_attention(eng.h, ptr(qs), ptr(ks), ptr(vs), B, heads, Tq, Tk, Cc // heads,
                                  C.c_float(scale), 1 if v_transposed else 0, ptr(o)))
    torch.cuda.synchronize()
    return o.cpu()


def softmax_rows(eng, s):
    rows, cols = s.shape
    p = torch.empty_like(s, device="cuda")
    ss = dev(s)
    check(eng.lib.cd_op_softmax_rows(eng.h, ptr(ss), rows, cols, ptr(p)))
    torch.cuda.synchronize()
    return p.cpu()


def timestep_embedding(eng, t, dim, mode):
    B = t.shape[0]
    out = torch.empty((B, dim), device="cuda", dtype=torch.float32)
    ts = dev(t)
    check(eng.lib.cd_op_timestep_embedding(eng.h, ptr(ts), B, dim, mode, ptr(out)))
    torch.cuda.synchronize()
    return out.cpu()


def sched_step(eng, mode, kind, coef_row, x0=None, xt=None, eps_hat=None, cfg=False, g=1.0, noise=None,
               eps_in=None, is_last=False):
    coef = _ffi.coef_array([coef_row])
    ref = xt if xt is not None else x0
    B, Cc, H, W = ref.shape
    xt_d = dev(xt) if xt is not None else torch.empty((B, Cc, H, W), device="cuda")
    z = torch.zeros((B, Cc, H, W), device="cuda")
    x0_d = dev(x0) if x0 is not None else None
    eh = dev(eps_hat) if eps_hat is not None else None
    nz = dev(noise) if noise is not None else None
    ei = dev(eps_in) if eps_in is not None else None
    check(eng.lib.cd_op_sched_step(eng.h, mode, kind, C.c_void_p(coef.ctypes.data), ptr(x0_d), ptr(xt_d), ptr(eh),
                                   int(cfg), C.c_float(g), ptr(nz), ptr(ei), int(is_last), B, Cc, H * W, ptr(z)))
    torch.cuda.synchronize()
    return xt_d.cpu(), z.cpu()


def probe(eng, which, nfloats):
    out = torch.zeros(nfloats, device="cuda", dtype=torch.float32)
    check(eng.lib.cd_op_probe(eng.h, which, None, ptr(out), nfloats * 4))
    torch.cuda.synchronize()
    return out.cpu().numpy()
