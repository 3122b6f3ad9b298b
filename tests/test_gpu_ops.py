"""Per-kernel parity: every HIP kernel (called through the C ABI) against a plain PyTorch fp32
reference of the same op on identical, bf16-representable inputs.

Tolerances: the kernels keep bf16 operands / fp32 accumulation and round outputs to bf16 once, so
max-abs error is bounded by ~2^-8 of the output scale (REL_TOL) and the mean error by MEAN_TOL.
Scheduler kernels are fp32 with the reference's operation order and must be bit-exact.
"""
import math
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import _ops
from _ops import bf16_round as r16

pytestmark = pytest.mark.gpu

REL_TOL = 2.0e-2   # max |err| / max |ref|
MEAN_TOL = 6.0e-3  # mean |err| / mean |ref|


def _check(report, name, got, ref, rel=REL_TOL, mean=MEAN_TOL):
    st = _ops.err_stats(got, ref)
    report.add(name, **st)
    assert st["finite"], name
    assert st["rel_to_max"] < rel, (name, st)
    assert st["mean_rel"] < mean, (name, st)


def test_probe_mfma_layout(engine, report):
    out = _ops.probe(engine, 0, 3072).reshape(3, 64, 16)
    lane = np.arange(64)[:, None]
    reg = np.arange(16)[None, :]
    rows = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    cols = np.broadcast_to(lane & 31, (64, 16))
    report.add("probe_mfma", rowmap_ok=bool((out[0] == rows + 1).all()), colmap_ok=bool((out[1] == cols + 1).all()),
               ksum=float(out[2].mean()))
    assert (out[0] == rows + 1).all(), out[0][:4]
    assert (out[1] == cols + 1).all(), out[1][:4]
    assert (out[2] == 136.0).all()


def test_probe_tr_read(engine, report):
    # informational: records the ds_read_b64_tr_b16 lane->element map for the attention V^T path
    out = _ops.probe(engine, 1, 256).reshape(64, 4)
    report.add("probe_tr_b16", lanes0_3=out[:4].tolist(), lane16=out[16].tolist(), lane63=out[63].tolist())
    assert np.isfinite(out).all()


def test_probe_tr_read_attention_pattern(engine, report):
    # the LDS transpose read as k_attention addresses it (token-major V tile, row stride 96): lane l must receive, for
    # its d column (dt*32 + l%32), the four keys base + 4*(l//32) + 0..3 and the four 8 keys further on
    out = _ops.probe(engine, 2, 1024).reshape(64, 2, 2, 4)  # lane, sel (kh=s2=dt=sel), piece, j
    lane = np.arange(64)[:, None, None, None]
    sel = np.arange(2)[None, :, None, None]
    piece = np.arange(2)[None, None, :, None]
    j = np.arange(4)[None, None, None, :]
    key = sel * 32 + 16 * sel + 8 * piece + 4 * (lane >> 5) + j
    d = sel * 32 + (lane & 31)
    want = key * 64 + d
    report.add("probe_tr_attn", ok=bool((out == want).all()), lane0=out[0].reshape(-1).tolist(),
               lane37=out[37].reshape(-1).tolist())
    assert (out == want).all(), (out[0], want[0], out[37], want[37])


CONV_CASES = [
    # name, B, C0, C1, H, W, N, k, stride, pad, asym, up, bias, rowvec, resid, act, tile
    ("3x3_64_64_16", 2, 64, 0, 16, 16, 64, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("3x3_320_320_16_t1", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 1),
    ("3x3_320_320_16_t2", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 2),
    ("3x3_320_320_16_t3", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 3),
    ("3x3_stride2_pad1", 2, 128, 0, 16, 16, 128, 3, 2, 1, False, False, True, False, False, 0, 0),
    ("3x3_stride2_asym", 2, 128, 0, 16, 16, 128, 3, 2, 0, True, False, True, False, False, 0, 0),
    ("3x3_upsample", 2, 128, 0, 8, 8, 128, 3, 1, 1, False, True, True, False, False, 0, 0),
    ("3x3_concat_k64", 2, 128, 64, 16, 16, 192, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("3x3_concat_k32", 2, 64, 32, 16, 16, 96, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("1x1_concat_skip", 2, 128, 64, 16, 16, 64, 1, 1, 0, False, False, True, False, False, 0, 0),
    ("1x1_silu", 1, 256, 0, 12, 12, 256, 1, 1, 0, False, False, True, False, False, 1, 0),
    ("1x1_gelu", 1, 64, 0, 8, 8, 96, 1, 1, 0, False, False, True, False, False, 2, 0),
    ("3x3_cin4_pad32", 2, 4, 0, 16, 16, 64, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("3x3_cout4", 2, 64, 0, 16, 16, 4, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("3x3_cout3_cin128", 1, 128, 0, 24, 24, 3, 3, 1, 1, False, False, True, False, False, 0, 0),
    ("3x3_ragged_M", 1, 64, 0, 10, 10, 64, 3, 1, 1, False, False, True, False, True, 0, 0),
    ("3x3_1280_8x8", 2, 1280, 0, 8, 8, 1280, 3, 1, 1, False, False, True, True, True, 0, 0),
    ("3x3_2560_1280_skip", 1, 1280, 1280, 8, 8, 1280, 3, 1, 1, False, False, True, False, False, 0, 0),
    # 320- / 256-wide block tiles (chunked epilogue, 64x160 and 64x128 wave tiles); ragged M, N = 640 (two 320 tiles),
    # N not a multiple of the tile (masked columns), residual + time-embedding row vector, split-K on top
    ("3x3_320_320_16_t20", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 20),
    ("3x3_320_320_16_t21", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 21),
    ("3x3_320_320_16_t23", 2, 320, 0, 16, 16, 320, 3, 1, 1, False, False, True, True, True, 0, 23),
    ("3x3_320_640_t22", 2, 320, 0, 16, 16, 640, 3, 1, 1, False, False, True, True, True, 0, 22),
    ("1x1_640_640_t20_silu", 3, 640, 0, 10, 10, 640, 1, 1, 0, False, False, True, False, True, 1, 20),
    ("3x3_concat_320_t21_ragged", 1, 192, 128, 10, 10, 320, 3, 1, 1, False, False, True, False, True, 0, 21),
    ("3x3_N192_t22_masked", 2, 64, 0, 16, 16, 192, 3, 1, 1, False, False, True, False, False, 0, 22),
    ("3x3_1280_8x8_t20_split4", 2, 1280, 0, 8, 8, 1280, 3, 1, 1, False, False, True, True, True, 0, 20 | (4 << 8)),
    ("3x3_up_640_t23", 1, 640, 0, 8, 8, 640, 3, 1, 1, False, True, True, False, False, 0, 23),
    # a patch-embedding conv (CLIP vision tower: kernel = stride = patch): 256 taps - more than the 8 x 8 tap masks of the
    # channel-major K order hold, so it must take the tap-major order (round 4: the first mask version broke this layer)
    ("16x16_patch_stride16", 2, 32, 0, 64, 64, 96, 16, 16, 0, False, False, True, False, False, 0, 0),
    ("5x5_pad2", 1, 64, 0, 12, 12, 64, 5, 1, 2, False, False, True, False, False, 0, 0),
    # split-K (tile | split << 8): K ranges that start mid-tap / in the second concat source, ragged M,
    # more splits than K steps (empty ranges), fused epilogue after the fix-up
    ("3x3_1280_8x8_split4", 2, 1280, 0, 8, 8, 1280, 3, 1, 1, False, False, True, True, True, 0, 8 | (4 << 8)),
    ("3x3_2560_1280_skip_split7", 1, 1280, 1280, 8, 8, 1280, 3, 1, 1, False, False, True, False, False, 0, 2 | (7 << 8)),
    ("3x3_ragged_M_split2", 1, 64, 0, 10, 10, 64, 3, 1, 1, False, False, True, False, True, 0, 3 | (2 << 8)),
    ("3x3_ragged_M_split8", 1, 64, 0, 10, 10, 64, 3, 1, 1, False, False, True, False, True, 0, 1 | (8 << 8)),
    ("3x3_concat_k32_split3", 2, 64, 32, 16, 16, 96, 3, 1, 1, False, False, True, False, False, 0, 2 | (3 << 8)),
    ("3x3_stride2_split2_silu", 2, 128, 0, 16, 16, 128, 3, 2, 1, False, False, True, False, False, 1, 3 | (2 << 8)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(engine, report, case):
    name, B, C0, C1, H, W, N, k, stride, pad, asym, up, has_bias, has_rv, has_res, act, tile = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
    x0 = r16(torch.randn(B, C0, H, W, generator=g))
    x1 = r16(torch.randn(B, C1, H, W, generator=g)) if C1 else None
    Cin = C0 + C1
    w = r16(torch.randn(N, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    bias = torch.randn(N, generator=g) * 0.5 if has_bias else None
    x = torch.cat([x0, x1], 1) if C1 else x0
    if up:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if asym:
        x = F.pad(x, (0, 1, 0, 1))
    ref = F.conv2d(x, w, bias, stride=stride, padding=0 if asym else pad)
    rv = torch.randn(B, N, generator=g) if has_rv else None
    if rv is not None:
        ref = ref + rv[:, :, None, None]
    if act == 1:
        ref = F.silu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    res = r16(torch.randn(ref.shape, generator=g)) if has_res else None
    if res is not None:
        ref = ref + res
    got = _ops.conv2d(engine, x0, w, x1=x1, stride=stride, pad=pad, asym=asym, up=up, bias=bias, rowvec=rv,
                      resid=res, act=act, tile=tile)
    # the op wrapper returns the fp32 epilogue result (out_f32 path): only operand rounding remains
    _check(report, "conv2d/" + name, got, ref, rel=5e-3, mean=2e-3)


def test_conv_splitk_repeatable(engine, report):
    # the last-arriving block sums the partial tiles in split order and resets the tile counter: repeated
    # launches (and different tile configurations with the same split) give identical bits
    g = torch.Generator().manual_seed(11)
    x = r16(torch.randn(2, 640, 8, 8, generator=g))
    w = r16(torch.randn(640, 640, 3, 3, generator=g) / math.sqrt(640 * 9))
    outs = [_ops.conv2d(engine, x, w, pad=1, tile=t | (5 << 8)) for t in (2, 2, 3, 8, 2)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = F.conv2d(x, w, None, padding=1)
    _check(report, "conv2d/splitk_repeat", outs[0], ref, rel=5e-3, mean=2e-3)


# ---- the 16-bit epilogue of k_conv_gemm (what every ResBlock / transformer projection runs; test_conv2d above goes through
# the fp32-output path): stores and residual loads through row-bounded buffer descriptors, bias / time-embedding rows from the
# LDS tables the prologue copies, GroupNorm statistics per 32-row block - on every tile family, with and without each operand,
# rows beyond M, an image size that is not a power of two (the tables' row index is then a division), columns beyond N
FAST_EPI_CASES = [
    # name, B, C, H, W, N, k, bias, rowvec, resid, stats, tile
    ("t20_all", 2, 64, 16, 16, 320, 3, True, True, True, True, 20),
    ("t20_plain", 2, 64, 16, 16, 640, 1, False, False, False, False, 20),
    ("t20_rowvec_only", 3, 64, 8, 8, 320, 1, False, True, False, True, 20),
    ("t20_resid_ragged", 3, 64, 10, 10, 320, 3, True, False, True, False, 20),
    ("t20_rows96_per_image", 5, 64, 8, 12, 320, 1, True, True, True, True, 20),
    ("t20_N200_masked", 2, 64, 16, 16, 200, 1, True, True, True, True, 20),
    ("t23_all", 2, 64, 16, 16, 320, 3, True, True, True, True, 23),
    ("t21_all", 2, 64, 16, 16, 320, 1, True, True, True, True, 21),
    ("t22_all", 2, 64, 16, 16, 512, 1, True, True, True, True, 22),
    ("t22_rows96", 3, 64, 8, 12, 256, 3, True, True, False, True, 22),
    ("t5_all", 2, 64, 16, 16, 128, 3, True, True, True, True, 5),
    ("t6_all", 2, 64, 16, 16, 256, 1, True, True, True, True, 6),
    ("t18_16waves", 2, 64, 16, 16, 128, 1, True, True, True, True, 18),
    ("t19_16waves", 2, 64, 16, 16, 128, 3, True, True, True, True, 19),
    ("t1_all", 2, 64, 16, 16, 128, 3, True, True, True, True, 1),
    ("t3_rows96_ragged_tile", 3, 64, 8, 12, 64, 1, True, True, True, True, 3),
    ("t9_no_tables", 2, 64, 16, 16, 64, 1, True, True, True, True, 9),
    ("t24_two_per_cu", 2, 64, 16, 16, 256, 1, True, True, True, True, 24),
    ("t20_split3", 2, 320, 8, 8, 320, 3, True, True, True, True, 20 | (3 << 8)),
]


@pytest.mark.parametrize("case", FAST_EPI_CASES, ids=[c[0] for c in FAST_EPI_CASES])
def test_conv2d_16bit_epilogue(engine, report, case):
    name, B, C, H, W, N, k, has_bias, has_rv, has_res, stats, tile = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
    x = r16(torch.randn(B, C, H, W, generator=g))
    w = r16(torch.randn(N, C, k, k, generator=g) / math.sqrt(C * k * k))
    bias = torch.randn(N, generator=g) * 0.5 if has_bias else None
    rv = torch.randn(B, N, generator=g) if has_rv else None
    ref = F.conv2d(x, w, bias, padding=k // 2)
    if rv is not None:
        ref = ref + rv[:, :, None, None]
    res = r16(torch.randn(ref.shape, generator=g)) if has_res else None
    if res is not None:
        ref = ref + res
    out = _ops.conv2d16(engine, x, w, pad=k // 2, bias=bias, rowvec=rv, resid=res, tile=tile, want_stats=stats)
    got, st = out if stats else (out, None)
    _check(report, "conv2d16/" + name, got, ref, rel=5e-3, mean=2e-3)
    if stats:  # sums over 32-row blocks of the NHWC row order, of the values before their 16-bit rounding
        rows = ref.permute(0, 2, 3, 1).reshape(-1, N).double()
        assert rows.shape[0] % 32 == 0
        blk = rows.reshape(-1, 32, N)
        want = torch.stack([blk.sum(1), (blk * blk).sum(1)], 1).float()
        err = (st - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        report.add("conv2d16_stats/" + name, rel=err)
        assert err < 2e-3, (name, err)


def test_conv_geglu(engine, report):
    g = torch.Generator().manual_seed(7)
    M, K, N = 512, 320, 2560  # GEGLU.proj: dim -> 2*inner (attention.py:37-44)
    x = r16(torch.randn(1, K, M, 1, generator=g))
    w = r16(torch.randn(N, K, generator=g) / math.sqrt(K))
    b = torch.randn(N, generator=g) * 0.3
    h = F.linear(x[0, :, :, 0].t(), w, b)
    val, gate = h.chunk(2, dim=-1)
    ref = (val * F.gelu(gate)).t()[None, :, :, None]
    for tile in (1, 2, 22, 6):
        got = _ops.conv2d(engine, x, w, pad=0, bias=b, geglu=True, tile=tile)
        _check(report, "conv2d/geglu_t%d" % tile, got, ref, rel=5e-3, mean=3e-3)


# ---- the streaming K = 320 linear kernel (csrc/lin_stream.hip, tile id 30): the 1x1 / nn.Linear layers of the
# 320-channel level (attention.py:37-44,171-200,211-215; proj_in / proj_out). Every case is checked against torch fp32
# on 16-bit-rounded operands AND bit for bit against a conv_gemm.hip tile on the same operands (same k-ascending
# reduction, same epilogue arithmetic: the autotuner may pick either without changing a result).
LIN_STREAM_CASES = [
    # name, B, H, W, N, bias, resid, geglu, stats, conv_gemm tile to compare with
    ("n320_bias", 1, 32, 32, 320, True, False, False, False, 2),
    ("n320_resid_ragged_strip", 3, 16, 24, 320, True, True, False, False, 20),
    ("n640_qk_nobias", 1, 32, 32, 640, False, False, False, False, 1),
    ("n960_qkv", 2, 16, 16, 960, False, False, False, False, 2),
    ("n2560_geglu", 1, 32, 32, 2560, True, False, True, False, 22),
    ("n320_resid_stats", 2, 32, 32, 320, True, True, False, True, 20),
    ("n320_stats_only", 1, 32, 16, 320, True, False, False, True, 2),
    # more strips than CUs: workgroups 0-15 take a second strip (A prefetched during the first, register hand-over)
    ("n320_resid_272_strips", 17, 64, 64, 320, True, True, False, False, 20),
    ("n2560_geglu_272_strips", 17, 64, 64, 2560, True, False, True, False, 22),
    ("n640_600_strips", 75, 32, 64, 640, False, False, False, False, 20),
]


@pytest.mark.parametrize("case", LIN_STREAM_CASES, ids=[c[0] for c in LIN_STREAM_CASES])
def test_lin_stream(engine, report, case):
    name, B, H, W, N, has_bias, has_res, geglu, stats, ref_tile = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
    K = 320
    x = r16(torch.randn(B, K, H, W, generator=g))
    w = r16(torch.randn(N, K, 1, 1, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.5 if has_bias else None
    Nout = N // 2 if geglu else N
    res = r16(torch.randn(B, Nout, H, W, generator=g)) if has_res else None
    kw = dict(pad=0, bias=bias, resid=res, geglu=geglu, want_stats=stats)
    got = _ops.conv2d16(engine, x, w, tile=30, **kw)
    alt = _ops.conv2d16(engine, x, w, tile=ref_tile, **kw)
    if stats:
        (got, st), (alt, st_alt) = got, alt
    assert torch.equal(got, alt), (got - alt).abs().max().item()
    big = B * H * W > 20000
    if not big or not geglu:  # the large GEGLU case is pinned by the bit comparison alone (115 GFLOP on the host)
        ref = F.conv2d(x, w, bias)
        if geglu:
            val, gate = ref.chunk(2, dim=1)
            ref = val * F.gelu(gate)
        if res is not None:
            ref = ref + res
        _check(report, "lin_stream/" + name, got, ref, rel=5e-3, mean=2e-3)
        if stats:
            rows = ref.permute(0, 2, 3, 1).reshape(-1, 32, Nout)  # 32-row blocks of the [M][N] matrix
            want = torch.stack([rows.sum(1), (rows * rows).sum(1)], 1)
            scale = want.abs().max().item()
            assert (st - want).abs().max().item() < 2e-3 * scale, (st - want).abs().max().item() / scale
            assert (st_alt - want).abs().max().item() < 2e-3 * scale


@pytest.mark.parametrize("geglu", [False, True], ids=["plain", "geglu"])
def test_lin_stream_layernorm_fold(engine, report, geglu):
    """LayerNorm folded into the projection (attention.py:211-215: attn2(norm2(x)), ff(norm3(x))): the kernel normalises
    the rows in registers (statistics only), gain and bias travel in the weights: y = LN(x; gamma, beta) W^T + b =
    ((x - mean) rstd) (W gamma)^T + (b + W beta). 65536 rows (the 64 x 64 level at batch 16)."""
    g = torch.Generator().manual_seed(77 + int(geglu))
    B, H, W, K = 16, 64, 64, 320
    N = 2560 if geglu else 320
    x = r16(torch.randn(B, K, H, W, generator=g) * 1.7 + 0.3 * torch.randn(B, 1, H, W, generator=g))
    w = r16(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.5
    gamma, beta = 1.0 + 0.2 * torch.randn(K, generator=g), 0.3 * torch.randn(K, generator=g)
    rows = x.permute(0, 2, 3, 1).reshape(-1, K)
    ref = F.linear(F.layer_norm(rows, (K,), gamma, beta, 1e-5), w, bias)
    if geglu:
        val, gate = ref.chunk(2, dim=1)
        ref = val * F.gelu(gate)
    w_f = r16(w * gamma[None, :])             # what k_fold_ln produces
    b_f = bias + w @ beta
    got = _ops.conv2d16(engine, x, w_f[:, :, None, None], pad=0, bias=b_f, geglu=geglu, tile=30, act=0x400)
    got = got.permute(0, 2, 3, 1).reshape(-1, ref.shape[1])
    _check(report, "lin_stream/ln_fold_%s" % ("geglu" if geglu else "plain"), got, ref, rel=8e-3, mean=4e-3)


GN_CASES = [("c320", 2, 320, 16, 16, 1e-5, True, False), ("c64", 2, 64, 8, 8, 1e-6, False, False),
            ("c2560", 1, 2560, 8, 8, 1e-5, True, False), ("c128_film", 2, 128, 16, 16, 1e-5, True, True),
            ("c32", 1, 32, 32, 32, 1e-6, True, False), ("c1920", 1, 1920, 16, 16, 1e-5, True, False),
            ("c128_big", 1, 128, 96, 96, 1e-6, True, False)]


@pytest.mark.parametrize("case", GN_CASES, ids=[c[0] for c in GN_CASES])
def test_groupnorm(engine, report, case):
    name, B, C, H, W, eps, silu, film = case
    g = torch.Generator().manual_seed(11)
    x = r16(torch.randn(B, C, H, W, generator=g) * 2.0 + 0.5)
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gamma, beta, eps)
    fl = None
    if film:
        fl = 0.3 * torch.randn(B, 2 * C, generator=g)
        scale, shift = fl.chunk(2, dim=1)
        ref = ref * (1 + scale[:, :, None, None]) + shift[:, :, None, None]
    if silu:
        ref = F.silu(ref)
    got = _ops.groupnorm(engine, x, gamma, beta, eps, silu=silu, film=fl)
    _check(report, "groupnorm/" + name, got, ref, rel=1.5e-2, mean=4e-3)


@pytest.mark.parametrize("C,rows", [(320, 300), (640, 301), (1280, 297), (64, 3), (768, 77), (2048, 35)])
def test_layernorm(engine, report, C, rows):
    # rows: not multiples of the 16 / 8 rows a workgroup covers (4 waves x 4 or 2 rows); C: 1 - 4 vectors per lane
    g = torch.Generator().manual_seed(13)
    x = r16(torch.randn(rows, C, generator=g) * 1.5 + 0.3)
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    got = _ops.layernorm(engine, x, gamma, beta)
    _check(report, "layernorm/c%d" % C, got, ref, rel=1.5e-2, mean=4e-3)


def test_16bit_stores_saturate(engine, report):
    """16-bit activation stores saturate at the format's largest finite value instead of
    overflowing to inf (fp16: +-65504). A LayerNorm with a huge gain drives every output past the range."""
    g = torch.Generator().manual_seed(29)
    x = r16(torch.randn(33, 320, generator=g))
    got = _ops.layernorm(engine, x, torch.full((320,), 3.0e5), torch.zeros(320))
    ref = F.layer_norm(x, (320,), torch.full((320,), 3.0e5), torch.zeros(320), 1e-5)
    fmt_max = 65504.0 if engine.lib.cd_act_format() == 1 else 3.3895313892515355e38
    report.add("saturation/layernorm", finite=bool(torch.isfinite(got).all()), max=float(got.abs().max()))
    assert torch.isfinite(got).all()
    assert float(got.abs().max()) <= fmt_max
    if engine.lib.cd_act_format() == 1:
        big = ref.abs() > 7.0e4
        assert big.any() and torch.equal(got[big], torch.sign(ref[big]) * fmt_max)


ATTN_CASES = [("d40_self", 2, 8, 256, 256, 40), ("d80_self", 1, 8, 256, 256, 80), ("d160_self", 1, 8, 64, 64, 160),
              ("d40_cross77", 2, 8, 256, 77, 40), ("d160_cross77", 1, 8, 64, 77, 160), ("d64_iddpm", 1, 4, 256, 256, 64),
              ("d32_1head", 1, 1, 64, 64, 32), ("d128_1head", 1, 1, 320, 320, 128), ("d40_long", 1, 2, 1024, 1024, 40)]


@pytest.mark.parametrize("vt", [False, True], ids=["v_token_major", "v_transposed"])
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(engine, report, case, vt):
    # both V layouts of k_attention: token-major (fused q|k|v output, LDS transpose reads) and pre-transposed V^T
    name, B, H, Tq, Tk, D = case
    g = torch.Generator().manual_seed(17)
    C = H * D
    q = r16(torch.randn(B, Tq, C, generator=g))
    k = r16(torch.randn(B, Tk, C, generator=g))
    v = r16(torch.randn(B, Tk, C, generator=g))
    scale = D ** -0.5
    qh = q.view(B, Tq, H, D).transpose(1, 2)
    kh = k.view(B, Tk, H, D).transpose(1, 2)
    vh = v.view(B, Tk, H, D).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    ref = (att @ vh).transpose(1, 2).reshape(B, Tq, C)
    got = _ops.attention(engine, q, k, v, H, scale, v_transposed=vt)
    # measured (fp16 storage): rel_to_max 3-10e-4, mean_rel 3-4e-4; bf16 storage is 8x coarser
    f = 1.0 if engine.lib.cd_act_format() == 1 else 8.0
    _check(report, "attention/%s/%s" % (name, "vt" if vt else "v"), got, ref, rel=4e-3 * f, mean=1.5e-3 * f)


@pytest.mark.parametrize("vt", [False, True], ids=["v_token_major", "v_transposed"])
@pytest.mark.parametrize("case", ["wide", "spike", "cross_spike", "wide_long", "spike_long"])
def test_attention_deferred_max(engine, report, case, vt):
    """The attention kernel keeps a stale reference maximum and only moves it when a row's scores outgrow it by 2^8
    (attn.hip). The move is a rare, data-dependent branch: bounded random scores never take it after the first
    tile, so these inputs force it - a wide score distribution (log2-unit std ~6: the maximum of almost every row
    grows past the threshold several times over 16 tiles) and single spiked keys late in the sequence, placed so
    that some rows of a 32-query wave move their maximum while their neighbours do not. The *_long cases (1024 queries,
    V pre-transposed) run on the 256-query (8-wave) workgroups of the 64 x 64 level."""
    g = torch.Generator().manual_seed(23)
    B, H, D = 1, 2, 40
    long_case = case.endswith("_long")
    case = case.replace("_long", "")
    Tq, Tk = (1024 if long_case else 512, 1024) if case != "cross_spike" else (256, 77)
    C = H * D
    amp = 2.0 if case == "wide" else 1.0
    q = torch.randn(B, Tq, C, generator=g) * amp
    k = torch.randn(B, Tk, C, generator=g) * amp
    v = torch.randn(B, Tk, C, generator=g)
    if case != "wide":
        for qrow, krow, gain in ((5, Tk - 3, 9.0), (37, Tk // 2 + 1, 14.0), (200, 70, 11.0)):
            for h in range(H):
                d = q[0, qrow, h * D:(h + 1) * D]
                k[0, krow, h * D:(h + 1) * D] = d / d.norm() * gain * (h + 1)
    q, k, v = r16(q), r16(k), r16(v)
    scale = D ** -0.5
    qh = q.view(B, Tq, H, D).transpose(1, 2).double()
    kh = k.view(B, Tk, H, D).transpose(1, 2).double()
    vh = v.view(B, Tk, H, D).transpose(1, 2).double()
    att = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    ref = (att @ vh).transpose(1, 2).reshape(B, Tq, C).float()
    got = _ops.attention(engine, q, k, v, H, scale, v_transposed=vt)
    f = 1.0 if engine.lib.cd_act_format() == 1 else 8.0  # measured: rel_to_max 4-10e-4, mean_rel 3-5e-4 (fp16)
    _check(report, "attention_deferred_max/%s%s/%s" % (case, "_long" if long_case else "", "vt" if vt else "v"), got, ref,
           rel=4e-3 * f, mean=2e-3 * f)


@pytest.mark.parametrize("vt", [False, True], ids=["v_token_major", "v_transposed"])
def test_attention_ragged_tail_is_zero_filled(engine, report, vt):
    """Keys beyond Tk in the last 64-key tile must arrive as zeros from the buffer descriptor's bounds check, not as
    whatever follows the tensor: 77 keys (the cross-attention length), batch of 2, and the SECOND batch element's K / V
    all NaN - for the first element the rows 77 ... 127 of its ragged tile are exactly those NaN rows if the check does
    not cover the tile offset (0 * NaN would poison every output of element 0)."""
    g = torch.Generator().manual_seed(31)
    B, H, D, Tq, Tk = 2, 8, 40, 128, 77
    C = H * D
    q = r16(torch.randn(B, Tq, C, generator=g))
    k = r16(torch.randn(B, Tk, C, generator=g))
    v = r16(torch.randn(B, Tk, C, generator=g))
    k[1] = float("nan")
    v[1] = float("nan")
    scale = D ** -0.5
    qh = q[:1].view(1, Tq, H, D).transpose(1, 2)
    kh = k[:1].view(1, Tk, H, D).transpose(1, 2)
    vh = v[:1].view(1, Tk, H, D).transpose(1, 2)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(1, Tq, C)
    got = _ops.attention(engine, q, k, v, H, scale, v_transposed=vt)[:1]
    f = 1.0 if engine.lib.cd_act_format() == 1 else 8.0
    _check(report, "attention_ragged_tail/%s" % ("vt" if vt else "v"), got, ref, rel=4e-3 * f, mean=1.5e-3 * f)


def test_attention_d40_long_ragged_keys_and_queries(engine, report):
    """The long d = 40 self-attention kernel (k_attention_d40: K / V^T tiles by LDS-DMA, K rows gathered in permuted key
    order, rows 32 .. 47 of O^T on the 16 x 16 x 32 matrix instruction) with a token count that is not a multiple of its
    64-key tiles or 256-query workgroups: 1100 = 17 tiles + 12 keys, 4 workgroups + 76 queries. The ragged tile's mask
    has to follow the key permutation, and rows beyond Tk must arrive as zeros: the second batch element's K / V are NaN
    (they follow element 0's rows in memory)."""
    g = torch.Generator().manual_seed(37)
    B, H, D, T = 2, 2, 40, 1100
    C = H * D
    q = r16(torch.randn(B, T, C, generator=g))
    k = r16(torch.randn(B, T, C, generator=g))
    v = r16(torch.randn(B, T, C, generator=g))
    k[1] = float("nan")
    v[1] = float("nan")
    scale = D ** -0.5
    qh = q[:1].view(1, T, H, D).transpose(1, 2).double()
    kh = k[:1].view(1, T, H, D).transpose(1, 2).double()
    vh = v[:1].view(1, T, H, D).transpose(1, 2).double()
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1) @ vh).transpose(1, 2).reshape(1, T, C).float()
    got = _ops.attention(engine, q, k, v, H, scale, v_transposed=True)[:1]
    f = 1.0 if engine.lib.cd_act_format() == 1 else 8.0
    _check(report, "attention_d40_long_ragged", got, ref, rel=4e-3 * f, mean=1.5e-3 * f)


def test_softmax_rows(engine, report):
    g = torch.Generator().manual_seed(19)
    s = torch.randn(70, 1000, generator=g) * 4
    got = _ops.softmax_rows(engine, s)
    _check(report, "softmax_rows", got, torch.softmax(s, -1), rel=1e-2, mean=5e-3)


@pytest.mark.parametrize("mode", [0, 1])
def test_timestep_embedding(engine, report, mode):
    t = torch.tensor([1.0, 11.0, 501.0, 981.0, 999.0])
    dim = 320 if mode == 0 else 128
    half = dim // 2
    if mode == 0:  # util.py:152-172
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        a = t[:, None] * freqs[None]
        ref = torch.cat([torch.cos(a), torch.sin(a)], -1)
    else:  # ddpm/diffusion.py:6-24
        e = math.log(10000) / (half - 1)
        f = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
        a = t[:, None] * f[None]
        ref = torch.cat([torch.sin(a), torch.cos(a)], 1)
    got = _ops.timestep_embedding(engine, t, dim, mode)
    st = _ops.err_stats(got, ref)
    report.add("timestep_embedding/mode%d" % mode, **st)
    assert st["max_abs"] < 2e-4, st  # device sin/cos of arguments up to ~1e3


def _coef(a_t, a_prev, sigma):
    f = np.float32
    a_t, a_prev, sigma = f(a_t), f(a_prev), f(sigma)
    return (np.sqrt(a_t), np.sqrt(f(1) - a_t), np.sqrt(a_prev), np.sqrt(f(1) - a_prev - sigma * sigma), sigma,
            np.sqrt(f(1) - a_t), f(1), 0)


def test_sched_steps_bit_exact(engine, report):
    """fp32 scheduler kernels vs the reference's tensor expressions (ddim.py:576-600, 634-645)."""
    g = torch.Generator().manual_seed(23)
    B, C, H, W = 2, 4, 16, 16
    x0, xt, e_u, e_c, nz, eps = [torch.randn(B, C, H, W, generator=g) for _ in range(6)]
    a_t, a_prev, sig = 0.4321, 0.4876, 0.0123
    co = _coef(a_t, a_prev, sig)
    at = torch.full((B, 1, 1, 1), float(np.float32(a_t)))
    ap = torch.full((B, 1, 1, 1), float(np.float32(a_prev)))
    sg = torch.full((B, 1, 1, 1), float(np.float32(sig)))
    s1 = torch.full((B, 1, 1, 1), float(np.sqrt(np.float32(1) - np.float32(a_t))))
    # init x_T
    ref_xT = at.sqrt() * x0 + (1 - at).sqrt() * nz
    got_xT, _ = _ops.sched_step(engine, 0, 0, co, x0=x0, noise=nz)
    assert torch.equal(got_xT, ref_xT)
    # encode step without / with CFG
    for cfg, gs in ((False, 1.0), (True, 3.0)):
        e_t = e_u + gs * (e_c - e_u) if cfg else e_c
        et_post = (xt - at.sqrt() * x0) / (1 - at).sqrt()
        x_next = ap.sqrt() * x0 + (1. - ap - sg ** 2).sqrt() * et_post + sg * nz
        pred_x0 = (xt - s1 * e_t) / at.sqrt()
        ref_eps = (x_next - ap.sqrt() * pred_x0 - (1. - ap - sg ** 2).sqrt() * e_t) / sg / 1.0
        eh = torch.cat([e_u, e_c], 0) if cfg else e_c
        got_xn, got_eps = _ops.sched_step(engine, 1, 0, co, x0=x0, xt=xt, eps_hat=eh, cfg=cfg, g=gs, noise=nz)
        assert torch.equal(got_xn, x_next), (got_xn - x_next).abs().max()
        assert torch.equal(got_eps, ref_eps), (got_eps - ref_eps).abs().max()
        # decode step with injected eps
        x_prev = ap.sqrt() * pred_x0 + (1. - ap - sg ** 2).sqrt() * e_t + sg * eps * 1.0
        got_xp, _ = _ops.sched_step(engine, 2, 0, co, xt=xt, eps_hat=eh, cfg=cfg, g=gs, eps_in=eps)
        assert torch.equal(got_xp, x_prev), (got_xp - x_prev).abs().max()
    # last encode step returns x0 without a draw
    got_xn, _ = _ops.sched_step(engine, 1, 0, co, x0=x0, xt=xt, eps_hat=e_c, noise=None, is_last=True)
    assert torch.equal(got_xn, x0)
    report.add("sched_steps_bit_exact", ok=True)


def test_sustained_mfma_rate_diagnostic_is_consistent(engine):
    """cd_op_bench_mfma_sustained (csrc/diag.hip): the rate it reports is the clock it reports x 1024 SIMDs x 1024 flop per
    cycle at 90-100 % duty, inside what an MI355X can do (bench.py puts it beside roofline.peak)."""
    tf, ghz = engine.mfma_sustained(150)
    assert 0.8 < ghz < 2.6 and 600 < tf < 2600, (tf, ghz)
    duty = tf / (ghz * 1024 * 1024 * 1e-3)
    assert 0.85 < duty < 1.03, (tf, ghz, duty)
