"""Diagnostic (GPU): lin_stream with fused GroupNorm statistics against conv_gemm and torch - where do they differ?"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cycle_diffusion_amd as cda  # noqa: E402
import _ops  # noqa: E402
from _ops import bf16_round as r16  # noqa: E402

eng = cda.Engine("cuda:0", workspace_bytes=4 << 30)
for (B, H, W, has_res) in ((1, 32, 16, False), (2, 32, 32, True)):
    g = torch.Generator().manual_seed(5)
    K = N = 320
    x = r16(torch.randn(B, K, H, W, generator=g))
    w = r16(torch.randn(N, K, 1, 1, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g) * 0.5
    res = r16(torch.randn(B, N, H, W, generator=g)) if has_res else None
    ref = F.conv2d(x, w, bias)
    if res is not None:
        ref = ref + res
    for tile in (30, 20, 2):
        for want in (False, True):
            out = _ops.conv2d16(eng, x, w, pad=0, bias=bias, resid=res, tile=tile, want_stats=want)
            y, st = out if want else (out, None)
            d = (y - ref).abs()
            m = d.permute(0, 2, 3, 1).reshape(-1, N)  # [M][N]
            bad = m > 0.05
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            msg = "M=%d res=%d tile=%d stats=%d: max err %.4f, bad rows %d (first %s), bad cols %d (first %s)" % (
                B * H * W, has_res, tile, want, d.max().item(), len(rows), rows[:12].tolist(), len(cols), cols[:12].tolist())
            if want:
                blocks = ref.permute(0, 2, 3, 1).reshape(-1, 32, N)
                wantst = torch.stack([blocks.sum(1), (blocks * blocks).sum(1)], 1)
                msg += "; stats rel err %.2e" % ((st - wantst).abs().max() / wantst.abs().max()).item()
            print(msg, flush=True)
