"""Time the fused attention kernel on one self-attention shape of the SD U-Net (default: 64x64 level, d = 40).

  python scripts/bench_attn.py [B] [T] [heads] [d] [iters]

Runs both V layouts of k_attention (token-major V with LDS transpose reads, and V^T pre-transposed by k_transpose_v);
kernel times: rocprofv3 --kernel-trace + scripts/kernel_breakdown.py (the wall time below includes the fp32 <-> 16-bit
layout conversions of the test entry point).
"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cycle_diffusion_amd as cda
from cycle_diffusion_amd._ffi import check, ptr
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = int(sys.argv[3]) if len(sys.argv) > 3 else 8
D = int(sys.argv[4]) if len(sys.argv) > 4 else 40
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
eng = cda.Engine("cuda:0", workspace_bytes=4 << 30)
g = torch.Generator().manual_seed(0)
q = torch.randn(B, T, H * D, generator=g).cuda()
k = torch.randn(B, T, H * D, generator=g).cuda()
v = torch.randn(B, T, H * D, generator=g).cuda()
o = torch.empty_like(q)
scale = D ** -0.5


def run(vt):
    check(eng.lib.cd_op_attention(eng.h, ptr(q), ptr(k), ptr(v), B, H, T, T, D, C.c_float(scale), vt, ptr(o)))


for vt in (0, 1):
    run(vt)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        run(vt)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / iters * 1e3
    print("B=%d T=%d H=%d d=%d %s: %.3f ms per call incl. layout conversions (kernel time: see rocprofv3)"
          % (B, T, H, D, "V^T" if vt else "V token-major", ms))
