"""Sweep the implicit-GEMM kernel over the contraction shapes of one SD-v1 U-Net forward (SURVEY.md
Appendix B, batch B' = 8 = the CFG decode pass of C2) and print TFLOP/s per shape + a weighted total.

  python scripts/bench_gemm.py [B] [iters] [only-substring]
"""
import ctypes as C
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cycle_diffusion_amd as cda
from cycle_diffusion_amd._ffi import check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""

# name, H(=W), C0, C1, N, k, stride, up, act, calls per forward
SHAPES = [
    ("conv3 320>320 @64", 64, 320, 0, 320, 3, 1, 0, 0, 7),
    ("conv3 640>640 @32", 32, 640, 0, 640, 3, 1, 0, 0, 6),
    ("conv3 1280>1280 @16", 16, 1280, 0, 1280, 3, 1, 0, 0, 7),
    ("conv3 1280>1280 @8", 8, 1280, 0, 1280, 3, 1, 0, 0, 11),
    ("conv3 2560>1280 @16", 16, 1280, 1280, 1280, 3, 1, 0, 0, 2),
    ("conv3 2560>1280 @8", 8, 1280, 1280, 1280, 3, 1, 0, 0, 3),
    ("conv3 1920>1280 @16", 16, 1280, 640, 1280, 3, 1, 0, 0, 1),
    ("conv3 1920>640 @32", 32, 1280, 640, 640, 3, 1, 0, 0, 1),
    ("conv3 1280>640 @32", 32, 640, 640, 640, 3, 1, 0, 0, 1),
    ("conv3 960>640 @32", 32, 640, 320, 640, 3, 1, 0, 0, 1),
    ("conv3 960>320 @64", 64, 640, 320, 320, 3, 1, 0, 0, 1),
    ("conv3 640>320 @64", 64, 320, 320, 320, 3, 1, 0, 0, 2),
    ("conv3 320>640 @32", 32, 320, 0, 640, 3, 1, 0, 0, 1),
    ("conv3 640>1280 @16", 16, 640, 0, 1280, 3, 1, 0, 0, 1),
    ("up conv3 1280 @16>32", 16, 1280, 0, 1280, 3, 1, 1, 0, 1),
    ("up conv3 640 @32>64", 32, 640, 0, 640, 3, 1, 1, 0, 1),
    ("down conv3 320 @64>32", 64, 320, 0, 320, 3, 2, 0, 0, 1),
    ("lin 320>320 T4096", 64, 320, 0, 320, 1, 1, 0, 0, 25),
    ("lin 320>640 (qk) T4096", 64, 320, 0, 640, 1, 1, 0, 0, 5),
    ("geglu 320>2560 T4096", 64, 320, 0, 2560, 1, 1, 0, 3, 5),
    ("lin 1280>320 (ff2) T4096", 64, 1280, 0, 320, 1, 1, 0, 0, 5),
    ("lin 640>640 T1024", 32, 640, 0, 640, 1, 1, 0, 0, 25),
    ("geglu 640>5120 T1024", 32, 640, 0, 5120, 1, 1, 0, 3, 5),
    ("lin 2560>640 (ff2) T1024", 32, 2560, 0, 640, 1, 1, 0, 0, 5),
    ("lin 1280>1280 T256", 16, 1280, 0, 1280, 1, 1, 0, 0, 25),
    ("geglu 1280>10240 T256", 16, 1280, 0, 10240, 1, 1, 0, 3, 5),
    ("lin 5120>1280 (ff2) T256", 16, 5120, 0, 1280, 1, 1, 0, 0, 5),
]
if len(sys.argv) > 5 and sys.argv[5] == "half":  # spatial dims of the B'=B shapes with half the rows: B/2
    pass

eng = cda.Engine("cuda:0", workspace_bytes=8 << 30)
tot_ms, tot_fl, tot_best = 0.0, 0.0, 0.0
TILES = [0] + ([int(t) for t in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(1, 20)))
print("shapes at B=%d; per-config TFLOP/s" % B)
for name, hw, c0, c1, n, k, stride, up, act, calls in SHAPES:
    if only and only not in name:
        continue
    ho = hw * 2 if up else hw // stride
    fl = 2.0 * B * ho * ho * n * k * k * (c0 + c1)
    res = {}
    if act != 3:
        act |= int(os.environ.get("GEMM_ACT_OR", "0"), 0)  # | 0x100 in-place residual, | 0x200 fused GroupNorm statistics
    for tile in TILES:
        if act == 3 and tile in (3, 8, 15, 18):
            continue
        ms = C.c_float()
        check(eng.lib.cd_op_bench_conv(eng.h, B, hw, hw, c0, c1, n, k, stride, up, act, tile, iters, C.byref(ms)))
        res[tile] = ms.value
    best = min((t for t in res if t != 0), key=lambda t: res[t]) if len(res) > 1 else 0
    line = " ".join("%d:%.0f" % (t, fl / res[t] / 1e9) for t in sorted(res))
    print("%-28s auto %7.3f ms %6.0f TF | best cfg %2d %6.0f TF | %s" % (
        name, res[0], fl / res[0] / 1e9, best, fl / res[best] / 1e9, line), flush=True)
    tot_ms += res[0] * calls
    tot_best += res[best] * calls
    tot_fl += fl * calls
print("weighted auto: %.2f ms per forward, %.1f TFLOP/s; with best configs: %.2f ms, %.1f TFLOP/s" % (
    tot_ms, tot_fl / tot_ms / 1e9, tot_best, tot_fl / tot_best / 1e9))
