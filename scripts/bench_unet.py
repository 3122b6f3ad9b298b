"""Time one SD-v1 U-Net forward (random init) at the C2 shapes; optional per-kernel profile via rocprofv3."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cycle_diffusion_amd as cda

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = cda.Engine("cuda:0")
t0 = time.time()
net = eng.create_net(cda.sd_v1_unet_desc())
eng.random_init(net, seed=0)
print("init %.1fs" % (time.time() - t0), flush=True)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 4, 64, 64, generator=g).cuda()
t = torch.full((B,), 500.0).cuda()
ctx = torch.randn(B, 77, 768, generator=g).cuda()
y = eng.unet_forward(net, x, t, ctx)
torch.cuda.synchronize()
print("finite", bool(torch.isfinite(y).all()), "std", float(y.std()), flush=True)
for _ in range(2):
    eng.unet_forward(net, x, t, ctx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    eng.unet_forward(net, x, t, ctx)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print("B=%d: %.2f ms/forward  -> %.1f TFLOP/s (803.3 GFLOP/sample)" % (B, ms, B * 803.3 / ms))
print("workspace high water %.2f GB" % (eng.workspace_high_water() / 2 ** 30))
if len(sys.argv) > 3 and sys.argv[3] == "gemmlog":  # CYCLEDIFF_GEMM_LOG=1: per-shape in-situ GEMM table
    eng.prof_enable(True)
    eng.unet_forward(net, x, t, ctx)
    torch.cuda.synchronize()
    print("conv_gemm in situ:", eng.prof_collect())
    eng.prof_enable(False)
