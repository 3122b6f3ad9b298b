"""Several engines on one GPU, driven concurrently from host threads (bench.py keeps 2-4 batches in flight this
way): each engine owns its stream, arena and split-K workspace, so results must equal the single-engine run
bit for bit - in particular the split-K partial tiles and arrival counters of two streams must never mix."""
import math
import threading

import pytest
import torch

import _ops
import cycle_diffusion_amd as cda
import golden_util as gu
from cycle_diffusion_amd import _ffi, schedule
from test_gpu_models import _load, tiny_sd_desc

pytestmark = pytest.mark.gpu


def _engines(n):
    out = []
    for _ in range(n):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            out.append((st, cda.Engine("cuda:0", workspace_bytes=1 << 30)))
    return out


def test_concurrent_splitk_convs_do_not_mix():
    engs = _engines(3)
    g = torch.Generator().manual_seed(21)
    w = _ops.bf16_round(torch.randn(640, 640, 3, 3, generator=g) / math.sqrt(640 * 9))
    xs = [_ops.bf16_round(torch.randn(2, 640, 8, 8, generator=g)) for _ in engs]
    tile = 2 | (6 << 8)  # 128x64 tiles, K split in 6: 20 tiles x 6 partials per launch
    solo = []
    for (st, e), x in zip(engs, xs):
        with torch.cuda.stream(st):
            solo.append(_ops.conv2d(e, x, w, pad=1, tile=tile))
    res = [[] for _ in engs]

    def work(i):
        st, e = engs[i]
        with torch.cuda.stream(st):
            for _ in range(40):
                res[i].append(_ops.conv2d(e, xs[i], w, pad=1, tile=tile))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(len(engs)):
        assert len(res[i]) == 40
        for y in res[i]:
            assert torch.equal(y, solo[i])
    for _, e in engs:
        e.close()


def test_concurrent_samplers_match_solo(report):
    fx = gu.load("latent_cycle_tiny")
    engs = _engines(2)
    nets = []
    for st, e in engs:
        with torch.cuda.stream(st):
            nets.append(_load(e, tiny_sd_desc(), fx)[0])
    x0, c, uc, _ = gu.latent_cycle_inputs()
    K = 30
    sch = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), K, 0.1)
    noises = [torch.randn(K, *x0.shape, generator=torch.Generator().manual_seed(40 + i)) for i in range(2)]

    def run(i, out):
        st, e = engs[i]
        with torch.cuda.stream(st):
            z = e.dpm_encode(nets[i], _ffi.CD_SCHED_DDIM, x0.cuda(), sch.coef_encode(), ctx_c=c.cuda(), guidance=1.0,
                             noise=noises[i].cuda())
            x = e.ddim_decode(nets[i], _ffi.CD_SCHED_DDIM, z, sch.coef_decode(), ctx_c=c.cuda(), ctx_uc=uc.cuda(),
                              guidance=3.0)
            st.synchronize()
            out[i] = (z.cpu(), x.cpu())

    solo = {}
    for i in range(2):
        run(i, solo)
    conc = {}
    ths = [threading.Thread(target=run, args=(i, conc)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(2):
        assert torch.equal(conc[i][0], solo[i][0]) and torch.equal(conc[i][1], solo[i][1])
    report.add("concurrency/samplers", engines=2, steps=K, identical=True)
    for _, e in engs:
        e.close()
