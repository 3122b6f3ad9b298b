"""Engine smoke on tiny random-init networks: every executor path launches, outputs are finite and
bit-reproducible run to run (deterministic reductions: required so the encode-time and decode-time
U-Net passes agree, SURVEY.md §7 'hard parts')."""
import pytest
import torch

import cycle_diffusion_amd as cda
from cycle_diffusion_amd import _ffi

pytestmark = pytest.mark.gpu


def _run_twice(fn):
    a = fn()
    b = fn()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, b)
    return a


def test_tiny_sd_unet(engine, report):
    d = cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=16, in_channels=4, out_channels=4, model_channels=64,
                      num_res_blocks=1, channel_mult=(1, 2), attn=(1, 2), num_heads=2,
                      use_spatial_transformer=True, context_dim=64)
    net = engine.create_net(d)
    engine.random_init(net, seed=1)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 16, 16, generator=g).cuda()
    t = torch.tensor([10.0, 500.0]).cuda()
    ctx = torch.randn(2, 77, 64, generator=g).cuda()
    y = _run_twice(lambda: engine.unet_forward(net, x, t, ctx))
    assert y.shape == (2, 4, 16, 16)
    report.add("smoke/tiny_sd_unet", std=float(y.std()))
    assert y.std() > 1e-3


def test_tiny_iddpm_unet(engine, report):
    d = cda.make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=32, in_channels=3, out_channels=6, model_channels=32,
                      num_res_blocks=1, channel_mult=(1, 2, 2), attn=(2,), num_heads=4, num_head_channels=32,
                      use_scale_shift_norm=True, resblock_updown=True)
    net = engine.create_net(d)
    engine.random_init(net, seed=2)
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(0)).cuda()
    t = torch.tensor([3.0, 700.0]).cuda()
    y = _run_twice(lambda: engine.unet_forward(net, x, t))
    assert y.shape == (2, 6, 32, 32)
    report.add("smoke/tiny_iddpm_unet", std=float(y.std()))


def test_tiny_ho_unet(engine, report):
    d = cda.ho_ddpm_desc(32, 32, (1, 2, 2), 1, (16,))
    net = engine.create_net(d)
    engine.random_init(net, seed=3)
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(0)).cuda()
    t = torch.tensor([49.0]).cuda()
    y = _run_twice(lambda: engine.unet_forward(net, x, t))
    assert y.shape == (1, 3, 32, 32)
    report.add("smoke/tiny_ho_unet", std=float(y.std()))


@pytest.mark.parametrize("ch", [32, 64])
def test_tiny_vae(engine, report, ch):
    # ch=64 -> mid block has 256 channels: exercises the materialised-score AttnBlock path
    d = cda.make_desc(_ffi.CD_NET_VAE_KL, image_size=0, in_channels=3, out_channels=3, model_channels=ch,
                      num_res_blocks=1, channel_mult=(1, 2, 4), z_channels=4, embed_dim=4, double_z=True)
    net = engine.create_net(d)
    engine.random_init(net, seed=4)
    img = (torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1).cuda()
    z = _run_twice(lambda: engine.vae_encode(net, img, sample=False))
    assert z.shape == (2, 4, 16, 16)
    rec = _run_twice(lambda: engine.vae_decode(net, z))
    assert rec.shape == (2, 3, 64, 64)
    report.add("smoke/tiny_vae_ch%d" % ch, z_std=float(z.std()), rec_std=float(rec.std()))
