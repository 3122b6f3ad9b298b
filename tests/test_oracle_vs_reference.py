"""CPU, only where /root/reference is mounted (skipped on the GPU box): the oracle restatement against the LIVE
reference modules on seeds and inputs that differ from the committed fixtures - the fixtures pin the oracle
everywhere, this catches a fixture that has gone stale against the reference or an oracle branch no fixture reaches
(per-sample timesteps, batch > 1 VAE, a second draw of weights)."""
import json

import pytest
import torch

import golden_util as gu
from oracle import nets, ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not mounted")


def _close(a, b, atol=3e-5):
    err = (a - b).abs().max().item()
    assert err < atol * max(1.0, b.abs().max().item()), err


def test_networks_match_live_reference_modules():
    from oracle import gen_golden as gg
    with torch.no_grad():
        u = gg.build_ref_sd_unet()
        ns, sd = gg.load_synth(u, 901)
        x, t, ctx = gu.rnd((3, 4, 16, 16), 31), torch.tensor([1, 501, 981]), gu.rnd((3, 77, 64), 32)
        _close(nets.openai_unet(sd, gu.TINY_SD_CFG, x, t, ctx), u(x, t, context=ctx))

        u2 = gg.build_ref_iddpm()
        ns, sd = gg.load_synth(u2, 902)
        x, t = gu.rnd((2, 3, 32, 32), 33), torch.tensor([999.0, 0.0])
        _close(nets.openai_unet(sd, gu.TINY_IDDPM_CFG, x, t), u2(x, t))

        v = gg.RefVAE()
        ns, sd = gg.load_synth(v, 903)
        img = torch.rand((3, 3, 64, 64), generator=torch.Generator().manual_seed(34)) * 2 - 1
        _close(nets.vae_encode_moments(sd, gu.TINY_VAE_CFG, img), v.moments(img))
        zz = gu.rnd((3, 4, 16, 16), 35, 0.7)
        _close(nets.vae_decode(sd, gu.TINY_VAE_CFG, zz), v.decode(zz))

        h = gg.build_ref_ho()
        ns, sd = gg.load_synth(h, 904)
        x, t = gu.rnd((2, 3, 32, 32), 36), torch.tensor([10.0, 870.0])
        _close(nets.ho_unet(sd, gu.TOY_HO_CFG, x, t), h(x, t))

        u3 = gg.build_ref_sd_unet(gg.TINY_LDM_UNCOND)
        ns, sd = gg.load_synth(u3, 905)
        x, t = gu.rnd((2, 3, 16, 16), 37), torch.tensor([3, 777])
        _close(nets.openai_unet(sd, gu.TINY_LDM_UNCOND_CFG, x, t), u3(x, t))


def test_committed_fixtures_still_match_the_reference():
    """re-run one network fixture and the schedule through the live reference: bit-for-bit what is committed"""
    from oracle import gen_golden as gg
    import numpy as np
    fx = gu.load("unet_tiny_sd")
    with torch.no_grad():
        u = gg.build_ref_sd_unet()
        u.load_state_dict(nets.synth_state_dict(json.loads(str(fx["names"])), int(fx["wseed"])))
        u.eval()
        x, t, ctx = gu.tiny_sd_inputs()
        assert np.array_equal(u(x, t, context=ctx).numpy(), fx["y"])
