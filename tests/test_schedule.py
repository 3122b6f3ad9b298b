"""CPU: host-side schedule logic of the product (cycle-diffusion_amd/schedule.py) against the reference
fixture and against the oracle's per-step tensors."""
import numpy as np
import torch

import golden_util as gu
from cycle_diffusion_amd import schedule
from oracle import samplers


def test_latent_tables_match_reference_fixture():
    fx = gu.load("schedule_sd_s99_eta0p1")
    ac = schedule.latent_alphas_cumprod()
    assert np.array_equal(ac.numpy(), fx["alphas_cumprod"])
    s = schedule.DDIMSchedule(ac, 99, 0.1)
    assert np.array_equal(s.timesteps, fx["timesteps"])
    assert np.array_equal(s.a, fx["a"])
    assert np.array_equal(s.a_prev, fx["a_prev"].astype(np.float32))
    assert np.array_equal(s.sigma, fx["sigma"].astype(np.float32))
    assert np.array_equal(s.r, fx["r"])


def test_latent_coefficients_are_the_reference_scalars():
    s = schedule.DDIMSchedule(schedule.latent_alphas_cumprod(), 99, 0.1)
    enc = s.coef_encode()
    assert len(enc) == 100 and len(s.coef_decode()) == 99
    for k in (0, 1, 50, 98):
        a_t = torch.full((1,), float(s.a[k]))
        a_p = torch.full((1,), float(s.a_prev[k]))
        sg = torch.full((1,), float(s.sigma[k]))
        assert enc["sa"][k] == a_t.sqrt().item()
        assert enc["s1a"][k] == (1 - a_t).sqrt().item()
        assert enc["sap"][k] == a_p.sqrt().item()
        assert enc["dirc"][k] == (1. - a_p - sg ** 2).sqrt().item()
        assert enc["t"][k] == s.timesteps[k]
    # skip_steps shortens the chain from the noisy end (ddim.py:468-473)
    assert len(s.coef_encode(skip_steps=15)) == 85
    assert s.coef_encode(skip_steps=15)["t"][83] == s.timesteps[83]


def test_pixel_schedule_matches_oracle_tensors():
    for st, eta in (("ddim", 0.1), ("ddpm", None)):
        p = schedule.PixelSchedule(50, 50, sample_type=st, eta=eta, refine_steps=5)
        b = torch.from_numpy(samplers.pixel_betas()).float()
        acp = (1.0 - b).cumprod(dim=0)
        assert np.array_equal(p.acp, acp.numpy())
        seq, seq_next = samplers.pixel_seq(50, 50)
        assert p.seq == seq and p.seq_next == seq_next
        enc, dec = p.coef_encode(), p.coef_decode()
        assert len(enc) == 50 and len(dec) == 50
        # spot-check one row against the tensor expressions of the wrapper
        i, j = seq[30], seq_next[30]
        row = dec[30]
        at, atn, bt = acp[i], acp[j], b[i]
        if st == "ddim":
            c1 = eta * ((1 - at / atn) * (1 - atn) / (1 - at)).sqrt()
            c2 = ((1 - atn) - c1 ** 2).sqrt()
            assert row["sigma"] == c1.item() and row["dirc"] == c2.item() and row["sap"] == atn.sqrt().item()
            assert dec[0]["sigma"] == 0.0 and dec[0]["sap"] == 1.0  # t_next = -1 -> alpha_bar_next = 1
        else:
            assert row["sa"] == (atn.sqrt() * bt / (1 - at)).item()
            assert row["r"] == (1 / torch.sqrt(1.0 - bt)).item()
            assert dec[0]["t_mask"] == 0.0 and dec[1]["t_mask"] == 1.0
        assert row["t"] == i
        assert len(p.coef_refine()) == 6
