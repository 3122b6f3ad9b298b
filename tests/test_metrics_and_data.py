"""Host-side evaluation pieces (CPU): the reference's image metrics (evaluation/utils.py) and the triplet
dataset transform (preprocess/translate_text512.py)."""
import json

import numpy as np
import pytest
import torch
from PIL import Image

from cycle_diffusion_amd.data.triplets import TripletDataset, collate, load_image
from cycle_diffusion_amd.utils import metrics


def test_psnr_matches_the_definition():
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(3, 16, 16, generator=g), torch.rand(3, 16, 16, generator=g)
    assert abs(float(metrics.calculate_psnr(a, b)) - 10 * np.log10(1.0 / float(((a - b) ** 2).mean()))) < 1e-5
    assert float(metrics.calculate_psnr(a, a)) == 100.0
    with pytest.raises(AssertionError):
        metrics.calculate_psnr(a * 2, b)


def test_ssim_against_a_direct_window_loop():
    g = torch.Generator().manual_seed(1)
    a = torch.rand(20, 23, generator=g) * 255
    b = (a + 20 * torch.randn(20, 23, generator=g)).clamp(0, 255)
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k /= k.sum()
    w = np.outer(k, k)
    A, B = a.double().numpy(), b.double().numpy()
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for y in range(20 - 10):
        for x in range(23 - 10):
            pa, pb = A[y:y + 11, x:x + 11], B[y:y + 11, x:x + 11]
            m1, m2 = (w * pa).sum(), (w * pb).sum()
            s11, s22, s12 = (w * pa * pa).sum() - m1 * m1, (w * pb * pb).sum() - m2 * m2, (w * pa * pb).sum() - m1 * m2
            vals.append(((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s11 + s22 + C2)))
    assert abs(float(metrics.ssim(a, b)) - float(np.mean(vals))) < 1e-9
    assert abs(float(metrics.ssim(a, a)) - 1.0) < 1e-12
    c3 = torch.stack([a, b, a], 2)
    assert abs(float(metrics.calculate_ssim(c3, c3)) - 1.0) < 1e-12


def test_triplet_dataset_center_crops_the_long_edge(tmp_path):
    arr = np.zeros((40, 60, 3), dtype=np.uint8)
    arr[:, 10:50] = 255  # the centre 40x40 crop is all white
    Image.fromarray(arr).save(tmp_path / "a.png")
    Image.fromarray(np.full((32, 32, 3), 128, dtype=np.uint8)).save(tmp_path / "b.png")
    meta = [{"img_path": "a.png", "encode_text": "s0", "decode_text": "t0"},
            {"img_path": "b.png", "encode_text": "s1", "decode_text": "t1"},
            {"img_path": "b.png", "encode_text": "s2", "decode_text": "t2"}]
    (tmp_path / "t.json").write_text(json.dumps(meta))
    img = load_image(str(tmp_path / "a.png"), 32)
    assert img.shape == (3, 32, 32) and float(img.min()) == 1.0
    ds = TripletDataset(str(tmp_path / "t.json"), 32, 1, 3)
    assert len(ds) == 2 and int(ds[0]["sample_id"]) == 1 and ds[1]["decode_text"] == "t2"
    batch = collate([ds[0], ds[1]])
    assert batch["original_image"].shape == (2, 3, 32, 32) and batch["encode_text"] == ["s1", "s2"]
    assert abs(float(batch["original_image"][0].mean()) - 128 / 255.0) < 1e-6


def test_pair_grid_visualizer_layout(tmp_path):
    """visualization/multi_image.py: pairs interleaved image by image, 8 per row, 2-pixel border, plus a 256-px copy"""
    from cycle_diffusion_amd.utils import visualize as viz
    orig = torch.zeros(5, 3, 16, 16)
    out = torch.ones(5, 3, 16, 16)
    full, small = viz.visualize((orig, out), "eval", str(tmp_path), 7)
    assert full.endswith("eval_000007.png") and small.endswith("eval_256_000007.png")
    im = np.asarray(Image.open(full))
    # 10 tiles -> 2 rows of 8 columns: (16 + 2) * 2 + 2 by (16 + 2) * 8 + 2
    assert im.shape == (38, 146, 3)
    assert im[2:18, 2:18].max() == 0 and im[2:18, 20:36].min() == 255  # original then translated
    assert im[:2].max() == 0  # border
    im2 = np.asarray(Image.open(small))
    assert im2.shape == ((256 + 2) * 2 + 2, (256 + 2) * 8 + 2, 3)
    g = viz.make_grid(torch.rand(3, 1, 4, 4), nrow=2)
    assert g.shape == (1, 14, 14)
