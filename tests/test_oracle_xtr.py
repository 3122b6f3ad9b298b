"""Pins oracle/xtr_text.py (LDM BERTEmbedder's x-transformers encoder): against the committed reference output
(tests/golden/xtr_text_tiny.npz, written by oracle/gen_golden.py from the reference's own module) and, where the
reference tree exists, against that module directly."""
import os

import pytest
import torch

import golden_util as gu
from oracle import ref_import, xtr_text as ox


def test_oracle_matches_reference_fixture():
    fx = gu.load("xtr_text_tiny")
    cfg = ox.XtrTextCfg(width=int(fx["width"]), layers=int(fx["layers"]), vocab=int(fx["vocab"]), positions=77)
    sd = ox.synth_state_dict(cfg, int(fx["wseed"]))
    with torch.no_grad():
        y = ox.xtr_text_forward(sd, cfg, torch.as_tensor(fx["ids"]))
    assert torch.allclose(y, torch.as_tensor(fx["y"]), atol=2e-5, rtol=1e-5), (y - torch.as_tensor(fx["y"])).abs().max()


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_oracle_matches_reference_module():
    ref_import.setup()
    from ldm.modules.x_transformer import Encoder, TransformerWrapper
    cfg = ox.XtrTextCfg(width=128, layers=3, vocab=500, positions=40)
    m = TransformerWrapper(num_tokens=cfg.vocab, max_seq_len=cfg.positions,
                           attn_layers=Encoder(dim=cfg.width, depth=cfg.layers), emb_dropout=0.0).eval()
    sd = ox.synth_state_dict(cfg, 3)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("to_logits") for k in missing)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.startswith("to_logits")} == \
        dict(ox.param_shapes(cfg))
    ids = torch.randint(0, cfg.vocab, (2, 33), generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        assert torch.allclose(ox.xtr_text_forward(sd, cfg, ids), m(ids, return_embeddings=True), atol=2e-5, rtol=1e-5)
