"""Pins oracle/clip_text.py against the dependency the reference actually calls: HF transformers' CLIPTextModel
(ldm/modules/encoders/modules.py:136-161). Runs on CPU; skipped if transformers cannot build the model."""
import pytest
import torch

from oracle import clip_text as oc


def _hf_model(cfg):
    tr = pytest.importorskip("transformers")
    hc = tr.CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.width, intermediate_size=cfg.mlp,
                           num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                           max_position_embeddings=cfg.positions, hidden_act="quick_gelu", layer_norm_eps=cfg.eps)
    return tr.CLIPTextModel(hc).eval()


def _to_hf_keys(model, sd):
    """The reference's transformers 4.19.2 (and every SD checkpoint) nests the module as `text_model.*`; newer
    transformers flatten CLIPTextModel. Map the oracle's 4.19-style names onto whatever is installed."""
    if any(k.startswith("text_model.") for k in model.state_dict()):
        return sd
    return {k[len("text_model."):]: v for k, v in sd.items()}


@pytest.mark.parametrize("cfg", [oc.ClipTextCfg(width=64, layers=2, heads=4, mlp=128, vocab=500, positions=77),
                                 oc.ClipTextCfg(width=128, layers=3, heads=2, mlp=256, vocab=1000, positions=20)],
                         ids=["w64", "w128"])
def test_oracle_matches_hf_clip_text_model(cfg):
    m = _hf_model(cfg)
    sd = oc.synth_state_dict(cfg, 5)
    missing, unexpected = m.load_state_dict(_to_hf_keys(m, sd), strict=False)
    # every oracle tensor is consumed; HF may list only non-persistent buffers (position_ids) as missing
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    ids = torch.randint(0, cfg.vocab, (3, cfg.positions), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = m(input_ids=ids).last_hidden_state
        got = oc.clip_text_forward(sd, cfg, ids)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), (got - ref).abs().max()


def test_param_names_cover_the_hf_state_dict():
    cfg = oc.ClipTextCfg(width=64, layers=2, heads=4, mlp=128, vocab=500, positions=77)
    m = _hf_model(cfg)
    hf = {k: tuple(v.shape) for k, v in m.state_dict().items() if "position_ids" not in k}
    assert {k: tuple(s) for k, s in _to_hf_keys(m, dict(oc.param_shapes(cfg))).items()} == hf
