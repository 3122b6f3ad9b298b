"""CLIP text transformer on the engine (csrc/clip_text.hip) vs the oracle (oracle/clip_text.py, itself pinned to HF
transformers' CLIPTextModel by tests/test_oracle_clip.py): a small configuration with ragged lengths, the full
ViT-L/14 text configuration of Stable Diffusion, and the wrapper-level embedder."""
import pytest
import torch

import cycle_diffusion_amd as cda
from cycle_diffusion_amd import _ffi
from oracle import clip_text as oc
from oracle import xtr_text as ox

pytestmark = pytest.mark.gpu

FMT = 1.0 if _ffi.load_library().cd_act_format() == 1 else 8.0


def _run(engine, cfg, B, L, seed):
    net = engine.create_net(cda.clip_text_desc(cfg.width, cfg.layers, cfg.heads, cfg.mlp, cfg.vocab, cfg.positions))
    assert [(n, tuple(s)) for n, s in engine.net_params(net)] == [(n, tuple(s)) for n, s in oc.param_shapes(cfg)] or \
        set(n for n, _ in engine.net_params(net)) == set(n for n, _ in oc.param_shapes(cfg))
    sd = oc.synth_state_dict(cfg, seed)
    n, first = engine.load_state_dict(net, sd)
    assert n == 0, first
    ids = torch.randint(0, cfg.vocab, (B, L), generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        ref = oc.clip_text_forward(sd, cfg, ids)
    got = engine.text_encode(net, ids).cpu()
    d = (got - ref).abs()
    return d.max().item() / ref.abs().max().item(), d.mean().item() / ref.abs().mean().item()


@pytest.mark.parametrize("L", [77, 20, 64, 65])
def test_clip_text_small(engine, report, L):
    cfg = oc.ClipTextCfg(width=128, layers=3, heads=2, mlp=256, vocab=1000, positions=77)
    rmax, rmean = _run(engine, cfg, 3, L, 11)
    report.add("clip_text/small_L%d" % L, rel_to_max=rmax, mean_rel=rmean)
    assert rmax < 8e-3 * FMT and rmean < 8e-3 * FMT, (rmax, rmean)


def test_clip_text_vit_l14_full_size(engine, report):
    """The text tower of openai/clip-vit-large-patch14: 12 layers x 768, 12 heads of 64, MLP 3072, 77 positions."""
    rmax, rmean = _run(engine, oc.ClipTextCfg(), 2, 77, 3)
    report.add("clip_text/vit_l14", rel_to_max=rmax, mean_rel=rmean)
    assert rmax < 8e-3 * FMT and rmean < 8e-3 * FMT, (rmax, rmean)


def test_frozen_clip_embedder_on_engine(engine):
    from cycle_diffusion_amd.gan_wrapper.text_encoders import BOS, EOS, FrozenCLIPEmbedderHIP, HashTokenizer
    ids = HashTokenizer()(["a photo of a cat", ""])
    assert ids.shape == (2, 77) and ids[0, 0] == BOS and ids[0, 6] == EOS and ids[1, 1] == EOS
    emb = FrozenCLIPEmbedderHIP(engine)
    c = emb(["a photo of a cat", "a photo of a dog", ""])
    assert c.shape == (3, 77, 768) and torch.isfinite(c).all()
    # causal transformer: the first token's state cannot depend on the text; later ones do
    assert torch.equal(c[0, 0], c[1, 0]) and not torch.equal(c[0, 5], c[1, 5])


# ------------------------------------------------------------------ LDM text encoder (BERTEmbedder.transformer)
def _run_xtr(engine, cfg, B, L, seed):
    net = engine.create_net(cda.bert_xtransformer_desc(cfg.width, cfg.layers, cfg.vocab, cfg.positions, cfg.heads,
                                                       cfg.dim_head))
    assert set(n for n, _ in engine.net_params(net)) == set(n for n, _ in ox.param_shapes(cfg))
    sd = ox.synth_state_dict(cfg, seed)
    n, first = engine.load_state_dict(net, sd)
    assert n == 0, first
    ids = torch.randint(0, cfg.vocab, (B, L), generator=torch.Generator().manual_seed(seed + 1))
    with torch.no_grad():
        ref = ox.xtr_text_forward(sd, cfg, ids)
    got = engine.text_encode(net, ids).cpu()
    d = (got - ref).abs()
    return d.max().item() / ref.abs().max().item(), d.mean().item() / ref.abs().mean().item()


def test_bert_xtransformer_small(engine, report):
    rmax, rmean = _run_xtr(engine, ox.XtrTextCfg(width=128, layers=3, vocab=500, positions=77), 3, 77, 21)
    report.add("xtr_text/small", rel_to_max=rmax, mean_rel=rmean)
    assert rmax < 8e-3 * FMT and rmean < 8e-3 * FMT, (rmax, rmean)


def test_bert_xtransformer_ldm_width(engine, report):
    """Width 1280 as in txt2img-1p4B-eval.yaml (BERTEmbedder n_embed 1280), 4 of the 32 layers: every layer has
    the same shapes, the depth only repeats them."""
    rmax, rmean = _run_xtr(engine, ox.XtrTextCfg(width=1280, layers=4, vocab=30522, positions=77), 2, 77, 22)
    report.add("xtr_text/w1280", rel_to_max=rmax, mean_rel=rmean)
    assert rmax < 8e-3 * FMT and rmean < 8e-3 * FMT, (rmax, rmean)


def test_bert_embedder_on_engine(engine):
    from cycle_diffusion_amd.gan_wrapper.text_encoders import BERTEmbedderHIP
    emb = BERTEmbedderHIP(engine, layers=2)
    c = emb(["a painting of a fox", ""])
    assert c.shape == (2, 77, 1280) and torch.isfinite(c).all()
