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


# ------------------------------------------------------------------ OpenAI CLIP towers + DirectionalCLIP ranker
def _oclip_nets(engine, cfg, seed):
    from oracle import openai_clip as ocl
    tnet = engine.create_net(cda.oclip_text_desc(cfg.t_width, cfg.t_layers, cfg.t_heads, cfg.vocab, cfg.positions, cfg.embed))
    vnet = engine.create_net(cda.oclip_vision_desc(cfg.v_width, cfg.v_layers, cfg.v_heads, cfg.res, cfg.patch, cfg.embed))
    assert set(n for n, _ in engine.net_params(tnet)) == set(n for n, _ in ocl.text_shapes(cfg))
    assert set(n for n, _ in engine.net_params(vnet)) == set(n for n, _ in ocl.vision_shapes(cfg))
    sd = {**ocl.synth_state_dict(ocl.vision_shapes(cfg), seed), **ocl.synth_state_dict(ocl.text_shapes(cfg), seed + 1)}
    for net in (tnet, vnet):
        n, first = engine.load_state_dict(net, sd)
        assert n == 0, first
    return tnet, vnet, sd


def _feat_err(got, ref):
    d = (got.cpu() - ref).abs()
    return d.max().item() / ref.abs().max().item()


def test_openai_clip_towers_small(engine, report):
    from oracle import openai_clip as ocl
    cfg = ocl.OClipCfg(embed=64, res=64, patch=16, v_width=128, v_layers=2, v_heads=2, t_width=64, t_layers=2, t_heads=1,
                       vocab=400, positions=24)
    tnet, vnet, sd = _oclip_nets(engine, cfg, 51)
    g = torch.Generator().manual_seed(52)
    img = torch.randn(3, 3, 64, 64, generator=g)
    ids = torch.randint(1, cfg.vocab - 1, (3, 24), generator=g)
    for b, pos in enumerate((5, 23, 11)):
        ids[b, pos] = cfg.vocab - 1
        ids[b, pos + 1:] = 0
    with torch.no_grad():
        ri, rt = ocl.encode_image(sd, cfg, img), ocl.encode_text(sd, cfg, ids)
    ei = _feat_err(engine.clip_image_features(vnet, img.cuda()), ri)
    et = _feat_err(engine.clip_text_features(tnet, ids), rt)
    report.add("oclip/small", image_rel=ei, text_rel=et)
    assert ei < 8e-3 * FMT and et < 8e-3 * FMT, (ei, et)


def test_openai_clip_vit_b32_and_directional_scores(engine, report):
    """Full ViT-B/32 configuration (both towers) and the ranker's scores on 512x512 images vs the oracle."""
    from cycle_diffusion_amd.gan_wrapper.ranker import DirectionalCLIPHIP, clip_preprocess
    from oracle import openai_clip as ocl
    cfg = ocl.OClipCfg()
    sd = {**ocl.synth_state_dict(ocl.vision_shapes(cfg), 61), **ocl.synth_state_dict(ocl.text_shapes(cfg), 62)}
    rk = DirectionalCLIPHIP(engine, state_dict=sd)
    g = torch.Generator().manual_seed(63)
    img, orig = torch.rand(2, 3, 512, 512, generator=g), torch.rand(2, 3, 512, 512, generator=g)
    src, tgt = ["a photo of a cat", "a red car"], ["a photo of a dog", "a blue car"]
    assert torch.allclose(clip_preprocess(img), ocl.preprocess(img), atol=1e-6)
    cs, ds = rk(img.cuda(), orig.cuda(), src, tgt)
    with torch.no_grad():
        fi, fo = ocl.encode_image(sd, cfg, ocl.preprocess(img)), ocl.encode_image(sd, cfg, ocl.preprocess(orig))
        fs, ft = ocl.encode_text(sd, cfg, rk.tokenize(src).long()), ocl.encode_text(sd, cfg, rk.tokenize(tgt).long())
        rcs, rds = ocl.directional_scores(fi, fo, fs, ft)
    e_img = _feat_err(rk.features(img=img.cuda()), fi)
    e_txt = _feat_err(rk.features(text=tgt), ft)
    report.add("oclip/vit_b32", image_rel=e_img, text_rel=e_txt, clip_score_err=float((cs.cpu() - rcs).abs().max()),
               dclip_score_err=float((ds.cpu() - rds).abs().max()))
    assert e_img < 8e-3 * FMT and e_txt < 8e-3 * FMT, (e_img, e_txt)
    assert (cs.cpu() - rcs).abs().max() < 5e-3 * FMT and (ds.cpu() - rds).abs().max() < 2e-2 * FMT
    # a folded ensemble call (latent_text_wrapper.forward): two candidates per sample, texts and sources encoded once
    cand = torch.cat([img, torch.rand(2, 3, 512, 512, generator=g)], dim=0).cuda()
    cs2, ds2 = rk.score_folded(cand, orig.cuda(), src, tgt, 2)
    cs1, ds1 = rk(cand, orig.cuda().repeat(2, 1, 1, 1), src * 2, tgt * 2)
    assert (cs2 - cs1).abs().max() < 1e-3 * FMT and (ds2 - ds1).abs().max() < 4e-3 * FMT
    assert (cs2[:2].cpu() - rcs).abs().max() < 5e-3 * FMT
