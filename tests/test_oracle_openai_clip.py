"""Pins oracle/openai_clip.py (the networks behind the reference's DirectionalCLIP ranker) against HF transformers'
CLIPModel - the same architecture under other parameter names - on seeded weights (CPU)."""
import pytest
import torch

from oracle import openai_clip as oc


def _hf(cfg):
    tr = pytest.importorskip("transformers")
    tc = tr.CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.t_width, intermediate_size=4 * cfg.t_width,
                           num_hidden_layers=cfg.t_layers, num_attention_heads=cfg.t_heads,
                           max_position_embeddings=cfg.positions, hidden_act="quick_gelu", projection_dim=cfg.embed,
                           eos_token_id=cfg.vocab - 1)
    vc = tr.CLIPVisionConfig(hidden_size=cfg.v_width, intermediate_size=4 * cfg.v_width, num_hidden_layers=cfg.v_layers,
                             num_attention_heads=cfg.v_heads, image_size=cfg.res, patch_size=cfg.patch,
                             hidden_act="quick_gelu", projection_dim=cfg.embed)
    return tr.CLIPModel(tr.CLIPConfig(text_config=tc.to_dict(), vision_config=vc.to_dict(),
                                      projection_dim=cfg.embed)).eval()


def _to_hf(sd, cfg):
    """openai/CLIP names -> HF CLIPModel names (fused in_proj split into q / k / v)."""
    out = {}

    def block(src, dst, D):
        w, b = sd[src + "attn.in_proj_weight"], sd[src + "attn.in_proj_bias"]
        for j, n in enumerate("qkv"):
            out[dst + "self_attn.%s_proj.weight" % n] = w[j * D:(j + 1) * D]
            out[dst + "self_attn.%s_proj.bias" % n] = b[j * D:(j + 1) * D]
        out[dst + "self_attn.out_proj.weight"] = sd[src + "attn.out_proj.weight"]
        out[dst + "self_attn.out_proj.bias"] = sd[src + "attn.out_proj.bias"]
        for a, c in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            out[dst + c + ".weight"] = sd[src + a + ".weight"]
            out[dst + c + ".bias"] = sd[src + a + ".bias"]

    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    out["vision_model.pre_layrnorm.weight"], out["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]
    out["vision_model.post_layernorm.weight"], out["vision_model.post_layernorm.bias"] = sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    for i in range(cfg.v_layers):
        block("visual.transformer.resblocks.%d." % i, "vision_model.encoder.layers.%d." % i, cfg.v_width)
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    out["text_projection.weight"] = sd["text_projection"].t().contiguous()
    for i in range(cfg.t_layers):
        block("transformer.resblocks.%d." % i, "text_model.encoder.layers.%d." % i, cfg.t_width)
    return out


def test_oracle_matches_hf_clip_model():
    cfg = oc.OClipCfg(embed=32, res=64, patch=16, v_width=64, v_layers=2, v_heads=4, t_width=64, t_layers=2, t_heads=2,
                      vocab=300, positions=20)
    sd = {**oc.synth_state_dict(oc.vision_shapes(cfg), 1), **oc.synth_state_dict(oc.text_shapes(cfg), 2)}
    m = _hf(cfg)
    missing, unexpected = m.load_state_dict(_to_hf(sd, cfg), strict=False)
    assert not unexpected and all(("position_ids" in k) or k == "logit_scale" for k in missing), (missing, unexpected)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, cfg.res, cfg.res, generator=g)
    ids = torch.randint(1, cfg.vocab - 1, (2, cfg.positions), generator=g)
    ids[0, 7] = cfg.vocab - 1   # end-of-text = the largest id, at different positions
    ids[1, 12] = cfg.vocab - 1
    with torch.no_grad():
        vi = m.get_image_features(pixel_values=img)
        ti = m.get_text_features(input_ids=ids)
        vi = vi if torch.is_tensor(vi) else vi.pooler_output
        ti = ti if torch.is_tensor(ti) else ti.pooler_output
        assert torch.allclose(oc.encode_image(sd, cfg, img), vi, atol=3e-5, rtol=1e-4)
        assert torch.allclose(oc.encode_text(sd, cfg, ids), ti, atol=3e-5, rtol=1e-4)


def test_directional_scores_and_preprocess_shapes():
    g = torch.Generator().manual_seed(0)
    f = [torch.randn(3, 16, generator=g) for _ in range(4)]
    cs, ds = oc.directional_scores(*f)
    assert cs.shape == (3,) and ds.shape == (3,) and (cs.abs() <= 1 + 1e-6).all() and (ds.abs() <= 1 + 1e-6).all()
    x = oc.preprocess(torch.rand(2, 3, 512, 512, generator=g))
    assert x.shape == (2, 3, 224, 224)
    x = oc.preprocess(torch.rand(1, 3, 300, 400, generator=g))
    assert x.shape == (1, 3, 224, 224)
