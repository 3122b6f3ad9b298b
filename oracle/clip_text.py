"""CPU oracle (TEST INFRASTRUCTURE ONLY - never imported by the product path) for the CLIP text transformer.

The reference obtains its conditioning from HF transformers' `CLIPTextModel` (FrozenCLIPEmbedder,
model/lib/stable_diffusion/ldm/modules/encoders/modules.py:136-161; environment pins transformers==4.19.2,
not vendored under /root/reference). This file restates that module's forward
(transformers/models/clip/modeling_clip.py: CLIPTextEmbeddings, CLIPAttention, CLIPMLP with quick_gelu,
CLIPEncoderLayer, CLIPTextTransformer.forward -> last_hidden_state) as a functional torch-fp32 program over
a state_dict with the HF names. Pin: tests/test_oracle_clip.py checks it against the installed
`transformers.CLIPTextModel` (same architecture code in 4.19 and 5.x) on seeded random weights.
"""
import math

import torch
import torch.nn.functional as F


class ClipTextCfg:
    def __init__(self, width=768, layers=12, heads=12, mlp=3072, vocab=49408, positions=77, eps=1e-5):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def param_shapes(cfg):
    """(name, shape) list in HF state_dict order (without the non-persistent position_ids buffer)."""
    D, M = cfg.width, cfg.mlp
    out = [("text_model.embeddings.token_embedding.weight", (cfg.vocab, D)),
           ("text_model.embeddings.position_embedding.weight", (cfg.positions, D))]
    for i in range(cfg.layers):
        p = "text_model.encoder.layers.%d." % i
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(p + "self_attn.%s.weight" % n, (D, D)), (p + "self_attn.%s.bias" % n, (D,))]
        out += [(p + "layer_norm1.weight", (D,)), (p + "layer_norm1.bias", (D,)),
                (p + "mlp.fc1.weight", (M, D)), (p + "mlp.fc1.bias", (M,)),
                (p + "mlp.fc2.weight", (D, M)), (p + "mlp.fc2.bias", (D,)),
                (p + "layer_norm2.weight", (D,)), (p + "layer_norm2.bias", (D,))]
    out += [("text_model.final_layer_norm.weight", (D,)), ("text_model.final_layer_norm.bias", (D,))]
    return out


def synth_state_dict(cfg, seed):
    """Seeded weights with trained-model-like scales: N(0, 0.02) embeddings / projections, LayerNorm gains near 1."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg):
        if "layer_norm" in name and name.endswith("weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif "embedding" in name:
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        else:
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(shape[1])
    return sd


def clip_text_forward(sd, cfg, ids):
    """ids [B, L] int64 -> last_hidden_state [B, L, width] (modeling_clip.py CLIPTextTransformer.forward)."""
    B, L = ids.shape
    D, H = cfg.width, cfg.heads
    dh = D // H
    # CLIPTextEmbeddings: token + learned absolute position
    h = sd["text_model.embeddings.token_embedding.weight"][ids] + \
        sd["text_model.embeddings.position_embedding.weight"][:L][None]
    # causal mask: -inf strictly above the diagonal (_build_causal_attention_mask / _create_4d_causal_attention_mask)
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(cfg.layers):
        p = "text_model.encoder.layers.%d." % i
        r = h
        x = F.layer_norm(h, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.eps)
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * dh ** -0.5
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, L, H, dh).transpose(1, 2) for t in (q, k, v))
        w = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(B, L, D)
        h = r + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        r = h
        x = F.layer_norm(h, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.eps)
        x = F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        x = x * torch.sigmoid(1.702 * x)  # quick_gelu
        h = r + F.linear(x, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return F.layer_norm(h, (D,), sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"],
                        cfg.eps)
