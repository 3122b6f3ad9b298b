"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the LDM text encoder: BERTEmbedder.transformer, i.e.
TransformerWrapper(num_tokens, max_seq_len, attn_layers=Encoder(dim, depth), emb_dropout=0)(tokens,
return_embeddings=True) of the reference's vendored x-transformers
(model/lib/latentdiff/ldm/modules/encoders/modules.py:75-98; ldm/modules/x_transformer.py:
TransformerWrapper.forward :600-640, AttentionLayers.forward :520-560 (pre-norm residual blocks 'a','f'),
Attention.forward :270-330 (8 heads x 64, bias-free q/k/v, scale dim_head**-0.5, softmax, to_out),
FeedForward :194-211 (Linear -> exact GELU -> Linear)). Functional torch-fp32 restatement over a state_dict with
the module's own names. Pin: tests/test_oracle_xtr.py runs the reference module itself where /root/reference
exists, and the committed fixture tests/golden/xtr_text_tiny.npz (oracle/gen_golden.py) everywhere else.
"""
import math

import torch
import torch.nn.functional as F


class XtrTextCfg:
    def __init__(self, width=1280, layers=32, vocab=30522, positions=77, heads=8, dim_head=64, eps=1e-5):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def param_shapes(cfg):
    D, I, M = cfg.width, cfg.heads * cfg.dim_head, 4 * cfg.width
    out = [("token_emb.weight", (cfg.vocab, D)), ("pos_emb.emb.weight", (cfg.positions, D))]
    for i in range(cfg.layers):
        a, f = "attn_layers.layers.%d." % (2 * i), "attn_layers.layers.%d." % (2 * i + 1)
        out += [(a + "0.weight", (D,)), (a + "0.bias", (D,)),
                (a + "1.to_q.weight", (I, D)), (a + "1.to_k.weight", (I, D)), (a + "1.to_v.weight", (I, D)),
                (a + "1.to_out.weight", (D, I)), (a + "1.to_out.bias", (D,)),
                (f + "0.weight", (D,)), (f + "0.bias", (D,)),
                (f + "1.net.0.0.weight", (M, D)), (f + "1.net.0.0.bias", (M,)),
                (f + "1.net.2.weight", (D, M)), (f + "1.net.2.bias", (D,))]
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    return out


def synth_state_dict(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg):
        if len(shape) == 1 and name.endswith("weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif "emb" in name:
            sd[name] = 0.02 * torch.randn(shape, generator=g)  # AbsolutePositionalEmbedding.init_: std 0.02
        else:
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(shape[1])
    return sd


def xtr_text_forward(sd, cfg, ids):
    """ids [B, L] int64 -> embeddings [B, L, width]"""
    B, L = ids.shape
    D, H, dh = cfg.width, cfg.heads, cfg.dim_head
    x = sd["token_emb.weight"][ids] + sd["pos_emb.emb.weight"][:L][None]
    for i in range(cfg.layers):
        a, f = "attn_layers.layers.%d." % (2 * i), "attn_layers.layers.%d." % (2 * i + 1)
        h = F.layer_norm(x, (D,), sd[a + "0.weight"], sd[a + "0.bias"], cfg.eps)
        q, k, v = (F.linear(h, sd[a + "1.to_%s.weight" % n]).view(B, L, H, dh).transpose(1, 2) for n in "qkv")
        w = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
        o = (w @ v).transpose(1, 2).reshape(B, L, H * dh)
        x = x + F.linear(o, sd[a + "1.to_out.weight"], sd[a + "1.to_out.bias"])
        h = F.layer_norm(x, (D,), sd[f + "0.weight"], sd[f + "0.bias"], cfg.eps)
        h = F.gelu(F.linear(h, sd[f + "1.net.0.0.weight"], sd[f + "1.net.0.0.bias"]))
        x = x + F.linear(h, sd[f + "1.net.2.weight"], sd[f + "1.net.2.bias"])
    return F.layer_norm(x, (D,), sd["norm.weight"], sd["norm.bias"], cfg.eps)
