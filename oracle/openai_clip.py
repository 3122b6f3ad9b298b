"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the DirectionalCLIP ranker's networks and score.

The reference ranks ensemble candidates with OpenAI CLIP ViT-B/32 (model/energy/clean_clip.py:7-41:
`clip.load("ViT-B/32")` from the un-vendored openai/CLIP package, README.md:82). This file restates that
package's clip/model.py in functional torch fp32 over a state_dict with ITS names: VisionTransformer.forward
(conv1 patch embedding, class token, positional embedding, ln_pre, transformer, ln_post(x[:, 0]) @ proj),
CLIP.encode_text (token + positional embedding, causal transformer, ln_final, x[arange, argmax(text)] @
text_projection), ResidualAttentionBlock (nn.MultiheadAttention with fused in_proj, QuickGELU MLP); and
DirectionalCLIP.__call__'s score arithmetic. Pin: tests/test_oracle_openai_clip.py maps the names onto HF
transformers' CLIPModel (same architecture) and compares features on seeded weights.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class OClipCfg:
    def __init__(self, embed=512, res=224, patch=32, v_width=768, v_layers=12, v_heads=12, t_width=512, t_layers=12,
                 t_heads=8, vocab=49408, positions=77, eps=1e-5):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _block_shapes(p, D):
    return [(p + "ln_1.weight", (D,)), (p + "ln_1.bias", (D,)),
            (p + "attn.in_proj_weight", (3 * D, D)), (p + "attn.in_proj_bias", (3 * D,)),
            (p + "attn.out_proj.weight", (D, D)), (p + "attn.out_proj.bias", (D,)),
            (p + "ln_2.weight", (D,)), (p + "ln_2.bias", (D,)),
            (p + "mlp.c_fc.weight", (4 * D, D)), (p + "mlp.c_fc.bias", (4 * D,)),
            (p + "mlp.c_proj.weight", (D, 4 * D)), (p + "mlp.c_proj.bias", (D,))]


def vision_shapes(cfg):
    D, T = cfg.v_width, (cfg.res // cfg.patch) ** 2 + 1
    out = [("visual.class_embedding", (D,)), ("visual.positional_embedding", (T, D)),
           ("visual.proj", (D, cfg.embed)), ("visual.conv1.weight", (D, 3, cfg.patch, cfg.patch)),
           ("visual.ln_pre.weight", (D,)), ("visual.ln_pre.bias", (D,))]
    for i in range(cfg.v_layers):
        out += _block_shapes("visual.transformer.resblocks.%d." % i, D)
    out += [("visual.ln_post.weight", (D,)), ("visual.ln_post.bias", (D,))]
    return out


def text_shapes(cfg):
    D = cfg.t_width
    out = [("positional_embedding", (cfg.positions, D)), ("text_projection", (D, cfg.embed)),
           ("token_embedding.weight", (cfg.vocab, D))]
    for i in range(cfg.t_layers):
        out += _block_shapes("transformer.resblocks.%d." % i, D)
    out += [("ln_final.weight", (D,)), ("ln_final.bias", (D,))]
    return out


def synth_state_dict(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes:
        if (".ln_" in name or name.startswith("ln_")) and name.endswith("weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif "embedding" in name:
            sd[name] = 0.02 * torch.randn(shape, generator=g)
        elif name.endswith("proj") or name == "text_projection":
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(shape[0])
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[name] = torch.randn(shape, generator=g) / math.sqrt(fan_in)
    return sd


def _transformer(sd, p, x, layers, heads, mask):
    B, L, D = x.shape
    dh = D // heads
    for i in range(layers):
        q = "%stransformer.resblocks.%d." % (p, i)
        h = F.layer_norm(x, (D,), sd[q + "ln_1.weight"], sd[q + "ln_1.bias"])
        qkv = F.linear(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"])
        a, b, c = (t.view(B, L, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        s = (a * dh ** -0.5) @ b.transpose(-1, -2)
        if mask is not None:
            s = s + mask
        o = (torch.softmax(s, dim=-1) @ c).transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(o, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), sd[q + "ln_2.weight"], sd[q + "ln_2.bias"])
        h = F.linear(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"])
    return x


def encode_image(sd, cfg, img):
    """img [B, 3, res, res] already preprocessed -> [B, embed] (VisionTransformer.forward)"""
    D = cfg.v_width
    x = F.conv2d(img, sd["visual.conv1.weight"], stride=cfg.patch)  # [B, D, g, g]
    x = x.flatten(2).transpose(1, 2)
    cls = sd["visual.class_embedding"][None, None].expand(x.shape[0], 1, D)
    x = torch.cat([cls, x], 1) + sd["visual.positional_embedding"][None]
    x = F.layer_norm(x, (D,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = _transformer(sd, "visual.", x, cfg.v_layers, cfg.v_heads, None)
    x = F.layer_norm(x[:, 0], (D,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"]


def encode_text(sd, cfg, ids):
    """ids [B, L] int64 -> [B, embed] (CLIP.encode_text)"""
    D, L = cfg.t_width, ids.shape[1]
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][:L][None]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    x = _transformer(sd, "", x, cfg.t_layers, cfg.t_heads, mask)
    x = F.layer_norm(x, (D,), sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), ids.argmax(dim=-1)] @ sd["text_projection"]


def preprocess(img, res=224):
    """clip_preprocess minus ToRGB / ToTensor (clean_clip.py:14-17): Resize(res, bicubic) -> CenterCrop(res) ->
    Normalize; img [B, 3, H, W] in [0, 1]. torchvision 0.12 (the reference's environment) resizes tensors with
    F.interpolate(mode="bicubic", align_corners=False) and no antialiasing."""
    H, W = img.shape[-2:]
    if H <= W:
        nh, nw = res, int(res * W / H)
    else:
        nh, nw = int(res * H / W), res
    x = F.interpolate(img, size=(nh, nw), mode="bicubic", align_corners=False)
    t, l = int(round((nh - res) / 2.0)), int(round((nw - res) / 2.0))
    x = x[..., t:t + res, l:l + res]
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1).to(x)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1).to(x)
    return (x - mean) / std


def directional_scores(img_f, orig_f, src_f, tgt_f):
    """clean_clip.py:24-41 on un-normalised features -> (clip_score, dclip_score), each [B]"""
    n = lambda t: t / t.norm(dim=-1, keepdim=True)
    img_f, orig_f, src_f, tgt_f = n(img_f), n(orig_f), n(src_f), n(tgt_f)
    return (img_f * tgt_f).sum(-1), (n(img_f - orig_f) * n(tgt_f - src_f)).sum(-1)
