"""CPU oracle: functional fp32 restatement of the networks on the CycleDiffusion hot path.

TEST INFRASTRUCTURE — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker. The product path never routes through it.

Parity pin: the reference has no tests or golden vectors; this restatement is pinned
(a) against the reference's own modules imported in-process (tests/test_oracle_vs_reference.py,
runs wherever /root/reference exists) and (b) against fixtures those modules produced, committed
under tests/golden/ by oracle/gen_golden.py.

Every function works on a flat state_dict keyed by the reference's parameter names and follows:
  UNetModel (SD / LDM)        model/lib/stable_diffusion/ldm/modules/diffusionmodules/openaimodel.py:413-742
  SpatialTransformer & co     model/lib/stable_diffusion/ldm/modules/attention.py:37-261
  UNetModel (improved DDPM)   model/lib/ddpm_ddim/models/improved_ddpm/unet.py:137-668
  Encoder / Decoder / Attn    model/lib/stable_diffusion/ldm/modules/diffusionmodules/model.py:33-568
  AutoencoderKL               model/lib/stable_diffusion/ldm/models/autoencoder.py:324-333
  DiagonalGaussianDistribution model/lib/stable_diffusion/ldm/modules/distributions/distributions.py:24-37
  DDPM (Ho et al.)            model/lib/ddpm_ddim/models/ddpm/diffusion.py:6-337
"""
import math

import torch
import torch.nn.functional as F


def _gn(sd, p, x, eps):
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def timestep_embedding_cos_sin(t, dim):
    """util.py:152-172 — cat([cos, sin]), freqs = exp(-ln(1e4) * k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def timestep_embedding_sin_cos(t, dim):
    """ddpm/diffusion.py:6-24 — cat([sin, cos]), divisor half-1."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -e)
    a = t.float()[:, None] * f[None, :]
    return torch.cat([torch.sin(a), torch.cos(a)], dim=1)


# ----------------------------------------------------------------------------- openai-style U-Net
class OpenAIUNetCfg:
    def __init__(self, in_channels, out_channels, model_channels, num_res_blocks, channel_mult, attn_ds,
                 num_heads=-1, num_head_channels=-1, use_spatial_transformer=False, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def _resblock(sd, p, x, emb, cfg, up=False, down=False):
    """openaimodel.py:255-275 / improved_ddpm/unet.py:236-258."""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = _conv(sd, p + ".in_layers.2", h)
    e = _lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    if cfg.use_scale_shift_norm:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(sd, p + ".out_layers.0", h, 1e-5) * (1 + scale) + shift
        h = _conv(sd, p + ".out_layers.3", F.silu(h))
    else:
        h = h + e
        h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = _conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def _mha(q, k, v, heads):
    """CrossAttention core (attention.py:176-193): q [B,N,C], k/v [B,M,C]."""
    B, N, C = q.shape
    d = C // heads
    qh = q.view(B, N, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    sim = torch.einsum("bhid,bhjd->bhij", qh, kh) * (d ** -0.5)
    out = torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), vh)
    return out.transpose(1, 2).reshape(B, N, C)


def _spatial_transformer(sd, p, x, ctx, heads):
    """attention.py:250-261 + BasicTransformerBlock 211-215 + GEGLU 37-44."""
    b, c, h, w = x.shape
    x_in = x
    x = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, 1e-6), padding=0)
    x = x.flatten(2).transpose(1, 2)  # b (h w) c
    t = p + ".transformer_blocks.0"
    n = F.layer_norm(x, (c,), sd[t + ".norm1.weight"], sd[t + ".norm1.bias"])
    a = _mha(_lin(sd, t + ".attn1.to_q", n), _lin(sd, t + ".attn1.to_k", n), _lin(sd, t + ".attn1.to_v", n), heads)
    x = _lin(sd, t + ".attn1.to_out.0", a) + x
    n = F.layer_norm(x, (c,), sd[t + ".norm2.weight"], sd[t + ".norm2.bias"])
    a = _mha(_lin(sd, t + ".attn2.to_q", n), _lin(sd, t + ".attn2.to_k", ctx), _lin(sd, t + ".attn2.to_v", ctx), heads)
    x = _lin(sd, t + ".attn2.to_out.0", a) + x
    n = F.layer_norm(x, (c,), sd[t + ".norm3.weight"], sd[t + ".norm3.bias"])
    val, gate = _lin(sd, t + ".ff.net.0.proj", n).chunk(2, dim=-1)
    x = _lin(sd, t + ".ff.net.2", val * F.gelu(gate)) + x
    x = x.transpose(1, 2).reshape(b, c, h, w)
    return _conv(sd, p + ".proj_out", x, padding=0) + x_in


def _attention_block_legacy(sd, p, x, heads):
    """improved_ddpm/unet.py:300-345 (AttentionBlock + QKVAttentionLegacy)."""
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(F.group_norm(xf.float(), 32, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5),
                   sd[p + ".qkv.weight"].reshape(3 * c, c, 1), sd[p + ".qkv.bias"])
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, -1).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    wgt = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    wgt = torch.softmax(wgt.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", wgt, v).reshape(b, -1, xf.shape[-1])
    hproj = F.conv1d(a, sd[p + ".proj_out.weight"].reshape(c, c, 1), sd[p + ".proj_out.bias"])
    return (xf + hproj).reshape(b, c, hh, ww)


def openai_unet(sd, cfg, x, t, context=None):
    """UNetModel.forward (openaimodel.py:710-742); structure from the constructor (:516-686)."""
    mc = cfg.model_channels
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding_cos_sin(t, mc))))

    def heads_for(ch):
        return cfg.num_heads if cfg.num_head_channels == -1 else ch // cfg.num_head_channels

    def attn(p, h, ch):
        if cfg.use_spatial_transformer:
            return _spatial_transformer(sd, p, h, context, heads_for(ch))
        return _attention_block_legacy(sd, p, h, heads_for(ch))

    hs = []
    h = _conv(sd, "input_blocks.0.0", x)
    hs.append(h)
    ch, ds, bi = mc, 1, 1
    nlev = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            p = "input_blocks.%d" % bi
            h = _resblock(sd, p + ".0", h, emb, cfg)
            ch = mult * mc
            if ds in cfg.attn_ds:
                h = attn(p + ".1", h, ch)
            hs.append(h)
            bi += 1
        if level != nlev - 1:
            p = "input_blocks.%d.0" % bi
            if cfg.resblock_updown:
                h = _resblock(sd, p, h, emb, cfg, down=True)
            else:
                h = _conv(sd, p + ".op", h, stride=2, padding=1)
            hs.append(h)
            bi += 1
            ds *= 2
    h = _resblock(sd, "middle_block.0", h, emb, cfg)
    h = attn("middle_block.1", h, ch)
    h = _resblock(sd, "middle_block.2", h, emb, cfg)
    oi = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            p = "output_blocks.%d" % oi
            h = torch.cat([h, hs.pop()], dim=1)
            h = _resblock(sd, p + ".0", h, emb, cfg)
            ch = mc * mult
            sub = 1
            if ds in cfg.attn_ds:
                h = attn(p + ".%d" % sub, h, ch)
                sub += 1
            if level and i == cfg.num_res_blocks:
                if cfg.resblock_updown:
                    h = _resblock(sd, p + ".%d" % sub, h, emb, cfg, up=True)
                else:
                    h = _conv(sd, p + ".%d.conv" % sub, F.interpolate(h, scale_factor=2, mode="nearest"))
                ds //= 2
            oi += 1
    return _conv(sd, "out.2", F.silu(_gn(sd, "out.0", h, 1e-5)))


# ----------------------------------------------------------------------------- pytorch_diffusion blocks
def _swish(x):
    return x * torch.sigmoid(x)


def _resnet_block(sd, p, x, temb=None):
    """model.py:121-143 / ddpm/diffusion.py:116-134."""
    h = _conv(sd, p + ".conv1", _swish(_gn(sd, p + ".norm1", x, 1e-6)))
    if temb is not None:
        h = h + _lin(sd, p + ".temb_proj", _swish(temb))[:, :, None, None]
    h = _conv(sd, p + ".conv2", _swish(_gn(sd, p + ".norm2", h, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def _attn_block(sd, p, x):
    """model.py:178-202: single head, d = C, scale C^-0.5."""
    h_ = _gn(sd, p + ".norm", x, 1e-6)
    q, k, v = (_conv(sd, p + "." + n, h_, padding=0) for n in "qkv")
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    h_ = torch.bmm(v.reshape(b, c, h * w), w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(sd, p + ".proj_out", h_, padding=0)


def _down(sd, p, x):
    """model.py:72-76: pad (0,1,0,1) then 3x3 stride 2."""
    return _conv(sd, p + ".conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0)


def _up(sd, p, x):
    return _conv(sd, p + ".conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))


class VAECfg:
    def __init__(self, ch, ch_mult, num_res_blocks, z_channels=4, embed_dim=4, in_channels=3, out_ch=3):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def vae_encode_moments(sd, cfg, x):
    """Encoder.forward (model.py:434-459) + quant_conv (autoencoder.py:324-328)."""
    h = _conv(sd, "encoder.conv_in", x)
    n = len(cfg.ch_mult)
    for l in range(n):
        for b in range(cfg.num_res_blocks):
            h = _resnet_block(sd, "encoder.down.%d.block.%d" % (l, b), h)
        if l != n - 1:
            h = _down(sd, "encoder.down.%d.downsample" % l, h)
    h = _resnet_block(sd, "encoder.mid.block_1", h)
    h = _attn_block(sd, "encoder.mid.attn_1", h)
    h = _resnet_block(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", _swish(_gn(sd, "encoder.norm_out", h, 1e-6)))
    return _conv(sd, "quant_conv", h, padding=0)


def posterior_sample(moments, noise=None):
    """distributions.py:24-37; noise=None -> mode()."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if noise is None:
        return mean
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def vq_quantize(z, codebook):
    """VectorQuantizer2.forward of taming-transformers (taming/modules/vqvae/quantize.py, git master - the version the
    reference's README installs, README.md:85-91; NOT vendored and not installed here, so this restatement of the
    published algorithm is unpinned by the reference: "parity unpinned" for this one operator) as
    VQModelInterface.decode calls it (model/lib/latentdiff/ldm/models/autoencoder.py:274-280):
    z [B, C, H, W] -> nearest codebook row per position by d = |z|^2 + |e|^2 - 2 z.e, returned with the
    straight-through form z + (z_q - z)."""
    zp = z.permute(0, 2, 3, 1).contiguous()
    zf = zp.view(-1, zp.shape[-1])
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1) - 2 * torch.einsum(
        "bd,dn->bn", zf, codebook.t())
    idx = torch.argmin(d, dim=1)
    zq = codebook[idx].view(zp.shape)
    zq = zp + (zq - zp)
    return zq.permute(0, 3, 1, 2).contiguous()


def vae_decode(sd, cfg, z):
    """post_quant_conv + Decoder.forward (autoencoder.py:330-333; model.py:535-568)."""
    h = _conv(sd, "decoder.conv_in", _conv(sd, "post_quant_conv", z, padding=0))
    h = _resnet_block(sd, "decoder.mid.block_1", h)
    h = _attn_block(sd, "decoder.mid.attn_1", h)
    h = _resnet_block(sd, "decoder.mid.block_2", h)
    n = len(cfg.ch_mult)
    for l in reversed(range(n)):
        for b in range(cfg.num_res_blocks + 1):
            h = _resnet_block(sd, "decoder.up.%d.block.%d" % (l, b), h)
        if l != 0:
            h = _up(sd, "decoder.up.%d.upsample" % l, h)
    return _conv(sd, "decoder.conv_out", _swish(_gn(sd, "decoder.norm_out", h, 1e-6)))


class HoCfg:
    def __init__(self, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, in_channels=3, out_ch=3):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def ho_unet(sd, cfg, x, t):
    """DDPM.forward (ddpm/diffusion.py:292-337)."""
    temb = _lin(sd, "temb.dense.1", _swish(_lin(sd, "temb.dense.0", timestep_embedding_sin_cos(t, cfg.ch))))
    n = len(cfg.ch_mult)
    res = cfg.resolution
    hs = [_conv(sd, "conv_in", x)]
    for l in range(n):
        for b in range(cfg.num_res_blocks):
            h = _resnet_block(sd, "down.%d.block.%d" % (l, b), hs[-1], temb)
            if res in cfg.attn_resolutions:
                h = _attn_block(sd, "down.%d.attn.%d" % (l, b), h)
            hs.append(h)
        if l != n - 1:
            hs.append(_down(sd, "down.%d.downsample" % l, hs[-1]))
            res //= 2
    h = hs[-1]
    h = _resnet_block(sd, "mid.block_1", h, temb)
    h = _attn_block(sd, "mid.attn_1", h)
    h = _resnet_block(sd, "mid.block_2", h, temb)
    for l in reversed(range(n)):
        for b in range(cfg.num_res_blocks + 1):
            h = _resnet_block(sd, "up.%d.block.%d" % (l, b), torch.cat([h, hs.pop()], dim=1), temb)
            if res in cfg.attn_resolutions:
                h = _attn_block(sd, "up.%d.attn.%d" % (l, b), h)
        if l != 0:
            h = _up(sd, "up.%d.upsample" % l, h)
            res *= 2
    return _conv(sd, "conv_out", _swish(_gn(sd, "norm_out", h, 1e-6)))


# ----------------------------------------------------------------------------- deterministic synthetic weights
ZERO_MODULE_SUFFIXES = ("out_layers.3.weight", "proj_out.weight", "out.2.weight")


def synth_state_dict(named_shapes, seed):
    """Seeded synthetic weights from a (name, shape) list — independent of any module class, so
    fixtures can be rebuilt without the reference. Scales follow SURVEY.md §8(d): torch's default-init
    scale for matrices (std 1/sqrt(3 fan_in)), and every tensor the reference creates with
    zero_module() (ResBlock out conv, transformer / attention proj_out, U-Net out conv) drawn from
    N(0, 0.02^2) instead of zeros (otherwise eps_hat == 0); norm gains ~ 1, biases small."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in named_shapes:
        shape = tuple(shape)
        if len(shape) == 1:
            is_gain = name.endswith(".weight")
            v = torch.randn(shape, generator=g) * (0.1 if is_gain else 0.05)
            sd[name] = (1.0 + v) if is_gain else v
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = 0.02 if name.endswith(ZERO_MODULE_SUFFIXES) else 1.0 / math.sqrt(3.0 * fan_in)
            sd[name] = torch.randn(shape, generator=g) * std
    return sd
