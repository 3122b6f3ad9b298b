"""Host-side handle on the HIP engine: device memory and stream come from PyTorch-ROCm, every
computation goes through the C ABI (include/cyclediff.h)."""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from ._ffi import NetDesc, check, ptr


def make_desc(kind, *, image_size, in_channels, out_channels, model_channels, num_res_blocks, channel_mult,
              attn=(), num_heads=-1, num_head_channels=-1, use_spatial_transformer=False, context_dim=0,
              transformer_depth=1, use_scale_shift_norm=False, resblock_updown=False, conv_resample=True,
              z_channels=0, embed_dim=0, double_z=False, precision=_ffi.CD_PREC_16, n_embed=0):
    d = NetDesc()
    d.precision = int(precision)
    d.n_embed = int(n_embed)
    d.kind = kind
    d.image_size = image_size
    d.in_channels, d.out_channels = in_channels, out_channels
    d.model_channels, d.num_res_blocks = model_channels, num_res_blocks
    d.n_mult = len(channel_mult)
    for i, m in enumerate(channel_mult):
        d.channel_mult[i] = int(m)
    d.n_attn = len(attn)
    for i, a in enumerate(attn):
        d.attn[i] = int(a)
    d.num_heads, d.num_head_channels = num_heads, num_head_channels
    d.use_spatial_transformer = int(use_spatial_transformer)
    d.context_dim = int(context_dim or 0)
    d.transformer_depth = transformer_depth
    d.use_scale_shift_norm = int(use_scale_shift_norm)
    d.resblock_updown = int(resblock_updown)
    d.conv_resample = int(conv_resample)
    d.z_channels, d.embed_dim, d.double_z = z_channels, embed_dim, int(double_z)
    return d


# ---- the architectures the reference ships configs for ------------------------------------------
def sd_v1_unet_desc(image_size=64):
    """model/lib/stable_diffusion/configs/stable-diffusion/v1-inference.yaml:29-44"""
    return make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=image_size, in_channels=4, out_channels=4,
                     model_channels=320, num_res_blocks=2, channel_mult=(1, 2, 4, 4), attn=(4, 2, 1),
                     num_heads=8, use_spatial_transformer=True, context_dim=768)


def ldm_text_unet_desc(image_size=32):
    """model/lib/latentdiff/configs/latent-diffusion/txt2img-1p4B-eval.yaml (context 1280)"""
    return make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=image_size, in_channels=4, out_channels=4,
                     model_channels=320, num_res_blocks=2, channel_mult=(1, 2, 4, 4), attn=(4, 2, 1),
                     num_heads=8, use_spatial_transformer=True, context_dim=1280)


def kl_f8_vae_desc():
    """v1-inference.yaml:46-65 (first_stage_config ddconfig)"""
    return make_desc(_ffi.CD_NET_VAE_KL, image_size=0, in_channels=3, out_channels=3, model_channels=128,
                     num_res_blocks=2, channel_mult=(1, 2, 4, 4), z_channels=4, embed_dim=4, double_z=True)


def vq_f4_vae_desc():
    """first_stage_config of the unconditional LDMs (model/lib/latentdiff/models/ldm/{celeba256,ffhq256}/config.yaml):
    VQModelInterface, embed_dim 3, n_embed 8192, ch 128, ch_mult (1, 2, 4), 2 res blocks, double_z False"""
    return make_desc(_ffi.CD_NET_VAE_KL, image_size=0, in_channels=3, out_channels=3, model_channels=128,
                     num_res_blocks=2, channel_mult=(1, 2, 4), z_channels=3, embed_dim=3, double_z=False, n_embed=8192)


def ldm_uncond_unet_desc(image_size=64):
    """unet_config of celeba256 / ffhq256 (same files): UNetModel with AttentionBlocks (legacy QKV order), 224
    channels, mult (1, 2, 3, 4), attention at downsample rates 2 / 4 / 8, 32-channel heads, no conditioning"""
    return make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=image_size, in_channels=3, out_channels=3,
                     model_channels=224, num_res_blocks=2, channel_mult=(1, 2, 3, 4), attn=(8, 4, 2),
                     num_head_channels=32)


def clip_text_desc(width=768, layers=12, heads=12, mlp=3072, vocab=49408, positions=77):
    """HF CLIPTextConfig of "openai/clip-vit-large-patch14" (ldm/modules/encoders/modules.py:138-141)"""
    return make_desc(_ffi.CD_NET_CLIP_TEXT, image_size=positions, in_channels=vocab, out_channels=width,
                     model_channels=width, num_res_blocks=layers, channel_mult=(1,), num_heads=heads,
                     context_dim=mlp)


def oclip_text_desc(width=512, layers=12, heads=8, vocab=49408, positions=77, embed=512):
    """text tower of OpenAI CLIP ViT-B/32 (clip/model.py CLIP.__init__: transformer_width 512, heads 8, layers 12)"""
    return make_desc(_ffi.CD_NET_OCLIP_TEXT, image_size=positions, in_channels=vocab, out_channels=embed,
                     model_channels=width, num_res_blocks=layers, channel_mult=(1,), num_heads=heads,
                     context_dim=4 * width)


def oclip_vision_desc(width=768, layers=12, heads=12, resolution=224, patch=32, embed=512):
    """image tower of OpenAI CLIP ViT-B/32 (clip/model.py VisionTransformer: width 768, patch 32, 12 layers)"""
    return make_desc(_ffi.CD_NET_OCLIP_VISION, image_size=resolution, in_channels=3, out_channels=embed,
                     model_channels=width, num_res_blocks=layers, channel_mult=(1,), num_heads=heads,
                     context_dim=4 * width, z_channels=patch)


def bert_xtransformer_desc(width=1280, layers=32, vocab=30522, positions=77, heads=8, dim_head=64):
    """BERTEmbedder(n_embed=1280, n_layer=32) of txt2img-1p4B-eval.yaml: x-transformers Encoder (8 heads x 64, FF x4)"""
    return make_desc(_ffi.CD_NET_BERT_XTR, image_size=positions, in_channels=vocab, out_channels=width,
                     model_channels=width, num_res_blocks=layers, channel_mult=(1,), num_heads=heads,
                     num_head_channels=dim_head, context_dim=4 * width)


def afhq_iddpm_desc(image_size=256, precision=_ffi.CD_PREC_16):
    """improved_ddpm/script_util.py:5-22,45-104 (AFHQ_DICT; learn_sigma -> 6 output channels)"""
    return make_desc(_ffi.CD_NET_UNET_OPENAI, image_size=image_size, in_channels=3, out_channels=6,
                     model_channels=128, num_res_blocks=1, channel_mult=(1, 1, 2, 2, 4, 4),
                     attn=(image_size // 16,), num_heads=4, num_head_channels=64,
                     use_scale_shift_norm=True, resblock_updown=True, precision=precision)


def ho_ddpm_desc(image_size, ch, ch_mult, num_res_blocks, attn_resolutions, in_channels=3, out_ch=3,
                 precision=_ffi.CD_PREC_16):
    """ddpm/diffusion.py:192-205 (config.model.*)"""
    return make_desc(_ffi.CD_NET_UNET_HO, image_size=image_size, in_channels=in_channels, out_channels=out_ch,
                     model_channels=ch, num_res_blocks=num_res_blocks, channel_mult=tuple(ch_mult),
                     attn=tuple(attn_resolutions), precision=precision)


class Engine:
    """One engine per rank / stream (cd_engine_create)."""

    def __init__(self, device="cuda:0", workspace_bytes=None):
        self.lib = _ffi.load_library()
        if not torch.cuda.is_available():
            raise _ffi.EngineError("no HIP device visible to PyTorch: the CycleDiffusion engine has no CPU fallback")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        if workspace_bytes is None:
            free, _total = torch.cuda.mem_get_info(self.device)
            # the bump arena is reserved up front: 32 GB covers the largest folded-ensemble VAE decode (32 images
            # at 512x512) many times over and leaves room for several engines per GPU (bench.py replicas)
            workspace_bytes = int(min(32 << 30, free * 0.45))
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        check(self.lib.cd_engine_create(C.c_void_p(self.stream.cuda_stream), C.c_size_t(workspace_bytes), C.byref(h)))
        self.h = h
        self._descs = {}

    def synchronize(self):
        """Wait for the engine's stream without spinning (cd_engine_synchronize)."""
        check(self.lib.cd_engine_synchronize(self.h))

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.lib.cd_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- networks
    def create_net(self, desc):
        nid = C.c_int(-1)
        check(self.lib.cd_net_create(self.h, C.byref(desc), C.byref(nid)))
        self._descs[nid.value] = desc
        return nid.value

    def net_params(self, net):
        n = C.c_int()
        check(self.lib.cd_net_param_count(self.h, net, C.byref(n)))
        out = []
        buf = C.create_string_buffer(512)
        nd = C.c_int()
        shp = (C.c_int64 * 4)()
        for i in range(n.value):
            check(self.lib.cd_net_param_info(self.h, net, i, buf, 512, C.byref(nd), shp))
            out.append((buf.value.decode(), tuple(int(shp[j]) for j in range(nd.value))))
        return out

    def load_param(self, net, name, tensor):
        t = tensor.detach().to(torch.float32).cpu().contiguous()
        shp = (C.c_int64 * max(1, t.dim()))(*t.shape)
        check(self.lib.cd_net_load_param(self.h, net, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), shp))

    def load_state_dict(self, net, sd, prefix="", strict=True):
        """Load weights keyed by the reference's state_dict names (txt2img.py:25-42)."""
        names = [n for n, _ in self.net_params(net)]
        for n in names:
            key = prefix + n
            if key in sd:
                self.load_param(net, n, sd[key])
            elif strict:
                raise KeyError("missing parameter %s" % key)
        return self.missing(net)

    def missing(self, net):
        n = C.c_int()
        buf = C.create_string_buffer(512)
        check(self.lib.cd_net_missing_params(self.h, net, C.byref(n), buf, 512))
        return n.value, buf.value.decode()

    _SYNTH_CACHE = {}  # (seed, names+shapes) -> tensors; lets several engines of one process share the host work

    def random_init(self, net, seed=0, std=0.02, cache=False):
        """Synthetic weights for benchmarks: N(0, fan-in scaled) matrices, unit norms.
        (There are no checkpoints in the tree)"""
        params = self.net_params(net)
        key = (seed, tuple((n, tuple(s)) for n, s in params))
        if cache and key in Engine._SYNTH_CACHE:
            for (name, _), t in zip(params, Engine._SYNTH_CACHE[key]):
                self.load_param(net, name, t)
            n, first = self.missing(net)
            assert n == 0, first
            return
        made = []
        g = torch.Generator().manual_seed(seed)
        for name, shape in params:
            if len(shape) == 1:
                base = name.rsplit(".", 1)[0]
                is_norm = name.endswith("weight") and (".norm" in name or "in_layers.0" in name or "out_layers.0" in name
                                                       or name.startswith("out.0") or "norm_out" in name
                                                       or "layer_norm" in name or ".ln_" in name
                                                       or name.startswith("ln_"))
                if is_norm:
                    t = 1.0 + 0.05 * torch.randn(shape, generator=g)
                else:
                    t = 0.02 * torch.randn(shape, generator=g)
                del base
            else:
                fan_in = int(np.prod(shape[1:]))
                t = torch.randn(shape, generator=g) * (1.0 / np.sqrt(fan_in))
            self.load_param(net, name, t)
            if cache:
                made.append(t)
        if cache:
            Engine._SYNTH_CACHE[key] = made
        n, first = self.missing(net)
        assert n == 0, first

    # ---- forward passes
    def _f32(self, t):
        assert t.is_cuda and t.dtype == torch.float32
        return t.contiguous()

    def unet_forward(self, net, x, t, ctx=None):
        x, t = self._f32(x), self._f32(t.float())
        ctx = self._f32(ctx) if ctx is not None else None
        out_ch = self._out_channels(net)
        y = torch.empty((x.shape[0], out_ch, x.shape[2], x.shape[3]), device=x.device, dtype=torch.float32)
        check(self.lib.cd_unet_forward(self.h, net, ptr(x), ptr(t), ptr(ctx), x.shape[0],
                                       ctx.shape[1] if ctx is not None else 0, ptr(y)))
        return y

    def _out_channels(self, net):
        return self._descs[net].out_channels

    def text_encode(self, net, tokens):
        """tokens [B, L] integer ids -> last_hidden_state [B, L, width] fp32 (FrozenCLIPEmbedder.forward)."""
        ids = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        B, L = ids.shape
        out = torch.empty((B, L, self._descs[net].model_channels), device=self.device, dtype=torch.float32)
        check(self.lib.cd_text_encode(self.h, net, ptr(ids), B, L, ptr(out)))
        return out

    def clip_text_features(self, net, tokens):
        """model.encode_text(tokens) of OpenAI CLIP: [B, L] ids -> [B, embed] fp32 (un-normalised)."""
        ids = tokens.to(device=self.device, dtype=torch.int32).contiguous()
        B, L = ids.shape
        out = torch.empty((B, self._descs[net].out_channels), device=self.device, dtype=torch.float32)
        check(self.lib.cd_clip_text_features(self.h, net, ptr(ids), B, L, ptr(out)))
        return out

    def clip_image_features(self, net, img):
        """model.encode_image(img) of OpenAI CLIP: preprocessed [B, 3, R, R] fp32 -> [B, embed] fp32."""
        img = self._f32(img)
        d = self._descs[net]
        assert img.shape[1:] == (3, d.image_size, d.image_size), img.shape
        out = torch.empty((img.shape[0], d.out_channels), device=self.device, dtype=torch.float32)
        check(self.lib.cd_clip_image_features(self.h, net, ptr(img), img.shape[0], ptr(out)))
        return out

    def vae_encode(self, net, img, noise=None, seed=0, sample=True, scale=0.18215):
        img = self._f32(img)
        d = self._descs[net]
        B, _, R, _ = img.shape
        f = 2 ** (d.n_mult - 1)
        z = torch.empty((B, d.embed_dim, R // f, R // f), device=img.device, dtype=torch.float32)
        if noise is not None:
            noise = self._f32(noise)
        check(self.lib.cd_vae_encode(self.h, net, ptr(img), ptr(noise), C.c_uint64(seed), B, R, int(sample),
                                     C.c_float(scale), ptr(z)))
        return z

    def vae_decode(self, net, z, scale=0.18215, out_mul=1.0, out_add=0.0):
        z = self._f32(z)
        d = self._descs[net]
        B, _, hl, _ = z.shape
        f = 2 ** (d.n_mult - 1)
        img = torch.empty((B, d.out_channels, hl * f, hl * f), device=z.device, dtype=torch.float32)
        check(self.lib.cd_vae_decode(self.h, net, ptr(z), B, hl, C.c_float(scale), C.c_float(out_mul),
                                     C.c_float(out_add), ptr(img)))
        return img

    def dpm_encode(self, net, kind, x0, coef, ctx_c=None, ctx_uc=None, guidance=1.0, noise=None, seed=0,
                   last_uses_x0=True):
        """coef: numpy struct array (STEP_COEF_DTYPE) with K+1 rows; returns z [B, K+1, C, H, W]."""
        x0 = self._f32(x0)
        K = len(coef) - 1
        B, Cc, H, W = x0.shape
        z = torch.empty((B, K + 1, Cc, H, W), device=x0.device, dtype=torch.float32)
        coef = np.ascontiguousarray(coef)
        L = ctx_c.shape[1] if ctx_c is not None else (ctx_uc.shape[1] if ctx_uc is not None else 0)
        check(self.lib.cd_dpm_encode(self.h, net, kind, ptr(x0),
                                     ptr(self._f32(ctx_c)) if ctx_c is not None else None,
                                     ptr(self._f32(ctx_uc)) if ctx_uc is not None else None,
                                     L, C.c_float(guidance), B, K, C.c_void_p(coef.ctypes.data),
                                     ptr(self._f32(noise)) if noise is not None else None,
                                     C.c_uint64(seed), int(last_uses_x0), ptr(z)))
        return z

    def ddim_decode(self, net, kind, z, coef, n_eps=None, ctx_c=None, ctx_uc=None, guidance=1.0, noise_tail=None,
                    seed=0):
        """z [B, T, C, H, W]; coef K rows; returns x [B, C, H, W]. `guidance`: one scale, or a sequence / tensor of B
        scales (each neither 0 nor 1: cd_ddim_decode_v) when the samples of the batch differ in it."""
        z = self._f32(z)
        B, T, Cc, H, W = z.shape
        K = len(coef)
        if n_eps is None:
            n_eps = T - 1
        x = torch.empty((B, Cc, H, W), device=z.device, dtype=torch.float32)
        coef = np.ascontiguousarray(coef)
        L = ctx_c.shape[1] if ctx_c is not None else (ctx_uc.shape[1] if ctx_uc is not None else 0)
        cc = ptr(self._f32(ctx_c)) if ctx_c is not None else None
        cu = ptr(self._f32(ctx_uc)) if ctx_uc is not None else None
        nt = ptr(self._f32(noise_tail)) if noise_tail is not None else None
        if isinstance(guidance, (int, float)):
            check(self.lib.cd_ddim_decode(self.h, net, kind, ptr(z), T, n_eps, cc, cu, L, C.c_float(guidance), B, K,
                                          C.c_void_p(coef.ctypes.data), nt, C.c_uint64(seed), ptr(x)))
        else:
            g = torch.as_tensor(guidance, dtype=torch.float32).to(z.device).contiguous()
            if g.numel() != B or bool(((g == 0) | (g == 1)).any()):
                raise ValueError("per-sample guidance: B scales, none of them 0 or 1")
            check(self.lib.cd_ddim_decode_v(self.h, net, kind, ptr(z), T, n_eps, cc, cu, L, ptr(g), B, K,
                                            C.c_void_p(coef.ctypes.data), nt, C.c_uint64(seed), ptr(x)))
            self._keep = g  # the launches read it asynchronously
        return x

    def cycle_translate(self, net, kind, x0, coef_enc, coef_dec, enc_ctx_c=None, enc_ctx_uc=None, enc_guidance=1.0,
                        dec_ctx_c=None, dec_ctx_uc=None, dec_guidance=1.0, n_dec=1, noise=None, seed=0, last_uses_x0=True):
        """The coupled loop (cd_cycle_translate): dpm_encode and ddim_decode of the same network over the whole chain with ONE
        forward per step over [encoder rows | decoder rows]. x0 [B, C, H, W]; dec_ctx_* [n_dec * B, L, Dc] (decoder row
        j * B + b decodes encoder sample b); dec_guidance one scale or n_dec * B of them (none 0 or 1).
        Returns (z [B, K+1, C, H, W], x [n_dec * B, C, H, W])."""
        x0 = self._f32(x0)
        K = len(coef_dec)
        assert len(coef_enc) == K + 1
        B, Cc, H, W = x0.shape
        z = torch.empty((B, K + 1, Cc, H, W), device=x0.device, dtype=torch.float32)
        x = torch.empty((n_dec * B, Cc, H, W), device=x0.device, dtype=torch.float32)
        coef_enc, coef_dec = np.ascontiguousarray(coef_enc), np.ascontiguousarray(coef_dec)
        ctxs = [self._f32(t) if t is not None else None for t in (enc_ctx_c, enc_ctx_uc, dec_ctx_c, dec_ctx_uc)]
        L = next((t.shape[1] for t in ctxs if t is not None), 0)
        for t, rows in zip(ctxs, (B, B, n_dec * B, n_dec * B)):
            assert t is None or t.shape[0] == rows, (t.shape, rows)
        gvec, gscalar = None, 1.0
        if isinstance(dec_guidance, (int, float)):
            gscalar = float(dec_guidance)
        else:
            gvec = torch.as_tensor(dec_guidance, dtype=torch.float32).to(x0.device).contiguous()
            if gvec.numel() != n_dec * B or bool(((gvec == 0) | (gvec == 1)).any()):
                raise ValueError("per-sample guidance: n_dec * B scales, none of them 0 or 1")
        nz = self._f32(noise) if noise is not None else None
        check(self.lib.cd_cycle_translate(self.h, net, kind, ptr(x0), ptr(ctxs[0]), ptr(ctxs[1]), C.c_float(enc_guidance),
                                          ptr(ctxs[2]), ptr(ctxs[3]), C.c_float(gscalar), ptr(gvec), L, B, n_dec, K,
                                          C.c_void_p(coef_enc.ctypes.data), C.c_void_p(coef_dec.ctypes.data), ptr(nz),
                                          C.c_uint64(seed), int(last_uses_x0), ptr(z), ptr(x)))
        self._keep = (gvec, ctxs, nz)  # the launches read them asynchronously
        return z, x

    def pix_refine(self, net, kind, x, coef, noise=None, seed=0):
        x = self._f32(x).clone()
        R = len(coef) - 1
        coef = np.ascontiguousarray(coef)
        check(self.lib.cd_pix_refine(self.h, net, kind, ptr(x), x.shape[0], R, C.c_void_p(coef.ctypes.data),
                                     ptr(self._f32(noise)) if noise is not None else None, C.c_uint64(seed)))
        return x

    def prof_enable(self, on=True):
        check(self.lib.cd_prof_enable(self.h, int(on)))

    def prof_collect(self):
        """(launches, total_ms, total_flops) of the implicit-GEMM kernel since prof_enable."""
        n, ms, fl = C.c_int(), C.c_double(), C.c_double()
        check(self.lib.cd_prof_collect(self.h, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def mfma_sustained(self, target_ms=300):
        """(TFLOP/s, GHz) of a bare 16-bit MFMA loop on every CU for ~target_ms: what this device's matrix cores sustain
        under its power cap (csrc/diag.hip). Measurement support for bench.py, not part of the path."""
        tf, ghz = C.c_float(), C.c_float()
        check(self.lib.cd_op_bench_mfma_sustained(self.h, int(target_ms), C.byref(tf), C.byref(ghz)))
        return tf.value, ghz.value

    def workspace_high_water(self):
        v = C.c_size_t()
        check(self.lib.cd_engine_workspace_high_water(self.h, C.byref(v)))
        return v.value
