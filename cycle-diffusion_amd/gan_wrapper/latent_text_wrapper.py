"""Text-conditioned latent wrappers on the HIP engine: drop-ins for
  SDStochasticTextWrapper          model/gan_wrapper/stable_diffusion_stochastic_text_wrapper.py:102-253
  LatentDiffStochasticTextWrapper  model/gan_wrapper/latentdiff_stochastic_text_wrapper.py
Same constructor kwargs (the [gan] section of config/experiments/*.cfg), same
encode(image, encode_text) -> z_ensemble and forward(z_ensemble, original_img, encode_text,
decode_text) -> img contracts, same ensemble ordering (trial -> encoder scale -> skip_steps, then
decoder scales) and z layout torch.stack(z_list, dim=1).view(bsz, -1).

Conditioning enters through `cond_stage` (a callable list[str] -> [B, 77, context_dim]). With a real checkpoint
the text encoder stored in it is used, as in the reference: CLIP ViT-L/14 text (SD) or the BERT-tokenizer
x-transformer (LDM), both on the engine (text_encoders.py), and a missing tokenizer vocabulary is an error.
`[gan] text_encoder = clip | bert` forces one. Only in synthetic-weight runs (CYCLEDIFF_SYNTHETIC_WEIGHTS=1, no
checkpoint) a deterministic stand-in embedding is used, and SAYS SO in its name. `[gan] ranker = directional_clip`
is the reference's DirectionalCLIP on the engine (ranker.py); its ViT-B/32 weights come from `ranker_path` /
CYCLEDIFF_CLIP_RANKER (the `clip.load("ViT-B/32")` state_dict saved with torch.save).

Ensemble members that share (guidance scale, skip) differ only in their noise: they are folded into the batch
dimension of ONE engine call - same draws, same member order, larger GEMMs.
"""
import hashlib
import os

import numpy as np

import torch

from .. import _ffi, schedule
from ..engine import kl_f8_vae_desc, ldm_text_unet_desc, sd_v1_unet_desc
from ..runtime import get_engine, load_or_init_weights, read_checkpoint, synthetic_allowed


class StandInTextEmbedder:
    """Deterministic N(0,1) embedding per string — NOT a text encoder; keeps tensor shapes and the
    call pattern of get_learned_conditioning (ddpm.py:545-556) when no CLIP/BERT weights exist."""

    def __init__(self, context_dim, length=77):
        self.context_dim, self.length = context_dim, length

    def __call__(self, texts):
        out = []
        for t in texts:
            seed = int.from_bytes(hashlib.sha256(t.encode("utf-8")).digest()[:8], "little") % (2 ** 63)
            g = torch.Generator().manual_seed(seed)
            out.append(torch.randn(self.length, self.context_dim, generator=g))
        return torch.stack(out, 0)


TEXT_PRECISIONS = {"fp16": _ffi.CD_PREC_16, "16": _ffi.CD_PREC_16, "fp32": _ffi.CD_PREC_F32, "fp32x3": _ffi.CD_PREC_F32X3}


class _LatentStochasticTextWrapper(torch.nn.Module):
    # subclass constants
    UNET_DESC = None
    RESOLUTION = None
    SAMPLE_POSTERIOR = True
    LINEAR_START, LINEAR_END = 0.00085, 0.0120
    SCALE_FACTOR = 0.18215
    VAE_DESC = staticmethod(kl_f8_vae_desc)
    MAX_FOLD = 32  # samples per engine call when ensemble members are folded into the batch
    # first-stage calls are cut at this many pixels (32 images of 512 x 512): the KL-f8 decoder's 512 x 512 x 128-channel
    # level is 64 MiB per image and tensor in 16 bits (256 MiB in fp32), several of them live at once - a look-ahead fold of
    # 64 images through ONE decode would need ~25 GB of workspace for 1.5 % of the path's FLOPs
    VAE_MAX_PIXELS = 32 * 512 * 512
    COUPLE_MAX_TOKENS = 96 * 64 * 64  # rows x latent tokens of one coupled forward (translate())

    def __init__(self, source_model_type, custom_steps, eta, white_box_steps, skip_steps,
                 encoder_unconditional_guidance_scales=None, decoder_unconditional_guidance_scales=None,
                 n_trials=None, cond_stage=None, ranker=None, device=None, text_encoder=None,
                 noise_on_cpu=False, fold_ensemble=True, ranker_path=None, precision="fp16", couple=True):
        super().__init__()
        # `[gan] precision`: arithmetic of the U-Net AND (round 5) of the first stage - the reference's `precision = "full"` covers
        # both (sd_wrapper:117, autoencoder.py:324-333); the text towers stay 16-bit (their output is the conditioning, rounded
        # once). 'fp16' (default):
        # 16-bit storage, the benchmarked engine (55 dB against the reference on C2, 99-step self-cycle 2e-2 rms); 'fp32': the
        # reference's own arithmetic (`precision = "full"`, stable_diffusion_stochastic_text_wrapper.py:117) - fp32 storage,
        # fp32 matrix instructions, fp32 flash attention; 'fp32x3': the same network with every GroupNorm- / LayerNorm-fed
        # convolution and projection as three-term split-fp16 products (include/cyclediff.h CD_PREC_F32X3)
        if str(precision) not in TEXT_PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(TEXT_PRECISIONS))
        self.precision = str(precision)
        if self.precision == "fp32x3" and _ffi.load_library().cd_act_format() != 1:
            raise ValueError("precision='fp32x3' needs the fp16 build of the library (this is the bf16 build): use 'fp32'")
        self.encoder_unconditional_guidance_scales = encoder_unconditional_guidance_scales
        self.decoder_unconditional_guidance_scales = decoder_unconditional_guidance_scales
        self.n_trials = n_trials
        self.eta, self.custom_steps = eta, custom_steps
        self.white_box_steps, self.skip_steps = white_box_steps, skip_steps
        self.resolution = self.RESOLUTION
        # parity runs draw every noise tensor on the CPU in the reference's order; throughput
        # runs draw on the device
        self.noise_on_cpu, self.fold_ensemble = bool(noise_on_cpu), bool(fold_ensemble)
        # `[gan] couple = False`: translate() runs encode() then forward() as two loops (the reference's own order of work).
        # The coupled loop is taken while its forward - [encoder rows | decoder rows] x latent tokens - stays below
        # COUPLE_MAX_TOKENS: measured on MI355X (profiles/r6_coupled_loop_ab.json) it is +28 % at a batch of 4 (12 rows of
        # 64 x 64 tokens per forward instead of 4, then 8), +7 % on config 3 (192 rows of 32 x 32) and -1.4 % at 64 images of
        # 512 x 512 (192 rows of 64 x 64: a 503 MB activation per 320-channel tensor no longer stays in the 256 MB Infinity
        # Cache between producer and consumer, where the two loops run 64 and 128 rows)
        self.couple = bool(couple) and os.environ.get("CYCLEDIFF_COUPLE", "1") != "0"
        self.couple_max_tokens = int(os.environ.get("CYCLEDIFF_COUPLE_MAX_TOKENS", self.COUPLE_MAX_TOKENS))
        self.last_translate_coupled = None
        self.noise_source = None
        self.engine = get_engine(device)
        udesc = self.UNET_DESC()
        udesc.precision = TEXT_PRECISIONS[self.precision]
        self.channels, self.image_size = udesc.in_channels, udesc.image_size
        self.unet = self.engine.create_net(udesc)
        vdesc = self.VAE_DESC()
        vdesc.precision = TEXT_PRECISIONS[self.precision]
        self.vae = self.engine.create_net(vdesc)
        self.vae_factor = 2 ** (vdesc.n_mult - 1)
        ckpt = self.checkpoint_path(source_model_type)
        ckpt_sd = read_checkpoint(ckpt)
        self.weights_origin = load_or_init_weights(self.engine, ckpt, {
            self.unet: "model.diffusion_model.", self.vae: "first_stage_model."}, state_dict=ckpt_sd)
        if cond_stage is None and text_encoder is None and ckpt_sd is not None:
            # a real checkpoint carries its text encoder (cond_stage_model.*): use it, as the reference does
            text_encoder = "clip" if udesc.context_dim == 768 else "bert"
        strict_tok = ckpt_sd is not None or not synthetic_allowed()
        if cond_stage is None and text_encoder == "clip":
            # `[gan] text_encoder = clip`: FrozenCLIPEmbedder on the engine (768-wide contexts: the SD U-Net)
            from .text_encoders import FrozenCLIPEmbedderHIP
            assert udesc.context_dim == 768, "the CLIP ViT-L/14 text encoder conditions the SD-v1 U-Net"
            cond_stage = FrozenCLIPEmbedderHIP(self.engine, state_dict=ckpt_sd,  # cond_stage_model.transformer.*
                                               require_vocab=strict_tok)
        elif cond_stage is None and text_encoder == "bert":
            # `[gan] text_encoder = bert`: BERTEmbedder of LDM text2img-large (1280-wide contexts)
            from .text_encoders import BERTEmbedderHIP
            assert udesc.context_dim == 1280, "the BERT / x-transformer encoder conditions the LDM text2img U-Net"
            cond_stage = BERTEmbedderHIP(self.engine, state_dict=ckpt_sd, require_vocab=strict_tok)
        if cond_stage is None:
            if not synthetic_allowed():
                raise RuntimeError("no text encoder: pass cond_stage / text_encoder, or load a checkpoint that has one")
            cond_stage = StandInTextEmbedder(udesc.context_dim)
        self.cond_stage = cond_stage
        n_candidates = (n_trials or 1) * len(skip_steps or [0]) * len(encoder_unconditional_guidance_scales or [1]) * \
            len(decoder_unconditional_guidance_scales or [1])
        if ranker is None and n_candidates > 1:
            # the reference builds its DirectionalCLIP unconditionally (sd_wrapper:140) and uses it when there is more
            # than one candidate (:213-235): an ensemble config needs no extra key here either
            ranker = "directional_clip"
        if ranker == "directional_clip":  # `[gan] ranker = directional_clip`: the reference's DirectionalCLIP on the engine
            from .ranker import DirectionalCLIPHIP
            rpath = ranker_path or os.environ.get("CYCLEDIFF_CLIP_RANKER")
            rsd = None
            if rpath:
                rsd = torch.load(rpath, map_location="cpu")
                rsd = rsd.get("state_dict", rsd) if isinstance(rsd, dict) else rsd.state_dict()
            elif not synthetic_allowed():
                raise FileNotFoundError("ranker = directional_clip needs the CLIP ViT-B/32 state_dict: set `ranker_path` "
                                        "or CYCLEDIFF_CLIP_RANKER (or CYCLEDIFF_SYNTHETIC_WEIGHTS=1 for random towers)")
            ranker = DirectionalCLIPHIP(self.engine, state_dict=rsd, require_vocab=rsd is not None)
        self.ranker = ranker
        self.alphas_cumprod = schedule.latent_alphas_cumprod(1000, self.LINEAR_START, self.LINEAR_END)
        # `next(self.parameters()).device` must work (sd_wrapper:251-253) and DDP must be able to wrap us
        self._anchor = torch.nn.Parameter(torch.zeros(1, device=self.engine.device), requires_grad=True)

    # ---- conditioning (sd_wrapper:28-36)
    def get_condition(self, text, bs):
        assert isinstance(text, list) and isinstance(text[0], str)
        uc = self.cond_stage(bs * [""]).to(self.device, torch.float32)
        c = self.cond_stage(text).to(self.device, torch.float32)
        return c, uc

    def _schedule(self):
        return schedule.DDIMSchedule(self.alphas_cumprod, self.custom_steps, self.eta)

    def _vae_batch(self):
        """images per first-stage call (fp32 / split-mode activations are twice the bytes of the 16-bit ones)"""
        px = self.VAE_MAX_PIXELS if getattr(self, "precision", "fp16") == "fp16" else self.VAE_MAX_PIXELS // 2
        res = getattr(self, "resolution", None) or self.RESOLUTION or 512
        return max(1, px // (res * res))

    def _randn(self, shape, cpu=None):
        """One noise draw. `noise_source` (a callable shape -> CPU tensor; parity tests) replaces the generator: it lets
        a test hand every sample of a batch its own stream, e.g. the stream a fixture was made with."""
        if self.noise_source is not None:
            return self.noise_source(tuple(shape)).to(self.device, torch.float32)
        if self.noise_on_cpu if cpu is None else cpu:
            return torch.randn(shape).to(self.device)
        return torch.randn(shape, device=self.device)

    @staticmethod
    def _chunks(idx, per_call):
        """split a group into the fewest engine calls of at most `per_call` members, as evenly as possible (75 members
        at 32 per call -> 25 + 25 + 25 rather than 32 + 32 + 11: no small tail call, one batch shape per group)"""
        n_calls = -(-len(idx) // per_call)
        size = -(-len(idx) // n_calls)
        return [idx[i:i + size] for i in range(0, len(idx), size)]

    def _groups(self, keys):
        """indices of ensemble members grouped by key, in first-appearance order; singletons when folding is off"""
        if not self.fold_ensemble:
            return [[i] for i in range(len(keys))]
        order, groups = [], {}
        for i, k in enumerate(keys):
            if k not in groups:
                groups[k] = []
                order.append(k)
            groups[k].append(i)
        return [groups[k] for k in order]

    # ---- encode (sd_wrapper:169-206)
    def _encode_front(self, image, encode_text):
        """first stage, conditioning and the members' noise in the reference's draw order -> (x0, c, uc, members)"""
        image = (image - 0.5) * 2.0
        assert image.shape[2] == image.shape[3] == self.resolution
        image = image.to(self.device, torch.float32)
        bsz = image.shape[0]
        noise = None
        if self.SAMPLE_POSTERIOR:
            # DiagonalGaussianDistribution.sample draws on the CPU and moves (distributions.py:36)
            h = self.resolution // self.vae_factor
            noise = self._randn((bsz, self.channels, h, h), cpu=True)
        per = self._vae_batch()
        x0 = torch.cat([self.engine.vae_encode(self.vae, image[i:i + per], noise=None if noise is None else noise[i:i + per],
                                               sample=self.SAMPLE_POSTERIOR, scale=self.SCALE_FACTOR)
                        for i in range(0, bsz, per)], 0)
        sch = self._schedule()
        assert self.eta > 0
        c, uc = self.get_condition(encode_text, bsz)
        # members in the reference's order: trial -> encoder scale -> skip; noise drawn member by member in the
        # draw order of _ddpm_ddim_encoding: randn_like(x0) then K-1 x randn(shape) (ddim.py:479,599)
        members = []
        for _trial in range(self.n_trials):
            for enc_scale in self.encoder_unconditional_guidance_scales:
                for skip in self.skip_steps:
                    K = len(sch) - skip
                    # the DPM-Encoder loop breaks after white_box_steps - skip - 1 steps (ddim.py:486): the reference's
                    # configs set white_box_steps = custom_steps (the whole chain, n_loop = K); a shorter prefix leaves the
                    # rest of the decode to fresh noise (`eps=None`, ddim.py:437), -1 keeps x_T only
                    n_loop = self._white_box_loop(K, skip)
                    n_draw = K if n_loop == K else n_loop + 1  # x_T + one per executed step; index 0 draws nothing
                    if self.noise_on_cpu or self.noise_source is not None:
                        nz = torch.stack([self._randn(tuple(x0.shape)) for _ in range(n_draw)], 0)
                    else:
                        nz = self._randn((n_draw,) + tuple(x0.shape))
                    members.append((float(enc_scale), int(skip), nz))
        return x0, c, uc, members

    def encode(self, image, encode_text):
        x0, c, uc, members = self._encode_front(image, encode_text)
        bsz, sch = x0.shape[0], self._schedule()
        z_ensemble = [None] * len(members)
        per_call = max(1, self.MAX_FOLD // bsz)
        for grp in self._groups([(m[0], m[1]) for m in members]):
            enc_scale, skip = members[grp[0]][0], members[grp[0]][1]
            for idx in self._chunks(grp, per_call):
                n = len(idx)
                coef, K = sch.coef_encode(skip), len(sch) - skip
                n_loop = self._white_box_loop(K, skip)
                if n_loop < K:  # the first n_loop steps of the chain: table rows K-n_loop .. K-1 + the x_T row
                    coef = np.concatenate([coef[K - n_loop:K], coef[K:K + 1]])
                z = self.engine.dpm_encode(self.unet, _ffi.CD_SCHED_DDIM, x0.repeat(n, 1, 1, 1), coef,
                                           ctx_c=c.repeat(n, 1, 1), ctx_uc=uc.repeat(n, 1, 1), guidance=enc_scale,
                                           noise=torch.cat([members[i][2] for i in idx], dim=1), last_uses_x0=n_loop == K)
                for j, i in enumerate(idx):
                    z_ensemble[i] = z[j * bsz:(j + 1) * bsz].reshape(bsz, -1)
        return z_ensemble

    def _white_box_loop(self, K, skip):
        """DPM-Encoder steps that run for a chain of K steps (ddim.py:486 `if i < white_box_steps - skip_steps - 1`)."""
        if self.white_box_steps == -1:
            return 0
        return max(0, min(K, self.white_box_steps - skip - 1))

    # ---- generate (sd_wrapper:142-167)
    def generate(self, z_ensemble, decode_text, latents=None):
        """`latents` (translate()): {output slot: latent [bsz, C, h, w]} of candidates the coupled loop already decoded."""
        sch = self._schedule()
        n_dec = len(self.decoder_unconditional_guidance_scales)
        bsz = z_ensemble[0].shape[0]
        c, uc = self.get_condition(decode_text, bsz)
        latents = dict(latents or {})
        jobs = []  # (output slot, skip, decoder scale, z) in the reference's order: z member -> decoder scale
        for i, z in enumerate(z_ensemble):
            skip = self.skip_steps[i % len(self.skip_steps)]
            slots = self.white_box_steps - skip if self.white_box_steps != -1 else 1  # sd_wrapper:149-152
            zz = z.view(bsz, slots, self.channels, self.image_size, self.image_size)
            # decode steps beyond the white-box prefix draw fresh noise (ddim.py:437 `eps=None`): one tensor per step and
            # candidate, drawn here in candidate order so that folding candidates into one call changes nothing
            n_tail = (len(sch) - skip) - (slots - 1)
            for j, dec_scale in enumerate(self.decoder_unconditional_guidance_scales):
                tail = None
                if n_tail > 0:
                    shape = (bsz, self.channels, self.image_size, self.image_size)
                    if self.noise_on_cpu or self.noise_source is not None:
                        tail = torch.stack([self._randn(shape) for _ in range(n_tail)], 0)
                    else:
                        tail = self._randn((n_tail,) + shape)
                jobs.append((i * n_dec + j, int(skip), float(dec_scale), zz, tail))
        n_jobs = len(jobs)
        per_call = max(1, self.MAX_FOLD // bsz)
        todo = [jb for jb in jobs if jb[0] not in latents]

        # jobs that share the skip and the batch structure fold into one engine call; inside a classifier-free-guidance
        # group every sample carries its own scale (cd_ddim_decode_v), so the 5 guided scales of the reference's config
        # fill the calls instead of running 15 members at a time
        for grp in self._groups([(jb[1], self._kind(jb[2]), jb[2] if self._kind(jb[2]) != "cfg" else None) for jb in todo]):
            skip = todo[grp[0]][1]
            for idx in self._chunks(grp, per_call):
                n = len(idx)
                scales = [todo[i][2] for i in idx]
                if self._kind(scales[0]) == "cfg" and len(set(scales)) > 1:
                    guidance = torch.tensor([sc for sc in scales for _ in range(bsz)], dtype=torch.float32)
                else:
                    guidance = scales[0]
                x = self.engine.ddim_decode(self.unet, _ffi.CD_SCHED_DDIM,
                                            torch.cat([todo[i][3] for i in idx], dim=0).contiguous(),
                                            sch.coef_decode(skip), ctx_c=c.repeat(n, 1, 1), ctx_uc=uc.repeat(n, 1, 1),
                                            guidance=guidance,
                                            noise_tail=None if todo[idx[0]][4] is None else
                                            torch.cat([todo[i][4] for i in idx], dim=1).contiguous())
                for j, i in enumerate(idx):
                    latents[todo[i][0]] = x[j * bsz:(j + 1) * bsz]
        # decode_first_stage, then post_process (x+1)/2 fused into the final layout kernel; candidates in output order, in calls
        # of at most _vae_batch() images
        per = self._vae_batch()
        lat = torch.cat([latents[k] for k in range(n_jobs)], 0)
        img = torch.cat([self.engine.vae_decode(self.vae, lat[i:i + per].contiguous(), scale=self.SCALE_FACTOR,
                                                out_mul=0.5, out_add=0.5) for i in range(0, lat.shape[0], per)], 0)
        return [img[k * bsz:(k + 1) * bsz] for k in range(n_jobs)]

    @staticmethod
    def _kind(scale):  # which network batch a scale needs (ddim.py:550-559)
        return "cond" if scale == 1.0 else ("uncond" if scale == 0.0 else "cfg")

    # ---- encode + generate as ONE coupled loop (north_star; include/cyclediff.h cd_cycle_translate)
    def translate(self, image, encode_text, decode_text):
        """What Model.forward composes (model/text_unsupervised_translation.py:24-40): `self(encode(image, encode_text), image,
        encode_text, decode_text)`, with the DPM-Encoder and the decode of every ensemble member running as one loop - step k of
        both evaluates the same U-Net at the same timestep, and the decode step needs eps_k only after its forward, so each
        step is ONE forward over [encoder rows | decoder rows] (C2: 12 rows per step for a batch of 4 instead of 4, then 8).
        Same draws in the same order, same member order, same per-sample arithmetic as the two calls. Chains that leave part
        of the decode to fresh noise (white_box_steps shorter than the chain) take the two calls."""
        sch = self._schedule()
        whole = all(self._white_box_loop(len(sch) - sk, sk) == len(sch) - sk for sk in self.skip_steps)
        dec_scales = [float(sc) for sc in self.decoder_unconditional_guidance_scales]
        n_dec = len(dec_scales)
        kinds = [self._kind(sc) for sc in dec_scales]
        main = "cfg" if "cfg" in kinds else kinds[0]  # the decoder scales that ride with the encoder (the others decode from z)
        ride = [j for j in range(n_dec) if kinds[j] == main and (main == "cfg" or dec_scales[j] == dec_scales[kinds.index(main)])]
        bsz = image.shape[0]
        enc_cfg = any(self._kind(float(sc)) == "cfg" for sc in self.encoder_unconditional_guidance_scales)
        rows = bsz * ((2 if enc_cfg else 1) + len(ride) * (2 if main == "cfg" else 1))  # of one member's coupled forward
        self.last_translate_coupled = bool(self.couple and whole and rows * self.image_size ** 2 <= self.couple_max_tokens)
        if not self.last_translate_coupled:
            z_ensemble = self.encode(image, encode_text)
            return self.forward(z_ensemble, image, encode_text, decode_text)
        x0, c_src, uc, members = self._encode_front(image, encode_text)
        c_tgt, _ = self.get_condition(decode_text, bsz)
        z_ensemble, latents = [None] * len(members), {}
        per_call = max(1, min(self.MAX_FOLD // (bsz * (1 + len(ride))), self.couple_max_tokens // (rows * self.image_size ** 2)))
        for grp in self._groups([(m[0], m[1]) for m in members]):
            enc_scale, skip = members[grp[0]][0], members[grp[0]][1]
            for idx in self._chunks(grp, per_call):
                n, nr = len(idx), len(ride)
                if main == "cfg" and len({dec_scales[j] for j in ride}) > 1:
                    guidance = torch.tensor([dec_scales[j] for j in ride for _ in range(n * bsz)], dtype=torch.float32)
                else:
                    guidance = dec_scales[ride[0]]
                z, x = self.engine.cycle_translate(
                    self.unet, _ffi.CD_SCHED_DDIM, x0.repeat(n, 1, 1, 1), sch.coef_encode(skip), sch.coef_decode(skip),
                    enc_ctx_c=c_src.repeat(n, 1, 1), enc_ctx_uc=uc.repeat(n, 1, 1), enc_guidance=enc_scale,
                    dec_ctx_c=c_tgt.repeat(n * nr, 1, 1), dec_ctx_uc=uc.repeat(n * nr, 1, 1), dec_guidance=guidance, n_dec=nr,
                    noise=torch.cat([members[i][2] for i in idx], dim=1), last_uses_x0=True)
                for m, i in enumerate(idx):
                    z_ensemble[i] = z[m * bsz:(m + 1) * bsz].reshape(bsz, -1)
                    for jr, j in enumerate(ride):  # decoder row (jr * n + m) * bsz + b
                        latents[i * n_dec + j] = x[(jr * n + m) * bsz:(jr * n + m + 1) * bsz]
        img_ensemble = self.generate(z_ensemble, decode_text, latents=latents)
        return self._select(img_ensemble, image, encode_text, decode_text)

    def forward(self, z_ensemble, original_img, encode_text, decode_text):
        return self._select(self.generate(z_ensemble, decode_text), original_img, encode_text, decode_text)

    def _select(self, img_ensemble, original_img, encode_text, decode_text):
        """the candidate the reference returns (sd_wrapper:213-249): the only one, or per sample the directional-CLIP argmax"""
        assert len(img_ensemble) == len(self.decoder_unconditional_guidance_scales) * \
            len(self.encoder_unconditional_guidance_scales) * len(self.skip_steps) * self.n_trials
        if len(img_ensemble) == 1:
            return img_ensemble[0]
        if self.ranker is None:
            raise NotImplementedError(
                "ensemble of %d candidates needs a DirectionalCLIP ranker (model/energy/clean_clip.py) — out of "
                "scope for the hot path; pass ranker=callable(img, original_img, encode_text, decode_text)"
                % len(img_ensemble))
        def score(img):  # DirectionalCLIP returns (clip_score, dclip_score) and the reference ranks by the latter
            r = self.ranker(img, original_img, encode_text, decode_text)
            return r[1] if isinstance(r, (tuple, list)) else r
        if self.fold_ensemble:
            # the reference scores candidate by candidate (sd_wrapper:216-227); candidates are independent samples of the
            # ranker's batch, so they are scored MAX_FOLD images at a time (same per-sample scores, 1/32 of the calls)
            bsz, per_call = original_img.shape[0], max(1, self.MAX_FOLD // original_img.shape[0])
            parts = []
            for j0 in range(0, len(img_ensemble), per_call):
                chunk = img_ensemble[j0:j0 + per_call]
                n = len(chunk)
                if hasattr(self.ranker, "score_folded"):  # the built-in ranker encodes texts and source images once
                    sc = self.ranker.score_folded(torch.cat(chunk, dim=0), original_img, encode_text, decode_text, n)
                else:
                    sc = self.ranker(torch.cat(chunk, dim=0), original_img.repeat(n, 1, 1, 1), list(encode_text) * n,
                                     list(decode_text) * n)
                sc = sc[1] if isinstance(sc, (tuple, list)) else sc
                parts.append(sc.view(n, bsz).t())
            scores = torch.cat(parts, dim=1)
        else:
            scores = torch.stack([score(img) for img in img_ensemble], dim=1)
        best = torch.argmax(scores, dim=1)  # per-sample argmax over the ensemble (sd_wrapper:228-235)
        return torch.stack([img_ensemble[best[b].item()][b] for b in range(scores.shape[0])], dim=0)

    @property
    def device(self):
        return next(self.parameters()).device


class SDStochasticTextWrapper(_LatentStochasticTextWrapper):
    """gan_type = SDStochasticText (v1-inference.yaml; 512 px; posterior SAMPLED, ddpm.py:538)."""
    UNET_DESC = staticmethod(sd_v1_unet_desc)
    RESOLUTION = 512
    SAMPLE_POSTERIOR = True

    @staticmethod
    def checkpoint_path(source_model_type):  # sd_wrapper:24
        return os.path.join("ckpts", "stable_diffusion", source_model_type)


class LatentDiffStochasticTextWrapper(_LatentStochasticTextWrapper):
    """gan_type = LatentDiffStochasticText (txt2img-1p4B-eval.yaml; 256 px; posterior MEAN,
    model/lib/latentdiff/ldm/models/diffusion/ddpm.py:535-538; linear schedule 0.00085..0.012)."""
    UNET_DESC = staticmethod(ldm_text_unet_desc)
    RESOLUTION = 256
    SAMPLE_POSTERIOR = False

    @staticmethod
    def checkpoint_path(source_model_type):  # latentdiff_stochastic_text_wrapper.py:20-23
        if source_model_type != "text2img-large":
            raise ValueError(source_model_type)
        return os.path.join("ckpts", "ldm_models", source_model_type, "model.ckpt")
