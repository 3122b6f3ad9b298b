"""Host-side logic that needs no GPU: config parsing (utils/config_utils.py contract), the [gan] -> kwargs
remapping of get_gan_wrapper (model/gan_wrapper/get_gan_wrapper.py:3-30), ensemble grouping, the stand-in
tokenizers' framing, and that the product path refuses to run without a HIP device."""
import os
import types

import pytest
import torch

import cycle_diffusion_amd  # noqa: F401
from cycle_diffusion_amd.gan_wrapper import get_gan_wrapper as ggw
from cycle_diffusion_amd.gan_wrapper.latent_text_wrapper import _LatentStochasticTextWrapper
from cycle_diffusion_amd.gan_wrapper.text_encoders import BOS, EOS, BertHashTokenizer, HashTokenizer
from cycle_diffusion_amd.utils.config_utils import get_config, parse_string

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_string_casts_like_the_reference():
    assert parse_string("99") == 99 and isinstance(parse_string("99"), int)
    assert parse_string("0.1") == 0.1
    assert parse_string("True") is True and parse_string("false") is False and parse_string("None") is None
    assert parse_string("[0, 10]") == [0, 10]
    assert parse_string('"sd-v1-4.ckpt"') == "sd-v1-4.ckpt" and parse_string("sd-v1-4.ckpt") == "sd-v1-4.ckpt"


def test_bench_config_is_the_c2_gan_section():
    args = get_config("experiments/bench_sd_c2.cfg", config_root=os.path.join(ROOT, "config"))
    assert args.model.name == "text_unsupervised_translation"
    gan = dict(iter(args.gan))
    assert gan["gan_type"] == "SDStochasticText" and gan["custom_steps"] == 99 and gan["white_box_steps"] == 100
    assert gan["eta"] == 0.1 and gan["skip_steps"] == [0] and gan["n_trials"] == 1
    assert gan["encoder_unconditional_guidance_scales"] == [1] and gan["decoder_unconditional_guidance_scales"] == [3]
    assert args.gan.not_a_key is None  # unknown attributes read as None
    assert list(k for k, _ in args.gan) == sorted(gan)  # iteration order: sorted keys


def test_reference_ensemble_config_runs_as_written():
    """the reference's SD experiment (config/experiments/translate_text2img256_stable_diffusion_stochastic_1.cfg): 15
    trials x 6 skips x 6 decoder scales = 540 candidates per image; bench.py's c2e workload prices exactly that"""
    args = get_config("experiments/translate_text2img256_stable_diffusion_stochastic_1.cfg",
                      config_root=os.path.join(ROOT, "config"))
    gan = dict(iter(args.gan))
    assert gan["gan_type"] == "SDStochasticText" and gan["n_trials"] == 15 and gan["eta"] == 0.1
    assert gan["skip_steps"] == [15, 20, 25, 30, 40, 50]
    assert gan["decoder_unconditional_guidance_scales"] == [1, 1.5, 2, 3, 4, 5]
    assert gan["encoder_unconditional_guidance_scales"] == [1]
    n = gan["n_trials"] * len(gan["skip_steps"]) * len(gan["decoder_unconditional_guidance_scales"])
    assert n == 540
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    wl = bench.WORKLOADS["c2e"]
    steps = sum(gan["custom_steps"] - s for s in gan["skip_steps"])  # 414 sampler steps per trial and scale
    fwd = 15 * steps * (1 + 1 + 2 * 5)  # encode at scale 1, decode at scale 1 (one forward) and at 5 CFG scales (two)
    assert wl["flop_per_image"] == bench.F_VAE_ENC + fwd * bench.F_UNET + 540 * bench.F_VAE_DEC


def test_get_gan_wrapper_kwarg_remapping(monkeypatch):
    seen = {}

    class Fake:
        def __init__(self, **kw):
            seen.update(kw)

    fake_mod = types.SimpleNamespace(DDPMDDIMWrapper=Fake)
    monkeypatch.setitem(__import__("sys").modules, "cycle_diffusion_amd.gan_wrapper.ddpm_ddim_wrapper", fake_mod)

    class A:  # minimal stand-in for the parsed [gan] section
        gan_type = "DDPM_DDIM"
        items = [("custom_steps", 50), ("gan_type", "DDPM_DDIM"), ("sample_type", "ddim"),
                 ("source_model_type", "cat"), ("target_model_type", "dog")]

        def __iter__(self):
            return iter(self.items)

    ggw.get_gan_wrapper(A(), target=False)
    assert seen == {"custom_steps": 50, "sample_type": "ddim", "source_model_type": "cat"}
    seen.clear()
    ggw.get_gan_wrapper(A(), target=True)  # target_* keys are renamed to source_*; source_* dropped
    assert seen == {"custom_steps": 50, "sample_type": "ddim", "source_model_type": "dog"}


def test_ensemble_grouping_preserves_first_appearance_order():
    w = types.SimpleNamespace(fold_ensemble=True)
    keys = [(1.0, 0), (1.0, 4), (1.0, 0), (3.0, 0), (1.0, 4)]
    assert _LatentStochasticTextWrapper._groups(w, keys) == [[0, 2], [1, 4], [3]]
    w.fold_ensemble = False
    assert _LatentStochasticTextWrapper._groups(w, keys) == [[0], [1], [2], [3], [4]]


def test_ensemble_groups_are_cut_into_even_engine_calls():
    cut = _LatentStochasticTextWrapper._chunks
    assert [len(c) for c in cut(list(range(75)), 32)] == [25, 25, 25]  # 5 guided scales x 15 trials of one skip
    assert [len(c) for c in cut(list(range(15)), 32)] == [15]
    assert [len(c) for c in cut(list(range(33)), 32)] == [17, 16]
    assert sum(cut(list(range(75)), 32), []) == list(range(75))  # order preserved


def test_generate_folds_the_reference_ensemble_into_the_expected_engine_calls():
    """generate() on the reference's SD config (15 trials x 6 skips, 6 decoder scales) with a recording stand-in for the
    engine: per skip one conditional-only call of 15 members (scale 1: ddim.py:550) and three guided calls of 25 members
    carrying one scale per sample; every candidate lands in the slot of the reference's loop order (member -> scale)."""
    from cycle_diffusion_amd import schedule
    skips, scales, trials = [15, 20, 25, 30, 40, 50], [1, 1.5, 2, 3, 4, 5], 15
    w = object.__new__(_LatentStochasticTextWrapper)
    torch.nn.Module.__init__(w)
    w.skip_steps, w.decoder_unconditional_guidance_scales, w.n_trials = skips, scales, trials
    w.white_box_steps, w.custom_steps, w.eta, w.fold_ensemble = 100, 99, 0.1, True
    w.channels, w.image_size = 4, 2
    w.alphas_cumprod = schedule.latent_alphas_cumprod(1000, 0.00085, 0.0120)
    w.cond_stage = lambda texts: torch.zeros(len(texts), 77, 8)
    w.unet = w.vae = 0
    w._anchor = torch.nn.Parameter(torch.zeros(1))
    calls = []

    class Eng:
        def ddim_decode(self, net, kind, z, coef, ctx_c=None, ctx_uc=None, guidance=1.0, **kw):
            calls.append((z.shape[0], len(coef), guidance))
            return z[:, 0].clone()  # each member's x_T carries its tag

        def vae_decode(self, net, x, **kw):
            return x

    w.engine = Eng()
    z_ens = []
    for i in range(trials * len(skips)):  # member i: every element = i
        K = 100 - skips[i % len(skips)]
        z_ens.append(torch.full((1, K * 4 * 2 * 2), float(i)))
    imgs = w.generate(z_ens, ["target"])
    assert len(imgs) == 540
    for i in range(90):
        for j in range(6):
            assert float(imgs[i * 6 + j].flatten()[0]) == float(i)  # slot = member * n_scales + scale index
    assert len(calls) == 6 * 4
    for skip in skips:
        mine = [c for c in calls if c[1] == 99 - skip]
        assert sorted(c[0] for c in mine) == [15, 25, 25, 25]
        cond = [c for c in mine if c[0] == 15][0]
        assert cond[2] == 1.0
        guided = sorted(sum((c[2].tolist() for c in mine if c[0] == 25), []))
        assert guided == sorted([float(s) for s in scales[1:]] * trials)


def test_stand_in_tokenizers_keep_the_reference_framing():
    ids = HashTokenizer()(["a photo of a cat", "", "A Photo"])
    assert ids.shape == (3, 77) and ids.dtype == torch.int32
    assert ids[0, 0] == BOS and ids[0, 6] == EOS and (ids[0, 6:] == EOS).all()
    assert ids[1, 0] == BOS and ids[1, 1] == EOS
    assert ids[0, 1] == ids[2, 1] and ids[0, 2] == ids[2, 2]  # case-insensitive, one stable id per word
    assert ((ids > 0) & (ids < 49408)).all()
    b = BertHashTokenizer()(["hello world", ""])
    assert b[0, 0] == 101 and b[0, 3] == 102 and (b[0, 4:] == 0).all() and b[1, 1] == 102
    long = HashTokenizer()([" ".join(["w%d" % i for i in range(200)])])
    assert long[0, 76] == EOS and long[0, 75] != EOS  # truncated to 75 words + BOS/EOS


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_wrappers_refuse_to_run_without_a_gpu():
    from cycle_diffusion_amd import _ffi
    from cycle_diffusion_amd.gan_wrapper.ddpm_ddim_wrapper import DDPMDDIMWrapper
    with pytest.raises(_ffi.EngineError):
        DDPMDDIMWrapper(source_model_type="toy32", sample_type="ddpm", custom_steps=4, es_steps=4)


def test_ema_shadow_replaces_the_raw_unet_weights():
    """celeba256 / ffhq256 leave use_ema at True: the reference samples inside model.ema_scope(), i.e. on the
    shadow buffers `model_ema.<name without dots>` (ldm/modules/ema.py:17-21,46-53)."""
    from cycle_diffusion_amd.runtime import apply_ema_shadow
    raw = {"model.diffusion_model.input_blocks.0.0.weight": torch.zeros(2), "model.diffusion_model.out.2.bias": torch.zeros(3),
           "first_stage_model.encoder.conv_in.weight": torch.full((1,), 7.0)}
    sd = dict(raw)
    sd["model_ema.diffusion_modelinput_blocks00weight"] = torch.ones(2)
    sd["model_ema.diffusion_modelout2bias"] = torch.full((3,), 2.0)
    sd["model_ema.decay"], sd["model_ema.num_updates"] = torch.tensor(0.9999), torch.tensor(5)
    out = apply_ema_shadow(sd)
    assert torch.equal(out["model.diffusion_model.input_blocks.0.0.weight"], torch.ones(2))
    assert torch.equal(out["model.diffusion_model.out.2.bias"], torch.full((3,), 2.0))
    assert torch.equal(out["first_stage_model.encoder.conv_in.weight"], torch.full((1,), 7.0))  # first stage untouched
    assert torch.equal(sd["model.diffusion_model.out.2.bias"], torch.zeros(3))  # the input dict is not modified
    with pytest.raises(KeyError):
        apply_ema_shadow(raw)  # no shadow weights: the reference would sample from random-init clones
    with pytest.raises(KeyError):
        apply_ema_shadow(dict(raw, **{"model_ema.decay": torch.tensor(0.9)}))


def test_16_bit_ddim_is_refused_by_name():
    """a 16-bit network inside the pixel 'ddim' chain reproduces the reference to ~15 dB only; the wrapper
    refuses the combination instead of accepting it silently (the check precedes engine creation: no GPU needed)."""
    from cycle_diffusion_amd.gan_wrapper.ddpm_ddim_wrapper import DDPMDDIMWrapper
    with pytest.raises(ValueError, match="allow_lossy_ddim"):
        DDPMDDIMWrapper(source_model_type="toy32", sample_type="ddim", eta=0.1, custom_steps=4, es_steps=4,
                        precision="fp16")
    with pytest.raises(ValueError, match="precision must be"):
        DDPMDDIMWrapper(source_model_type="toy32", sample_type="ddim", eta=0.1, custom_steps=4, es_steps=4,
                        precision="fp8")


def test_unconditional_ldm_wrapper_refuses_the_lossy_16bit_engine_by_default():
    """The celeba256 / ffhq256 configs sample with eta 0.1 over up to 999 steps: the 16-bit engine ends a 99-step chain
    at 25 dB against the reference's latent (73-76 dB on the fp32 path / split mode), so - like DDPMDDIMWrapper's
    16-bit 'ddim' - it must be asked for by name and acknowledged; the default is the split mode. Checked before any
    engine is created (no GPU needed)."""
    import inspect
    from cycle_diffusion_amd.gan_wrapper.latent_wrapper import LatentDiffStochasticWrapper
    # default None = 'fp32x3' on the fp16 build of the library, 'fp32' on the bf16 build (which has no split mode)
    assert inspect.signature(LatentDiffStochasticWrapper.__init__).parameters["precision"].default is None
    from cycle_diffusion_amd import _ffi
    from cycle_diffusion_amd.engine import ldm_uncond_unet_desc
    d16 = ldm_uncond_unet_desc()
    d16.precision = _ffi.CD_PREC_16
    with pytest.raises(ValueError, match="allow_lossy_16bit"):  # an explicit 16-bit descriptor is the same lossy engine
        LatentDiffStochasticWrapper("celeba256", custom_steps=99, eta=0.1, white_box_steps=100, unet_desc=d16)
    with pytest.raises(ValueError, match="allow_lossy_16bit"):
        LatentDiffStochasticWrapper("celeba256", custom_steps=99, eta=0.1, white_box_steps=100, precision="fp16")
    with pytest.raises(ValueError, match="precision must be one of"):
        LatentDiffStochasticWrapper("celeba256", custom_steps=99, eta=0.1, white_box_steps=100, precision="tf32")


def test_ranker_scores_a_folded_ensemble_with_texts_and_sources_encoded_once():
    """DirectionalCLIPHIP.score_folded (ranker.py) = __call__ on the repeated inputs (what a user-supplied ranker gets from
    a folded ensemble call, latent_text_wrapper.forward), with one text / source-image encoder pass instead of n
    (reference scoring: model/energy/clean_clip.py:24-31, per sample). Towers replaced by a deterministic stand-in."""
    import torch
    from cycle_diffusion_amd.gan_wrapper import ranker as R

    class FakeEngine:
        device = torch.device("cpu")

        def __init__(self):
            self.calls = {"text": 0, "image": 0}
            g = torch.Generator().manual_seed(3)
            self.wt, self.wi = torch.randn(77, 16, generator=g), torch.randn(3 * 224 * 224, 16, generator=g) / 100

        def clip_text_features(self, net, ids):
            self.calls["text"] += ids.shape[0]
            return (torch.sin(ids.double() * 0.37) @ self.wt.double()).float()

        def clip_image_features(self, net, x):
            self.calls["image"] += x.shape[0]
            return (x.reshape(x.shape[0], -1).double() @ self.wi.double()).float()  # (fp64: batch-size independent sums)

    rk = R.DirectionalCLIPHIP.__new__(R.DirectionalCLIPHIP)
    rk.engine, rk.text, rk.vision = FakeEngine(), None, None
    rk.tokenize = R._ClipTokenize()
    g = torch.Generator().manual_seed(0)
    bsz, n = 2, 3
    orig = torch.rand(bsz, 3, 64, 64, generator=g)
    cands = torch.rand(n * bsz, 3, 64, 64, generator=g)
    enc, dec = ["a photo of a cat", "a dog"], ["a photo of a tiger", "a wolf"]
    ref = rk(cands, orig.repeat(n, 1, 1, 1), enc * n, dec * n)
    per_call = dict(rk.engine.calls)
    rk.engine.calls = {"text": 0, "image": 0}
    got = rk.score_folded(cands, orig, enc, dec, n)
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    assert per_call == {"text": 2 * n * bsz, "image": 2 * n * bsz}
    assert rk.engine.calls == {"text": 2 * bsz, "image": (n + 1) * bsz}


@pytest.mark.parametrize("wb", [100, 40, -1])
def test_white_box_prefix_reaches_the_engine_as_a_truncated_table(wb):
    """encode() / generate() with `white_box_steps` at, below and without the chain (ddim.py:486; sd_wrapper:149-152) on a
    recording stand-in for the engine: the DPM-Encoder call gets the first n steps of the chain (table rows K-n .. K-1 by
    their timesteps + the x_T row, no `index 0 returns x0` special case, n + 1 draws), z has n + 1 slots, and the decode
    call gets the K - n fresh-noise tensors the steps beyond the prefix consume."""
    from cycle_diffusion_amd import schedule
    skip, steps = 20, 99
    w = object.__new__(_LatentStochasticTextWrapper)
    torch.nn.Module.__init__(w)
    w.skip_steps, w.encoder_unconditional_guidance_scales, w.decoder_unconditional_guidance_scales = [skip], [1.0], [3.0]
    w.n_trials, w.white_box_steps, w.custom_steps, w.eta, w.fold_ensemble = 1, wb, steps, 0.1, True
    w.channels, w.image_size, w.resolution, w.vae_factor = 4, 2, 16, 8
    w.noise_on_cpu, w.noise_source = True, None
    w.alphas_cumprod = schedule.latent_alphas_cumprod(1000, 0.00085, 0.0120)
    w.cond_stage = lambda texts: torch.zeros(len(texts), 77, 8)
    w.unet = w.vae = 0
    w._anchor = torch.nn.Parameter(torch.zeros(1))
    rec = {}

    class Eng:
        def vae_encode(self, net, image, **kw):
            return torch.zeros(image.shape[0], 4, 2, 2)

        def dpm_encode(self, net, kind, x0, coef, noise=None, last_uses_x0=True, **kw):
            rec["enc"] = (coef.copy(), noise.shape, last_uses_x0)
            return torch.zeros(x0.shape[0], len(coef), 4, 2, 2)

        def ddim_decode(self, net, kind, z, coef, noise_tail=None, **kw):
            rec["dec"] = (z.shape, len(coef), None if noise_tail is None else tuple(noise_tail.shape))
            return z[:, 0].clone()

        def vae_decode(self, net, x, **kw):
            return x

    w.engine = Eng()
    K = steps - skip
    n = K if wb == 100 else (0 if wb == -1 else wb - skip - 1)
    z = w.encode(torch.rand(1, 3, 16, 16), ["source"])
    full = w._schedule().coef_encode(skip)
    coef, nshape, last = rec["enc"]
    assert len(coef) == n + 1 and nshape[0] == (K if n == K else n + 1) and last == (n == K)
    t_field = coef.dtype.names[-1]
    assert list(coef[t_field][:n]) == list(full[t_field][K - n:K])  # the chain's first n steps, by timestep
    assert z[0].shape == (1, (n + 1) * 4 * 2 * 2)
    w.generate(z, ["target"])
    zshape, k_dec, tail = rec["dec"]
    assert zshape[1] == n + 1 and k_dec == K and tail == (None if n == K else (K - n, 1, 4, 2, 2))


def test_bench_launch_sets():
    """bench.py issues K queued steps in launch sets of `cap` steps plus one tail set (equal-sized sets were measured and are
    slower: a fold that is not a power of two leaves the 256 CUs a partial round of row tiles)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.launch_sets(20, 16) == [16, 4] and b.launch_sets(16, 16) == [16] and b.launch_sets(8, 16) == [8]
    assert b.launch_sets(33, 16) == [16, 16, 1] and b.launch_sets(5, 2) == [2, 2, 1] and b.launch_sets(0, 4) == []


def test_main_fold_look_ahead_is_invariant_to_fold_and_sharding():
    """main.py's look-ahead (`--fold N`: N dataloader batches in one model() call) with one noise stream per SAMPLE: the
    output of every sample must be BIT-identical for any fold, any batch size, and whether the batches run on one rank or
    are sharded over two (ShardSampler order) - including batch sizes that do not divide the dataset, where the sampler's
    wrap-around duplicates land in other batch positions than their originals (round-5 advisor: with per-batch streams
    n = 10, bs = 4, world 1 vs 2 gave different images for samples 2..5 and the duplicate overwrote the original). Checked
    with a stand-in model whose output is the image plus the noise its wrapper drew through the `noise_source` hook (CPU;
    the GPU twin is test_main_driver_fold_look_ahead_matches_...)."""
    import importlib.util
    import os
    import torch
    from cycle_diffusion_amd.parallel import shard_indices, shard_padding
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_main", os.path.join(root, "main.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)

    class W:
        noise_source = None

    class M:
        def __init__(self):
            self.gan_wrapper = W()

        def __call__(self, sample_id, original_image, encode_text=None, decode_text=None):
            w = self.gan_wrapper
            n1 = w.noise_source((original_image.shape[0], 3, 4, 4))  # two draws per call, as a sampler makes many
            n2 = w.noise_source((original_image.shape[0], 3, 4, 4))
            tag = torch.tensor([float(len(t)) for t in decode_text]).view(-1, 1, 1, 1)
            return (original_image, original_image + n1 + 0.5 * n2 + tag), torch.zeros(original_image.shape[0]), {}

    n_items = 10
    g = torch.Generator().manual_seed(0)
    imgs = torch.rand(n_items, 3, 4, 4, generator=g)

    def collate(idx, pad):
        return {"sample_id": torch.tensor(idx), "original_image": imgs[idx], "is_padding": list(pad),
                "encode_text": ["s%d" % i for i in idx], "decode_text": ["t" * (i + 1) for i in idx]}

    def run(fold, world, bs):
        out, n_pad = {}, 0
        for rank in range(world):
            # a generator, as main.py hands it over: the loop must pull `fold` batches at a time
            batches = (collate(ix, pd) for ix, pd in zip(shard_indices(n_items, bs, world, rank),
                                                         shard_padding(n_items, bs, world, rank)))
            for batch, orig, img in drv.folded_calls(M(), batches, fold, 42, torch.device("cpu")):
                assert torch.equal(orig, imgs[batch["sample_id"]])
                for j, sid in enumerate(batch["sample_id"].tolist()):
                    if batch["is_padding"][j]:  # main.py skips these rows; they must still REPRODUCE their original
                        n_pad += 1
                        pads.append((sid, img[j]))
                        continue
                    assert sid not in out  # every real sample is produced exactly once
                    out[sid] = img[j]
        assert n_pad == -(-n_items // (bs * world)) * bs * world - n_items
        return torch.stack([out[i] for i in range(n_items)])

    pads = []
    base = run(1, 1, 2)
    assert (base - imgs).abs().max() > 0.1  # noise really entered
    for fold, world, bs in ((2, 1, 2), (3, 1, 2), (5, 1, 2), (1, 2, 2), (2, 2, 2), (1, 1, 4), (1, 2, 4), (2, 2, 4), (3, 1, 3),
                            (1, 2, 3), (16, 1, 4), (1, 1, 7)):
        assert torch.equal(run(fold, world, bs), base), (fold, world, bs)
    assert pads and all(torch.equal(img, base[sid]) for sid, img in pads)
