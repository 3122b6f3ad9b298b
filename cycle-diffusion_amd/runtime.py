"""Per-process engine registry and weight provisioning."""
import os
import warnings

import torch

from .engine import Engine

_ENGINES = {}


def get_engine(device=None):
    """One engine (= one HIP stream + workspace) per (device, current torch stream) per process: one process
    per GPU, and normally one stream. A caller that wants several batches in flight on one GPU (bench.py) builds
    each model replica under its own `torch.cuda.stream(s)`; every replica then gets its own engine bound to s."""
    if device is None:
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(device)
    if not torch.cuda.is_available():
        from ._ffi import EngineError
        raise EngineError("no HIP device visible to PyTorch: the CycleDiffusion engine has no CPU fallback")
    key = (str(dev), int(torch.cuda.current_stream(dev).cuda_stream))
    if key not in _ENGINES:
        _ENGINES[key] = Engine(str(dev))
    return _ENGINES[key]


def read_checkpoint(ckpt_path):
    """The reference's checkpoint formats: pl_sd["state_dict"] (txt2img.py:25-42) or a plain dict
    (ddpm_ddim_wrapper.py:378-379). None when the file does not exist."""
    if not (ckpt_path and os.path.exists(ckpt_path)):
        return None
    sd = torch.load(ckpt_path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    return sd


def apply_ema_shadow(sd, model_prefix="model.", ema_prefix="model_ema."):
    """What `with model.ema_scope():` does to the weights the samplers see (ddpm.py:171-185 -> LitEma.copy_to,
    ldm/modules/ema.py:46-53): every parameter `<model_prefix><name>` is replaced by the shadow buffer
    `<ema_prefix><name with the dots removed>` (ema.py:17-21). Models whose config leaves `use_ema` at its default
    True (celeba256 / ffhq256) are evaluated on the shadow weights. A checkpoint without shadow buffers raises:
    the reference would silently run on the random-init clones LitEma took at construction."""
    shadow = {k: v for k, v in sd.items() if k.startswith(ema_prefix)}
    if not any(k not in (ema_prefix + "decay", ema_prefix + "num_updates") for k in shadow):
        raise KeyError("use_ema model, but the checkpoint holds no %s* shadow weights" % ema_prefix)
    out = dict(sd)
    n = 0
    for k in sd:
        if k.startswith(model_prefix) and not k.startswith(ema_prefix):
            s = ema_prefix + k[len(model_prefix):].replace(".", "")
            if s in shadow:
                out[k] = shadow[s]
                n += 1
    if n == 0:
        raise KeyError("no %s* parameter has a shadow buffer under %s*" % (model_prefix, ema_prefix))
    return out


def synthetic_allowed():
    return os.environ.get("CYCLEDIFF_SYNTHETIC_WEIGHTS", "0") == "1"


def load_or_init_weights(engine, ckpt_path, nets_and_prefixes, seed=0, state_dict=None):
    """Load a reference checkpoint by its state_dict names (txt2img.py:25-42: pl_sd["state_dict"];
    ddpm_ddim_wrapper.py:378-379: plain dict). A missing checkpoint is an error, as in the reference (torch.load
    raises). Seeded synthetic weights (identical on every rank) are OPT-IN: CYCLEDIFF_SYNTHETIC_WEIGHTS=1, which
    bench.py, the tests and `main.py --synthetic-weights` set - there are no checkpoints in this tree.
    Returns the origin string recorded in metrics.json."""
    sd = state_dict if state_dict is not None else read_checkpoint(ckpt_path)
    if sd is not None:
        for net, prefix in nets_and_prefixes.items():
            n, first = engine.load_state_dict(net, sd, prefix=prefix, strict=True)
            if n:
                raise KeyError("checkpoint %s lacks %d tensors, first: %s" % (ckpt_path, n, first))
        return ckpt_path
    if not synthetic_allowed():
        raise FileNotFoundError("checkpoint %s not found (set CYCLEDIFF_SYNTHETIC_WEIGHTS=1 to run on seeded synthetic "
                                "weights instead)" % ckpt_path)
    warnings.warn("checkpoint %s not found: using seeded synthetic weights (CYCLEDIFF_SYNTHETIC_WEIGHTS=1)" % ckpt_path)
    share = os.environ.get("CYCLEDIFF_SHARE_SYNTH", "0") == "1"  # bench.py: several replicas, same seeds
    for i, net in enumerate(nets_and_prefixes):
        engine.random_init(net, seed=seed + i, cache=share)
    return "synthetic(seed=%d)" % seed
