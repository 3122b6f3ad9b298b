"""ctypes binding of libcyclediff.so (include/cyclediff.h).

PyTorch is only the host shell here: it owns device memory and the stream; every tensor crosses
the boundary as a raw device pointer. There is no CPU fallback: without the library or without a
HIP device every call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcyclediff.so")

CD_NET_UNET_OPENAI, CD_NET_UNET_HO, CD_NET_VAE_KL, CD_NET_CLIP_TEXT, CD_NET_BERT_XTR = 1, 2, 3, 4, 5
CD_NET_OCLIP_TEXT, CD_NET_OCLIP_VISION = 6, 7
CD_SCHED_DDIM, CD_SCHED_DDPM = 0, 1
CD_PREC_16, CD_PREC_F32, CD_PREC_F32X3 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_GELU, ACT_GEGLU = 0, 1, 2, 3


class NetDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("image_size", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
        ("model_channels", C.c_int), ("num_res_blocks", C.c_int),
        ("n_mult", C.c_int), ("channel_mult", C.c_int * 8),
        ("n_attn", C.c_int), ("attn", C.c_int * 8),
        ("num_heads", C.c_int), ("num_head_channels", C.c_int),
        ("use_spatial_transformer", C.c_int), ("context_dim", C.c_int), ("transformer_depth", C.c_int),
        ("use_scale_shift_norm", C.c_int), ("resblock_updown", C.c_int), ("conv_resample", C.c_int),
        ("z_channels", C.c_int), ("embed_dim", C.c_int), ("double_z", C.c_int),
        ("precision", C.c_int), ("n_embed", C.c_int), ("reserved", C.c_int * 6),
    ]


class StepCoef(C.Structure):
    _fields_ = [("sa", C.c_float), ("s1a", C.c_float), ("sap", C.c_float), ("dirc", C.c_float),
                ("sigma", C.c_float), ("r", C.c_float), ("t_mask", C.c_float), ("t", C.c_int32)]


STEP_COEF_DTYPE = np.dtype([("sa", "<f4"), ("s1a", "<f4"), ("sap", "<f4"), ("dirc", "<f4"), ("sigma", "<f4"),
                            ("r", "<f4"), ("t_mask", "<f4"), ("t", "<i4")])

_VP, _I, _F, _U64, _SZ, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES); this table is also what
# tests/test_abi.py checks against the header
SIGNATURES = {
    "cd_last_error": [],
    "cd_version": [],
    "cd_act_format": [],
    "cd_engine_create": [_VP, _SZ, C.POINTER(_VP)],
    "cd_engine_destroy": [_VP],
    "cd_engine_synchronize": [_VP],
    "cd_engine_workspace_high_water": [_VP, C.POINTER(_SZ)],
    "cd_prof_enable": [_VP, _I],
    "cd_prof_collect": [_VP, C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "cd_net_create": [_VP, C.POINTER(NetDesc), C.POINTER(_I)],
    "cd_net_param_count": [_VP, _I, C.POINTER(_I)],
    "cd_net_param_info": [_VP, _I, _I, C.c_char_p, _I, C.POINTER(_I), C.POINTER(_I64)],
    "cd_net_load_param": [_VP, _I, C.c_char_p, _VP, _I, C.POINTER(_I64)],
    "cd_net_missing_params": [_VP, _I, C.POINTER(_I), C.c_char_p, _I],
    "cd_unet_forward": [_VP, _I, _VP, _VP, _VP, _I, _I, _VP],
    "cd_text_encode": [_VP, _I, _VP, _I, _I, _VP],
    "cd_clip_text_features": [_VP, _I, _VP, _I, _I, _VP],
    "cd_clip_image_features": [_VP, _I, _VP, _I, _VP],
    "cd_vae_encode": [_VP, _I, _VP, _VP, _U64, _I, _I, _I, _F, _VP],
    "cd_vae_decode": [_VP, _I, _VP, _I, _I, _F, _F, _F, _VP],
    "cd_dpm_encode": [_VP, _I, _I, _VP, _VP, _VP, _I, _F, _I, _I, _VP, _VP, _U64, _I, _VP],
    "cd_ddim_decode": [_VP, _I, _I, _VP, _I, _I, _VP, _VP, _I, _F, _I, _I, _VP, _VP, _U64, _VP],
    "cd_ddim_decode_v": [_VP, _I, _I, _VP, _I, _I, _VP, _VP, _I, _VP, _I, _I, _VP, _VP, _U64, _VP],
    "cd_cycle_translate": [_VP, _I, _I, _VP, _VP, _VP, _F, _VP, _VP, _F, _VP, _I, _I, _I, _I, _VP, _VP, _VP, _U64, _I, _VP,
                           _VP],
    "cd_pix_refine": [_VP, _I, _I, _VP, _I, _I, _VP, _VP, _U64],
    "cd_op_pack_conv_weight": [_VP, _VP, _I, _I, _I, _I, _I, C.POINTER(_VP), C.POINTER(_I), C.POINTER(_I)],
    "cd_op_free": [_VP, _VP],
    "cd_op_conv2d": [_VP, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP],
    "cd_op_conv2d_16": [_VP, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP,
                        _VP],
    "cd_op_groupnorm": [_VP, _VP, _I, _I, _I, _I, _I, _F, _VP, _VP, _VP, _I, _VP],
    "cd_op_layernorm": [_VP, _VP, _I, _I, _VP, _VP, _F, _VP],
    "cd_op_attention": [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _F, _I, _VP],
    "cd_op_softmax_rows": [_VP, _VP, _I64, _I, _VP],
    "cd_op_timestep_embedding": [_VP, _VP, _I, _I, _I, _VP],
    "cd_op_sched_step": [_VP, _I, _I, _VP, _VP, _VP, _VP, _I, _F, _VP, _VP, _I, _I, _I, _I, _VP],
    "cd_op_bench_conv": [_VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float)],
    "cd_op_bench_mfma_sustained": [_VP, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)],
    "cd_op_probe": [_VP, _I, _VP, _VP, _SZ],
}
_RESTYPES = {"cd_last_error": C.c_char_p}

_lib = None


class EngineError(RuntimeError):
    pass


def load_library(build_if_missing=True):
    """dlopen libcyclediff.so (building it in-tree if it is missing and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise EngineError("libcyclediff.so has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        import importlib.util
        spec = importlib.util.spec_from_file_location("_cd_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    # tile / split-K choices for the reference networks' GEMM shapes, measured once on MI355X (conv_gemm.hip)
    os.environ.setdefault("CYCLEDIFF_TUNE_DEFAULT", os.path.join(_HERE, "tune_gfx950.txt"))
    # CYCLEDIFF_LIB: A/B timing of two builds of the SAME ABI on one box (scripts/); never a CPU fallback
    lib = C.CDLL(os.environ.get("CYCLEDIFF_LIB") or LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # raises AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise EngineError(load_library().cd_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) pointer of a contiguous torch tensor / None."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise EngineError("tensor crossing the C ABI must be contiguous")
    return C.c_void_p(t.data_ptr())


def coef_array(rows):
    """list of (sa, s1a, sap, dirc, sigma, r, t_mask, t) -> contiguous numpy struct array."""
    arr = np.zeros(len(rows), dtype=STEP_COEF_DTYPE)
    for i, r in enumerate(rows):
        arr[i] = tuple(r)
    return arr
