"""Build libcyclediff.so (gfx950 only) in-tree with hipcc.

Used by __graft_entry__.build() and by the tests' session fixture. The library has no torch
dependency: it links only against the HIP runtime, so it cross-compiles on a GPU-less box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libcyclediff.so")
SOURCES = ["sched.hip", "elementwise.hip", "norm.hip", "conv_gemm.hip", "lin_stream.hip", "attn.hip", "f32_path.hip", "st_f32.hip", "diag.hip", "engine.hip",
           "unet_openai.hip", "nets_ho_vae.hip", "clip_text.hip", "capi.hip"]
HEADERS = ["common.h", "kernels.h", "engine.h", os.path.join("..", "..", "include", "cyclediff.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None):
    """variant=None: the product library (fp16 storage). variant="bf16": the same sources with -DCD_ACT_FP16=0
    into lib/libcyclediff_bf16.so - only for the measurement-hygiene bench line (CYCLEDIFF_LIB=... python bench.py)."""
    objdir, lib, defines = OBJDIR, LIB, []
    if variant == "bf16":
        objdir = os.path.join(HERE, "build", "bf16")
        lib = os.path.join(LIBDIR, "libcyclediff_bf16.so")
        defines = ["-DCD_ACT_FP16=0"]
    elif variant == "probe":  # phase-timing instrumentation of k_conv_gemm (scripts/probe_report.py); never the product
        objdir = os.path.join(HERE, "build", "probe")
        lib = os.path.join(LIBDIR, "libcyclediff_probe.so")
        defines = ["-DCD_PROBE"]
    elif variant == "pack8old":  # round-6 A/B: the f2bf-based pack8 / pack2 (measurement only)
        objdir = os.path.join(HERE, "build", "pack8old")
        lib = os.path.join(LIBDIR, "libcyclediff_pack8old.so")
        defines = ["-DCD_PACK8_F2BF"]
    elif variant == "geluold":  # round-6 A/B: the select form of gelu_fast (measurement only)
        objdir = os.path.join(HERE, "build", "geluold")
        lib = os.path.join(LIBDIR, "libcyclediff_geluold.so")
        defines = ["-DCD_GELU_SELECT_FORM"]
    elif variant in ("ntst", "ntld"):  # round-6 A/B: streaming (nt) output stores / residual loads of the GEMM epilogue (measurement only)
        objdir = os.path.join(HERE, "build", variant)
        lib = os.path.join(LIBDIR, "libcyclediff_%s.so" % variant)
        defines = ["-DCD_EPI_STORE_AUX=2"] if variant == "ntst" else ["-DCD_EPI_RESID_AUX=2"]
    elif variant is not None:
        raise ValueError("unknown build variant %r" % (variant,))
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            extra = ["-ffp-contract=off"] if s == "sched.hip" else []  # scheduler math: reference op order, no FMA
            if s == "conv_gemm.hip":  # the register epilogue of the 256x320 tile is 10 fully unrolled 32x32 blocks
                extra = ["-mllvm", "-pragma-unroll-threshold=1048576"]
            jobs.append([hipcc] + FLAGS + defines + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    _variant = next((v for v in ("bf16", "probe", "pack8old", "geluold", "ntst", "ntld") if "--" + v in sys.argv), None)
    print(build(force="--force" in sys.argv, verbose=True, variant=_variant))
