"""Stage the reference's own Python modules for the hot path into oracle/_ref/ (git-ignored, NOT gpurun-ignored) so that
the GPU box - where /root/reference does not exist - can time the REFERENCE ITSELF on its host cores
(bench.py: cpu_baseline.kind = "reference").

  python -m oracle.stage_reference          (also run by __graft_entry__.build() when /root/reference is present)

What is staged is found, not listed: the modules the CPU baseline needs are imported from the reference tree where they
lie (oracle/ref_import.py) and every file that import pulled in from under /root/reference is copied, path preserved,
plus the `__init__.py` files of its packages. Nothing under oracle/_ref/ is ever committed (.gitignore) and nothing in
the product imports it. TEST / MEASUREMENT INFRASTRUCTURE ONLY.
"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DST = os.path.join(ROOT, "oracle", "_ref")
SRC = "/root/reference"


def stage(dst=DST, src=SRC, verbose=False):
    if not os.path.isdir(os.path.join(src, "model", "lib", "stable_diffusion")):
        return 0
    os.environ["CYCLEDIFF_REFERENCE"] = src
    from oracle import ref_import
    assert ref_import.REF == src, "oracle.ref_import was imported with another reference root"
    with ref_import.session():
        import ldm.models.diffusion.ddim  # noqa: F401  DDIMSampler (ddpm_ddim_encoding / sample_with_eps)
        import ldm.modules.diffusionmodules.model  # noqa: F401  Encoder / Decoder
        import ldm.modules.diffusionmodules.openaimodel  # noqa: F401  UNetModel
        import ldm.modules.distributions.distributions  # noqa: F401  DiagonalGaussianDistribution
        files = set()
        for mod in list(sys.modules.values()):
            f = (getattr(mod, "__dict__", None) or {}).get("__file__") or ""
            if isinstance(f, str) and f.startswith(src + os.sep) and f.endswith(".py"):
                files.add(f)
    for f in sorted(files):  # package markers along the way
        d = os.path.dirname(f)
        while d.startswith(src + os.sep):
            init = os.path.join(d, "__init__.py")
            if os.path.exists(init):
                files.add(init)
            d = os.path.dirname(d)
    if os.path.isdir(dst):
        shutil.rmtree(dst)
    for f in sorted(files):
        out = os.path.join(dst, os.path.relpath(f, src))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        shutil.copyfile(f, out)
        if verbose:
            print("staged", os.path.relpath(f, src))
    with open(os.path.join(dst, "STAGED_FROM"), "w") as fh:
        fh.write("%s (%d files, never committed: see oracle/stage_reference.py)\n" % (src, len(files)))
    return len(files)


if __name__ == "__main__":
    print("staged %d reference files into %s" % (stage(verbose=True), DST))
