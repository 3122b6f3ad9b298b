"""Registers / LDS / scratch of every kernel in an object or shared library (AMDGPU metadata notes).
  python scripts/kernel_resources.py cycle-diffusion_amd/build/conv_gemm.o [name-substring]
`kernel_rows(path)` is what tests/test_kernel_resources.py checks after a build (no scratch, no spilled VGPRs anywhere)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count",
        "vgpr_spill_count", "sgpr_spill_count", "name", "max_flat_workgroup_size")


def kernel_rows(path):
    """One dict per kernel of the gfx950 code object embedded in `path` (a host object / shared library) or of `path` itself."""
    tmp = tempfile.mkdtemp()
    out, fat = os.path.join(tmp, "dev.co"), os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat],
                   capture_output=True)
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        out = path
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", out], capture_output=True, text=True).stdout
    cur, rows = {}, []
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k in KEYS:
            if k == "agpr_count" and "vgpr_count" in cur:  # keys are sorted: .agpr_count opens a kernel's record
                rows.append(cur)
                cur = {}
            cur[k] = v
    if cur.get("vgpr_count"):
        rows.append(cur)
    for r_ in rows:
        r_["demangled"] = subprocess.run(["c++filt", r_.get("name", "")], capture_output=True, text=True).stdout.strip()
    return rows


if __name__ == "__main__":
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    for r_ in kernel_rows(sys.argv[1]):
        n = r_["demangled"]
        if only and only not in n:
            continue
        print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s spill v%s s%s wg %s" % (
            n[:90], r_.get("vgpr_count"), r_.get("agpr_count"), r_.get("sgpr_count"), r_.get("private_segment_fixed_size"),
            r_.get("vgpr_spill_count"), r_.get("sgpr_spill_count"), r_.get("max_flat_workgroup_size")))
