"""Registers / LDS / scratch of every kernel in an object or shared library (AMDGPU metadata notes).
  python scripts/kernel_resources.py cycle-diffusion_amd/build/conv_gemm.o [name-substring]"""
import re
import subprocess
import sys

path = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else ""
# the device code object is embedded in a host object: extract the gfx950 bundle first
import os, tempfile
tmp = tempfile.mkdtemp()
out = os.path.join(tmp, "dev.co")
fat = os.path.join(tmp, "fat.bin")
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
    out = path
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
cur = {}
rows = []
for line in txt.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip().strip("'")
    if k == "agpr_count" and cur.get("name"):
        pass
    if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count",
             "vgpr_spill_count", "sgpr_spill_count", "name", "max_flat_workgroup_size"):
        if k == "agpr_count" and "vgpr_count" in cur:  # keys are sorted: .agpr_count opens a kernel's record
            rows.append(cur); cur = {}
        cur[k] = v
    if k == "vgpr_spill_count":
        pass
if cur.get("vgpr_count"):
    rows.append(cur)
for r_ in rows:
    n = subprocess.run(["c++filt", r_.get("name", "")], capture_output=True, text=True).stdout.strip()
    if only and only not in n:
        continue
    if len(sys.argv) > 3:
        print(out)
    print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s spill v%s s%s wg %s" % (
        n[:90], r_.get("vgpr_count"), r_.get("agpr_count"), r_.get("sgpr_count"), r_.get("private_segment_fixed_size"),
        r_.get("vgpr_spill_count"), r_.get("sgpr_spill_count"), r_.get("max_flat_workgroup_size")))
