"""The shipped tile table (cycle-diffusion_amd/tune_gfx950.txt) and the tool that maintains it.

Every line is a 14-integer GEMM shape key + the chosen configuration: tile id in the low byte, split-K factor in
bits 8-15, bit 16 = the 32-deep K-step variant. The ids must exist in csrc/conv_gemm.hip's kCfgs, keys must be unique
(the engine's std::map would silently keep one), and split factors / BK = 32 flags must be ones the launcher accepts -
a bad entry only shows up on the GPU as a CD_CHECK failure in the middle of a sampler call."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "cycle-diffusion_amd", "tune_gfx950.txt")
SRC = os.path.join(ROOT, "cycle-diffusion_amd", "csrc", "conv_gemm.hip")


def _cfg_ids():
    src = open(SRC).read()
    body = src[src.index("const CfgInfo kCfgs[] = {"):src.index("constexpr int kNumCfgs")]
    ids = {int(m.group(1)): int(m.group(3)) for m in re.finditer(r"\{(\d+), (\d+), (\d+), (\d+), \"", body)}  # id -> BN
    bk64_only = {int(x) for x in re.findall(r"id == (\d+)", src[src.index("inline bool cfg_needs_bk64"):][:120])}
    return ids, bk64_only


def _rows():
    rows = []
    for ln in open(TABLE):
        v = ln.split()
        assert len(v) == 15, ln
        rows.append([int(x) for x in v])
    return rows


def test_table_entries_are_valid_configurations():
    ids, bk64_only = _cfg_ids()
    assert len(ids) >= 23 and 20 in ids and ids[20] == 320
    rows = _rows()
    assert len(rows) > 600
    keys = set()
    for r in rows:
        key, val = tuple(r[:14]), r[14]
        assert key not in keys, "duplicate shape key %s" % (key,)
        keys.add(key)
        tile, split, bk32 = val & 0xff, (val >> 8) & 0xff, (val >> 16) & 1
        M, N, K, KH, C0, C1 = key[:6]
        if tile == 30:  # the streaming K = 320 linear kernel (csrc/lin_stream.hip): only where it is defined
            assert val == 30 and K == 320 and KH == 1 and C0 == 320 and C1 == 0, (key, val)
            assert key[6] == 1 and key[7] == 0 and key[9] == 1 and key[10] == 0, (key, val)  # stride, up, nbatch, f32
            assert N % 64 == 0 and 320 <= N <= 2560 and M % 32 == 0 and key[8] in (0, 3) and not key[13] & 2, (key, val)
            continue
        assert tile in ids, (key, val)
        assert val >> 17 == 0 and 0 <= split <= 16, (key, val)
        assert K == KH * KH * (C0 + C1) or KH == 1, key
        assert C0 % 32 == 0 and C1 % 32 == 0, key
        if tile in bk64_only:  # 320-wide tiles: 64-deep K steps only, channel counts multiples of 64
            assert not bk32 and C0 % 64 == 0 and C1 % 64 == 0, (key, val)
        if bk32:
            assert split <= 1, (key, val)  # the tuner never combines the BK = 32 variants with split-K
        if key[8] == 3:  # GEGLU epilogue: value and gate halves meet in one 64-column chunk
            assert N % 64 == 0, key


def test_merge_tune_overrides_and_keeps_order(tmp_path):
    a, b, out = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "o.txt"
    k1 = "1 2 3 1 32 0 1 0 0 1 0 8 8 0"
    k2 = "4 5 6 1 64 0 1 0 0 1 0 8 8 1"
    k3 = "7 8 9 3 64 0 1 0 0 1 0 8 8 0"
    a.write_text("%s 2\n%s 20\n" % (k1, k2))
    b.write_text("%s 276\nnot a table line\n%s 5\n" % (k2, k3))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "merge_tune.py"), str(a), str(b), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_text().splitlines() == [k1 + " 2", k2 + " 276", k3 + " 5"]
