"""Per-shape comparison of two CYCLEDIFF_GEMM_LOG tables (scripts/bench_unet.py ... gemmlog): python scripts/gemmlog_diff.py a.txt b.txt"""
import re
import sys


def load(path):
    out = {}
    for ln in open(path):
        m = re.match(r"\s+(M\d+ .*?) \| (.*?)\s+n=\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) us/launch\s+([\d.]+) TF/s", ln)
        if m:
            out[m.group(1)] = (m.group(2).strip(), int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)))
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
rows = []
for k in a:
    if k in b:
        rows.append((a[k][2] - b[k][2], k))
        ta += a[k][2]; tb += b[k][2]
for d, k in sorted(rows, reverse=True):
    x, y = a[k], b[k]
    print("%-42s n=%3d  %-22s %7.1f us %6.0f TF | %-22s %7.1f us %6.0f TF | %+6.3f ms" % (k[:42], x[1], x[0][:22], x[3], x[4], y[0][:22], y[3], y[4], -d))
print("total %.3f -> %.3f ms" % (ta, tb))
