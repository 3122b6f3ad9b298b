"""In-path re-tuning of the short-K GEMM shapes of the SD U-Net.

The autotuner times a candidate with back-to-back launches of the same shape: operands are L2-hot, which hides the
Infinity-Cache latency the 1x1 / linear layers (K <= 1280: 5-20 K steps per tile) see in the real forward, where the A
operand has just been written by another kernel and 84 MB do not fit L2. This script measures candidates where they
run: for every candidate configuration it writes a variant of the tile table in which ALL short-K shapes use that
candidate, runs one real U-Net forward at B' = 16 and 32 with per-launch HIP events (CYCLEDIFF_GEMM_LOG=1,
scripts/bench_unet.py), and keeps, per shape, the configuration with the lowest in-situ time. Output: the merged
table (same format as cycle-diffusion_amd/tune_gfx950.txt) plus a report.

  python scripts/inpath_tune.py <out_table> <report.txt>          (on the MI355X box; ~20 s per candidate and batch)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = os.path.join(ROOT, "cycle-diffusion_amd", "tune_gfx950.txt")
BK32 = 1 << 16
# tile ids: conv_gemm.hip kCfgs. Deeper rings / smaller K steps / wider tiles than the warm tuner tends to pick.
CANDIDATES = [22 | BK32, 5 | BK32, 10 | BK32, 16 | BK32, 6 | BK32, 20, 23, 13 | BK32, 1 | BK32]
if os.environ.get("INPATH_CANDIDATES"):  # e.g. "65560,65561,26" = 24 | BK32, 25 | BK32, 26
    CANDIDATES = [int(x) for x in os.environ["INPATH_CANDIDATES"].split(",")]
BATCHES = [int(x) for x in os.environ.get("INPATH_BATCHES", "16,32").split(",")]
LINE = re.compile(r"\s+M(\d+) N(\d+) K(\d+) k(\d) s(\d)( up)?( cat)? z(\d+) act(\d) \| (.*?)\s+n=\s*(\d+)\s+([\d.]+) ms\s+([\d.]+) us/launch")


def read_table(path):
    rows = []
    for ln in open(path):
        v = [int(x) for x in ln.split()]
        if len(v) == 15:
            rows.append(v)
    return rows


MIN_M = int(os.environ.get("INPATH_MIN_M", "4096"))
MAX_K = int(os.environ.get("INPATH_MAX_K", "1280"))
ONLY_M = [int(x) for x in os.environ.get("INPATH_ONLY_M", "").split(",") if x]  # restrict to the row counts of one batch size


def is_target(k):
    M, N, K, KH, C0, C1, stride, up, act, nbatch = k[:10]
    return KH == 1 and K <= MAX_K and nbatch == 1 and M >= MIN_M and K % 32 == 0 and (not ONLY_M or M in ONLY_M)


def label_key(k):  # what the GEMM log prints for a table key
    M, N, K, KH, C0, C1, stride, up, act, nbatch = k[:10]
    return (M, N, K, KH, stride, 1 if up else 0, 1 if C1 else 0, nbatch, act)


def run_forward(table_path, B):
    env = dict(os.environ, CYCLEDIFF_TUNE_DEFAULT=table_path, CYCLEDIFF_GEMM_LOG="1", PYTHONPATH=ROOT)
    env.pop("CYCLEDIFF_TUNE_CACHE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_unet.py"), str(B), os.environ.get("INPATH_ITERS", "1"), "gemmlog"],
                       env=env, capture_output=True, text=True, timeout=600)
    out = {}
    for ln in r.stderr.splitlines():
        m = LINE.match(ln)
        if m:
            M, N, K, k, s, up, cat, z, act, cfg, n, ms, us = m.groups()
            key = (int(M), int(N), int(K), int(k), int(s), 1 if up else 0, 1 if cat else 0, int(z), int(act))
            out[key] = out.get(key, 0.0) + float(ms)  # same label under two table keys (resid / rowvec flags): summed
    if not out:
        raise RuntimeError("no GEMM log in the forward's stderr:\n" + r.stderr[-2000:])
    return out


def main():
    out_table, report = sys.argv[1], sys.argv[2]
    base = read_table(BASE)
    targets = sorted({label_key(r[:14]) for r in base if is_target(r[:14])})
    results = {}  # cand -> {label: ms}
    for cand in [None] + CANDIDATES:
        rows = [r[:14] + [cand if (cand is not None and is_target(r[:14])) else r[14]] for r in base]
        path = "/tmp/tune_variant.txt"
        with open(path, "w") as fh:
            for r in rows:
                fh.write(" ".join(str(x) for x in r) + "\n")
        merged = {}
        for B in BATCHES:
            for k, ms in run_forward(path, B).items():
                merged[k] = merged.get(k, 0.0) + ms
        results[cand] = merged
        tot = sum(ms for k, ms in merged.items() if k in targets)
        print("candidate %s: short-K shapes %.3f ms per forward set B' = %s" % (cand, tot, BATCHES), flush=True)
    best = {}
    for t in targets:
        opts = [(results[c].get(t, 1e30), c) for c in results]
        best[t] = min(opts, key=lambda x: x[0])
    with open(out_table, "w") as fh:
        for r in base:
            v = r[14]
            if is_target(r[:14]):
                c = best[label_key(r[:14])][1]
                if c is not None:
                    v = c
            fh.write(" ".join(str(x) for x in r[:14] + [v]) + "\n")
    with open(report, "w") as fh:
        base_tot = sum(results[None].get(t, 0) for t in targets)
        new_tot = sum(best[t][0] for t in targets)
        fh.write("short-K shapes (k1, K <= 1280, M >= 4096): %d; in-situ time per forward pair: %.3f ms with the warm "
                 "tuner's choices -> %.3f ms with the best in-path candidate per shape\n" % (len(targets), base_tot, new_tot))
        for t in targets:
            fh.write("%s base %.3f ms -> %.3f ms with %s | all: %s\n" % (
                t, results[None].get(t, 0), best[t][0], best[t][1],
                " ".join("%s:%.3f" % (c, results[c].get(t, 0)) for c in results)))
    print(open(report).readline())


if __name__ == "__main__":
    main()
