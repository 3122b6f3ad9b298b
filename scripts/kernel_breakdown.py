"""Per-kernel time table from a rocprofv3 --kernel-trace CSV (kernel_trace.csv).

  python scripts/kernel_breakdown.py <dir-or-csv> [skip_first_n_dispatches | @kernel_name]

With `@name` the table covers one period of the trace: from the second-to-last to the last dispatch of
kernel `name` (e.g. @k_timestep_embedding = exactly one U-Net forward).

Prints one line per kernel (template arguments kept, argument lists and the cd::(anonymous namespace)::
prefix removed): launches, total ms, mean us, share.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = name.replace("cd::(anonymous namespace)::", "").replace("cd::gemm_detail::", "").replace("cd::", "")
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:  # cut at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    src = sys.argv[1]
    arg = sys.argv[2] if len(sys.argv) > 2 else "0"
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if arg.startswith("@"):
        hits = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == arg[1:]]
        if len(hits) < 2:
            raise SystemExit("need two dispatches of %s" % arg[1:])
        t0, t1 = int(rows[hits[-2]]["Start_Timestamp"]), int(rows[hits[-1]]["Start_Timestamp"])
        rows = rows[hits[-2]:hits[-1]]
        print("period %.3f ms wall" % ((t1 - t0) * 1e-6))
    else:
        rows = rows[int(arg):]
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in rows:
        k = short(r["Kernel_Name"])
        tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        cnt[k] += 1
    total = sum(tot.values())
    print("kernels %d  sum %.3f ms" % (len(rows), total))
    for k in sorted(tot, key=lambda k: -tot[k]):
        print("%-64s n=%5d %9.3f ms %8.1f us %5.1f%%" % (k[:64], cnt[k], tot[k], tot[k] / cnt[k] * 1e3, 100 * tot[k] / total))


if __name__ == "__main__":
    main()
