"""Per (kernel, grid size) time table from a rocprofv3 --kernel-trace CSV: which SHAPES of a streaming kernel run at what
duration (the per-kernel average of kernel_breakdown.py mixes the 64 x 64 level with the 8 x 8 one).

  python scripts/kernel_by_grid.py <dir-or-csv> <kernel-name-substring> [...more substrings]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, subs = sys.argv[1], sys.argv[2:]
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"]
                if not any(s in name for s in subs):
                    continue
                key = (next(s for s in subs if s in name), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                       r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), r.get("Workgroup_Size_X", ""))
                d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                agg[key][0] += 1
                agg[key][1] += d
    for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-28s grid %-9s %-6s %-6s wg %-5s n=%6d  total %10.1f ms  mean %8.1f us" % (key + (n, us * 1e-3, us / n)))


if __name__ == "__main__":
    main()
