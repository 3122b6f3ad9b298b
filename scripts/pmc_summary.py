"""Sum rocprofv3 --pmc counter_collection.csv per kernel: python scripts/pmc_summary.py <csv> [name-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if sub in k:
        print(k)
        for c, v in agg[k].items():
            print("   %-28s %16.0f  per-dispatch %14.0f (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
