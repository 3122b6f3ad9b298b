"""HBM-side traffic of the k_conv_gemm family (incl. its streaming K = 320 member k_lin_stream) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.

  python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-byte requests at
64 B, so a wide coalesced streaming read reports half its bytes -> x2 correction on the fetch side; the
counter unit is KB. WRITE_SIZE is reported uncorrected (uncalibrated on gfx950). Infinity-Cache hits are
counted as memory-side traffic.
"""
import csv
import json
import sys


def total(path, counter):
    n, s = 0, 0.0
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and ("k_conv_gemm" in r["Kernel_Name"] or "k_lin_stream" in r["Kernel_Name"]):
                n += 1
                s += float(r["Counter_Value"])
    return n, s


nf, fetch_kb = total(sys.argv[1], "FETCH_SIZE")
nw, write_kb = total(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "k_conv_gemm (all tile instantiations) + k_lin_stream (tile 30 of the same family)",
    "launches_fetch_pass": nf, "launches_write_pass": nw,
    "fetch_bytes_raw": fetch_kb * 1024.0, "fetch_bytes_corrected": 2.0 * fetch_kb * 1024.0,
    "write_bytes_raw": write_kb * 1024.0,
    "bytes_per_launch": (2.0 * fetch_kb * 1024.0 / max(nf, 1)) + (write_kb * 1024.0 / max(nw, 1)),
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
