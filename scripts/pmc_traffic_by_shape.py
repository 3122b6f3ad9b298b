"""Per-shape HBM-side traffic of the GEMM family against its algorithmic bytes (round-5 verdict item 5).

  python scripts/pmc_traffic_by_shape.py <fetch_counter_collection.csv> <write_counter_collection.csv> <gemm_trace.log> <out.json>

Inputs: the two rocprofv3 --pmc passes of `scripts/bench_unet.py B 1` (FETCH_SIZE, WRITE_SIZE; one counter per pass, as the
guide's HBM section prescribes) and the stderr of one of them run with CYCLEDIFF_GEMM_TRACE=1 (one line per GEMM launch, in
launch order). The LAST forward of each pass is taken: its k_conv_gemm / k_lin_stream dispatches, in dispatch order, are the
trace's last lines one to one. Algorithmic bytes of a launch: the activation tensor it gathers from (once), its weights
(once), its output (+ the residual it reads). Corrections as scripts/pmc_traffic.py: FETCH_SIZE x 2 on gfx950, unit KB."""
import csv
import json
import sys
from collections import OrderedDict


def dispatches(path, counter):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and ("k_conv_gemm" in r["Kernel_Name"] or "k_lin_stream" in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    return [v for _, v in rows]


def main():
    fetch = dispatches(sys.argv[1], "FETCH_SIZE")
    write = dispatches(sys.argv[2], "WRITE_SIZE")
    trace = [ln.split("]", 1)[1].strip() for ln in open(sys.argv[3], errors="replace") if ln.startswith("[gemm_trace]")]
    # forwards of bench_unet.py: identical launch sequences; the period is the shortest tail that repeats itself
    per = next((q for q in range(50, len(trace) // 2 + 1) if trace[-q:] == trace[-2 * q:-q]), len(trace))
    last = trace[-per:]
    assert len(fetch) >= per and len(write) >= per, (len(fetch), len(write), per)
    f, w = fetch[-per:], write[-per:]
    shapes = OrderedDict()
    for t, fk, wk in zip(last, f, w):
        key, tile = [x.strip() for x in t.split("|")]
        M, N, K, KH, stride, up, cat, nb, act, resid, of32, chm = [int(x) for x in key.split()]
        cin = K // (KH * KH)
        rows_in = M * stride * stride // (4 if up else 1)  # input pixels the gather touches
        n_out = N // 2 if act == 3 else N
        alg_r = nb * (rows_in * cin * 2 + N * K * 2 + (M * n_out * 2 if resid else 0))
        alg_w = nb * M * n_out * (4 if of32 else 2)
        e = shapes.setdefault(t, dict(shape=dict(M=M, N=N, K=K, k=KH, stride=stride, up=up, cat=cat, nbatch=nb, act=act,
                                                 resid=resid, channel_major_k=chm), tile=tile, launches=0, fetch_bytes=0.0,
                                      write_bytes=0.0, algorithmic_read_bytes=0, algorithmic_write_bytes=0))
        e["launches"] += 1
        e["fetch_bytes"] += 2.0 * fk * 1024.0
        e["write_bytes"] += wk * 1024.0
        e["algorithmic_read_bytes"] += alg_r
        e["algorithmic_write_bytes"] += alg_w
    rows = sorted(shapes.values(), key=lambda e: -(e["fetch_bytes"] + e["write_bytes"]))
    for e in rows:
        e["read_ratio"] = e["fetch_bytes"] / max(1, e["algorithmic_read_bytes"])
        e["total_ratio"] = (e["fetch_bytes"] + e["write_bytes"]) / max(1, e["algorithmic_read_bytes"] + e["algorithmic_write_bytes"])
    tot = {k: sum(e[k] for e in rows) for k in ("launches", "fetch_bytes", "write_bytes", "algorithmic_read_bytes",
                                                "algorithmic_write_bytes")}
    out = {"what": "HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included) per GEMM shape of ONE U-Net forward "
                   "against the shape's algorithmic bytes", "launches_per_forward": per, "total": tot,
           "total_read_ratio": tot["fetch_bytes"] / tot["algorithmic_read_bytes"],
           "total_ratio": (tot["fetch_bytes"] + tot["write_bytes"]) / (tot["algorithmic_read_bytes"] + tot["algorithmic_write_bytes"]),
           "bytes_per_launch": (tot["fetch_bytes"] + tot["write_bytes"]) / per, "shapes": rows}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print("launches/forward %d  read x%.2f  total x%.2f  %.1f MB/launch" % (per, out["total_read_ratio"], out["total_ratio"],
                                                                            out["bytes_per_launch"] / 1e6))
    for e in rows[:12]:
        s = e["shape"]
        print("  M%d N%d K%d k%d%s%s n=%d  read x%.2f total x%.2f  %s" % (s["M"], s["N"], s["K"], s["k"], " up" if s["up"] else "",
              " chm" if s["channel_major_k"] else "", e["launches"], e["read_ratio"], e["total_ratio"], e["tile"]))


if __name__ == "__main__":
    main()
