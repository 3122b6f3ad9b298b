"""Phase timing of k_conv_gemm from the probe build's s_memtime stamps (lib/libcyclediff_probe.so, -DCD_PROBE).

  python scripts/probe_report.py run  <outdir>      # on the GPU box: runs the shapes below, writes dumps + report.txt
  python scripts/probe_report.py show <dump.bin>    # report of one dump

Every wave records: kernel entry, prologue issued, first tile landed (first barrier passed), per K step the time it
ARRIVES at the counted-vmcnt + barrier and the time it PASSES it (sum / max of the waits, sum of the compute parts, and the
first 32 steps individually), K loop end, ring drained (vmcnt(0) + barrier), epilogue staging (accumulators -> LDS) and row
passes (LDS -> bias / residual / statistics -> 16-B stores) per column chunk, stores issued, stores landed."""
import os
import struct
import subprocess
import sys

import numpy as np

W = 48
# MFMAs (32x32x16, 32 cycles each) one wave issues per 64-deep K step; x waves per SIMD below
MFMA_PER_STEP = {"256x320 w4x2 s2": 40, "128x320 w4x2 s2": 20, "256x256 w4x2 s2": 32, "256x128 w4x2 s3": 16}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# name, B, H, C0, C1, N, k, act (| 0x100 residual, | 0x200 statistics), tiles
CASES = [
    # (..., tiles, probe-build debug bits: 1 = epilogue without global stores, 2 = without statistics, 4 = A operand of every
    # tile gathered from the first rows of the tensor, i.e. L2-resident)
    ("conv3 320>320 @64 B32 +stats", 32, 64, 320, 0, 320, 3, 0x200, [20, 23], 0),
    ("conv3 320>320 @64 B32 +stats, A L2-RESIDENT", 32, 64, 320, 0, 320, 3, 0x200, [20, 23], 4),
    ("conv3 320>320 @64 B32 plain, NO STORES", 32, 64, 320, 0, 320, 3, 0, [20], 1),
    ("conv3 320>320 @64 B32 plain, NO STORES, A L2-RESIDENT", 32, 64, 320, 0, 320, 3, 0, [20], 5),
    ("conv3 320>320 @64 B32 +resid +stats", 32, 64, 320, 0, 320, 3, 0x300, [20], 0),
    ("lin 640>640 @32 B32 +resid", 32, 32, 640, 0, 640, 1, 0x100, [20, 22], 0),
    ("lin 640>640 @32 B32 +resid, A L2-RESIDENT", 32, 32, 640, 0, 640, 1, 0x100, [20], 4),
    ("conv3 1280>1280 @16 B32 +stats", 32, 16, 1280, 0, 1280, 3, 0x200, [20, 23], 0),
    ("conv3 640>640 @32 B32 +stats", 32, 32, 640, 0, 640, 3, 0x200, [20], 0),
    ("conv3 640>640 @32 B32 +stats, A L2-RESIDENT", 32, 32, 640, 0, 640, 3, 0x200, [20], 4),
]


def load(path):
    with open(path, "rb") as f:
        head = f.readline().decode()
        raw = np.frombuffer(f.read(), dtype=np.uint64)
    raw = raw[: raw.size // W * W].reshape(-1, W)
    return head.strip(), raw


def report(path, out=sys.stdout):
    head, raw = load(path)
    ms = float(head.split("ms=")[1].split()[0])
    live = raw[raw[:, 0] != 0]
    if live.size == 0:
        print(head, "\n  no stamps (product build?)", file=out)
        return
    t0, prol, first, swait, scomp, maxw, loop, drain, stage, rows, issued, done, nst = (live[:, i].astype(np.float64) for i in range(13))
    xcc = (live[:, 13] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = live[:, 13].astype(np.int64) & 0xffffffff
    # s_memtime counters of different XCDs have different origins: time is taken relative to the XCD's first entry, and
    # the tick rate from the XCD's own span (first entry -> last store landed) against the launch duration
    # s_memtime counters are per CU (different origins): phases are differences inside one wave; the 100 MHz
    # s_memrealtime stamps (same origin everywhere) calibrate ticks -> us and place the waves on one time axis
    rt0, rt1 = live[:, 14].astype(np.float64), live[:, 15].astype(np.float64)
    rate = (done - t0) / np.maximum(rt1 - rt0, 1.0) * 100.0  # ticks per us, per wave
    tick_per_us = float(np.median(rate))
    us = lambda x: x / tick_per_us
    start = (rt0 - rt0.min()) / 100.0
    base = t0 - start * tick_per_us
    rnd = np.zeros(len(start), dtype=int)
    order = np.sort(start)
    gaps = np.diff(order)
    if len(gaps) and gaps.max() > 0.2 * ms * 1e3:
        cut = order[np.argmax(gaps)] + gaps.max() / 2
        rnd = (start > cut).astype(int)
    print(head, file=out)
    print("  waves %d  s_memtime %.0f ticks/us = shader clock %.2f GHz (p5 %.2f, p95 %.2f over waves); launch span by the 100 MHz "
          "clock %.1f us, average launch %.1f us" % (len(live), tick_per_us, tick_per_us / 1e3, np.percentile(rate, 5) / 1e3,
                                                    np.percentile(rate, 95) / 1e3, (rt1.max() - rt0.min()) / 100.0, ms * 1e3), file=out)
    n = nst.mean()

    def line(name, v, mask):
        v = us(v[mask])
        print("    %-34s mean %8.2f  min %8.2f  p50 %8.2f  max %8.2f us" % (name, v.mean(), v.min(), np.median(v), v.max()), file=out)

    for r in sorted(set(rnd.tolist())):
        m = rnd == r
        print("  round %d: %d waves, start %.1f .. %.1f us, end %.1f .. %.1f us after the first wave's entry" % (
            r, m.sum(), start[m].min(), start[m].max(), us(done[m] - base[m]).min(), us(done[m] - base[m]).max()), file=out)
        line("entry -> prologue issued", prol - t0, m)
        line("prologue issued -> first tile", first - prol, m)
        line("K loop (first tile -> last MFMA)", loop - first, m)
        line("  of it waiting at the barrier", swait, m)
        line("  of it issuing (loads+reads+MFMA)", scomp, m)
        line("  per K step: wait", swait / np.maximum(nst - 1, 1), m)
        line("  per K step: issue", scomp / np.maximum(nst, 1), m)
        line("  longest single wait", maxw, m)
        line("ring drain (vmcnt(0) + barrier)", drain - loop, m)
        line("epilogue: acc -> LDS staging", stage, m)
        line("epilogue: row passes + stores", rows, m)
        line("epilogue total (drain -> issued)", issued - drain, m)
        line("store tail (issued -> landed)", done - issued, m)
        line("whole wave", done - t0, m)
        mf = 32.0 * nst[m].mean() * MFMA_PER_STEP.get(head.split('cfg="')[1].split('"')[0], 0) * 2  # two waves share a SIMD
        if mf:
            print("    MFMA pipe time of the tile's K loop %.2f us = %.0f %% of the whole wave" % (
                mf / tick_per_us, 100.0 * mf / (done[m] - t0[m]).mean()), file=out)
    # one wave's first 32 K steps
    i = len(live) // 3
    lg = live[i, 16:48].view(np.uint32).astype(np.float64).reshape(32, 2)
    k = int(min(32, nst[i]))
    comp = [lg[j, 0] - (lg[j - 1, 1] if j else (prol[i] - t0[i])) for j in range(k)]
    wait = [lg[j, 1] - lg[j, 0] for j in range(k)]
    print("  wave %d (K steps %d): per-step issue / wait in ticks, first %d steps" % (i, int(nst[i]), k), file=out)
    print("    issue: " + " ".join("%5.0f" % c for c in comp), file=out)
    print("    wait : " + " ".join("%5.0f" % c for c in wait), file=out)
    print("", file=out)


def _worker(outdir, probe):
    """all cases in ONE process (one engine) of the product or of the probe build; prints `ci tile ms` lines"""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import cycle_diffusion_amd as cda
    from cycle_diffusion_amd._ffi import check
    eng = cda.Engine("cuda:0", workspace_bytes=8 << 30)
    for ci, (name, B, hw, c0, c1, n, k, act, tiles, dbg) in enumerate(CASES):
        for tile in tiles:
            os.environ["CYCLEDIFF_PROBE_DBG"] = str(dbg)
            if probe:
                os.environ["CYCLEDIFF_PROBE_OUT"] = os.path.join(outdir, "c%d_t%d.bin" % (ci, tile))
            ms = C.c_float()
            try:
                check(eng.lib.cd_op_bench_conv(eng.h, B, hw, hw, c0, c1, n, k, 1, 0, act, tile, 20, C.byref(ms)))
                print("RES %d %d %.5f" % (ci, tile, ms.value), flush=True)
            except Exception as e:  # noqa: BLE001
                print("RES %d %d nan %s" % (ci, tile, str(e)[:200]), flush=True)


def run(outdir):
    os.makedirs(outdir, exist_ok=True)
    res = {}
    for probe in (False, True):
        env = dict(os.environ)
        env.pop("CYCLEDIFF_PROBE_OUT", None)
        env.pop("CYCLEDIFF_LIB", None)
        if probe:
            env["CYCLEDIFF_LIB"] = os.path.join(ROOT, "cycle-diffusion_amd", "lib", "libcyclediff_probe.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker", outdir, "1" if probe else "0"], env=env,
                           capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("RES "):
                f = ln.split()
                res[(int(f[1]), int(f[2]), probe)] = float(f[3])
        if r.returncode != 0:
            print(r.stderr[-2000:])
    with open(os.path.join(outdir, "report.txt"), "w") as rep:
        for ci, (name, B, hw, c0, c1, n, k, act, tiles, dbg) in enumerate(CASES):
            for tile in tiles:
                fl = 2.0 * B * hw * hw * n * k * k * (c0 + c1)
                ms_prod, ms_probe = res.get((ci, tile, False), float("nan")), res.get((ci, tile, True), float("nan"))
                print("== %s (act 0x%x) | tile %d | product build %.1f us = %.0f TFLOP/s | probe build %.1f us" % (
                    name, act, tile, ms_prod * 1e3, fl / ms_prod / 1e9, ms_probe * 1e3), file=rep, flush=True)
                dump = os.path.join(outdir, "c%d_t%d.bin" % (ci, tile))
                if os.path.exists(dump):
                    report(dump, rep)
                    if not (ci == 0 and tile == 20):
                        os.remove(dump)  # keep one raw dump as a sample


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    elif sys.argv[1] == "worker":
        _worker(sys.argv[2], sys.argv[3] == "1")
    else:
        report(sys.argv[2])
