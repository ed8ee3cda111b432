#!/usr/bin/env python3
"""Device-resident timing of every BASELINE.json config (and the reference bench shapes) with HIP
events on the launch stream; prints one JSON line per workload with the roofline fraction.
Not the driver contract (that is bench.py); this fills DESIGN.md's measurement table."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import synth
from ndrustfft_amd import (DctHandler, FftHandler, R2cFftHandler, _lib, nddct1, nddct2, nddct3, nddct4, ndfft, ndfft_r2c,
                           ndifft, ndifft_r2c)

PEAK = 8000.0


RAMP_MS = 300.0


def timeit(fn, steps, warmup=5, ramp_ms=None):
    ramp_ms = RAMP_MS if ramp_ms is None else ramp_ms
    # keep the device busy for ramp_ms first: the clocks take ~30-40 ms of sustained work to settle
    import time
    t0 = time.perf_counter()
    while True:
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= ramp_ms:
            break
    # two consecutive batches of `steps` launches, the faster one counts: from an idle GPU the first ~0.15 s run 20-25 % slow
    # (tools/clock_timeline.py, profiles/r07/r07c_clock_timeline.jsonl), and a ramp that ends a little early would otherwise decide a row
    best = None
    for _ in range(2):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / steps
        best = t if best is None else min(best, t)
    return best


def timeit3(fn, steps, warmup=5, ramp_ms=None, batches=3):
    """(median, min) over `batches` consecutive batches of `steps` launches after the ramp: the table's `us` is the median, `us_min` the minimum (what
    --compare uses against the minima of earlier tables: like with like -- round 5)."""
    ramp_ms = RAMP_MS if ramp_ms is None else ramp_ms
    import time
    t0 = time.perf_counter()
    while True:
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if (time.perf_counter() - t0) * 1e3 >= ramp_ms:
            break
    ts = []
    for _ in range(batches):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / steps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


MALL_BYTES = 256 << 20
NO_REREAD = False
COLD_FOOTPRINT = 768 << 20   # the inputs an HBM-sourced row rotates over add up to at least this (3 x the 256 MiB Infinity Cache)
# --pairs K: rotate K distinct (in, out) pairs so that every launch finds its input in HBM, not in the Infinity Cache.
# 0 (default since round 5) = per row: ceil(768 MiB / input bytes), at least 2, at most 64; 1 for inputs of 768 MiB and more (nothing to rotate: they do not fit the cache)
PAIRS = 0
ROWS = []   # every row printed by this process (for --compare)
REP = 0


def gpu_state():
    """Best-effort snapshot of the GPU's clocks and power from sysfs (readable without privileges): a compute-bound row measured at a lower
    core clock is a different measurement, and the table should say so (round 3's 154 vs 124 us for the same Rader row in one run)."""
    import glob
    out = {}
    try:
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            def rd(name):
                try:
                    return int(open(os.path.join(h, name)).read().strip())
                except Exception:
                    return None
            f = [rd(f"freq{i}_input") for i in range(1, 11)]
            lab = []
            for i in range(1, 11):
                try:
                    lab.append(open(os.path.join(h, f"freq{i}_label")).read().strip())
                except Exception:
                    lab.append(None)
            sclk = [v for v, l in zip(f, lab) if v and l and l.startswith("sclk")]
            mclk = [v for v, l in zip(f, lab) if v and l and l.startswith("mclk")]
            if sclk:
                out["sclk_mhz"] = round(sum(sclk) / len(sclk) / 1e6); out["sclk_min_mhz"] = round(min(sclk) / 1e6)
            if mclk:
                out["mclk_mhz"] = round(mclk[0] / 1e6)
            pw = rd("power1_average") or rd("power1_input")
            if pw:
                out["power_w"] = round(pw / 1e6)
            if out:
                break
    except Exception:
        pass
    return out


def emit(row):
    row["rep"] = REP
    row.update(gpu_state())
    ROWS.append(row)
    print(json.dumps(row), flush=True)


def set_switch(name, value):
    """The library parses its environment switches once: change + ndfft_reload_switches (csrc/switches.h)."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    _lib.default().reload_switches()


ROW_FILTER = ""   # --row: only rows whose name contains this


def pairs_for(in_bytes):
    if PAIRS > 0:
        return PAIRS
    if in_bytes >= COLD_FOOTPRINT:
        return 1
    return max(2, min(64, -(-COLD_FOOTPRINT // max(in_bytes, 1))))


def run(name, fn, x, y, h, axis, points, steps):
    """One row.  `us` / `frac_of_8TBs` are HBM-SOURCED: the call walks over rotating (in, out) pairs whose inputs add up to >= 768 MiB, so no launch finds its
    input in the 256 MiB Infinity Cache (rows whose input alone is >= 768 MiB need no rotation).  `us_reread` / `frac_reread` time the same call on ONE pair
    re-used every launch (what the tables of rounds 1-4 showed).  Each is the median of three batches; `us_min` / `us_reread_min` are the minima."""
    if ROW_FILTER and ROW_FILTER not in name:
        return
    L = _lib.default()
    in_bytes = x.numel() * x.element_size()
    nbytes = in_bytes + y.numel() * y.element_size()
    K = pairs_for(in_bytes)
    row = {"workload": name, "algorithmic_bytes": nbytes, "pairs": K, "footprint_bytes": K * nbytes,
           "timing": "3 batches of %d launches after a %d ms ramp: us = median, us_min = min" % (steps, int(RAMP_MS))}
    if K > 1:
        xs = [x] + [x.clone() for _ in range(K - 1)]; ys = [y] + [torch.empty_like(y) for _ in range(K - 1)]
        cnt = [0]
        def go():
            k = cnt[0] % K; cnt[0] += 1
            fn(xs[k], ys[k], h, axis)
        t, tmin = timeit3(go, steps)
        row["policy"] = L.last_input_policy()
        del xs, ys
        if NO_REREAD:
            tw, twmin = t, tmin
        else:
            tw, twmin = timeit3(lambda: fn(x, y, h, axis), steps)
            row["policy_reread"] = L.last_input_policy()
    else:
        t, tmin = timeit3(lambda: fn(x, y, h, axis), steps)
        row["policy"] = L.last_input_policy()
        tw, twmin = t, tmin
    gbs = nbytes / t / 1e9
    row.update({"us": round(t * 1e6, 2), "us_min": round(tmin * 1e6, 2), "GFFT-points/s": round(points / t / 1e9, 2), "GB/s": round(gbs, 1),
                "frac_of_8TBs": round(gbs / PEAK, 4), "us_reread": round(tw * 1e6, 2), "us_reread_min": round(twmin * 1e6, 2),
                "frac_reread": round(nbytes / tw / 1e9 / PEAK, 4), "path": L.last_path()})
    emit(row)


def compare(rows, prev_path, tol):
    """--compare: every row of this run against the same workload in an earlier table.  Prints the table's own noise first (each row is
    timed --repeat times, far apart: the spread between the repeats is what a difference must exceed to mean anything), then fails (exit 1)
    on any row whose BEST time is more than `tol` slower than the earlier table's best."""
    # Like with like (round 5): minima against minima, HBM-sourced against HBM-sourced, re-read against re-read.  Tables of rounds 1-4 hold ONE figure per row,
    # `us` = the faster of two batches (a minimum): a re-read figure, or -- rows named "... [cold: K rotating pairs]" -- an HBM-sourced one.
    import re
    prev = {}
    def note(key, v):
        if v is not None:
            prev[key] = min(prev.get(key, 1e30), v)
    for pth in prev_path.split(","):                 # several earlier tables: the best earlier value of every row
        for line in open(pth):
            line = line.strip()
            if line.startswith("{"):
                r = json.loads(line)
                if "workload" not in r or "us" not in r:
                    continue
                m = re.match(r"^(.*) \[cold: \d+ rotating pairs\]$", r["workload"])
                if "us_min" in r:                                    # this format
                    note((r["workload"], "hbm"), r["us_min"]); note((r["workload"], "reread"), r.get("us_reread_min"))
                elif m:
                    note((m.group(1), "hbm"), r["us"])
                else:
                    note((r["workload"], "reread"), r["us"])
    cur = {}
    for r in rows:
        if "us_min" in r:
            cur.setdefault((r["workload"], "hbm"), []).append(dict(r, us=r["us_min"]))
            if r.get("pairs", 1) > 1:
                cur.setdefault((r["workload"], "reread"), []).append(dict(r, us=r["us_reread_min"]))
        else:
            m = re.match(r"^(.*) \[cold: \d+ rotating pairs\]$", r["workload"])
            cur.setdefault((m.group(1), "hbm") if m else (r["workload"], "reread"), []).append(r)
    noise = []
    for w, rs in cur.items():
        if len(rs) > 1:
            us = [r["us"] for r in rs]
            noise.append((max(us) / min(us) - 1, w, us))
    if noise:
        noise.sort(reverse=True)
        med = sorted(n[0] for n in noise)[len(noise) // 2]
        print(f"# noise of this table (max/min - 1 over {len(next(iter(cur.values())))} repeats): median {med * 100:.1f} %, worst:", file=sys.stderr)
        for n, w, us in noise[:8]:
            print(f"#   {n * 100:5.1f} %  {w}  {us}", file=sys.stderr)
    bad = []; matched = 0
    for w, rs in cur.items():
        if w not in prev:
            continue
        matched += 1
        best = min(r["us"] for r in rs)
        if best > prev[w] * (1 + tol):
            bad.append((best / prev[w] - 1, w, prev[w], best, rs[0].get("path"), [r.get("sclk_mhz") for r in rs]))
    print(f"# compared {matched} rows with {prev_path}: {len(bad)} slower by more than {tol * 100:.0f} %", file=sys.stderr)
    for d, w, a, b, path, clk in sorted(bad, reverse=True):
        print(f"#   +{d * 100:5.1f} %  {w}: {a} -> {b} us  path={path} sclk={clk}", file=sys.stderr)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--only", default=""); ap.add_argument("--ramp-ms", type=float, default=300.0); ap.add_argument("--pairs", type=int, default=0, help="rotating (in, out) pairs per row; 0 = per row, inputs adding up to >= 768 MiB (HBM-sourced), see pairs_for()"); ap.add_argument("--preheat-s", type=float, default=0.0)
    ap.add_argument("--row", default="", help="only rows whose name contains this text")
    ap.add_argument("--no-reread", action="store_true", help="skip the re-read measurement of every row (profiling runs: rocprofv3's per-kernel averages then belong to the HBM-sourced launches)")
    ap.add_argument("--repeat", type=int, default=1, help="run the whole selected table this many times, one after the other (rows carry `rep`): the spread between repeats is the table's noise")
    ap.add_argument("--compare", default="", help="an earlier table (jsonl): exit 1 if any row's best time is more than --tolerance slower than there")
    ap.add_argument("--tolerance", type=float, default=0.06)
    ap.add_argument("--rows-from", default="", help="with --compare: do not measure, compare this table (jsonl) instead (runs anywhere, no GPU)")
    a = ap.parse_args()
    if a.rows_from:
        rows = [json.loads(l) for l in open(a.rows_from) if l.strip().startswith("{")]
        sys.exit(compare([r for r in rows if "workload" in r and "us" in r], a.compare, a.tolerance))
    global REP, ROW_FILTER
    ROW_FILTER = a.row
    rc = 0
    for REP in range(max(1, a.repeat)):
        table(a)
    if a.compare:
        rc = compare(ROWS, a.compare, a.tolerance)
    sys.exit(rc)


def table(a):
    if a.preheat_s > 0:
        # The FIRST process on a fresh box runs LDS- / latency-bound kernels 15-25 % slower for its first tens of seconds (core clock; HBM-bound kernels are
        # unaffected): `nddct1` cfg4 234 us in the first process, 184 us in every later one (profiles/r04/r04s_abab_pow2real_cfg4.txt).  Keep the GPU busy first.
        import time
        xh = torch.from_numpy(synth.complex_array((16627, 1009))).to(torch.device("cuda:0")); yh = torch.empty_like(xh); hh = FftHandler(1009)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.preheat_s:
            for _ in range(50):
                ndfft(xh, yh, hh, 1)
            torch.cuda.synchronize()
        del xh, yh
    global RAMP_MS, PAIRS, NO_REREAD
    RAMP_MS = a.ramp_ms; PAIRS = a.pairs; NO_REREAD = a.no_reread
    dev = torch.device("cuda:0")
    want = lambda k: (not a.only) or a.only in k
    if want("cfg2"):
        x = torch.from_numpy(synth.complex_array((4096, 4096))).to(dev); y = torch.empty_like(x)
        run("cfg2 ndfft axis=1 4096x4096 c128", ndfft, x, y, FftHandler(4096), 1, x.numel(), a.steps)
        run("cfg2' ndifft axis=1 4096x4096 c128", ndifft, x, y, FftHandler(4096), 1, x.numel(), a.steps)
    if want("cfg3"):
        n = 8192; m = n // 2 + 1
        x = torch.from_numpy(synth.real_array((n, n), np.float32)).to(dev)
        w = torch.empty((m, n), dtype=torch.complex64, device=dev); o = torch.empty_like(w)
        hr = R2cFftHandler(n, np.float32); hc = FftHandler(n, np.float32)
        run("cfg3A ndfft_r2c axis=0 8192x8192 f32", ndfft_r2c, x, w, hr, 0, n * n, a.steps)
        run("cfg3B ndfft axis=1 4097x8192 c64", ndfft, w, o, hc, 1, m * n, a.steps)
        run("cfg3A' ndifft_r2c axis=0 -> 8192x8192 f32", ndifft_r2c, w, x, hr, 0, n * n, a.steps)
        xr = torch.from_numpy(synth.real_array((n, n), np.float32)).to(dev); wr = torch.empty((n, m), dtype=torch.complex64, device=dev)
        run("(rows) ndfft_r2c axis=1 8192x8192 f32", ndfft_r2c, xr, wr, hr, 1, n * n, a.steps)
    if a.only == "cfg3A_only":
        n = 8192; m = n // 2 + 1
        x = torch.from_numpy(synth.real_array((n, n), np.float32)).to(dev)
        w = torch.empty((m, n), dtype=torch.complex64, device=dev)
        run("cfg3A ndfft_r2c axis=0 8192x8192 f32", ndfft_r2c, x, w, R2cFftHandler(n, np.float32), 0, n * n, a.steps)
    if a.only == "cfg3Ap_only":
        n = 8192; m = n // 2 + 1
        w = torch.from_numpy(synth.complex_array((m, n), np.complex64)).to(dev)
        x = torch.empty((n, n), dtype=torch.float32, device=dev)
        run("cfg3A' ndifft_r2c axis=0 -> 8192x8192 f32", ndifft_r2c, w, x, R2cFftHandler(n, np.float32), 0, n * n, a.steps)
    if want("cfg4"):
        x = torch.from_numpy(synth.real_array((256, 256, 512))).to(dev); y = torch.empty_like(x)
        h = DctHandler(512)
        for nm, fn in (("nddct2", nddct2), ("nddct3", nddct3), ("nddct1", nddct1), ("nddct4", nddct4)):
            run(f"cfg4 {nm} axis=2 256x256x512 f64", fn, x, y, h, 2, x.numel(), a.steps)
        h1 = DctHandler(256)
        run("cfg4' nddct2 axis=1 256x256x512 f64", nddct2, x, y, h1, 1, x.numel(), a.steps)
        run("cfg4'' nddct2 axis=0 256x256x512 f64", nddct2, x, y, h1, 0, x.numel(), a.steps)
    if want("cfg5"):
        x = torch.from_numpy(synth.complex_array((8192, 4096))).to(dev); y = torch.empty_like(x)
        run("cfg5 shard ndfft axis=1 8192x4096 c128", ndfft, x, y, FftHandler(4096), 1, x.numel(), a.steps)
    if want("hostpath"):
        import time
        x = synth.complex_array((4096, 4096)); y = np.zeros_like(x); h = FftHandler(4096)
        ndfft(x, y, h, 1)
        t0 = time.perf_counter(); reps = 3
        for _ in range(reps):
            ndfft(x, y, h, 1)
        t = (time.perf_counter() - t0) / reps
        print(json.dumps({"workload": "hostpath ndfft_exec (pageable host arrays, PCIe both ways) 4096x4096 c128", "us": round(t * 1e6, 1),
                          "GFFT-points/s": round(x.size / t / 1e9, 3), "host_GB/s": round(2 * x.nbytes / t / 1e9, 2)}), flush=True)
        from ndrustfft_amd import pinned_empty
        xp = pinned_empty(x.shape, x.dtype); xp[...] = x; yp = pinned_empty(y.shape, y.dtype)
        ndfft(xp, yp, h, 1)
        assert np.abs(yp - y).max() <= 1e-12 * np.abs(y).max()
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps):
            ndfft(xp, yp, h, 1)
        t = (time.perf_counter() - t0) / reps
        print(json.dumps({"workload": "hostpath ndfft_exec (PINNED host arrays from ndfft_host_alloc: upload || transform || download over 8 row chunks) 4096x4096 c128",
                          "us": round(t * 1e6, 1), "GFFT-points/s": round(x.size / t / 1e9, 3), "host_GB/s": round(2 * x.nbytes / t / 1e9, 2)}), flush=True)
    if want("longlanes"):
        for n, rows, cdt, rdt in ((1 << 16, 256, np.complex128, np.float64), (1 << 20, 16, np.complex128, np.float64), (6000, 2048, np.complex128, np.float64),
                                  (1 << 20, 32, np.complex64, np.float32)):
            x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
            run(f"long ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), max(a.steps // 3, 3))
        x = torch.from_numpy(synth.real_array((64, 1 << 18))).to(dev); y = torch.empty_like(x)
        hd = DctHandler(1 << 18)
        run("long nddct2 axis=1 64x262144 f64", nddct2, x, y, hd, 1, x.numel(), max(a.steps // 3, 3))
        run("long nddct3 axis=1 64x262144 f64", nddct3, x, y, hd, 1, x.numel(), max(a.steps // 3, 3))
        run("long nddct4 axis=1 64x262144 f64", nddct4, x, y, hd, 1, x.numel(), max(a.steps // 3, 3))
        x1 = torch.from_numpy(synth.real_array((63, (1 << 18) + 1))).to(dev); y1 = torch.empty_like(x1)
        run("long nddct1 axis=1 63x262145 f64", nddct1, x1, y1, DctHandler((1 << 18) + 1), 1, x1.numel(), max(a.steps // 3, 3))
        del x1, y1
        xh = torch.from_numpy(synth.complex_array((64, (1 << 17) + 1))).to(dev); hr = R2cFftHandler(1 << 18)
        run("long ndfft_r2c axis=1 64x262144 f64", ndfft_r2c, x, xh, hr, 1, (x.numel() + 2 * xh.numel()) // 2, max(a.steps // 3, 3))
        run("long ndifft_r2c axis=1 64x262144 f64", ndifft_r2c, xh, y, hr, 1, (x.numel() + 2 * xh.numel()) // 2, max(a.steps // 3, 3))
        del x, y, xh
        # smooth lane lengths that are NOT powers of two (round 6: two passes on hiprtc-specialised four-step kernels; the six-pass transpose route before)
        for n, rows in ((196608, 85), (1000000, 16)):
            x = torch.from_numpy(synth.complex_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"long-smooth ndfft axis=1 {rows}x{n} complex128", ndfft, x, y, FftHandler(n), 1, x.numel(), max(a.steps // 3, 3))
            del x, y
        x = torch.from_numpy(synth.real_array((85, 196608))).to(dev); y = torch.empty_like(x)
        run("long-smooth nddct2 axis=1 85x196608 f64", nddct2, x, y, DctHandler(196608), 1, x.numel(), max(a.steps // 3, 3))
        run("long-smooth nddct3 axis=1 85x196608 f64", nddct3, x, y, DctHandler(196608), 1, x.numel(), max(a.steps // 3, 3))
        xh = torch.empty((85, 196608 // 2 + 1), dtype=torch.complex128, device=dev)
        run("long-smooth ndfft_r2c axis=1 85x196608 f64", ndfft_r2c, x, xh, R2cFftHandler(196608), 1, (x.numel() + 2 * xh.numel()) // 2, max(a.steps // 3, 3))
    if a.only == "radercol":
        x = torch.from_numpy(synth.real_array((512, 256 * 256))).to(dev); y = torch.empty_like(x)
        run("nddct1 axis=0 512x65536 f64", nddct1, x, y, DctHandler(512), 0, x.numel(), a.steps)
        x = torch.from_numpy(synth.complex_array((1009, 16384))).to(dev); y = torch.empty_like(x)
        run("ndfft axis=0 1009x16384 c128", ndfft, x, y, FftHandler(1009), 0, x.numel(), a.steps)
        x = torch.from_numpy(synth.complex_array((127, 131072), np.complex64)).to(dev); y = torch.empty_like(x)
        run("ndfft axis=0 127x131072 c64", ndfft, x, y, FftHandler(127, np.float32), 0, x.numel(), a.steps)
        return
    if a.only == "rader3":
        for n in (47, 139, 235, 311, 590):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n), np.complex64)).to(dev); y = torch.empty_like(x)
            run(f"rader3 ndfft axis=1 {rows}x{n} complex64", ndfft, x, y, FftHandler(n, np.float32), 1, x.numel(), a.steps)
        return
    if a.only == "rader2":
        for rad in ("1", "0"):
            set_switch("NDFFT_RADER", rad)
            for n, cdt, rdt in ((306, np.complex128, np.float64), (513, np.complex128, np.float64), (532, np.complex128, np.float64), (2336, np.complex128, np.float64),
                                (103, np.complex128, np.float64), (137, np.complex128, np.float64), (2466, np.complex128, np.float64), (513, np.complex64, np.float32), (103, np.complex64, np.float32)):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"rader2[NDFFT_RADER={rad}] ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
        set_switch("NDFFT_RADER", None)
        return
    if a.only == "r2crows":
        for n in (256, 1024, 8192):
            rows = (1 << 26) // n
            x = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
            run(f"r2crows ndfft_r2c f32 {rows}x{n}", ndfft_r2c, x, w, R2cFftHandler(n, np.float32), 1, x.numel(), a.steps)
            xc = torch.from_numpy(synth.complex_array((rows // 2, n), np.complex64)).to(dev); yc = torch.empty_like(xc)
            run(f"r2crows ndfft c64 {rows // 2}x{n}", ndfft, xc, yc, FftHandler(n, np.float32), 1, xc.numel(), a.steps)
        return
    if a.only == "bluesweep":
        for n, cdt, rdt in ((59, np.complex128, np.float64), (83, np.complex128, np.float64), (107, np.complex128, np.float64), (227, np.complex128, np.float64), (263, np.complex128, np.float64),
                            (479, np.complex128, np.float64), (515, np.complex128, np.float64), (983, np.complex128, np.float64), (1283, np.complex128, np.float64), (2039, np.complex128, np.float64),
                            (227, np.complex64, np.float32), (515, np.complex64, np.float32), (983, np.complex64, np.float32)):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
            run(f"bluesweep ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
        x = torch.from_numpy(synth.real_array((65536, 228))).to(dev); y = torch.empty_like(x)
        run("bluesweep nddct1 axis=1 65536x228 f64", nddct1, x, y, DctHandler(228), 1, x.numel(), a.steps)
        return
    if a.only == "oddreal":
        for v in ("1", "0"):
            set_switch("NDFFT_PLAIN", v)
            for n in (63, 125, 243, 625, 1001, 3003):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.real_array((rows, n))).to(dev); y = torch.empty_like(x)
                run(f"oddreal[NDFFT_PLAIN={v}] nddct2 axis=1 {rows}x{n} f64", nddct2, x, y, DctHandler(n), 1, x.numel(), a.steps)
                xf = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
                run(f"oddreal[NDFFT_PLAIN={v}] ndfft_r2c axis=1 {rows}x{n} f32", ndfft_r2c, xf, w, R2cFftHandler(n, np.float32), 1, xf.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((625, 16384))).to(dev); y = torch.empty_like(x)
            run(f"oddreal[NDFFT_PLAIN={v}] nddct2 axis=0 625x16384 f64", nddct2, x, y, DctHandler(625), 0, x.numel(), a.steps)
        set_switch("NDFFT_PLAIN", None)
        return
    if a.only == "shortplan":
        for n in (72, 80, 96, 100, 120, 144, 160, 200, 250):
            rows = (1 << 24) // n
            for cdt, rdt in ((np.complex64, np.float32), (np.complex128, np.float64)):
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"shortplan ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((rows, 2 * n))).to(dev); y = torch.empty_like(x)
            run(f"shortplan nddct2 axis=1 {rows}x{2 * n} f64", nddct2, x, y, DctHandler(2 * n), 1, x.numel(), a.steps)
            xf = torch.from_numpy(synth.real_array((rows, 2 * n), np.float32)).to(dev); w = torch.empty((rows, n + 1), dtype=torch.complex64, device=dev)
            run(f"shortplan ndfft_r2c axis=1 {rows}x{2 * n} f32", ndfft_r2c, xf, w, R2cFftHandler(2 * n, np.float32), 1, xf.numel(), a.steps)
        return
    if a.only == "c64short":
        for n in (72, 80, 96, 100, 120, 144, 200, 250):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n), np.complex64)).to(dev); y = torch.empty_like(x)
            run(f"c64short ndfft axis=1 {rows}x{n} complex64", ndfft, x, y, FftHandler(n, np.float32), 1, x.numel(), a.steps)
        return
    if a.only == "c2cplan":
        for n in (72, 80, 120, 300, 360, 600, 1200, 1500, 3000, 6000, 12000):
            for cdt, rdt in ((np.complex128, np.float64), (np.complex64, np.float32)):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"c2cplan ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
        for n in (120, 300, 1500):
            x = torch.from_numpy(synth.complex_array((n, (1 << 24) // n // 8 * 8))).to(dev); y = torch.empty_like(x)
            run(f"c2cplan ndfft axis=0 {n}x{x.shape[1]} complex128", ndfft, x, y, FftHandler(n), 0, x.numel(), a.steps)
        return
    if a.only == "realplan":
        # real-data transforms whose inner FFT is smooth but not a power of two: rows and column tiles of the specialised kernel
        for n in (72, 96, 100, 120, 144, 200, 300, 360, 500, 1000, 1200, 2000, 3000, 6000):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.real_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"realplan nddct2 axis=1 {rows}x{n} f64", nddct2, x, y, DctHandler(n), 1, x.numel(), a.steps)
            xf = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
            run(f"realplan ndfft_r2c axis=1 {rows}x{n} f32", ndfft_r2c, xf, w, R2cFftHandler(n, np.float32), 1, xf.numel(), a.steps)
        for n in (100, 500, 1000):
            cols = (1 << 24) // n
            x = torch.from_numpy(synth.real_array((n, cols))).to(dev); y = torch.empty_like(x)
            run(f"realplan nddct2 axis=0 {n}x{cols} f64", nddct2, x, y, DctHandler(n), 0, x.numel(), a.steps)
            xf = torch.from_numpy(synth.real_array((n, cols), np.float32)).to(dev); w = torch.empty((n // 2 + 1, cols), dtype=torch.complex64, device=dev)
            run(f"realplan ndfft_r2c axis=0 {n}x{cols} f32", ndfft_r2c, xf, w, R2cFftHandler(n, np.float32), 0, xf.numel(), a.steps)
        return
    if a.only == "jitcol":
        for n, cols, cdt, rdt in ((1000, 16384, np.complex128, np.float64), (1000, 16384, np.complex64, np.float32), (264, 65536, np.complex128, np.float64), (3000, 4096, np.complex128, np.float64),
                                  (96, 131072, np.complex128, np.float64)):
            x = torch.from_numpy(synth.complex_array((n, cols), cdt)).to(dev); y = torch.empty_like(x)
            run(f"jitcol ndfft axis=0 {n}x{cols} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 0, x.numel(), a.steps)
        x = torch.from_numpy(synth.real_array((1000, 16384))).to(dev); y = torch.empty_like(x)
        run("jitcol nddct2 axis=0 1000x16384 f64", nddct2, x, y, DctHandler(1000), 0, x.numel(), a.steps)
        return
    if a.only == "raderbig":
        for rad in ("1", "0"):
            set_switch("NDFFT_RADER", rad)
            for n, cdt, rdt in ((8191, np.complex128, np.float64), (7001, np.complex128, np.float64), (8191, np.complex64, np.float32), (16001, np.complex64, np.float32)):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"raderbig[NDFFT_RADER={rad}] ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), max(a.steps // 4, 3))
        set_switch("NDFFT_RADER", None)
        return
    if a.only == "radersweep":
        for n, cdt, rdt in ((1009, np.complex128, np.float64), (127, np.complex128, np.float64), (511, np.complex128, np.float64), (2017, np.complex128, np.float64),
                            (4001, np.complex128, np.float64), (1009, np.complex64, np.float32)):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
            run(f"ndfft {n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
        x = torch.from_numpy(synth.real_array((256, 256, 512))).to(dev); y = torch.empty_like(x)
        run("nddct1 axis=2 256x256x512 f64", nddct1, x, y, DctHandler(512), 2, x.numel(), a.steps)
        return
    if want("primes"):
        # lengths with a prime factor > 13: Rader / Good-Thomas (rader_kernel.h) against Bluestein (NDFFT_RADER=0)
        for rad in ("1", "0"):
            set_switch("NDFFT_RADER", rad)
            tag = "rader" if rad == "1" else "bluestein"
            for n, cdt, rdt in ((1009, np.complex128, np.float64), (97, np.complex128, np.float64), (127, np.complex128, np.float64), (257, np.complex128, np.float64),
                                (511, np.complex128, np.float64), (2017, np.complex128, np.float64), (4001, np.complex128, np.float64), (3027, np.complex128, np.float64),
                                (1009, np.complex64, np.float32), (511, np.complex64, np.float32), (4001, np.complex64, np.float32)):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"primes[{tag}] ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((256, 256, 512))).to(dev); y = torch.empty_like(x)
            run(f"primes[{tag}] nddct1 axis=2 256x256x512 f64", nddct1, x, y, DctHandler(512), 2, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((512, 256 * 256))).to(dev); y = torch.empty_like(x)
            run(f"primes[{tag}] nddct1 axis=0 512x65536 f64", nddct1, x, y, DctHandler(512), 0, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((16384, 2018))).to(dev); y = torch.empty((16384, 1010), dtype=torch.complex128, device=dev)
            run(f"primes[{tag}] ndfft_r2c axis=1 16384x2018 f64", ndfft_r2c, x, y, R2cFftHandler(2018), 1, x.numel(), a.steps)
        set_switch("NDFFT_RADER", None)
    if want("generic"):
        for n in (1000, 264, 1331, 1009, 3000, 96):
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"generic ndfft axis=1 {rows}x{n} c128", ndfft, x, y, FftHandler(n), 1, x.numel(), a.steps)
        x = torch.from_numpy(synth.complex_array((1000, 16384))).to(dev); y = torch.empty_like(x)
        run("generic ndfft axis=0 1000x16384 c128", ndfft, x, y, FftHandler(1000), 0, x.numel(), a.steps)
        x = torch.from_numpy(synth.real_array((16384, 1000))).to(dev); y = torch.empty_like(x)
        run("generic nddct2 axis=1 16384x1000 f64", nddct2, x, y, DctHandler(1000), 1, x.numel(), a.steps)
        x = torch.from_numpy(synth.complex_array((16384, 1000), np.complex64)).to(dev); y = torch.empty_like(x)
        run("generic ndfft axis=1 16384x1000 c64", ndfft, x, y, FftHandler(1000, np.float32), 1, x.numel(), a.steps)
    if want("fft2d"):
        # two-axis transforms with the work array resident in HBM (examples/fft2.rs, rfft2.rs at scale)
        for n, cdt, rdt in ((4096, np.complex128, np.float64), (8192, np.complex64, np.float32), (1024, np.complex128, np.float64)):
            x = torch.from_numpy(synth.complex_array((n, n), cdt)).to(dev); w = torch.empty_like(x); y = torch.empty_like(x)
            h = FftHandler(n, rdt)
            def fft2(_x, _y, _h, _ax):
                ndfft(_x, w, _h, 1); ndfft(w, _y, _h, 0)
            t, tmin = timeit3(lambda: fft2(x, y, h, 0), a.steps)
            nbytes = 4 * x.numel() * x.element_size()
            emit({"workload": f"fft2 {n}x{n} {np.dtype(cdt).name} (axis 1 then axis 0, work array in HBM)", "us": round(t * 1e6, 1), "us_min": round(tmin * 1e6, 1), "pairs": 1,
                  "GFFT-points/s": round(2 * x.numel() / t / 1e9, 1), "algorithmic_bytes(2 passes)": nbytes,
                  "GB/s": round(nbytes / t / 1e9, 1), "frac_of_8TBs": round(nbytes / t / 1e9 / PEAK, 4), "path": _lib.default().last_path()})
    if want("refbench"):
        for n in (128, 264, 512, 1024):
            x = torch.from_numpy(synth.bench_fill_complex((n, n))).to(dev); y = torch.empty_like(x)
            run(f"refbench fft2d n={n} axis=0 c128", ndfft, x, y, FftHandler(n), 0, n * n, a.steps)
        for n in (129, 265, 513, 1025):
            x = torch.from_numpy(np.arange(n * n, dtype=np.float64).reshape(n, n)).to(dev); y = torch.empty_like(x)
            run(f"refbench dct2d n={n} axis=0 f64", nddct1, x, y, DctHandler(n), 0, n * n, a.steps)
    if want("smallsweep"):
        # short dense lanes: the wavefront kernel (wave_kernel.h); NDFFT_WAVE=0 shows the kernels it replaces
        for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
            for n in (2, 4, 8, 16, 32, 64):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"small ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
    if want("smallcol"):
        # short lanes along a STRIDED axis (middle axis of a 3-D array, adjacent lanes contiguous)
        for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
            for n in (2, 3, 4, 6, 8, 11, 12, 13, 16, 32, 64):
                outer = (1 << 24) // (n * 4096)
                x = torch.from_numpy(synth.complex_array((outer, n, 4096), cdt)).to(dev); y = torch.empty_like(x)
                run(f"smallcol ndfft axis=1 {outer}x{n}x4096 {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
    if want("tinyrow"):
        # very short NON-power-of-two dense lanes (and n = 16 padded): the thread-per-lane kernel, LDS-staged
        for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
            for n in (3, 5, 6, 7, 9, 10, 11, 12, 13):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"tinyrow ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
    if want("tinyreal"):
        # real-data transforms on very short lanes: thread-per-lane, the transform as a dense matrix (tinymat_kernel.h)
        for n in (4, 6, 8, 9, 12, 16):
            rows = (1 << 25) // n
            x = torch.from_numpy(synth.real_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"tinyreal nddct2 axis=1 {rows}x{n} f64", nddct2, x, y, DctHandler(n), 1, x.numel(), a.steps)
            xc = torch.from_numpy(synth.real_array((rows // 4096, n, 4096))).to(dev); yc = torch.empty_like(xc)
            run(f"tinyreal nddct2 axis=1 {rows // 4096}x{n}x4096 f64", nddct2, xc, yc, DctHandler(n), 1, xc.numel(), a.steps)
            xf = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
            run(f"tinyreal ndfft_r2c axis=1 {rows}x{n} f32", ndfft_r2c, xf, w, R2cFftHandler(n, np.float32), 1, xf.numel(), a.steps)
            run(f"tinyreal ndifft_r2c axis=1 -> {rows}x{n} f32", ndifft_r2c, w, xf, R2cFftHandler(n, np.float32), 1, xf.numel(), a.steps)
    if want("midc2c"):
        # C2C lanes of 14..100 points: thread-per-lane two-factor kernels (reg_kernel.h) vs the general register kernel
        for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
            for n in (14, 17, 18, 20, 23, 24, 30, 31, 34, 36, 40, 46, 48, 56, 60, 62, 63, 64, 72, 80, 96, 100):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"midc2c ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)
                xc = torch.from_numpy(synth.complex_array((rows // 2048, n, 2048), cdt)).to(dev); yc = torch.empty_like(xc)
                run(f"midc2c ndfft axis=1 {rows // 2048}x{n}x2048 {np.dtype(cdt).name}", ndfft, xc, yc, FftHandler(n, rdt), 1, xc.numel(), a.steps)
    if want("midreal"):
        # real-data transforms on medium-short lanes (beyond the thread-per-lane kernels, below the pow2 real kernels)
        for n in (18, 20, 24, 30, 32, 48, 64, 96, 100, 128):
            rows = (1 << 25) // n
            x = torch.from_numpy(synth.real_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"midreal nddct2 axis=1 {rows}x{n} f64", nddct2, x, y, DctHandler(n), 1, x.numel(), a.steps)
            xf = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
            run(f"midreal ndfft_r2c axis=1 {rows}x{n} f32", ndfft_r2c, xf, w, R2cFftHandler(n, np.float32), 1, xf.numel(), a.steps)
            xc = torch.from_numpy(synth.complex_array((rows // 2, n))).to(dev); yc = torch.empty_like(xc)
            run(f"midreal ndfft axis=1 {rows // 2}x{n} c128", ndfft, xc, yc, FftHandler(n), 1, xc.numel(), a.steps)
    if want("landscape"):
        # the whole lane-length landscape on dense rows, 2^24 points per call: which kernel serves which n, and how well
        sizes = (2, 3, 4, 6, 8, 12, 16, 17, 24, 31, 32, 48, 63, 64, 96, 100, 128, 256, 500, 512, 1000, 1024, 2048, 4096, 8192, 16384)
        for n in sizes:
            rows = (1 << 24) // n
            x = torch.from_numpy(synth.complex_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"landscape ndfft c128 n={n}", ndfft, x, y, FftHandler(n), 1, x.numel(), a.steps)
            x = torch.from_numpy(synth.complex_array((rows, n), np.complex64)).to(dev); y = torch.empty_like(x)
            run(f"landscape ndfft c64 n={n}", ndfft, x, y, FftHandler(n, np.float32), 1, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((rows, n))).to(dev); y = torch.empty_like(x)
            run(f"landscape nddct2 f64 n={n}", nddct2, x, y, DctHandler(n), 1, x.numel(), a.steps)
            x = torch.from_numpy(synth.real_array((rows, n), np.float32)).to(dev); w = torch.empty((rows, n // 2 + 1), dtype=torch.complex64, device=dev)
            run(f"landscape ndfft_r2c f32 n={n}", ndfft_r2c, x, w, R2cFftHandler(n, np.float32), 1, x.numel(), a.steps)
    if want("pow2sweep"):
        for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
            for n in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
                rows = (1 << 24) // n
                x = torch.from_numpy(synth.complex_array((rows, n), cdt)).to(dev); y = torch.empty_like(x)
                run(f"pow2 ndfft axis=1 {rows}x{n} {np.dtype(cdt).name}", ndfft, x, y, FftHandler(n, rdt), 1, x.numel(), a.steps)


if __name__ == "__main__":
    main()
