#!/usr/bin/env python3
"""Why does the SAME workload time differently inside one run of tools/bench_configs.py (round 3: Rader 16627x1009 c128 154 us as the first row of
its section, 124 us four rows into another)?  This script times each of the four rows the round-3 review flagged in windows of ~0.2 s for several
seconds, twice: starting from an idle GPU, and right after 3 s of the HBM-bound cfg2 kernel -- and prints the core clock / power sysfs shows
beside every window.  One JSON line per window.

    python tools/clock_timeline.py [--seconds 6]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import synth
from bench_configs import gpu_state
from ndrustfft_amd import FftHandler, _lib, ndfft


def windows(tag, fn, seconds, per=20):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        e0.record()
        for _ in range(per):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / per
        if us * per < 150e3:                         # windows of ~0.2 s whatever the kernel
            per = max(per, int(per * 200e3 / max(us * per, 1.0)))
        row = {"case": tag, "t_s": round(time.perf_counter() - t0, 2), "us": round(us, 2), "launches": per}
        row.update(gpu_state())
        print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=6.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.default()
    xc = torch.from_numpy(synth.complex_array((4096, 4096))).to(dev); yc = torch.empty_like(xc); hc = FftHandler(4096)
    cases = []
    x = torch.from_numpy(synth.complex_array((16627, 1009))).to(dev); y = torch.empty_like(x); h = FftHandler(1009)
    cases.append(("rader 16627x1009 c128", lambda x=x, y=y, h=h: ndfft(x, y, h, 1)))
    x = torch.from_numpy(synth.complex_array((32832, 511))).to(dev); y = torch.empty_like(x); h = FftHandler(511)
    cases.append(("primes 32832x511 c128", lambda x=x, y=y, h=h: ndfft(x, y, h, 1)))
    x = torch.from_numpy(synth.complex_array((81, 100, 2048), np.complex64)).to(dev); y = torch.empty_like(x); h = FftHandler(100, np.float32)
    cases.append(("jit_col 81x100x2048 c64 axis=1", lambda x=x, y=y, h=h: ndfft(x, y, h, 1)))
    x = torch.from_numpy(synth.complex_array((16384, 1000), np.complex64)).to(dev); y = torch.empty_like(x); h = FftHandler(1000, np.float32)
    cases.append(("rows 16384x1000 c64", lambda x=x, y=y, h=h: ndfft(x, y, h, 1)))
    for tag, fn in cases:
        fn(); torch.cuda.synchronize()
        print(json.dumps({"case": tag, "path": L.last_path(), "policy": L.last_input_policy()}), flush=True)
    for tag, fn in cases:
        time.sleep(2.0)                                           # idle GPU: clocks fall back
        windows(tag + " | from idle", fn, a.seconds)
        windows("cfg2 4096x4096 c128 (HBM-bound) before " + tag, lambda: ndfft(xc, yc, hc, 1), 3.0)
        windows(tag + " | after 3 s of cfg2", fn, a.seconds)


if __name__ == "__main__":
    main()
