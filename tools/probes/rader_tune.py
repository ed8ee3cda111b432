"""Sweeps explicit FFT_(p-1) configurations of the Rader kernel (NDFFT_RADER_CFG / NDFFT_RADER_LPB, jit.hip) for a few lengths
and prints the time of each: the data behind the configuration heuristic in jit.hip (rader_choose).  GPU only."""
import itertools
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch

import synth
from ndrustfft_amd import DctHandler, FftHandler, _lib, nddct1, ndfft

RADS = (16, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2)
LPB_SWEEP = os.environ.get("RADER_TUNE_LPB", "0") == "1"


def lists(m, maxr=16, depth=0):
    if m == 1:
        yield ()
        return
    if depth >= 5:
        return
    for r in RADS:
        if r <= maxr and m % r == 0:
            for rest in lists(m // r, r, depth + 1):
                yield (r,) + rest


def configs(M, emax):
    out = []
    for rl in lists(M):
        if len(rl) > min(len(x) for x in lists(M)) + 1:
            continue
        tpls = set()
        for r in rl:
            for s in range(1, 5):
                tpls.add(-(-(M // r) // s))
            tpls.add(2 * (M // r))          # half the threads idle in that pass: more threads for the stage / PRE / POST loops of short lanes
        for tpl in sorted(tpls):
            if tpl < 1 or tpl > 1024:
                continue
            e = max(-(-(M // r) // tpl) * r for r in rl)
            work = sum(-(-(M // r) // tpl) * tpl * r for r in rl) / (M * len(rl))
            if e <= emax and work <= (2.1 if M < 200 else 1.25):
                out.append((tpl, rl, e, work))
    return out


def timeit(fn, steps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def main():
    dev = torch.device("cuda:0")
    cases = [(int(x.split(":")[0]), x.split(":")[1]) for x in sys.argv[1:]] or [(1009, "c128"), (2017, "c128"), (127, "c128"), (4001, "c128"), (1009, "c64")]
    for F, kind in cases:
        dct = kind == "dct"
        cdt, rdt = (np.complex128, np.float64) if kind in ("c128", "dct") else (np.complex64, np.float32)
        # largest prime factor
        p, m = 1, F
        f = 2
        while f * f <= m:
            while m % f == 0:
                p, m = f, m // f
            f += 1
        if m > 1:
            p = m
        M = p - 1
        rows = (1 << 24) // F
        if dct:
            from ndrustfft_amd import nddct2
            x = torch.from_numpy(synth.real_array((rows // 2, 2 * F))).to(dev)
        else:
            x = torch.from_numpy(synth.complex_array((rows, F), cdt)).to(dev)
        y = torch.empty_like(x)
        res = []
        mc = F // p
        for tpl, rl, e, work in configs(M, 21 if kind in ("c128", "dct") else 32):
            if work > (2.1 if M < 200 else 1.16):
                continue
            lt = tpl * mc
            lpbs = sorted({l for l in (1, 2, 3, 4, 6, 8, 12, 16, 64 // lt, 128 // lt, 192 // lt, 256 // lt) if l >= 1 and l * lt <= 512}) if LPB_SWEEP else (0,)
            for lpb in lpbs:
                os.environ["NDFFT_RADER_CFG"] = f"{tpl}:" + ".".join(map(str, rl))
                if lpb:
                    os.environ["NDFFT_RADER_LPB"] = str(lpb)
                else:
                    os.environ.pop("NDFFT_RADER_LPB", None)
                h = DctHandler(2 * F) if dct else FftHandler(F, rdt)
                fn = (lambda: nddct2(x, y, h, 1)) if dct else (lambda: ndfft(x, y, h, 1))
                try:
                    fn()
                    path = _lib.default().last_path()
                    t = timeit(fn) if path.startswith("rader") else float("nan")
                except Exception as ex:      # noqa: BLE001
                    path, t = f"error {ex}", float("nan")
                res.append((t, tpl, rl, e, work, path, lpb))
                print(json.dumps({"F": F, "kind": kind, "lpb": lpb, "tpl": tpl, "radix": rl, "e": e, "work": round(work, 3), "us": round(t, 1), "path": path}), flush=True)
        res.sort(key=lambda r: (r[0] != r[0], r[0]))
        print(f"## best for F={F} {kind}: " + "; ".join(f"{r[0]:.1f}us tpl={r[1]} lpb={r[6]} {'.'.join(map(str, r[2]))} e={r[3]}" for r in res[:6]), flush=True)


if __name__ == "__main__":
    main()
