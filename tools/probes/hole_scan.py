"""Scans lane lengths for performance holes: times `ndfft` (c128 rows, ~2^23 points per call) for every n in a range and prints the fraction of 8 TB/s and
the path, worst first.  hole_scan.py <lo> <hi> [step] [op: ndfft | nddct2 | r2c] [f32]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch

import synth
from ndrustfft_amd import DctHandler, FftHandler, R2cFftHandler, _lib, nddct2, ndfft, ndfft_r2c

lo, hi = int(sys.argv[1]), int(sys.argv[2])
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
op = sys.argv[4] if len(sys.argv) > 4 else "ndfft"
F32 = len(sys.argv) > 5 and sys.argv[5] == "f32"
RDT, CDT, TC = (np.float32, np.complex64, torch.complex64) if F32 else (np.float64, np.complex128, torch.complex128)
dev = torch.device("cuda:0")
PTS = 1 << 23
base_c = torch.from_numpy(synth.complex_array((PTS,), CDT)).to(dev)
base_r = torch.from_numpy(synth.real_array((PTS,), RDT)).to(dev)
res = []
for n in range(lo, hi + 1, step):
    rows = PTS // n
    if op == "ndfft":
        x = base_c[: rows * n].view(rows, n); y = torch.empty_like(x); h = FftHandler(n, RDT); fn = lambda: ndfft(x, y, h, 1)
    elif op == "nddct2":
        x = base_r[: rows * n].view(rows, n); y = torch.empty_like(x); h = DctHandler(n, RDT); fn = lambda: nddct2(x, y, h, 1)
    else:
        x = base_r[: rows * n].view(rows, n); y = torch.empty((rows, n // 2 + 1), dtype=TC, device=dev); h = R2cFftHandler(n, RDT); fn = lambda: ndfft_r2c(x, y, h, 1)
    fn(); fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 8
    nbytes = x.numel() * x.element_size() + y.numel() * y.element_size()
    res.append((nbytes / t / 8e12, n, t * 1e6, _lib.default().last_path()))
res.sort()
for fr, n, us, path in res:
    print(json.dumps({"n": n, "frac": round(fr, 3), "us": round(us, 1), "path": path}))
