"""hiprtc column tiles (smooth non-power-of-two lengths): lanes per tile (NDFFT_JIT_COL_LPB, read per launch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import DctHandler, R2cFftHandler, FftHandler, _lib, nddct2, ndfft_r2c, ndifft_r2c, ndfft
dev = torch.device("cuda:0")
def t(fn, *a, steps=20):
    for _ in range(5): fn(*a)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
    tr = torch.from_numpy(np.zeros(1, rdt)).dtype; tc = torch.from_numpy(np.zeros(1, cdt)).dtype
    for n in (600, 1000, 1500, 2000):
        cols = (1 << 24) // n // 16 * 16
        xc = torch.randn((n, cols), dtype=tc, device=dev); yc = torch.empty_like(xc)
        xr = torch.randn((n, cols), dtype=tr, device=dev); yr = torch.empty_like(xr)
        xh = torch.randn((n // 2 + 1, cols), dtype=tc, device=dev)
        for name, fn, a, b, h in (("ndfft", ndfft, xc, yc, FftHandler(n, rdt)), ("nddct2", nddct2, xr, yr, DctHandler(n, rdt)), ("ndfft_r2c", ndfft_r2c, xr, xh, R2cFftHandler(n, rdt)),
                                  ("ndifft_r2c", ndifft_r2c, xh, yr, R2cFftHandler(n, rdt))):
            res = []
            for lpb in os.environ.get("LPBS", "0,4,8,16").split(","):
                if lpb == "0": os.environ.pop("NDFFT_JIT_COL_LPB", None)
                else: os.environ["NDFFT_JIT_COL_LPB"] = lpb
                try: us = t(fn, a, b, h, 0); res.append(f"{lpb}:{us:7.1f}({_lib.default().last_path()})")
                except Exception as e: res.append(f"{lpb}: fail")
            nb = a.numel() * a.element_size() + b.numel() * b.element_size()
            print(f"{name:10s} axis=0 {n}x{cols} {np.dtype(rdt).name} [{nb >> 20} MiB]: " + "  ".join(res), flush=True)
