"""long real-data lanes: nddct2 / nddct3 / ndfft_r2c / ndifft_r2c on 64 x 262144 f64 (and f32), timing per call (HIP events)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import DctHandler, R2cFftHandler, FftHandler, _lib, nddct1, nddct2, nddct3, nddct4, ndfft_r2c, ndifft_r2c, ndfft
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
    L, n = 64, 1 << 18
    x = torch.randn((L, n), dtype=tr, device=dev); y = torch.empty_like(x)
    xh = torch.randn((L, n // 2 + 1), dtype=tc, device=dev)
    hd = DctHandler(n, rdt); hr = R2cFftHandler(n, rdt)
    only = os.environ.get("LONG_REAL_ONLY", "").split(",") if os.environ.get("LONG_REAL_ONLY") else None
    for name, fn, a, b, h in (("nddct2", nddct2, x, y, hd), ("nddct3", nddct3, x, y, hd), ("nddct4", nddct4, x, y, hd), ("ndfft_r2c", ndfft_r2c, x, xh, hr), ("ndifft_r2c", ndifft_r2c, xh, y, hr)):
        if only and name not in only: continue
        us = t(fn, a, b, h, 1)
        nbytes = a.numel() * a.element_size() + b.numel() * b.element_size()
        print(f"{name:11s} {np.dtype(rdt).name} {L}x{n}: {us:8.1f} us  {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s  path={_lib.default().last_path()}", flush=True)
    if only and "ndfft" not in only: continue
    xc = torch.randn((32, 1 << 20), dtype=tc, device=dev); yc = torch.empty_like(xc)
    us = t(ndfft, xc, yc, FftHandler(1 << 20, rdt), 1)
    print(f"ndfft       {np.dtype(cdt).name} 32x{1 << 20}: {us:8.1f} us  {2 * xc.numel() * xc.element_size() / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
