"""Which long-column cases are still on the narrow XCD tiles, and how fast are they?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, R2cFftHandler, DctHandler, ndfft, ndfft_r2c, nddct2, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
def t(name, fn, x, y, h, axis):
    s = timeit(lambda: fn(x, y, h, axis), 40)
    nb = x.numel() * x.element_size() + y.numel() * y.element_size()
    print(f"{name:52s} {s*1e6:8.1f} us {nb/s/8e12*100:5.1f}%  {_lib.default().last_path()}", flush=True)
for n, cdt, rdt in ((2048, torch.complex128, np.float64), (2048, torch.complex64, np.float32)):
    x = torch.randn((n, 8192), device=dev, dtype=cdt); y = torch.empty_like(x)
    t(f"ndfft axis=0 ({n},8192) {cdt}", ndfft, x, y, FftHandler(n, rdt), 0)
for n, tdt, cdt, rdt in ((4096, torch.float32, torch.complex64, np.float32), (4096, torch.float64, torch.complex128, np.float64), (8192, torch.float64, torch.complex128, np.float64)):
    x = torch.rand((n, 4096), device=dev, dtype=tdt); y = torch.empty((n // 2 + 1, 4096), device=dev, dtype=cdt)
    t(f"ndfft_r2c axis=0 ({n},4096) {tdt}", ndfft_r2c, x, y, R2cFftHandler(n, rdt), 0)
for n, tdt, rdt in ((4096, torch.float64, np.float64), (8192, torch.float32, np.float32), (2048, torch.float64, np.float64)):
    x = torch.rand((n, 4096), device=dev, dtype=tdt); y = torch.empty_like(x)
    t(f"nddct2 axis=0 ({n},4096) {tdt}", nddct2, x, y, DctHandler(n, rdt), 0)
