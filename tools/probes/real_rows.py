"""Row R2C / C2R / DCT kernels on arrays that fit the Infinity Cache (2^25 real points), per n and dtype."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import R2cFftHandler, DctHandler, ndfft_r2c, ndifft_r2c, nddct2, nddct3, nddct4, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
def t(name, fn, x, y, h):
    s = timeit(lambda: fn(x, y, h, 1), 40)
    nb = x.numel() * x.element_size() + y.numel() * y.element_size()
    print(f"{name:40s} {s*1e6:8.1f} us {nb/s/8e12*100:5.1f}%  {_lib.default().last_path()}", flush=True)
for tdt, cdt, rdt, tot in ((torch.float32, torch.complex64, np.float32, 1 << 25), (torch.float64, torch.complex128, np.float64, 1 << 24)):
    for n in (128, 512, 1024, 2048, 4096, 8192, 16384):
        rows = tot // n
        x = torch.rand((rows, n), device=dev, dtype=tdt); y = torch.empty((rows, n // 2 + 1), device=dev, dtype=cdt)
        h = R2cFftHandler(n, rdt)
        t(f"r2c {rows}x{n} {rdt.__name__}", ndfft_r2c, x, y, h)
        t(f"c2r {rows}x{n} {rdt.__name__}", ndifft_r2c, y, x, h)
        if n in (512, 4096):
            z = torch.empty_like(x); hd = DctHandler(n, rdt)
            t(f"dct2 {rows}x{n} {rdt.__name__}", nddct2, x, z, hd)
            t(f"dct3 {rows}x{n} {rdt.__name__}", nddct3, x, z, hd)
            t(f"dct4 {rows}x{n} {rdt.__name__}", nddct4, x, z, hd)
