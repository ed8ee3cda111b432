import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import R2cFftHandler, ndfft_r2c, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
for tdt, cdt, rdt, tot in ((torch.float32, torch.complex64, np.float32, 1 << 25), (torch.float64, torch.complex128, np.float64, 1 << 24)):
    for n in (128, 256, 512, 1024, 2048, 4096, 8192):
        rows = tot // n
        x = torch.rand((rows, n), device=dev, dtype=tdt); y = torch.empty((rows, n // 2 + 1), device=dev, dtype=cdt)
        h = R2cFftHandler(n, rdt)
        s = timeit(lambda: ndfft_r2c(x, y, h, 1), 40)
        nb = x.numel() * x.element_size() + y.numel() * y.element_size()
        print(f"r2c {rows}x{n} {rdt.__name__:8s} {s*1e6:8.1f} us {nb/s/8e12*100:5.1f}%", flush=True)
