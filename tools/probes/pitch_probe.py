"""Is the column four-step sensitive to a power-of-two row pitch (HBM channel camping)?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, R2cFftHandler, ndfft, ndfft_r2c, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
h = R2cFftHandler(8192, np.float32)
for w in (8192, 8192 + 32, 8192 + 64, 8192 + 96, 8192 + 128, 8000, 9000):
    x = torch.rand((8192, w), device=dev, dtype=torch.float32); y = torch.empty((4097, w), device=dev, dtype=torch.complex64)
    s = timeit(lambda: ndfft_r2c(x, y, h, 0), 40)
    nb = x.numel() * 4 + y.numel() * 8
    print(f"r2c axis0 8192x{w}: {s*1e6:7.1f} us  {nb/s/1e9:7.0f} GB/s  {_lib.default().last_path()}", flush=True)
hc = FftHandler(4096)
for w in (4096, 4096 + 16, 4096 + 48, 4000):
    x = torch.randn((4096, w), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
    s = timeit(lambda: ndfft(x, y, hc, 0), 40)
    nb = 2 * x.numel() * 16
    print(f"c2c axis0 4096x{w} c128: {s*1e6:7.1f} us  {nb/s/1e9:7.0f} GB/s  {_lib.default().last_path()}", flush=True)
