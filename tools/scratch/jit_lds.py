import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, ndfft, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
x = torch.randn((1000, 16384), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
h = FftHandler(1000)
ndfft(x, y, h, 0); torch.cuda.synchronize()
ref = torch.fft.fft(x, dim=0)
print("path", _lib.default().last_path(), "err", float((y - ref).abs().max() / ref.abs().max()))
t = timeit(lambda: ndfft(x, y, h, 0), 50)
print("us", t * 1e6, "GB/s", 2 * x.numel() * 16 / t / 1e9)
