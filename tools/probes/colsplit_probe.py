"""Probe: how fast are the existing column kernels on the two passes a column four-step would run?"""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, R2cFftHandler, ndfft, ndfft_r2c, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
def t(name, fn, x, y, h, axis):
    s = timeit(lambda: fn(x, y, h, axis), 50)
    nb = x.numel() * x.element_size() + y.numel() * y.element_size()
    print(f"{name:60s} {s*1e6:8.1f} us  {nb/s/1e9:7.0f} GB/s {nb/s/8e12*100:5.1f}%  {_lib.default().last_path()}", flush=True)
W = 8192
for n1, n2 in ((128, 64), (64, 128), (256, 32)):
    # R2C split of n = 8192 real
    x = torch.rand((n1, n2 * W), device=dev, dtype=torch.float32); y = torch.empty((n1 // 2 + 1, n2 * W), device=dev, dtype=torch.complex64)
    t(f"pass1 r2c axis0 ({n1},{n2}x{W}) f32", ndfft_r2c, x, y, R2cFftHandler(n1, np.float32), 0)
    x = torch.randn((n1 // 2 + 1, n2, W), device=dev, dtype=torch.complex64); y = torch.empty_like(x)
    t(f"pass2 c2c axis1 ({n1//2+1},{n2},{W}) c64", ndfft, x, y, FftHandler(n2, np.float32), 1)
    # C2C split of n = 8192 complex
    x = torch.randn((n1, n2 * W), device=dev, dtype=torch.complex64); y = torch.empty_like(x)
    t(f"pass1 c2c axis0 ({n1},{n2}x{W}) c64", ndfft, x, y, FftHandler(n1, np.float32), 0)
    x = torch.randn((n1, n2, W), device=dev, dtype=torch.complex64); y = torch.empty_like(x)
    t(f"pass2 c2c axis1 ({n1},{n2},{W}) c64", ndfft, x, y, FftHandler(n2, np.float32), 1)
# f64 4096 columns of a 4096x4096 c128
for n1, n2 in ((64, 64),):
    x = torch.randn((n1, n2 * 4096), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
    t(f"pass1 c2c axis0 ({n1},{n2}x4096) c128", ndfft, x, y, FftHandler(n1), 0)
    x = torch.randn((n1, n2, 4096), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
    t(f"pass2 c2c axis1 ({n1},{n2},4096) c128", ndfft, x, y, FftHandler(n2), 1)
