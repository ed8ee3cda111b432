"""Specialised f32 C2C row kernels at eight smooth lengths: time, fraction of 8 TB/s and the kernel path (the half vs whole-complex
LDS exchange A/B this script was written for is settled: the knob is gone, the measured setting is the code)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, ndfft, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
for n in (96, 500, 1000, 1536, 3000, 6000, 10000, 12288):
    rows = (1 << 25) // n
    x = torch.randn((rows, n), device=dev, dtype=torch.complex64); y = torch.empty_like(x)
    h = FftHandler(n, np.float32)
    s = timeit(lambda: ndfft(x, y, h, 1), 40)
    print(f"c64 {rows}x{n}: {s*1e6:7.1f} us {2*x.numel()*8/s/8e12*100:5.1f}%  {_lib.default().last_path()}", flush=True)
