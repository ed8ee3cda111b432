"""Times one transform on the GPU: one_time.py <name> <n> [f32]   (2^23 points per call, 20 launches)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch

import synth
from ndrustfft_amd import DctHandler, FftHandler, _lib, nddct2, nddct3, ndfft

name, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
rows = (1 << 23) // n
if name == "ndfft":
    F32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
    x = torch.from_numpy(synth.complex_array((rows, n), np.complex64 if F32 else np.complex128)).to(dev); h = FftHandler(n, np.float32 if F32 else np.float64); f = ndfft
else:
    x = torch.from_numpy(synth.real_array((rows, n))).to(dev); h = DctHandler(n); f = {"nddct2": nddct2, "nddct3": nddct3}[name]
y = torch.empty_like(x)
for _ in range(50):
    f(x, y, h, 1)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f(x, y, h, 1)
e1.record(); torch.cuda.synchronize()
print(name, n, round(e0.elapsed_time(e1) * 1e3 / 20, 1), "us", _lib.default().last_path())
