"""Host-side cost of one nd* call on device arrays (Python + ctypes + C ABI + launch), tiny problem."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ndrustfft_amd import FftHandler, ndfft
dev = torch.device("cuda", 0)
x = torch.randn((4, 64), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
h = FftHandler(64)
for _ in range(200): ndfft(x, y, h, 1)
torch.cuda.synchronize()
N = 5000
t0 = time.perf_counter()
for _ in range(N): ndfft(x, y, h, 1)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time per call: {(t1-t0)/N*1e6:.1f} us; incl. drain: {(t2-t0)/N*1e6:.1f} us")
