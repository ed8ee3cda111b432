"""nddct1 at power-of-two n (F = n - 1: 31, 63, 127, 255, 511, 1023, 2047, 4095, 8191, 16383): which kernel serves it and how well, 2^24 points per call, HBM-sourced / re-read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth
import bench_configs as bc
from ndrustfft_amd import DctHandler, nddct1, nddct2
dev = torch.device("cuda:0")
for rdt in (np.float64, np.float32):
    for e in range(5, 15):
        n = 1 << e
        rows = (1 << 24) // n
        x = torch.from_numpy(synth.real_array((rows, n), rdt)).to(dev); y = torch.empty_like(x)
        bc.run(f"nddct1 axis=1 {rows}x{n} {np.dtype(rdt).name}", nddct1, x, y, DctHandler(n, rdt), 1, x.numel(), 30)
        bc.run(f"nddct2 axis=1 {rows}x{n} {np.dtype(rdt).name}", nddct2, x, y, DctHandler(n, rdt), 1, x.numel(), 30)
