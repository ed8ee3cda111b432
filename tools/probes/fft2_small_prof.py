import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, ndfft
dev = torch.device("cuda:0")
for n in (512, 1024, 2048):
    x = torch.randn((n, n), dtype=torch.complex128, device=dev); w = torch.empty_like(x); y = torch.empty_like(x)
    h = FftHandler(n)
    for _ in range(200):
        ndfft(x, w, h, 1); ndfft(w, y, h, 0)
    torch.cuda.synchronize()
