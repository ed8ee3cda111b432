"""kernel-level view of the real four-step: nddct2 / ndfft_r2c on 64 x 262144 f64 and f32 (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import DctHandler, R2cFftHandler, nddct2, nddct3, ndfft_r2c, ndifft_r2c
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["nddct2", "ndfft_r2c"]
for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
    tr = torch.from_numpy(np.zeros(1, rdt)).dtype; tc = torch.from_numpy(np.zeros(1, cdt)).dtype
    L, n = 64, 1 << 18
    x = torch.randn((L, n), dtype=tr, device=dev); y = torch.empty_like(x)
    xh = torch.randn((L, n // 2 + 1), dtype=tc, device=dev)
    hd = DctHandler(n, rdt); hr = R2cFftHandler(n, rdt)
    for name, fn, a, b, h in (("nddct2", nddct2, x, y, hd), ("nddct3", nddct3, x, y, hd), ("ndfft_r2c", ndfft_r2c, x, xh, hr), ("ndifft_r2c", ndifft_r2c, xh, y, hr)):
        if name in which:
            for _ in range(20): fn(a, b, h, 1)
    torch.cuda.synchronize()
