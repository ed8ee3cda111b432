"""nddct1 long lanes (n - 1 a power of two): the real four-step on the even extension (round 5) against the packed route (NDFFT_REAL_FOURSTEP=0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth
from ndrustfft_amd import DctHandler, nddct1, _lib
from bench_configs import timeit, set_switch
dev = torch.device("cuda:0")
for rdt in (np.float64, np.float32):
    for e in (16, 17, 18, 20):
        n = (1 << e) + 1
        rows = max(2, (1 << 24) // n)
        x = torch.from_numpy(synth.real_array((rows, n), rdt)).to(dev); y = torch.empty_like(x)
        for v in ("1", "0", "1", "0"):
            set_switch("NDFFT_REAL_FOURSTEP", v)
            h = DctHandler(n, rdt)
            nddct1(x, y, h, 1); torch.cuda.synchronize()
            t = timeit(lambda: nddct1(x, y, h, 1), 20, ramp_ms=100)
            nb = 2 * x.numel() * x.element_size()
            print(f"nddct1 {rows}x{n} {np.dtype(rdt).name} REAL_FOURSTEP={v}: {t*1e6:8.1f} us  {nb/t/8e12:.3f}  {_lib.default().last_path()}", flush=True)
set_switch("NDFFT_REAL_FOURSTEP", None)
