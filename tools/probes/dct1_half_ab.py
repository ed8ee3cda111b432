"""nddct1 with F = n - 1 prime: the half-length Rader convolution (RaderCfg::half) on / off (developer build: NDFFT_RADER_HALF), 2^24 points per call, re-read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth
from ndrustfft_amd import DctHandler, nddct1, _lib
from bench_configs import timeit
dev = torch.device("cuda:0")
for n in [int(v) for v in sys.argv[1:]] or (128, 258, 1010, 8192, 98):
    for rdt in (np.float64, np.float32):
        x = torch.from_numpy(synth.real_array(((1 << 24) // n, n), rdt)).to(dev); y = torch.empty_like(x)
        for half in ("1", "0", "1", "0"):
            os.environ["NDFFT_RADER_HALF"] = half
            h = DctHandler(n, rdt)
            nddct1(x, y, h, 1); torch.cuda.synchronize()
            t = timeit(lambda: nddct1(x, y, h, 1), 30, ramp_ms=100)
            print(f"n={n} {np.dtype(rdt).name} half={half}: {t*1e6:8.1f} us  {_lib.default().last_path()}", flush=True)
