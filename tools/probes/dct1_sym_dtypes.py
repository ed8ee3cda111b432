import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch, synth
from ndrustfft_amd import DctHandler, nddct1, _lib
from bench_configs import timeit
dev = torch.device("cuda:0")
for n in [int(v) for v in sys.argv[1:]] or (512, 220, 514):
    for rdt in (np.float32, np.float64):
        x = torch.from_numpy(synth.real_array(((1 << 25) // n, n), rdt)).to(dev); y = torch.empty_like(x)
        for sym in ("1", "0", "1", "0"):
            os.environ["NDFFT_RADER_SYM"] = sym
            h = DctHandler(n, rdt)
            nddct1(x, y, h, 1); torch.cuda.synchronize()
            t = timeit(lambda: nddct1(x, y, h, 1), 30, ramp_ms=100)
            print(f"n={n} {np.dtype(rdt).name} sym={sym}: {t*1e6:8.1f} us  {_lib.default().last_path()}", flush=True)
