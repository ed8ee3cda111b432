"""Is the output of a pass (written with nt stores) cold for the next pass?  y = ndfft(x); z = ndifft(y) on 4096 x 4096 c128 (and 1024 x 4096),
per-pair time with the input hint AUTO (the residency model: streaming loads for y), CACHED (plain loads) and COLD, A-B-A-B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, _lib, ndfft, ndifft
L = _lib.default(); dev = torch.device("cuda:0")
for rows in (4096, 1024, 16384):
    n = 4096
    x = torch.randn((rows, n), dtype=torch.complex128, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    h = FftHandler(n)
    def pair():
        ndfft(x, y, h, 1); ndifft(y, z, h, 1)
    def run(steps=100):
        for _ in range(20): pair()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps): pair()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps
    for rep in range(2):
        res = []
        for name, hint in (("AUTO", _lib.INPUT_AUTO), ("CACHED", _lib.INPUT_CACHED), ("COLD", _lib.INPUT_COLD)):
            L.check(L.c.ndfft_set_input_hint(hint)); res.append(f"{name} {run():.1f} us")
        print(f"{rows}x{n} c128 fft->ifft pair:", ", ".join(res), flush=True)
    L.c.ndfft_set_input_hint(_lib.INPUT_AUTO)
