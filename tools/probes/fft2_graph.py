"""fft2 (ndfft axis 1 into a work array, then axis 0: examples/fft2.rs:23-27) on device-resident arrays: eager calls vs the same two calls replayed from a HIP graph"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, _lib, ndfft
dev = torch.device("cuda:0")
for n, cdt, rdt in ((512, torch.complex128, np.float64), (1024, torch.complex128, np.float64), (2048, torch.complex128, np.float64), (4096, torch.complex128, np.float64), (8192, torch.complex64, np.float32)):
    x = torch.randn((n, n), dtype=cdt, device=dev); w = torch.empty_like(x); y = torch.empty_like(x)
    h = FftHandler(n, rdt)
    def fft2():
        ndfft(x, w, h, 1); ndfft(w, y, h, 0)
    fft2(); torch.cuda.synchronize()
    ref = torch.fft.fft2(x)
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    def timed(fn, k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / k
    timed(fft2, 50)
    eager = sorted(timed(fft2, 100) for _ in range(5))[2]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fft2()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fft2()
    torch.cuda.synchronize()
    timed(g.replay, 5)
    graph = sorted(timed(g.replay, 20) for _ in range(5))[2] / 10
    nbytes = 4 * x.numel() * x.element_size()
    print(f"fft2 {n}x{n} {str(cdt).replace('torch.', '')}: eager {eager:7.1f} us ({nbytes / eager / 8e6:.2f} of 8 TB/s), graph replay {graph:7.1f} us ({nbytes / graph / 8e6:.2f}), rel err vs torch.fft {err:.1e}", flush=True)
