import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from ndrustfft_amd import FftHandler, R2cFftHandler, DctHandler, ndfft, ndfft_r2c, nddct2, _lib
from bench_configs import timeit
dev = torch.device("cuda", 0)
for n, cdt, rdt in ((10000, torch.complex128, np.float64), (12000, torch.complex128, np.float64), (18000, torch.complex128, np.float64),
                    (15000, torch.complex64, np.float32), (30000, torch.complex64, np.float32), (24576, torch.complex64, np.float32)):
    rows = (1 << 24) // n
    x = torch.randn((rows, n), device=dev, dtype=cdt); y = torch.empty_like(x)
    h = FftHandler(n, rdt)
    ndfft(x, y, h, 1); torch.cuda.synchronize()
    ref = torch.fft.fft(x.to(torch.complex128), dim=1)
    err = float((y - ref).abs().max() / ref.abs().max())
    t = timeit(lambda: ndfft(x, y, h, 1), 20)
    nb = 2 * x.numel() * x.element_size()
    print(f"ndfft {rows}x{n} {cdt}: {_lib.default().last_path()} err {err:.1e} {t*1e6:.0f} us {nb/t/8e12*100:.1f}%", flush=True)
for n, rdt, tdt in ((12000, np.float64, torch.float64), (20000, np.float32, torch.float32)):
    rows = (1 << 24) // n
    x = torch.rand((rows, n), device=dev, dtype=tdt); y = torch.empty_like(x)
    h = DctHandler(n, rdt)
    nddct2(x, y, h, 1); torch.cuda.synchronize()
    t = timeit(lambda: nddct2(x, y, h, 1), 20)
    nb = 2 * x.numel() * x.element_size()
    print(f"nddct2 {rows}x{n} {tdt}: {_lib.default().last_path()} {t*1e6:.0f} us {nb/t/8e12*100:.1f}%", flush=True)
