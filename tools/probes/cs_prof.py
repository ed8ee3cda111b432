import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ndrustfft_amd import FftHandler, R2cFftHandler, ndfft, ndifft, ndfft_r2c, ndifft_r2c
dev = torch.device("cuda", 0)
which = sys.argv[1]
if which == "r2c":
    x = torch.rand((8192, 8192), device=dev, dtype=torch.float32); y = torch.empty((4097, 8192), device=dev, dtype=torch.complex64)
    h = R2cFftHandler(8192, np.float32)
    for _ in range(300): ndfft_r2c(x, y, h, 0)
    torch.cuda.synchronize()
    for _ in range(300): ndifft_r2c(y, x, h, 0)
elif which == "c128":
    x = torch.randn((4096, 4096), device=dev, dtype=torch.complex128); y = torch.empty_like(x)
    h = FftHandler(4096)
    for _ in range(300): ndfft(x, y, h, 0)
elif which == "c64":
    x = torch.randn((8192, 8192), device=dev, dtype=torch.complex64); y = torch.empty_like(x)
    h = FftHandler(8192, np.float32)
    for _ in range(300): ndfft(x, y, h, 0)
torch.cuda.synchronize()
