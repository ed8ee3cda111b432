import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import synth
from ndrustfft_amd import FftHandler, ndfft, _lib
for n in (10000, 18000, 19200):
    rows = (1 << 17) // n + 9
    x = synth.complex_array((rows, n)); y = np.zeros_like(x)
    ndfft(x, y, FftHandler(n), 1)
    print(n, _lib.default().last_path(), np.abs(y - np.fft.fft(x, axis=1)).max() / np.abs(y).max())
