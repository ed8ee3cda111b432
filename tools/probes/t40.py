import sys, os, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from ndrustfft_amd import FftHandler, ndfft, _lib
for n, rows in ((40, 1675), (40, 2048), (36, 1857), (48, 1402), (45, 1493), (56, 1207)):
    x = synth.complex_array((rows, n)); y = np.zeros_like(x)
    ndfft(x, y, FftHandler(n), 1)
    ref = np.fft.fft(x, axis=1)
    err = np.abs(y - ref).max(axis=1) / np.abs(ref).max()
    bad = np.nonzero(err > 1e-10)[0]
    print(n, rows, _lib.default().last_path(), "bad lanes:", len(bad), bad[:12], "first bad lane cols:", (np.nonzero(np.abs(y[bad[0]] - ref[bad[0]]) > 1e-9)[0][:12] if len(bad) else ""))
