import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle_ctypes as orc
import synth
print("threads", orc.num_threads(), "sched_getaffinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
rows, n = 1024, 4096
x = synth.complex_array((rows, n)); y = np.zeros_like(x); h = orc.FftHandler(n)
for fn, nm in ((orc.ndfft, "serial"), (orc.ndfft_par, "par")):
    fn(x, y, h, 1)
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 2.0:
        fn(x, y, h, 1); reps += 1
    el = time.perf_counter() - t0
    print(nm, f"{rows*n*reps/el/1e9:.3f} GFFT-pts/s", f"{el/reps/rows*1e6:.1f} us per lane-call (wall/rows)")
