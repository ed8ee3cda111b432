"""Randomized parity run aimed at the routes round 3 changed late: column tiles of 4 / 8 / 16 lanes, column four-step from shorter lanes, hiprtc 4-lane tiles,
strided long DCT lanes, the real four-step (both directions, DCT-IV) with random lane counts, offsets and pitches.  Every case is checked against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_suite as ps
from ndrustfft_amd import _lib
L = _lib.default()
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "7")))
names = ["ndfft", "ndifft", "ndfft_r2c", "ndifft_r2c", "nddct1", "nddct2", "nddct3", "nddct4"]
paths = {}
t0 = time.time()
count = int(os.environ.get("FUZZ_COUNT", "260"))
for it in range(count):
    name = names[rng.integers(len(names))]
    rdt = (np.float64, np.float32)[rng.integers(2)]
    if rng.random() < 0.35:      # long contiguous lanes
        n = int(2 ** rng.integers(16, 20)); n += 1 if name == "nddct1" and rng.random() < 0.5 else 0
        shape, axis = (int(rng.integers(1, 9)), n), 1
    else:                        # strided mid-size / long lanes
        n = int(rng.choice([512, 1024, 2048, 4096, 8192, 600, 1000, 1500, 2000, 3000, 16384]))
        if name == "nddct1": n += int(rng.integers(2))
        inner = int(rng.integers(8, 260))
        while n * inner > (1 << 22): inner = max(8, inner // 2)
        if rng.random() < 0.3: shape, axis = (int(rng.integers(2, 4)), n, max(8, inner // 3)), 1
        else: shape, axis = (n, inner), 0
    norm = ("Default", "None")[rng.integers(2)]
    p = ps.run_case(L, name, shape, axis, rdt, norm=norm, offset=int(rng.integers(100)))
    paths[p] = paths.get(p, 0) + 1
print(f"{count} cases ok in {time.time() - t0:.0f} s")
print(sorted(paths.items(), key=lambda kv: -kv[1]))
