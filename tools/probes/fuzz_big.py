"""Randomized parity run with larger arrays (2^21 points): the specialised, column four-step and Bluestein paths."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_suite as ps
from ndrustfft_amd import _lib
L = _lib.default()
tot = {}
for seed in (11, 12):
    p = ps.fuzz(L, seed=seed, count=150, max_points=1 << 21,
                lengths=(60, 96, 100, 127, 128, 210, 256, 264, 385, 500, 511, 512, 840, 1000, 1009, 1024, 2000, 2048, 2520, 3003, 4096, 4099,
                         6000, 8192, 10000, 16384))
    for k, v in p.items(): tot[k] = tot.get(k, 0) + v
    print("seed", seed, "ok", flush=True)
print(sorted(tot.items(), key=lambda kv: -kv[1]))
