"""Longer randomized parity run than the test suite's (several seeds)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_suite as ps
from ndrustfft_amd import _lib
L = _lib.default()
tot = {}
for seed in (101, 202, 303, 404):
    p = ps.fuzz(L, seed=seed, count=400, lengths=(1, 2, 3, 5, 8, 11, 15, 16, 20, 27, 36, 49, 60, 64, 81, 100, 121, 125, 128, 144, 169, 180, 210, 240,
                                                256, 289, 320, 360, 385, 420, 511, 512, 540, 625, 720, 768, 840, 1000, 1023, 1024, 1155, 1331, 1536,
                                                2000, 2047, 2048, 2310, 2520, 3125, 4095, 4096, 4098, 6000, 6561, 8191, 8192, 8193, 10007, 12000,
                                                16384, 16385, 20000, 32768, 40000, 65536))
    for k, v in p.items(): tot[k] = tot.get(k, 0) + v
    print("seed", seed, "ok", flush=True)
print(sorted(tot.items(), key=lambda kv: -kv[1]))
