#!/usr/bin/env python3
"""Extended random parity run on an MI355X (beyond the fixed seeds of tests/test_gpu_parity.py): the parity suite's fuzz generator with fresh seeds,
small and large calls (host arrays: the small-call path, the plain path and the chunk pipeline all occur).  Exit 1 on the first mismatch.
    python tools/fuzz_extended.py [first_seed=1001] [n_seeds=6] [cases=400]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_suite as ps
from ndrustfft_amd import _lib

L = _lib.default()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1001
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 400
tot = collections.Counter()
t0 = time.time()
for s in range(first, first + nseeds):
    big = (s - first) % 3 == 2
    paths = ps.fuzz(L, seed=s, count=cases // 4 if big else cases, max_points=(1 << 21) if big else (1 << 17))
    tot.update(paths if isinstance(paths, dict) else {})
    print(f"seed {s}: ok ({'2^21' if big else '2^17'} points max), {time.time() - t0:.0f} s", flush=True)
print("paths seen:", dict(sorted(tot.items(), key=lambda kv: -kv[1])))
