#!/usr/bin/env python3
"""Column tiles of short f32 C2C lanes: the rows' re-planned recipe against the default one (jit.hip: jit_choose_col), on every length < 256 where
the two differ.  One process = one setting (developer knobs are read once): run with a DEV=1 build,

    NDFFT_MI355X_LIB=$PWD/ndrustfft_amd/csrc/libndfft_mi355x_dev.so NDFFT_JIT_COL_ALT=0 python tools/sweep_col_recipes.py
    NDFFT_MI355X_LIB=$PWD/ndrustfft_amd/csrc/libndfft_mi355x_dev.so NDFFT_JIT_COL_ALT=1 python tools/sweep_col_recipes.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import synth
from bench_configs import timeit
from ndrustfft_amd import FftHandler, _lib, ndfft

L = _lib.default()
dev = torch.device("cuda:0")
tag = os.environ.get("NDFFT_JIT_COL_ALT", "default")
lengths = [int(a) for a in sys.argv[1:]] or [n for n in range(97, 256) if "col_tpl" in L.explain_plan(0, 0, n)] or [98, 99, 100, 110, 121, 126, 132, 135, 140, 143, 144, 147, 154, 156, 160, 162, 176, 189, 192, 220, 225, 231, 242]
for n in lengths:
    for inner in (2048, 64):
        outer = max(1, (1 << 24) // (n * inner))
        x = torch.from_numpy(synth.complex_array((outer, n, inner), np.complex64)).to(dev); y = torch.empty_like(x)
        h = FftHandler(n, np.float32)
        t = timeit(lambda: ndfft(x, y, h, 1), 40)
        print(json.dumps({"alt": tag, "n": n, "shape": [outer, n, inner], "us": round(t * 1e6, 2), "path": L.last_path(), "recipe": L.explain_plan(0, 0, n).strip()[20:]}), flush=True)
