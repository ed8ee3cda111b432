"""Runs one transform on the GPU and prints the path that served it (debug helper): one_case.py <name> <n> <rows> [f32]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np

import parity_suite as ps
from ndrustfft_amd import _lib

name, n, rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rdt = np.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else np.float64
L = _lib.default()
print(name, n, rows, rdt.__name__, "->", ps.run_case(L, name, (rows, n), 1, rdt))
