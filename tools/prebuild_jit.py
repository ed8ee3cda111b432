#!/usr/bin/env python3
"""Runs every nd* call of the reference's own test / bench / example lengths once, so that the library compiles (hiprtc) the specialised kernels those lengths
ask for -- which kernels they are is an exec-time choice, hence an MI355X is needed -- and records the kernels' source texts in a MANIFEST:

    NDFFT_JIT_DUMP_SRC=gpurun_out/jit_manifest.txt NDFFT_JIT_CACHE=gpurun_out/jit_prebuilt NDFFT_JIT_PREBUILT=0 python tools/prebuild_jit.py

then copy the manifest to ndrustfft_amd/csrc/jit_prebuilt/manifest.txt (tracked).  The code objects themselves are BUILD products since round 6:
__graft_entry__.build() compiles every manifest entry with hiprtc (ndfft_jit_prebuild, no GPU needed) into ndrustfft_amd/csrc/jit_prebuilt/ (looked up read-only by
jit.hip after the user's cache), under names that hash the CURRENT kernel headers -- an edit of a kernel header no longer leaves stale objects behind.
Lengths: 128 / 264 / 512 / 1024 (benches/ndrustfft.rs:6), 129 / 265 / 513 / 1025 (:7), 3 / 6 (tests, examples); ops: ndfft, ndfft_r2c (+ inverses),
nddct1..4; axis 0 and axis 1; the bench shapes n x n and a large batch (4096 lanes), f64 and f32.  Power-of-two inner lengths and n <= 16 run
ahead-of-time kernels and need nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from ndrustfft_amd import DctHandler, FftHandler, R2cFftHandler, _lib, nddct1, nddct2, nddct3, nddct4, ndfft, ndfft_r2c, ndifft, ndifft_r2c

assert os.environ.get("NDFFT_JIT_CACHE"), "set NDFFT_JIT_CACHE to the directory to fill"
os.makedirs(os.environ["NDFFT_JIT_CACHE"], exist_ok=True)
dev = torch.device("cuda:0")
seen = {}
for rdt, cdt in ((np.float64, np.complex128), (np.float32, np.complex64)):
    tr, tc = torch.from_numpy(np.zeros(1, rdt)).dtype, torch.from_numpy(np.zeros(1, cdt)).dtype
    for n in (3, 6, 128, 264, 512, 1024, 129, 265, 513, 1025):
        m = n // 2 + 1
        for batch in (n, 4096):
            for axis in (0, 1):
                shp = lambda k: (k, batch) if axis == 0 else (batch, k)
                xc = torch.randn(shp(n), dtype=tc, device=dev); yc = torch.empty_like(xc)
                xr = torch.randn(shp(n), dtype=tr, device=dev); yr = torch.empty_like(xr)
                xh = torch.empty(shp(m), dtype=tc, device=dev)
                hf, hr, hd = FftHandler(n, rdt), R2cFftHandler(n, rdt), DctHandler(n, rdt)
                for name, fn, a, b, h in (("ndfft", ndfft, xc, yc, hf), ("ndifft", ndifft, xc, yc, hf), ("ndfft_r2c", ndfft_r2c, xr, xh, hr),
                                          ("ndifft_r2c", ndifft_r2c, xh, yr, hr), ("nddct1", nddct1, xr, yr, hd), ("nddct2", nddct2, xr, yr, hd),
                                          ("nddct3", nddct3, xr, yr, hd), ("nddct4", nddct4, xr, yr, hd)):
                    fn(a, b, h, axis)
                    seen[(np.dtype(rdt).name, name, n, batch, axis)] = _lib.default().last_path()
torch.cuda.synchronize()
files = sorted(f for f in os.listdir(os.environ["NDFFT_JIT_CACHE"]) if f.endswith(".hsaco"))
jit = {k: v for k, v in seen.items() if any(t in v for t in ("jit", "reg_", "regreal", "rader", "blue", "plain"))}
print(f"{len(files)} code objects in {os.environ['NDFFT_JIT_CACHE']}; {len(jit)} of {len(seen)} calls ran a specialised kernel")
for k, v in sorted(jit.items()):
    print("  ", *k, "->", v)
