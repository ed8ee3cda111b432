#!/usr/bin/env python3
"""Generate tests/golden/numpy_scipy_vectors.npz.

The reference (Rust) cannot run in the build container, and its in-tree known answers are
numpy/scipy values rounded to 3-5 decimals (tests/golden/reference_vectors.json).  This script
regenerates full-precision vectors *of the same definitions* with numpy (pocketfft) / scipy, on
seeded inputs, for the sizes SURVEY.md section 7 step 1 lists.  numpy/scipy are third-party
truth sources, not reference files; everything is computed in float64 regardless of the case's
dtype (an f32 case stores the f32-rounded input and the f64 answer for that rounded input).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.fft as sf

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import synth  # noqa: E402

SIZES = list(range(1, 18)) + [24, 30, 37, 64, 100, 128, 129, 264, 265, 512, 513]
LANES = 2


def main():
    out = {}
    for dt_name, rdt, cdt in (("f64", np.float64, np.complex128), ("f32", np.float32, np.complex64)):
        for n in SIZES:
            m = n // 2 + 1
            xc = synth.complex_array((LANES, n), cdt, offset=1000 * n)
            xr = synth.real_array((LANES, n), rdt, offset=7000 * n)
            xh = synth.complex_array((LANES, m), cdt, offset=13000 * n)
            xc64, xr64, xh64 = xc.astype(np.complex128), xr.astype(np.float64), xh.astype(np.complex128)
            key = f"{dt_name}_n{n}"
            out[key + "_c_in"] = xc
            out[key + "_r_in"] = xr
            out[key + "_h_in"] = xh
            # ndfft: unnormalised forward (src/lib.rs:313-318)
            out[key + "_fft"] = np.fft.fft(xc64, axis=1)
            # ndifft, Default: 1/n after (src/lib.rs:321-338)
            out[key + "_ifft"] = np.fft.ifft(xc64, axis=1)
            # ndfft_r2c (src/lib.rs:497-503)
            out[key + "_r2c"] = np.fft.rfft(xr64, axis=1)
            # ndifft_r2c, Default (src/lib.rs:506-531): DC (and even-n Nyquist) imag ignored, 1/n
            out[key + "_c2r"] = np.fft.irfft(xh64, n=n, axis=1)
            # nddctK, Default = x2 pre-scale = scipy's unnormalised dct (src/lib.rs:688-741)
            for k in (1, 2, 3, 4):
                if k == 1 and n < 2:
                    continue
                out[key + f"_dct{k}"] = sf.dct(xr64, type=k, axis=1)
    path = os.path.join(os.path.dirname(__file__), "numpy_scipy_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; numpy", np.__version__)


if __name__ == "__main__":
    main()
