#!/usr/bin/env python3
"""Reproducer (MI355X, ROCm 7.2): exec-masked global loads into a register array that spills to AGPRs lose their values.

reg_kernel.h's RegReal keeps a whole real lane, its inner FFT and its outputs in one thread's registers (n = 40 / 48 in f64: 256 VGPRs plus AGPR
spill, no scratch).  Its dense-row form stages a workgroup's lanes through LDS with NI coalesced loads per thread; in the LAST workgroup some of
those positions lie past the array.  Written as predicated loads (`if (g <= last) raw[k] = in[g]`) every lane of that workgroup came out wrong;
written with clamped addresses (every thread executes every load) the results are right.  The product uses the clamp; this script builds BOTH
forms with hiprtc and compares each with numpy / scipy.  The predicated form is compiled only by a DEVELOPER build of the library (round 4):

    make -C ndrustfft_amd/csrc DEV=1                                  # -> libndfft_mi355x_dev.so (developer knobs become environment values)
    python tools/repro_masked_tail.py                                 # product form (clamped)
    NDFFT_MI355X_LIB=ndrustfft_amd/csrc/libndfft_mi355x_dev.so NDFFT_REPRO_MASKED_TAIL=1 python tools/repro_masked_tail.py     # predicated form

tests/test_gpu_parity.py::test_regreal_tail_workgroup checks the product form on the same shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scipy.fft as sf


def run_case(n, lanes, masked):
    """max relative error per workgroup of 256 / 128 / 64 lanes (whatever the kernel chose) of nddct2 on `lanes` dense rows of n points, f64"""
    import torch
    from ndrustfft_amd import DctHandler, _lib, nddct2
    # the predicated form exists only in a developer build of the library (make -C ndrustfft_amd/csrc DEV=1 -> libndfft_mi355x_dev.so,
    # NDFFT_MI355X_LIB points ndrustfft_amd at it); developer knobs are read once, so one process = one form
    if masked: assert os.environ.get("NDFFT_REPRO_MASKED_TAIL") == "1", "run a separate process with NDFFT_REPRO_MASKED_TAIL=1 on a DEV=1 build"
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (lanes, n))
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros_like(xd)
    nddct2(xd, yd, DctHandler(n), 1); torch.cuda.synchronize()
    path = _lib.default().last_path()
    ref = sf.dct(x, type=2, axis=1)
    err = np.abs(yd.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max()
    return path, err


def main():
    masked_build = os.environ.get("NDFFT_REPRO_MASKED_TAIL") == "1"    # honoured only by a DEV=1 build of the library (developer knobs are read once)
    for n in (40, 48, 24):
        for masked in ((True,) if masked_build else (False,)):
            lanes = 65536 // n * 2 + 37                      # >= 2^16 points (the kernel is only specialised for real work) and a partial last workgroup
            path, err = run_case(n, lanes, masked)
            bad = np.nonzero(err > 1e-10)[0]
            print(f"nddct2 f64 n={n} lanes={lanes} path={path} tail loads {'PREDICATED' if masked else 'clamped   '}: "
                  f"{len(bad)} wrong lanes" + (f" (first {bad[0]}, last {bad[-1]}, max rel err {err.max():.2e})" if len(bad) else f" (max rel err {err.max():.1e})"))


if __name__ == "__main__":
    main()
