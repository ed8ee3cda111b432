#!/usr/bin/env python3
"""Generate tests/golden/baseline_lengths.npz -- independent pins at the lane lengths the BASELINE configs run.

make_golden.py stops at n = 513; BASELINE.json's configs transform lanes of 4096 (cfg2 / cfg5, c128), 8192 (cfg3-A f32 R2C,
cfg3-B c64 C2C, and the way back: C2R), 512 (cfg4, f64 DCT-I..IV) and the single-workgroup limit 16384.  At those lengths
the oracle used to be checked by nothing but itself.  This script writes, per case, seeded inputs (tests/synth.py) and THREE
independent truths of the reference's definitions (src/lib.rs:313-318, 321-338, 497-503, 506-531, 688-741):

  *_np    numpy pocketfft / scipy.fft in float64 on every lane (a different FFT implementation),
  *_ld    the DEFINITION summed directly in long double (O(n^2), twiddles e^{-2 pi i jk/n} from mpmath at 40 digits and rounded
          once to long double) on lane 0, rounded to float64,
  *_mp    the definition summed in mpmath (40 digits) for 12 output bins of lane 0 (bins listed in *_mp_bins).

f32 cases store the f32-rounded inputs; the truths are float64 answers for those rounded inputs.  numpy, scipy and mpmath
are third-party packages of the build container, not reference files.  Run from the repo root:
    python tests/golden/make_golden_baseline.py
"""
import os
import sys

import mpmath as mp
import numpy as np
import scipy.fft as sf

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import synth  # noqa: E402

mp.mp.dps = 40
LD = np.longdouble
CLD = np.clongdouble


def unit_roots(n):
    """e^{-2 pi i k/n}, k < n, as long double (from mpmath: exact to the last long-double bit)."""
    re = np.empty(n, LD); im = np.empty(n, LD)
    for k in range(n):
        a = -2 * mp.pi * k / n
        re[k] = LD(mp.nstr(mp.cos(a), 30)); im[k] = LD(mp.nstr(mp.sin(a), 30))
    return re + 1j * im


def dft_ld(x, sign, w):
    """X[k] = sum_j x[j] w^(sign jk), summed in long double, 256 outputs per block."""
    n = x.size
    xl = x.astype(CLD)
    ww = w if sign < 0 else np.conj(w)
    out = np.empty(n, CLD)
    j = np.arange(n, dtype=np.int64)
    for k0 in range(0, n, 256):
        k = np.arange(k0, min(n, k0 + 256), dtype=np.int64)
        out[k0:k0 + k.size] = (ww[(k[:, None] * j[None, :]) % n] * xl[None, :]).sum(axis=1)
    return out


def dft_mp_bins(x, sign, bins):
    n = x.size
    out = []
    for k in bins:
        s = mp.mpc(0)
        for j in range(n):
            s += mp.mpc(float(x[j].real), float(x[j].imag)) * mp.expjpi(mp.mpf(sign * 2 * ((j * int(k)) % n)) / n)
        out.append(complex(s))
    return np.asarray(out, np.complex128)


def hermitian_full(xh, n):
    """The length-n spectrum the reference's C2R sees (src/lib.rs:514-521): imaginary parts of DC (and of Nyquist, n even) dropped."""
    m = n // 2 + 1
    full = np.zeros(n, np.complex128)
    h = xh.astype(np.complex128).copy()
    h[0] = h[0].real
    if n % 2 == 0:
        h[m - 1] = h[m - 1].real
    full[:m] = h
    full[m:] = np.conj(h[1:n - m + 1][::-1])
    return full


def dct_ld(k, x):
    """scipy's unnormalised DCT-I..IV definitions (= the reference's Default normalisation, src/lib.rs:1206, 1257, 1308, 1359) in long double."""
    n = x.size
    xl = x.astype(LD)
    pi = LD(mp.nstr(mp.pi, 30))
    j = np.arange(n, dtype=LD)
    out = np.empty(n, LD)
    for kk in range(n):
        if k == 1:
            out[kk] = xl[0] + (-1) ** kk * xl[n - 1] + 2 * (xl[1:n - 1] * np.cos(pi * j[1:n - 1] * kk / (n - 1))).sum()
        elif k == 2:
            out[kk] = 2 * (xl * np.cos(pi * kk * (2 * j + 1) / (2 * n))).sum()
        elif k == 3:
            out[kk] = xl[0] + 2 * (xl[1:] * np.cos(pi * j[1:] * (2 * kk + 1) / (2 * n))).sum()
        else:
            out[kk] = 2 * (xl * np.cos(pi * (2 * j + 1) * (2 * kk + 1) / (4 * n))).sum()
    return out.astype(np.float64)


def main():
    out = {}
    bins_of = lambda n: np.unique(np.asarray([0, 1, 2, 3, n // 7, n // 3, n // 2 - 1, n // 2, n // 2 + 1, n - 3, n - 2, n - 1]) % n)
    # ---- C2C: cfg2 / cfg5 (4096 c128), cfg3-B (8192 c64), the one-workgroup limit (16384, both dtypes) -----------------
    for n, dt, lanes in ((4096, "f64", 3), (8192, "f32", 2), (8192, "f64", 1), (16384, "f64", 1), (16384, "f32", 1)):
        cdt = np.complex128 if dt == "f64" else np.complex64
        x = synth.complex_array((lanes, n), cdt, offset=31 * n)
        x64 = x.astype(np.complex128)
        w = unit_roots(n)
        key = f"c2c_{dt}_n{n}"
        out[key + "_in"] = x
        out[key + "_fft_np"] = np.fft.fft(x64, axis=1)
        out[key + "_ifft_np"] = np.fft.ifft(x64, axis=1)
        out[key + "_fft_ld"] = dft_ld(x64[0], -1, w).astype(np.complex128)
        out[key + "_ifft_ld"] = (dft_ld(x64[0], +1, w) / n).astype(np.complex128)
        b = bins_of(n)
        out[key + "_mp_bins"] = b
        out[key + "_fft_mp"] = dft_mp_bins(x64[0], -1, b)
        print(key, "pocketfft vs long double:", np.abs(out[key + "_fft_np"][0] - out[key + "_fft_ld"]).max() / np.abs(out[key + "_fft_ld"]).max(),
              " long double vs mpmath bins:", np.abs(out[key + "_fft_ld"][b] - out[key + "_fft_mp"]).max() / np.abs(out[key + "_fft_ld"]).max(), flush=True)
    # ---- R2C / C2R: cfg3-A and its way back (8192 f32); f64 at the same length ------------------------------------------
    for n, dt, lanes in ((8192, "f32", 2), (8192, "f64", 1)):
        rdt = np.float64 if dt == "f64" else np.float32
        cdt = np.complex128 if dt == "f64" else np.complex64
        m = n // 2 + 1
        xr = synth.real_array((lanes, n), rdt, offset=77 * n)
        xh = synth.complex_array((lanes, m), cdt, offset=131 * n)
        w = unit_roots(n)
        key = f"real_{dt}_n{n}"
        out[key + "_r_in"] = xr; out[key + "_h_in"] = xh
        out[key + "_r2c_np"] = np.fft.rfft(xr.astype(np.float64), axis=1)
        out[key + "_c2r_np"] = np.fft.irfft(xh.astype(np.complex128), n=n, axis=1)
        out[key + "_r2c_ld"] = dft_ld(xr[0].astype(np.complex128), -1, w)[:m].astype(np.complex128)
        out[key + "_c2r_ld"] = (dft_ld(hermitian_full(xh[0], n), +1, w) / n).real.astype(np.float64)
        b = bins_of(m)
        out[key + "_mp_bins"] = b
        out[key + "_r2c_mp"] = dft_mp_bins(xr[0].astype(np.complex128), -1, b)
        print(key, "rfft pocketfft vs long double:", np.abs(out[key + "_r2c_np"][0] - out[key + "_r2c_ld"]).max() / np.abs(out[key + "_r2c_ld"]).max(),
              " irfft:", np.abs(out[key + "_c2r_np"][0] - out[key + "_c2r_ld"]).max() / np.abs(out[key + "_c2r_ld"]).max(),
              " mpmath bins:", np.abs(out[key + "_r2c_ld"][b] - out[key + "_r2c_mp"]).max() / np.abs(out[key + "_r2c_ld"]).max(), flush=True)
    # ---- DCT-I..IV: cfg4 (512 f64) -----------------------------------------------------------------------------------------
    n = 512
    xr = synth.real_array((3, n), np.float64, offset=977 * n)
    out["dct_f64_n512_in"] = xr
    for k in (1, 2, 3, 4):
        out[f"dct_f64_n512_dct{k}_np"] = sf.dct(xr, type=k, axis=1)
        out[f"dct_f64_n512_dct{k}_ld"] = dct_ld(k, xr[0])
        print("dct", k, "scipy vs long double:", np.abs(out[f"dct_f64_n512_dct{k}_np"][0] - out[f"dct_f64_n512_dct{k}_ld"]).max() /
              np.abs(out[f"dct_f64_n512_dct{k}_ld"]).max(), flush=True)
    path = os.path.join(os.path.dirname(__file__), "baseline_lengths.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays; numpy", np.__version__, "mpmath", mp.__version__)


if __name__ == "__main__":
    main()
