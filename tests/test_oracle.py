"""CPU tests that PIN THE ORACLE (no GPU): against the reference's own known answers, against
numpy/scipy full-precision vectors, and against the long-double definitions; plus the iterator's
three strategies, the three normalisation points and the restated panics."""
import numpy as np
import pytest

import synth
from helpers import GOLDEN_SIZES, assert_close, cdt_of, rel_global
from oracle import oracle_ctypes as orc


def _c(a):
    return np.asarray(a, dtype=np.float64)


# ---- reference's own unit tests, restated one for one (src/lib.rs:903-1406) ----------------
@pytest.mark.parametrize("par", [False, True])
def test_fft(refvec, par):                                   # test_fft / test_fft_par
    m = _c(refvec["test_matrix"]["data"]); v = (m + 1j * m)
    sol = _c(refvec["fft_axis1"]["re"]) + 1j * _c(refvec["fft_axis1"]["im"])
    vhat = np.zeros((6, 6), np.complex128); h = orc.FftHandler(6)
    f, b = (orc.ndfft_par, orc.ndifft_par) if par else (orc.ndfft, orc.ndifft)
    v2 = np.zeros_like(v)
    f(v, vhat, h, 1); b(vhat, v2, h, 1)
    assert np.abs(vhat - sol).max() < 1e-3 and np.abs(v2 - v).max() < 1e-3
    assert orc.last_strategy() == 1


def test_fft_f_layout(refvec):                               # test_fft_f_layout (strategy iii)
    m = _c(refvec["test_matrix"]["data"]); v = np.asfortranarray(m + 1j * m)
    sol = _c(refvec["fft_axis1"]["re"]) + 1j * _c(refvec["fft_axis1"]["im"])
    vhat = np.zeros((6, 6), np.complex128); h = orc.FftHandler(6)
    orc.ndfft(v, vhat, h, 1); assert orc.last_strategy() == 3
    assert np.abs(vhat - sol).max() < 1e-3
    v2 = np.zeros((6, 6), np.complex128, order="F")
    orc.ndifft(vhat, v2, h, 1); assert orc.last_strategy() == 3
    assert np.abs(v2 - v).max() < 1e-3


@pytest.mark.parametrize("par", [False, True])
def test_fft_r2c(refvec, par):                               # test_fft_r2c / _par
    v = _c(refvec["test_matrix"]["data"])
    sol = _c(refvec["rfft_axis1"]["re"]) + 1j * _c(refvec["rfft_axis1"]["im"])
    vhat = np.zeros((6, 4), np.complex128); h = orc.R2cFftHandler(6)
    f, b = (orc.ndfft_r2c_par, orc.ndifft_r2c_par) if par else (orc.ndfft_r2c, orc.ndifft_r2c)
    v2 = np.zeros_like(v)
    f(v, vhat, h, 1); b(vhat, v2, h, 1)
    assert np.abs(vhat - sol).max() < 1e-3 and np.abs(v2 - v).max() < 1e-3


def test_ifft_c2r_first_last_element(refvec):
    d = refvec["c2r_first_last"]; h = orc.R2cFftHandler(6)
    for key_in, key_out in (("first_in", "first_out"), ("last_in", "last_out")):
        a = _c(d[key_in]); vhat = a[:, 0] + 1j * a[:, 1]
        v = np.zeros(6)
        orc.ndifft_r2c(vhat, v, h, 0)
        assert np.abs(v - _c(d[key_out])).max() < 1e-3


@pytest.mark.parametrize("par", [False, True])
def test_fft_r2c_odd(refvec, par):
    v = _c(refvec["r2c_odd_roundtrip"]["data"]); vhat = np.zeros((3, 2), np.complex128)
    h = orc.R2cFftHandler(3); v2 = np.zeros_like(v)
    f, b = (orc.ndfft_r2c_par, orc.ndifft_r2c_par) if par else (orc.ndfft_r2c, orc.ndifft_r2c)
    f(v, vhat, h, 1); b(vhat, v2, h, 1)
    assert np.abs(v2 - v).max() < 1e-3


@pytest.mark.parametrize("par", [False, True])
@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_dct(refvec, k, par):                                # test_dct1..4 / _par
    v = _c(refvec["test_matrix"]["data"]); sol = _c(refvec[f"dct{k}_axis1"]["data"])
    vhat = np.zeros_like(v); h = orc.DctHandler(6)
    f = getattr(orc, f"nddct{k}_par" if par else f"nddct{k}")
    f(v, vhat, h, 1)
    assert np.abs(vhat - sol).max() < 1e-3


# ---- examples' known answers ---------------------------------------------------------------
def test_example_fft2(refvec):                               # examples/fft2.rs
    d = refvec["example_fft2"]; m = _c(d["data"]); v = m + 1j * m
    sol = _c(d["re"]) + 1j * _c(d["im"])
    work = np.zeros_like(v); vhat = np.zeros_like(v)
    h0, h1 = orc.FftHandler(3), orc.FftHandler(3)
    orc.ndfft(v, work, h1, 1); orc.ndfft(work, vhat, h0, 0)
    assert orc.last_strategy() == 2
    assert np.abs(vhat - sol).max() < d["abs_tol"]
    w2 = np.zeros_like(v); v2 = np.zeros_like(v)
    orc.ndifft(vhat, w2, h0, 0); orc.ndifft(w2, v2, h1, 1)
    assert np.abs(v2 - v).max() < d["abs_tol"]


def test_example_rfft2(refvec):                              # examples/rfft2.rs
    d = refvec["example_rfft2"]; v = _c(d["data"]); sol = _c(d["re"]) + 1j * _c(d["im"])
    work = np.zeros((3, 2), np.complex128); vhat = np.zeros_like(work)
    h0, h1 = orc.FftHandler(3), orc.R2cFftHandler(3)
    orc.ndfft_r2c(v, work, h1, 1); orc.ndfft(work, vhat, h0, 0)
    assert np.abs(vhat - sol).max() < d["abs_tol"]
    w2 = np.zeros_like(work); v2 = np.zeros_like(v)
    orc.ndifft(vhat, w2, h0, 0); orc.ndifft_r2c(w2, v2, h1, 1)
    assert np.abs(v2 - v).max() < d["abs_tol"]


def test_example_fft_norm(refvec):                           # examples/fft_norm.rs
    d = refvec["example_fft_norm"]; x = _c(d["data"]); v = x + 1j * x

    def my_norm(lane):                                       # fn my_norm: 2/len
        lane *= 2.0 / lane.size

    for mode, fn, key in ((orc.NORM_DEFAULT, None, "default_roundtrip"), (orc.NORM_NONE, None, "none_roundtrip"),
                          (orc.NORM_CUSTOM, my_norm, "custom_2_over_n_roundtrip")):
        h = orc.FftHandler(3).normalization(mode, fn)
        vhat = np.zeros(3, np.complex128); v2 = np.zeros(3, np.complex128)
        orc.ndfft(v, vhat, h, 0); orc.ndifft(vhat, v2, h, 0)
        e = _c(d[key])
        assert np.abs(v2 - (e + 1j * e)).max() < 1e-12


def test_readme_r2c_6x4(refvec):                             # BASELINE.json configs[0]
    d = refvec["readme_r2c_6x4"]
    data = np.arange(24, dtype=np.float64).reshape(6, 4); vhat = np.zeros((4, 4), np.complex128)
    orc.ndfft_r2c(data, vhat, orc.R2cFftHandler(6), 0)
    assert orc.last_strategy() == 2
    assert np.abs(vhat - (_c(d["re"]) + 1j * _c(d["im"]))).max() < d["abs_tol"]


# ---- numpy/scipy full-precision vectors ------------------------------------------------------
@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_vs_numpy_scipy(npvec, dt, n):
    rdt = np.float64 if dt == "f64" else np.float32; cdt = cdt_of(rdt)
    tol = 1e-12 if dt == "f64" else 2e-5
    key = f"{dt}_n{n}"; m = n // 2 + 1
    xc, xr, xh = npvec[key + "_c_in"], npvec[key + "_r_in"], npvec[key + "_h_in"]
    h = orc.FftHandler(n, rdt); y = np.zeros_like(xc)
    orc.ndfft(xc, y, h, 1); assert_close(y, npvec[key + "_fft"], 1, tol, "fft")
    orc.ndifft(xc, y, h, 1); assert_close(y, npvec[key + "_ifft"], 1, tol, "ifft")
    hr = orc.R2cFftHandler(n, rdt); yr = np.zeros((2, m), cdt)
    orc.ndfft_r2c(xr, yr, hr, 1); assert_close(yr, npvec[key + "_r2c"], 1, tol, "r2c")
    xo = np.zeros((2, n), rdt)
    orc.ndifft_r2c(xh, xo, hr, 1); assert_close(xo, npvec[key + "_c2r"], 1, tol, "c2r")
    hd = orc.DctHandler(n, rdt)
    for k in (1, 2, 3, 4):
        if k == 1 and n < 2:
            continue
        getattr(orc, f"nddct{k}")(xr, xo, hd, 1)
        assert_close(xo, npvec[key + f"_dct{k}"], 1, tol * 4, f"dct{k}")


# ---- the BASELINE lane lengths: 4096 (cfg2 / cfg5), 8192 (cfg3-A R2C / C2R f32, cfg3-B c64), 512 (cfg4 DCT), 16384 ------------------
BL_C2C = [(4096, "f64"), (8192, "f32"), (8192, "f64"), (16384, "f64"), (16384, "f32")]


@pytest.mark.parametrize("n,dt", BL_C2C)
def test_baseline_lengths_c2c(blvec, n, dt):
    """ndfft / ndifft (src/lib.rs:313-338) at the lengths the BASELINE configs run, against three independent truths."""
    rdt = np.float64 if dt == "f64" else np.float32
    tol = 1e-12 if dt == "f64" else 2e-5
    key = f"c2c_{dt}_n{n}"; x = blvec[key + "_in"]
    h = orc.FftHandler(n, rdt); y = np.zeros_like(x)
    orc.ndfft(x, y, h, 1)
    assert_close(y, blvec[key + "_fft_np"], 1, tol, "fft vs pocketfft")
    assert_close(y[:1], blvec[key + "_fft_ld"][None, :], 1, tol, "fft vs long-double definition")
    b = blvec[key + "_mp_bins"]
    assert np.abs(y[0, b] - blvec[key + "_fft_mp"]).max() <= tol * np.abs(blvec[key + "_fft_ld"]).max(), "fft vs mpmath bins"
    orc.ndifft(x, y, h, 1)
    assert_close(y, blvec[key + "_ifft_np"], 1, tol, "ifft vs pocketfft")
    assert_close(y[:1], blvec[key + "_ifft_ld"][None, :], 1, tol, "ifft vs long-double definition")


@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_baseline_lengths_real(blvec, dt):
    """ndfft_r2c / ndifft_r2c (src/lib.rs:497-531) at n = 8192 (cfg3-A and the way back)."""
    n = 8192; m = n // 2 + 1
    rdt = np.float64 if dt == "f64" else np.float32; cdt = cdt_of(rdt)
    tol = 1e-12 if dt == "f64" else 2e-5
    key = f"real_{dt}_n{n}"; xr, xh = blvec[key + "_r_in"], blvec[key + "_h_in"]
    h = orc.R2cFftHandler(n, rdt)
    y = np.zeros((xr.shape[0], m), cdt); orc.ndfft_r2c(xr, y, h, 1)
    assert_close(y, blvec[key + "_r2c_np"], 1, tol, "r2c vs pocketfft")
    assert_close(y[:1], blvec[key + "_r2c_ld"][None, :], 1, tol, "r2c vs long-double definition")
    b = blvec[key + "_mp_bins"]
    assert np.abs(y[0, b] - blvec[key + "_r2c_mp"]).max() <= tol * np.abs(blvec[key + "_r2c_ld"]).max(), "r2c vs mpmath bins"
    xo = np.zeros((xh.shape[0], n), rdt); orc.ndifft_r2c(xh, xo, h, 1)
    assert_close(xo, blvec[key + "_c2r_np"], 1, tol, "c2r vs pocketfft")
    assert_close(xo[:1], blvec[key + "_c2r_ld"][None, :], 1, tol, "c2r vs long-double definition (DC / Nyquist imaginary parts dropped)")


def test_baseline_lengths_dct(blvec):
    """nddct1..4 (src/lib.rs:688-741) at n = 512 (cfg4)."""
    x = blvec["dct_f64_n512_in"]; h = orc.DctHandler(512); y = np.zeros_like(x)
    for k in (1, 2, 3, 4):
        getattr(orc, f"nddct{k}")(x, y, h, 1)
        assert_close(y, blvec[f"dct_f64_n512_dct{k}_np"], 1, 4e-12, f"dct{k} vs scipy")
        assert_close(y[:1], blvec[f"dct_f64_n512_dct{k}_ld"][None, :], 1, 4e-12, f"dct{k} vs long-double definition")


# ---- long-double definitions ------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 12, 17, 31, 37, 60, 74, 128, 221])
def test_vs_long_double_truth(n):
    x = synth.complex_array((n,), offset=n); y = np.zeros_like(x); h = orc.FftHandler(n).normalization(orc.NORM_NONE)
    orc.ndfft(x, y, h, 0); assert rel_global(y, orc.truth_dft(x, -1)) < 1e-13
    orc.ndifft(x, y, h, 0); assert rel_global(y, orc.truth_dft(x, +1)) < 1e-13
    xr = synth.real_array((n,), offset=n); yr = np.zeros_like(xr); hd = orc.DctHandler(n).normalization(orc.NORM_NONE)
    for k in (1, 2, 3, 4):
        getattr(orc, f"nddct{k}")(xr, yr, hd, 0)
        assert rel_global(yr, orc.truth_dct(k, xr)) < 1e-13, (n, k)


# ---- iterator strategies agree and are actually exercised ------------------------------------
def test_strategies_agree_3d():
    x = synth.complex_array((4, 5, 6)); ref = np.fft.fft(x, axis=1)
    h = orc.FftHandler(5)
    y = np.zeros_like(x); orc.ndfft(x, y, h, 1); assert orc.last_strategy() == 2
    assert rel_global(y, ref) < 1e-13
    yp = np.zeros_like(x); orc.ndfft_par(x, yp, h, 1); assert np.array_equal(y, yp)
    xf = np.asfortranarray(x); yf = np.zeros((4, 5, 6), np.complex128, order="F")
    orc.ndfft(xf, yf, h, 1); assert orc.last_strategy() == 3
    assert rel_global(yf, ref) < 1e-13
    # negative stride / broadcast-style views fall in strategy (iii)
    xs = x[::-1, :, ::2]; ys = np.zeros(xs.shape, np.complex128)
    orc.ndfft(xs, ys, h, 1); assert orc.last_strategy() == 3
    assert rel_global(ys, np.fft.fft(xs, axis=1)) < 1e-13
    y2 = np.zeros_like(x); h6 = orc.FftHandler(6)
    orc.ndfft(x, y2, h6, 2); assert orc.last_strategy() == 1
    assert rel_global(y2, np.fft.fft(x, axis=2)) < 1e-13


# ---- the three normalisation application points ----------------------------------
def test_normalization_points():
    n = 6
    seen = {}

    def spy(name):
        def f(lane):
            seen[name] = (lane.size, lane.dtype, lane.copy())
            lane *= 3.0
        return f

    # C2C inverse: AFTER, on the n-length output lane
    x = synth.complex_array((n,)); y = np.zeros_like(x)
    orc.ndifft(x, y, orc.FftHandler(n).normalization(orc.NORM_CUSTOM, spy("c2c")), 0)
    assert seen["c2c"][0] == n and rel_global(seen["c2c"][2], np.fft.ifft(x) * n) < 1e-13
    assert rel_global(y, np.fft.ifft(x) * n * 3) < 1e-13
    # forward C2C ignores normalisation entirely
    y0 = np.zeros_like(x); orc.ndfft(x, y0, orc.FftHandler(n).normalization(orc.NORM_CUSTOM, spy("fwd")), 0)
    assert "fwd" not in seen and rel_global(y0, np.fft.fft(x)) < 1e-13
    # C2R: BEFORE, on the m-length complex lane
    xh = synth.complex_array((n // 2 + 1,)); yr = np.zeros(n)
    orc.ndifft_r2c(xh, yr, orc.R2cFftHandler(n).normalization(orc.NORM_CUSTOM, spy("c2r")), 0)
    assert seen["c2r"][0] == n // 2 + 1 and np.array_equal(seen["c2r"][2], xh)
    assert rel_global(yr, np.fft.irfft(xh * 3, n) * n) < 1e-13
    # DCT: BEFORE, on the real input lane
    xr = synth.real_array((n,)); yd = np.zeros(n)
    orc.nddct2(xr, yd, orc.DctHandler(n).normalization(orc.NORM_CUSTOM, spy("dct")), 0)
    assert seen["dct"][0] == n and np.array_equal(seen["dct"][2], xr)
    import scipy.fft as sf
    assert rel_global(yd, sf.dct(xr * 3, type=2) / 2) < 1e-13
    # None: raw rustfft / realfft / rustdct scaling
    orc.nddct2(xr, yd, orc.DctHandler(n).normalization(orc.NORM_NONE), 0)
    assert rel_global(yd, sf.dct(xr, type=2) / 2) < 1e-13
    orc.ndifft_r2c(xh, yr, orc.R2cFftHandler(n).normalization(orc.NORM_NONE), 0)
    assert rel_global(yr, np.fft.irfft(xh, n) * n) < 1e-13


# ---- restated panics -----------------------------------------------------------------------------
def test_panics():
    x = np.zeros((3, 5), np.complex128); y = np.zeros((3, 5), np.complex128)
    with pytest.raises(orc.OraclePanic, match="Size mismatch in fft, got 5 expected 6"):
        orc.ndfft(x, y, orc.FftHandler(6), 1)
    with pytest.raises(orc.OraclePanic, match="Size mismatch in dct, got 5 expected 4"):
        orc.nddct1(np.zeros((3, 5)), np.zeros((3, 5)), orc.DctHandler(4), 1)
    with pytest.raises(orc.OraclePanic) as e:
        orc.ndfft(x, y, orc.FftHandler(5), 2)
    assert e.value.code == orc.PANIC_AXIS
    with pytest.raises(orc.OraclePanic) as e:
        orc.ndfft(x, np.zeros((4, 5), np.complex128), orc.FftHandler(5), 1)
    assert e.value.code == orc.PANIC_ZIP
    # r2c: output lane must be n/2+1
    with pytest.raises(orc.OraclePanic, match="Size mismatch in fft, got 6 expected 4"):
        orc.ndfft_r2c(np.zeros((2, 6)), np.zeros((2, 6), np.complex128), orc.R2cFftHandler(6), 1)
    # no lanes -> the closure never runs -> no panic even with a wrong handler
    orc.ndfft(np.zeros((0, 5), np.complex128), np.zeros((0, 5), np.complex128), orc.FftHandler(6), 1)
