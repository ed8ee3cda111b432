"""Parity cases shared by tests/test_gpu_parity.py (the real thing: gfx950 library on an MI355X,
through the C ABI) and tests/test_emul_parity.py (the same sources compiled for the host through
tests/emul, CPU container).  Every case compares against the CPU oracle on the same seeded inputs
and/or the committed golden vectors.  Tolerances are BASELINE.json's: 1e-10 (f64), 1e-4 (f32)."""
import contextlib
import os

import numpy as np

import synth
from helpers import TOL, assert_close, cdt_of
from ndrustfft_amd import _lib, api, handlers
from ndrustfft_amd.handlers import Normalization
from oracle import oracle_ctypes as orc

ORC_NORM = {"None": orc.NORM_NONE, "Default": orc.NORM_DEFAULT}

OPS = {
    # name: (api fn, oracle fn, handler class name, in complex, out complex)
    "ndfft": (api.ndfft, orc.ndfft, "FftHandler", True, True),
    "ndifft": (api.ndifft, orc.ndifft, "FftHandler", True, True),
    "ndfft_r2c": (api.ndfft_r2c, orc.ndfft_r2c, "R2cFftHandler", False, True),
    "ndifft_r2c": (api.ndifft_r2c, orc.ndifft_r2c, "R2cFftHandler", True, False),
    "nddct1": (api.nddct1, orc.nddct1, "DctHandler", False, False),
    "nddct2": (api.nddct2, orc.nddct2, "DctHandler", False, False),
    "nddct3": (api.nddct3, orc.nddct3, "DctHandler", False, False),
    "nddct4": (api.nddct4, orc.nddct4, "DctHandler", False, False),
}



@contextlib.contextmanager
def switches(L, **env):
    """Environment switches of the library for the duration of a block.  The library parses its switches ONCE (csrc/switches.h), so a
    change only takes effect through ndfft_reload_switches() -- a test hook, never called while another thread transforms."""
    old = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    L.reload_switches()
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        L.reload_switches()


def handlers_for(name, n, rdt, L, norm="Default"):
    cls = OPS[name][2]
    h = getattr(handlers, cls)(n, rdt, _library=L)
    o = getattr(orc, cls)(n, rdt)
    if norm != "Default":
        h = h.normalization(Normalization(norm)); o = o.normalization(ORC_NORM[norm])
    return h, o


def shapes_for(name, shape, axis):
    """(input shape, output shape) for op `name` on an array of `shape` (lane length n on `axis`)."""
    n = shape[axis]; m = n // 2 + 1
    sin, sout = list(shape), list(shape)
    if name == "ndfft_r2c":
        sout[axis] = m
    if name == "ndifft_r2c":
        sin[axis] = m
    return tuple(sin), tuple(sout)


def make_input(name, shape, rdt, offset=0):
    return synth.complex_array(shape, cdt_of(rdt), offset=offset) if OPS[name][3] else synth.real_array(shape, rdt, offset=offset)


def run_case(L, name, shape, axis, rdt, norm="Default", layout="C", tol=None, offset=0):
    """One nd* call on the library under test vs the oracle; returns the path the library took."""
    n = shape[axis]
    fn, ofn, _, in_c, out_c = OPS[name]
    sin, sout = shapes_for(name, shape, axis)
    x = make_input(name, sin, rdt, offset)
    odt = cdt_of(rdt) if out_c else np.dtype(rdt)
    if layout == "F":
        x = np.asfortranarray(x)
    y = np.zeros(sout, odt, order="F" if layout == "F" else "C")
    yo = np.zeros(sout, odt)
    h, o = handlers_for(name, n, rdt, L, norm)
    fn(x, y, h, axis)
    path = L.last_path()
    ofn(np.ascontiguousarray(x), yo, o, axis)
    assert_close(y, yo, axis, tol or TOL[np.dtype(rdt)], f"{name} shape={shape} axis={axis} {np.dtype(rdt)} norm={norm} path={path}")
    return path


# ---- reference's own unit tests restated against the library (src/lib.rs:903-1406) ------------
def reference_unit_tests(L, refvec):
    m = np.asarray(refvec["test_matrix"]["data"], np.float64)
    v = m + 1j * m
    sol = np.asarray(refvec["fft_axis1"]["re"]) + 1j * np.asarray(refvec["fft_axis1"]["im"])
    for f, b in ((api.ndfft, api.ndifft), (api.ndfft_par, api.ndifft_par)):      # test_fft, test_fft_par
        vhat = np.zeros((6, 6), np.complex128); v2 = np.zeros_like(v); h = handlers.FftHandler(6, _library=L)
        f(v, vhat, h, 1); b(vhat, v2, h, 1)
        assert np.abs(vhat - sol).max() < 1e-3 and np.abs(v2 - v).max() < 1e-3
    vf = np.asfortranarray(v)                                                    # test_fft_f_layout
    vhat = np.zeros((6, 6), np.complex128); h = handlers.FftHandler(6, _library=L)
    api.ndfft(vf, vhat, h, 1); assert np.abs(vhat - sol).max() < 1e-3
    v2 = np.zeros((6, 6), np.complex128, order="F"); api.ndifft(vhat, v2, h, 1)
    assert np.abs(v2 - v).max() < 1e-3
    solr = np.asarray(refvec["rfft_axis1"]["re"]) + 1j * np.asarray(refvec["rfft_axis1"]["im"])
    for f, b in ((api.ndfft_r2c, api.ndifft_r2c), (api.ndfft_r2c_par, api.ndifft_r2c_par)):   # test_fft_r2c(_par)
        vhat = np.zeros((6, 4), np.complex128); m2 = np.zeros_like(m); h = handlers.R2cFftHandler(6, _library=L)
        f(m, vhat, h, 1); b(vhat, m2, h, 1)
        assert np.abs(vhat - solr).max() < 1e-3 and np.abs(m2 - m).max() < 1e-3
    d = refvec["c2r_first_last"]; h = handlers.R2cFftHandler(6, _library=L)      # test_ifft_c2r_first_last_element
    for ki, ko in (("first_in", "first_out"), ("last_in", "last_out")):
        a = np.asarray(d[ki], np.float64); vh = a[:, 0] + 1j * a[:, 1]; out = np.zeros(6)
        api.ndifft_r2c(vh, out, h, 0)
        assert np.abs(out - np.asarray(d[ko])).max() < 1e-3
    v3 = np.asarray(refvec["r2c_odd_roundtrip"]["data"], np.float64)            # test_fft_r2c_odd(_par)
    vh = np.zeros((3, 2), np.complex128); v4 = np.zeros_like(v3); h = handlers.R2cFftHandler(3, _library=L)
    api.ndfft_r2c(v3, vh, h, 1); api.ndifft_r2c(vh, v4, h, 1)
    assert np.abs(v4 - v3).max() < 1e-3
    for k in (1, 2, 3, 4):                                                       # test_dct1..4(_par)
        sol_k = np.asarray(refvec[f"dct{k}_axis1"]["data"])
        for fn in (getattr(api, f"nddct{k}"), getattr(api, f"nddct{k}_par")):
            out = np.zeros_like(m); fn(m, out, handlers.DctHandler(6, _library=L), 1)
            assert np.abs(out - sol_k).max() < 1e-3, k


def reference_examples(L, refvec):
    d = refvec["example_fft2"]; m = np.asarray(d["data"], np.float64); v = m + 1j * m       # examples/fft2.rs
    sol = np.asarray(d["re"]) + 1j * np.asarray(d["im"])
    work = np.zeros_like(v); vhat = np.zeros_like(v)
    h0, h1 = handlers.FftHandler(3, _library=L), handlers.FftHandler(3, _library=L)
    api.ndfft(v, work, h1, 1); api.ndfft(work, vhat, h0, 0)
    assert np.abs(vhat - sol).max() < d["abs_tol"]
    w2 = np.zeros_like(v); v2 = np.zeros_like(v)
    api.ndifft(vhat, w2, h0, 0); api.ndifft(w2, v2, h1, 1)
    assert np.abs(v2 - v).max() < d["abs_tol"]
    d = refvec["example_rfft2"]; vr = np.asarray(d["data"], np.float64)                      # examples/rfft2.rs
    sol = np.asarray(d["re"]) + 1j * np.asarray(d["im"])
    work = np.zeros((3, 2), np.complex128); vhat = np.zeros_like(work)
    hr = handlers.R2cFftHandler(3, _library=L)
    api.ndfft_r2c(vr, work, hr, 1); api.ndfft(work, vhat, h0, 0)
    assert np.abs(vhat - sol).max() < d["abs_tol"]
    w2 = np.zeros_like(work); vr2 = np.zeros_like(vr)
    api.ndifft(vhat, w2, h0, 0); api.ndifft_r2c(w2, vr2, hr, 1)
    assert np.abs(vr2 - vr).max() < d["abs_tol"]
    d = refvec["example_fft_norm"]; x = np.asarray(d["data"], np.float64); v = x + 1j * x    # examples/fft_norm.rs

    def my_norm(lane):
        lane *= 2.0 / lane.size

    for norm, key in ((Normalization.default(), "default_roundtrip"), (Normalization.none(), "none_roundtrip"),
                      (Normalization.custom(my_norm), "custom_2_over_n_roundtrip")):
        h = handlers.FftHandler(3, _library=L).normalization(norm)
        vhat = np.zeros(3, np.complex128); v2 = np.zeros(3, np.complex128)
        api.ndfft(v, vhat, h, 0); api.ndifft(vhat, v2, h, 0)
        e = np.asarray(d[key]); assert np.abs(v2 - (e + 1j * e)).max() < 1e-12
    d = refvec["readme_r2c_6x4"]                                                              # BASELINE configs[0]
    data = np.arange(24, dtype=np.float64).reshape(6, 4); vhat = np.zeros((4, 4), np.complex128)
    api.ndfft_r2c(data, vhat, handlers.R2cFftHandler(6, _library=L), 0)
    assert np.abs(vhat - (np.asarray(d["re"]) + 1j * np.asarray(d["im"]))).max() < d["abs_tol"]


# ---- independent truths at the BASELINE lane lengths (tests/golden/make_golden_baseline.py) ------------------------
def baseline_length_fixtures(L, blvec, device=None):
    """The library under test against numpy / scipy (every lane), the long-double definition (lane 0) and mpmath (12 bins) at n = 4096 c128,
    8192 c64 / f32 R2C / C2R, 16384, 512 DCT-I..IV -- the lengths the BASELINE configs run.  Host arrays, or torch tensors on `device`."""
    def call(fn, x, yshape, ydt, h, axis):
        if device is None:
            y = np.zeros(yshape, ydt); fn(x, y, h, axis); return y
        import torch
        xd = torch.from_numpy(np.ascontiguousarray(x)).to(device); yd = torch.zeros(yshape, dtype=torch.from_numpy(np.zeros(1, ydt)).dtype, device=device)
        fn(xd, yd, h, axis); return yd.cpu().numpy()
    for n, dt in ((4096, "f64"), (8192, "f32"), (8192, "f64"), (16384, "f64"), (16384, "f32")):
        rdt = np.float64 if dt == "f64" else np.float32; tol = TOL[np.dtype(rdt)]
        key = f"c2c_{dt}_n{n}"; x = blvec[key + "_in"]; h = handlers.FftHandler(n, rdt, _library=L)
        y = call(api.ndfft, x, x.shape, x.dtype, h, 1)
        assert_close(y, blvec[key + "_fft_np"], 1, tol, f"{key} fft vs pocketfft")
        assert_close(y[:1], blvec[key + "_fft_ld"][None, :], 1, tol, f"{key} fft vs long double")
        b = blvec[key + "_mp_bins"]
        assert np.abs(y[0, b] - blvec[key + "_fft_mp"]).max() <= tol * np.abs(blvec[key + "_fft_ld"]).max(), f"{key} fft vs mpmath"
        y = call(api.ndifft, x, x.shape, x.dtype, h, 1)
        assert_close(y, blvec[key + "_ifft_np"], 1, tol, f"{key} ifft vs pocketfft")
        assert_close(y[:1], blvec[key + "_ifft_ld"][None, :], 1, tol, f"{key} ifft vs long double")
    for dt in ("f32", "f64"):
        n = 8192; m = n // 2 + 1
        rdt = np.float64 if dt == "f64" else np.float32; tol = TOL[np.dtype(rdt)]
        key = f"real_{dt}_n{n}"; xr, xh = blvec[key + "_r_in"], blvec[key + "_h_in"]; h = handlers.R2cFftHandler(n, rdt, _library=L)
        y = call(api.ndfft_r2c, xr, (xr.shape[0], m), cdt_of(rdt), h, 1)
        assert_close(y, blvec[key + "_r2c_np"], 1, tol, f"{key} r2c vs pocketfft")
        assert_close(y[:1], blvec[key + "_r2c_ld"][None, :], 1, tol, f"{key} r2c vs long double")
        xo = call(api.ndifft_r2c, xh, (xh.shape[0], n), rdt, h, 1)
        assert_close(xo, blvec[key + "_c2r_np"], 1, tol, f"{key} c2r vs pocketfft")
        assert_close(xo[:1], blvec[key + "_c2r_ld"][None, :], 1, tol, f"{key} c2r vs long double")
        # the strided axis (cfg3-A's layout): the same lanes as columns
        yt = call(api.ndfft_r2c, np.ascontiguousarray(xr.T), (m, xr.shape[0]), cdt_of(rdt), h, 0)
        assert_close(yt.T, blvec[key + "_r2c_np"], 1, tol, f"{key} r2c axis 0 vs pocketfft")
    x = blvec["dct_f64_n512_in"]; h = handlers.DctHandler(512, _library=L)
    for k in (1, 2, 3, 4):
        y = call(getattr(api, f"nddct{k}"), x, x.shape, x.dtype, h, 1)
        assert_close(y, blvec[f"dct_f64_n512_dct{k}_np"], 1, 1e-10, f"dct{k} n=512 vs scipy")
        assert_close(y[:1], blvec[f"dct_f64_n512_dct{k}_ld"][None, :], 1, 1e-10, f"dct{k} n=512 vs long double")


# ---- committed numpy/scipy golden vectors -------------------------------------------------------
def golden_vectors(L, npvec, dt, n):
    rdt = np.float64 if dt == "f64" else np.float32; cdt = cdt_of(rdt); tol = TOL[np.dtype(rdt)]
    key = f"{dt}_n{n}"; m = n // 2 + 1
    xc, xr, xh = npvec[key + "_c_in"], npvec[key + "_r_in"], npvec[key + "_h_in"]
    h = handlers.FftHandler(n, rdt, _library=L); y = np.zeros_like(xc)
    api.ndfft(xc, y, h, 1); assert_close(y, npvec[key + "_fft"], 1, tol, f"fft n={n}")
    api.ndifft(xc, y, h, 1); assert_close(y, npvec[key + "_ifft"], 1, tol, f"ifft n={n}")
    hr = handlers.R2cFftHandler(n, rdt, _library=L); yr = np.zeros((2, m), cdt)
    api.ndfft_r2c(xr, yr, hr, 1); assert_close(yr, npvec[key + "_r2c"], 1, tol, f"r2c n={n}")
    xo = np.zeros((2, n), rdt)
    api.ndifft_r2c(xh, xo, hr, 1); assert_close(xo, npvec[key + "_c2r"], 1, tol, f"c2r n={n}")
    hd = handlers.DctHandler(n, rdt, _library=L)
    for k in (1, 2, 3, 4):
        if k == 1 and n < 2:
            continue
        getattr(api, f"nddct{k}")(xr, xo, hd, 1)
        assert_close(xo, npvec[key + f"_dct{k}"], 1, tol, f"dct{k} n={n}")


# ---- layouts: the three iterator strategies + views ndarray allows -------------------------------
def layouts(L):
    seen = set()
    for name in OPS:
        for rdt in (np.float64, np.float32):
            seen.add(run_case(L, name, (5, 12), 1, rdt))                 # strategy (i)
            seen.add(run_case(L, name, (12, 5), 0, rdt))                 # strategy (ii), 2-D
            seen.add(run_case(L, name, (3, 10, 7), 1, rdt))              # strategy (ii), middle axis of 3-D
            seen.add(run_case(L, name, (3, 4, 9), 2, rdt))               # strategy (i), 3-D (rows() flattens)
            seen.add(run_case(L, name, (6, 9), 1, rdt, layout="F"))      # strategy (iii)
            seen.add(run_case(L, name, (6, 9), 0, rdt, layout="F"))      # strategy (iii), contiguous lanes
            seen.add(run_case(L, name, (5, 20), 1, rdt))                 # beyond the thread-per-lane kernels (n <= 16): LDS kernel
            seen.add(run_case(L, name, (20, 5), 0, rdt))
            seen.add(run_case(L, name, (4, 21, 3), 1, rdt, layout="F"))
    assert {"generic_row", "generic_col", "tiny_row", "tiny_col", "tinymat_row", "tinymat_col"} <= seen, seen
    # negative strides, stepped views, broadcast (stride 0) input, non-dense output view
    x = synth.complex_array((4, 8, 6))
    h = handlers.FftHandler(8, _library=L); o = orc.FftHandler(8)
    xv = x[::-1, :, ::2]
    y = np.zeros(xv.shape, np.complex128); yo = np.zeros_like(y)
    api.ndfft(xv, y, h, 1); orc.ndfft(xv, yo, o, 1); assert_close(y, yo, 1, 1e-10, "negative/stepped view")
    xb = np.broadcast_to(x[0:1], (3, 8, 6))
    y = np.zeros((3, 8, 6), np.complex128); yo = np.zeros_like(y)
    api.ndfft(xb, y, h, 1); orc.ndfft(xb, yo, o, 1); assert_close(y, yo, 1, 1e-10, "broadcast input")
    big = np.full((4, 8, 12), 7.5 + 0j); bigo = big.copy()
    api.ndfft(x, big[:, :, ::2], h, 1); orc.ndfft(x, bigo[:, :, ::2], o, 1)
    assert_close(big, bigo, 1, 1e-10, "strided output view"); assert np.all(big[:, :, 1::2] == 7.5)
    # 5-D with a permuted (non-mergeable) layout: exercises > 4 batch dims peeling
    x5 = synth.real_array((2, 3, 2, 6, 2, 3)).transpose(2, 0, 4, 3, 1, 5)
    y5 = np.zeros(x5.shape).transpose(1, 0, 2, 3, 5, 4).copy().transpose(1, 0, 2, 3, 5, 4)
    y5o = np.zeros(x5.shape)
    hd = handlers.DctHandler(6, _library=L); od = orc.DctHandler(6)
    api.nddct2(x5, y5, hd, 3); orc.nddct2(x5, y5o, od, 3); assert_close(y5, y5o, 3, 1e-10, "6-D permuted")


def interleaved_mut_views_two_threads(L, rounds=6):
    """Two host threads transform into INTERLEAVED mutable views of one allocation (even / odd columns, as
    ndarray's multi_slice_mut hands out; src/lib.rs:109 only requires `S: DataMut` of the view).  Rust guarantees
    each thread exclusivity of its view's OWN elements only, so the host path must never write -- not even
    write back unchanged -- an element outside its view: a whole-span download would overwrite the other thread's
    results with stale data."""
    import threading
    n = 64
    h = handlers.FftHandler(n, _library=L); o = orc.FftHandler(n)
    xs = [synth.complex_array((96, n, 5), offset=17 * k) for k in range(2)]
    # also negative strides and a hole pattern with a contiguous run > 1 (pairs of columns)
    views = [lambda a, k: a[:, :, k::2], lambda a, k: a[:, ::-1, k::2], lambda a, k: a.reshape(96, n, 5, 2)[:, :, :, k]]
    for mk in views:
        for _ in range(rounds):
            big = np.zeros((96, n, 10), np.complex128)
            errs = []
            def work(k):
                try:
                    for _rep in range(3):
                        api.ndfft(xs[k], mk(big, k), h, 1)
                except Exception as e:          # pragma: no cover
                    errs.append(e)
            ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
            [t.start() for t in ts]; [t.join() for t in ts]
            assert not errs, errs
            for k in range(2):
                yo = np.zeros((96, n, 5), np.complex128)
                orc.ndfft(xs[k], yo, o, 1)
                ref = np.zeros((96, n, 10), np.complex128); mk(ref, k)[...] = yo
                assert_close(mk(big, k), mk(ref, k), 1, 1e-10, f"interleaved view {k}")
    # holes are preserved bit for bit, also on a real-output op and a 1-D output with a step
    big = np.full((7, 40), -3.25); x = synth.real_array((7, 20))
    api.nddct2(x, big[:, 1::2], handlers.DctHandler(20, _library=L), 1)
    assert np.all(big[:, 0::2] == -3.25)
    yo = np.zeros((7, 20)); orc.nddct2(x, yo, orc.DctHandler(20), 1); assert_close(big[:, 1::2], yo, 1, 1e-10, "stepped dct out")


def host_pipeline_pageable(L, shapes=None):
    """ndfft_exec on ordinary (pageable) host arrays of >= 8 MiB whose dimension 0 is a batch dimension runs as a
    chunk pipeline through pinned bounce buffers (host copy pool || H2D || kernel || D2H); same results as the
    plain path (NDFFT_HOST_PIPE=0), uneven last chunk, padded rows, every op family."""
    shapes = shapes or (("ndfft", (1000, 1024), 1, np.float64), ("ndfft_r2c", (301, 8, 2048), 2, np.float32),
                        ("nddct2", (130, 96, 128), 1, np.float64), ("ndifft_r2c", (515, 2049), 1, np.float64))
    for name, shape, axis, rdt in shapes:
        sin, sout = shapes_for(name, shape, axis)
        x = make_input(name, sin, rdt)
        odt = cdt_of(rdt) if OPS[name][4] else np.dtype(rdt)
        h, o = handlers_for(name, shape[axis], rdt, L)
        y1 = np.zeros(sout, odt); y2 = np.zeros(sout, odt); yo = np.zeros(sout, odt)
        with switches(L, NDFFT_HOST_PIPE="1"):                   # force the pipeline whatever the size
            OPS[name][0](x, y1, h, axis)
        with switches(L, NDFFT_HOST_PIPE="0"):
            OPS[name][0](x, y2, h, axis)
        assert np.abs(y1 - y2).max() <= 50 * np.finfo(rdt).eps * np.abs(y2).max(), (name, shape)
        OPS[name][1](x, yo, o, axis)
        assert_close(y1, yo, axis, TOL[np.dtype(rdt)], f"host pipeline {name} {shape}")
    # rows with padding between them (stride[0] > row length) on both sides
    xb = synth.complex_array((600, 1100)); x = xb[:, :1024]
    yb = np.full((600, 1030), 9.0 + 0j); y = yb[:, :1024]
    h = handlers.FftHandler(1024, _library=L)
    with switches(L, NDFFT_HOST_PIPE="1"):
        api.ndfft(x, y, h, 1)
    yo = np.zeros((600, 1024), np.complex128); orc.ndfft(np.ascontiguousarray(x), yo, orc.FftHandler(1024), 1)
    assert_close(y, yo, 1, 1e-10, "host path, padded rows"); assert np.all(yb[:, 1024:] == 9.0)


def tiny_lanes(L):
    """C2C lanes of 2..13 and 16 points on the thread-per-lane kernel: dense rows (LDS-staged), strided axes with adjacent
    lanes contiguous (middle / first axis of C-layout arrays), F-layout and stepped views (direct), both directions and
    norms, workgroup tails -- including n = 6, the length of the reference's own unit tests (src/lib.rs:903-1406)."""
    seen = set()
    for rdt in (np.float64, np.float32):
        for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16):
            for name in ("ndfft", "ndifft"):
                seen.add(run_case(L, name, (300, n, 70), 1, rdt, offset=n))            # middle axis: tiny_col (n < 12) / jit_col
                seen.add(run_case(L, name, (n, 1000), 0, rdt, norm="None", offset=n))  # first axis
                if n & (n - 1):                                                        # (dense power-of-two rows: wave kernel)
                    seen.add(run_case(L, name, (777, n), 1, rdt, offset=3 * n))        # dense rows: tiny_row (n < 12)
                seen.add(run_case(L, name, (40, n), 1, rdt, layout="F"))               # F layout: lanes strided, neighbours n apart
            x = synth.complex_array((50, 2 * n + 3), cdt_of(rdt))[:, 1:2 * n + 1:2]    # stepped lanes, padded rows
            y = np.zeros((50, n), cdt_of(rdt)); yo = np.zeros_like(y)
            h, o = handlers_for("ndfft", n, rdt, L)
            api.ndfft(x, y, h, 1); orc.ndfft(np.ascontiguousarray(x), yo, o, 1)
            seen.add(L.last_path())
            assert_close(y, yo, 1, TOL[np.dtype(rdt)], f"tiny stepped n={n}")
    assert {"tiny_col", "tiny_row", "tiny_strided"} <= seen, seen


def reg_lanes(L, sizes=(14, 15, 17, 18, 19, 20, 21, 23, 24, 28, 29, 30, 31, 34, 36, 38, 40, 45, 46, 48, 49, 51, 56, 57, 58, 60, 62, 63), sizes_f32=(70, 72, 80, 90, 96), want=True):
    """C2C lanes of 14..64 (f64) / 96 (f32) points that factor into two butterflies: one thread per lane, two passes in
    registers (reg_kernel.h, hiprtc-specialised): dense rows (LDS-staged chunks, workgroup tails), strided axes, both
    directions and norms.  `want`: the path must be taken (False where only some sizes are built, as in the CPU emulation)."""
    for rdt in (np.float64, np.float32):
        for n in tuple(sizes) + (tuple(sizes_f32) if rdt == np.float32 else ()):
            rows = (1 << 16) // n + 37
            for name, norm in (("ndfft", "Default"), ("ndifft", "Default"), ("ndifft", "None")):
                p1 = run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=n)
                p2 = run_case(L, name, (5, n, rows // 4), 1, rdt, norm=norm, offset=2 * n)
                if want:
                    assert p2 == "reg_col" and (p1 == "reg_row" or n > 63), (n, rdt, p1, p2)


def regreal_lanes(L, sizes=(12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 25, 27, 30, 32, 33, 36, 40, 42, 45, 48), sizes_f32=(50, 56, 60, 64, 65, 72), want=True):
    """R2C / C2R / DCT-I..IV on lanes of 17..48 (f64) / 72 (f32) points, even and odd: one thread per lane with the raw
    lane, the inner FFT and the outputs in registers (reg_kernel.h: RegReal, hiprtc-specialised); dense rows, strided
    axes, norms.  `want`: every factorable size must take the path (False: only what the build instantiates)."""
    seen = set()
    for rdt in (np.float64, np.float32):
        for n in tuple(sizes) + (tuple(sizes_f32) if rdt == np.float32 else ()):
            rows = (1 << 16) // n + 29
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct1", "nddct2", "nddct3", "nddct4"):
                p1 = run_case(L, name, (rows, n), 1, rdt, offset=n)
                p2 = run_case(L, name, (3, n, rows // 2), 1, rdt, norm="None", offset=3 * n)
                seen.update((p1, p2))
    if want:
        assert {"regreal_row", "regreal_col"} <= seen, seen
    return seen


def tinymat_lanes(L):
    """R2C / C2R / DCT-I..IV on lanes of 2..16 points (thread-per-lane, the transform as a dense matrix): every op and n,
    both dtypes, None / Default norms, dense rows, strided axes, F layout, stepped views, workgroup tails."""
    seen = set()
    for rdt in (np.float64, np.float32):
        for n in range(2, 17):
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct1", "nddct2", "nddct3", "nddct4"):
                for norm in ("Default", "None"):
                    seen.add(run_case(L, name, (300, n), 1, rdt, norm=norm, offset=n))
                seen.add(run_case(L, name, (9, n, 70), 1, rdt, offset=2 * n))
                seen.add(run_case(L, name, (n, 300), 0, rdt, norm="None"))
                seen.add(run_case(L, name, (33, n), 1, rdt, layout="F"))
        # stepped input lanes, padded output rows
        x = synth.real_array((40, 2 * 12 + 1), rdt)[:, ::2][:, :12]; yb = np.full((40, 16), 3.0, rdt); y = yb[:, 2:14]; yo = np.zeros((40, 12), rdt)
        h, o = handlers_for("nddct2", 12, rdt, L)
        api.nddct2(x, y, h, 1); orc.nddct2(np.ascontiguousarray(x), yo, o, 1)
        seen.add(L.last_path())
        assert_close(y, yo, 1, TOL[np.dtype(rdt)], "tinymat stepped"); assert np.all(yb[:, :2] == 3.0) and np.all(yb[:, 14:] == 3.0)
    assert {"tinymat_row", "tinymat_col", "tinymat_strided"} <= seen, seen
    # the DC / Nyquist imaginary parts of a C2R input are ignored (lib.rs:516-521), n even and odd
    for n in (6, 7):
        m = n // 2 + 1
        z = synth.complex_array((5, m)); z2 = z.copy(); z2[:, 0] = z2[:, 0].real + 7j
        if n % 2 == 0: z2[:, -1] = z2[:, -1].real - 3j
        a = np.zeros((5, n)); b = np.zeros((5, n)); h = handlers.R2cFftHandler(n, _library=L)
        api.ndifft_r2c(z, a, h, 1); api.ndifft_r2c(z2, b, h, 1)
        assert np.array_equal(a, b)


def wave_short_lanes(L):
    """Dense C2C lanes of n = 2..64 (powers of two) run on the LDS-free wavefront kernel (wave_kernel.h: coalesced
    16-byte accesses + DPP / v_permlane swaps); every size, both dtypes, both directions and norms, chunk tails (a
    wavefront owns 512 elements: row counts that leave partial chunks), 3-D arrays; and the older kernels stay covered
    with NDFFT_WAVE=0."""
    for rdt in (np.float64, np.float32):
        for n in (2, 4, 8, 16, 32, 64):
            for rows in (1, 5, 512 // n, 512 // n + 1, 777):
                for name in ("ndfft", "ndifft"):
                    for norm in ("Default", "None"):
                        assert run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=rows * n) == "wave_reg", (n, rows)
        assert run_case(L, "ndfft", (6, 10, 32), 2, rdt) == "wave_reg"          # rows() flattens the leading dims
        assert run_case(L, "ndfft", (10, 32, 6), 1, rdt) != "wave_reg"          # strided lanes keep the column kernels
    # a view whose rows are padded (pitch != n) is not dense: older path
    x = synth.complex_array((9, 80))[:, :64]; y = np.zeros((9, 64), np.complex128); yo = np.zeros_like(y)
    api.ndfft(x, y, handlers.FftHandler(64, _library=L), 1); orc.ndfft(np.ascontiguousarray(x), yo, orc.FftHandler(64), 1)
    assert_close(y, yo, 1, 1e-10, "padded rows n=64")
    with switches(L, NDFFT_WAVE="0"):
        for rdt in (np.float64, np.float32):
            assert run_case(L, "ndfft", (37, 64), 1, rdt) == "pow2_reg"
            assert run_case(L, "ndifft", (37, 16), 1, rdt) == "tiny_row"
            with switches(L, NDFFT_TINY="0"):
                assert run_case(L, "ndifft", (37, 16), 1, rdt) == "generic_row"


def sharded_exec(L, device_ids, torch_device=None):
    """ndfft_exec_sharded (host arrays) and ndfft_exec_sharded_device (arrays resident on one device) over `device_ids`:
    every op, the split dimension in every position, uneven blocks, fewer lanes than devices, a single lane, views with
    negative strides and holes -- bit-identical to the single-device call, and the same panics."""
    import ctypes
    api.set_par_devices(device_ids)
    try:
        cases = [("ndfft", (6, 16), 1), ("ndfft", (5, 16), 1), ("ndifft", (12, 7, 3), 0), ("ndfft_r2c", (9, 10), 1),
                 ("ndifft_r2c", (4, 3, 10), 2), ("nddct1", (3, 9, 5), 1), ("nddct2", (4, 3, 8), 2), ("nddct3", (7, 8), 1),
                 ("nddct4", (8, 7), 0), ("ndfft", (1, 16), 1), ("ndfft", (16,), 0), ("ndfft", (3, 64, 40), 1)]
        for name, shape, axis in cases:
            for rdt in (np.float64, np.float32):
                sin, sout = shapes_for(name, shape, axis)
                x = make_input(name, sin, rdt)
                odt = cdt_of(rdt) if OPS[name][4] else np.dtype(rdt)
                h, o = handlers_for(name, shape[axis], rdt, L)
                y1 = np.zeros(sout, odt); y2 = np.zeros(sout, odt); yo = np.zeros(sout, odt)
                OPS[name][0](x, y1, h, axis)                               # single device
                getattr(api, name + "_par")(x, y2, h, axis)                # sharded
                assert L.last_path().startswith("sharded:"), L.last_path()
                assert np.array_equal(y1, y2), (name, shape, axis)
                OPS[name][1](x, yo, o, axis)
                assert_close(y2, yo, axis, TOL[np.dtype(rdt)], f"sharded {name} {shape}")
        # views: reversed input, output with holes; holes must survive
        x = synth.complex_array((9, 16, 6)); h = handlers.FftHandler(16, _library=L); o = orc.FftHandler(16)
        big = np.full((9, 16, 12), 2.5 + 0j); ref = big.copy()
        api.ndfft_par(x[::-1, :, :], big[:, :, ::2], h, 1); orc.ndfft(x[::-1, :, :], ref[:, :, ::2], o, 1)
        assert_close(big, ref, 1, 1e-10, "sharded strided views"); assert np.all(big[:, :, 1::2] == 2.5)
        # F-layout: the outermost dimension in memory is the LAST index
        xf = np.asfortranarray(synth.complex_array((16, 10))); yf = np.zeros((16, 10), np.complex128, order="F"); yo = np.zeros((16, 10), np.complex128)
        api.ndfft_par(xf, yf, h, 0); orc.ndfft(np.ascontiguousarray(xf), yo, o, 0); assert_close(yf, yo, 0, 1e-10, "sharded F layout")
        # several host threads issue sharded calls at once: their blocks interleave on the per-device workers
        import threading
        xs = [synth.complex_array((64, 128), offset=1000 * k) for k in range(4)]
        ys = [np.zeros((64, 128), np.complex128) for _ in range(4)]
        h128 = handlers.FftHandler(128, _library=L); errs = []
        def work(k):
            try:
                for _ in range(5):
                    api.ndfft_par(xs[k], ys[k], h128, 1)
            except Exception as e:          # pragma: no cover
                errs.append(e)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        [t.start() for t in ts]; [t.join() for t in ts]
        assert not errs, errs
        for k in range(4):
            yo = np.zeros((64, 128), np.complex128); orc.ndfft(xs[k], yo, orc.FftHandler(128), 1)
            assert_close(ys[k], yo, 1, 1e-10, f"concurrent sharded calls, thread {k}")
        # the reference's panics come out before any device starts
        import pytest
        with pytest.raises(_lib.Panic, match="Size mismatch in fft, got 15 expected 16"):
            api.ndfft_par(np.zeros((4, 15), np.complex128), np.zeros((4, 15), np.complex128), h, 1)
        # bad device id
        ids = (ctypes.c_int * 2)(0, 99)
        st = L.c.ndfft_exec_sharded(h._plan, 0, ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(x.ctypes.data), 3, api._i64((9, 16, 6)), api._i64((96, 6, 1)),
                                    api._i64((9, 16, 6)), api._i64((96, 6, 1)), 1, 1, 0.0, 2, ids)
        assert st == _lib.ERR_INVALID_ARG and b"out of range" in L.c.ndfft_last_error()
        if torch_device is not None:
            import torch
            for name, shape, axis in (("ndfft", (37, 256), 1), ("ndfft_r2c", (64, 5, 3), 0), ("nddct2", (10, 33, 128), 2)):
                sin, sout = shapes_for(name, shape, axis)
                x = make_input(name, sin, np.float64)
                h, o = handlers_for(name, shape[axis], np.float64, L)
                odt = np.complex128 if OPS[name][4] else np.float64
                yo = np.zeros(sout, odt); OPS[name][1](x, yo, o, axis)
                xd = torch.from_numpy(x).to(torch_device); yd = torch.zeros(sout, dtype=torch.from_numpy(yo).dtype, device=torch_device)
                getattr(api, name + "_par")(xd, yd, h, axis)
                assert L.last_path().startswith("sharded:")
                assert_close(yd.cpu().numpy(), yo, axis, 1e-10, f"sharded device-resident {name} {shape}")
    finally:
        api.set_par_devices(None)


def dev_sharded_fft2(L, shape, root, ids, real=False):
    """examples/fft2.rs:23-27 / rfft2.rs:29-33 on the native multi-device path: axis 1, then axis 0, both through ndfft_exec_sharded_device with the
    input, the work array and the output resident on `root` -- the second pass's blocks interleave in memory (packed dense images), which is the
    re-shard of a distributed fft2 done by the library itself.  Against numpy."""
    import ctypes
    from ndrustfft_amd import api
    r, c = shape
    if real:
        x = synth.real_array((r, c)); w_shape = (r, c // 2 + 1); ref = np.fft.fft(np.fft.rfft(x, axis=1), axis=0)
        h1 = handlers.R2cFftHandler(c, _library=L); op1 = _lib.OP_R2C
    else:
        x = synth.complex_array((r, c)); w_shape = (r, c); ref = np.fft.fft2(x)
        h1 = handlers.FftHandler(c, _library=L); op1 = _lib.OP_C2C_FWD
    h0 = handlers.FftHandler(r, _library=L)
    cids = (ctypes.c_int * len(ids))(*ids)
    def strides(sh): return api._i64([int(np.prod(sh[i + 1:])) for i in range(len(sh))])
    try:
        assert L.c.ndfft_set_device(root) == 0
        dx, dw, dy = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        wbytes = int(np.prod(w_shape)) * 16
        L.check(L.c.ndfft_dev_alloc(ctypes.byref(dx), x.nbytes)); L.check(L.c.ndfft_dev_alloc(ctypes.byref(dw), wbytes)); L.check(L.c.ndfft_dev_alloc(ctypes.byref(dy), wbytes))
        L.check(L.c.ndfft_dev_upload(dx, ctypes.c_void_p(x.ctypes.data), x.nbytes))
        L.check(L.c.ndfft_exec_sharded_device(h1._plan, op1, dx, dw, 2, api._i64(shape), strides(shape), api._i64(w_shape), strides(w_shape), 1, _lib.NORM_DEFAULT, 0.0, len(ids), cids, None))
        assert L.last_path().startswith("sharded:"), L.last_path()
        L.check(L.c.ndfft_exec_sharded_device(h0._plan, _lib.OP_C2C_FWD, dw, dy, 2, api._i64(w_shape), strides(w_shape), api._i64(w_shape), strides(w_shape), 0, _lib.NORM_DEFAULT, 0.0, len(ids), cids, None))
        got = np.empty(w_shape, np.complex128)
        L.check(L.c.ndfft_dev_download(ctypes.c_void_p(got.ctypes.data), dy, wbytes))
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), f"sharded fft2 {shape} real={real}"
        for p in (dx, dw, dy): L.check(L.c.ndfft_dev_free(p))
    finally:
        L.c.ndfft_set_device(0)


def dev_sharded_case(L, name, shape, axis, root, ids, rdt=np.float64, out_view=None, in_view=None, repeats=1, sentinel=7.25):
    """One ndfft_exec_sharded_device call on arrays resident on fake device `root`, blocks on `ids`.  out_view / in_view: (alloc_shape, index) --
    the array is a view into a larger allocation (holes); every element outside the view must keep the sentinel."""
    import ctypes
    from ndrustfft_amd import api
    sin, sout = shapes_for(name, shape, axis)
    x = make_input(name, sin, rdt)
    odt = cdt_of(rdt) if OPS[name][4] else np.dtype(rdt)
    h, o = handlers_for(name, shape[axis], rdt, L)
    yo = np.zeros(sout, odt); OPS[name][1](x, yo, o, axis)
    # host images of the two allocations
    if in_view is None: xa = np.ascontiguousarray(x); xv = xa
    else:
        xa = np.full(in_view[0], sentinel, x.dtype); xv = xa[in_view[1]]; assert xv.shape == x.shape; xv[...] = x
    if out_view is None: ya = np.full(sout, sentinel, odt); yv = ya
    else:
        ya = np.full(out_view[0], sentinel, odt); yv = ya[out_view[1]]; assert yv.shape == tuple(sout)
    opcode = {"ndfft": _lib.OP_C2C_FWD, "ndifft": _lib.OP_C2C_INV, "ndfft_r2c": _lib.OP_R2C, "ndifft_r2c": _lib.OP_C2R, "nddct1": _lib.OP_DCT1,
              "nddct2": _lib.OP_DCT2, "nddct3": _lib.OP_DCT3, "nddct4": _lib.OP_DCT4}[name]
    def off(v, a): return (v.__array_interface__["data"][0] - a.__array_interface__["data"][0])
    try:
        assert L.c.ndfft_set_device(root) == 0
        din, dout = ctypes.c_void_p(), ctypes.c_void_p()
        L.check(L.c.ndfft_dev_alloc(ctypes.byref(din), xa.nbytes)); L.check(L.c.ndfft_dev_alloc(ctypes.byref(dout), ya.nbytes))
        L.check(L.c.ndfft_dev_upload(din, ctypes.c_void_p(xa.ctypes.data), xa.nbytes))
        cids = (ctypes.c_int * len(ids))(*ids)
        for rep in range(repeats):
            L.check(L.c.ndfft_dev_upload(dout, ctypes.c_void_p(ya.ctypes.data), ya.nbytes))      # sentinel everywhere
            L.check(L.c.ndfft_exec_sharded_device(
                h._plan, opcode, ctypes.c_void_p(din.value + off(xv, xa)), ctypes.c_void_p(dout.value + off(yv, ya)), len(sin),
                api._i64(sin), api._i64([s // xv.itemsize for s in xv.strides]), api._i64(sout), api._i64([s // yv.itemsize for s in yv.strides]),
                axis, _lib.NORM_DEFAULT, 0.0, len(ids), cids, None))
            assert L.last_path().startswith("sharded:"), L.last_path()
            got = np.empty_like(ya)
            L.check(L.c.ndfft_dev_download(ctypes.c_void_p(got.ctypes.data), dout, ya.nbytes))
            gv = got if out_view is None else got[out_view[1]]
            assert_close(gv, yo, axis, TOL[np.dtype(rdt)], f"sharded device-resident {name} {shape} axis {axis} rep {rep}")
            if out_view is not None:
                mask = np.ones(out_view[0], bool); mask[out_view[1]] = False
                assert np.all(got[mask] == sentinel), "an element outside the output view was written"
        L.check(L.c.ndfft_dev_free(din)); L.check(L.c.ndfft_dev_free(dout))
    finally:
        L.c.ndfft_set_device(0)



def fuzz(L, seed, count, max_points=1 << 17, lengths=None):
    """Random op x lane length x shape x axis x dtype x norm x layout (C / F / stepped and reversed views, padded
    output views) against the oracle.  The lane lengths mix every dispatch class: powers of two, smooth,
    partial-round smooth, prime / Bluestein, DCT-I n - 1 classes, tiny and long."""
    rng = np.random.default_rng(seed)
    lengths = lengths or (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 21, 23, 24, 29, 30, 31, 32, 33, 34, 36, 40, 45, 48, 56, 60,
                          62, 63, 64, 65, 72, 96, 97, 100, 127, 128, 129, 210, 243, 256,
                          257, 264, 343, 500, 512, 513, 1000, 1009, 1024, 1025, 2048, 2187, 3000, 4096, 4097, 5000, 8192)
    names = list(OPS)
    paths = {}
    for it in range(count):
        name = names[rng.integers(len(names))]
        n = int(lengths[rng.integers(len(lengths))])
        if name == "nddct1" and n < 2:
            n = 2
        rdt = (np.float64, np.float32)[rng.integers(2)]
        ndim = int(rng.integers(1, 4))
        axis = int(rng.integers(ndim))
        other = max(1, max_points // max(n, 1))
        shape = [0] * ndim
        for d in range(ndim):
            if d == axis:
                shape[d] = n
            else:
                cap = max(1, int(round(other ** (1.0 / max(ndim - 1, 1)))))
                shape[d] = int(rng.integers(1, min(cap, 300) + 1))
        norm = ("Default", "None")[rng.integers(2)]
        sin, sout = shapes_for(name, tuple(shape), axis)
        view = int(rng.integers(4))          # 0 C, 1 F, 2 stepped / reversed input view, 3 padded output view
        x = make_input(name, sin, rdt, offset=it)
        odt = cdt_of(rdt) if OPS[name][4] else np.dtype(rdt)
        if view == 1:
            x = np.asfortranarray(x)
        if view == 2:
            big = make_input(name, tuple(2 * e for e in sin), rdt, offset=it)
            x = big[tuple(slice(None, None, -2) if rng.integers(2) else slice(0, None, 2) for _ in sin)]
        if view == 3:
            ybig = np.full(tuple(e + 2 for e in sout), 3.25, odt); y = ybig[tuple(slice(1, e + 1) for e in sout)]
        else:
            y = np.zeros(sout, odt, order="F" if view == 1 else "C")
        yo = np.zeros(sout, odt)
        h, o = handlers_for(name, n, rdt, L, norm)
        OPS[name][0](x, y, h, axis)
        path = L.last_path()
        OPS[name][1](np.ascontiguousarray(x), yo, o, axis)
        assert_close(np.ascontiguousarray(y), yo, axis, TOL[np.dtype(rdt)], f"fuzz #{it} seed={seed} {name} shape={shape} axis={axis} {np.dtype(rdt)} norm={norm} view={view} path={path}")
        if view == 3:
            edge = ybig.copy(); edge[tuple(slice(1, e + 1) for e in sout)] = 3.25
            assert np.all(edge == 3.25), f"fuzz #{it}: wrote outside the output view ({name} {shape} axis={axis} path={path})"
        paths[path] = paths.get(path, 0) + 1
    return paths


# ---- normalisation: None / Default / Custom at the reference's three application points --------
def normalization_modes(L):
    for name in OPS:
        for norm in ("None", "Default"):
            run_case(L, name, (4, 12), 1, np.float64, norm=norm)
            run_case(L, name, (4, 9), 1, np.float32, norm=norm)

    def triple(lane):
        lane *= 3.0

    n = 6
    for name in OPS:
        fn, ofn, cls, in_c, out_c = OPS[name]
        sin, sout = shapes_for(name, (3, n), 1)
        x = make_input(name, sin, np.float64)
        y = np.zeros(sout, np.complex128 if out_c else np.float64); yo = np.zeros_like(y)
        h = getattr(handlers, cls)(n, _library=L).normalization(Normalization.custom(triple))
        o = getattr(orc, cls)(n).normalization(orc.NORM_CUSTOM, triple)
        fn(x, y, h, 1); ofn(x, yo, o, 1)
        assert_close(y, yo, 1, 1e-10, f"custom norm {name}")


# ---- restated panics ---------------------------------------------------------------------------------
def panics(L):
    import pytest
    x = np.zeros((3, 5), np.complex128); y = np.zeros((3, 5), np.complex128)
    with pytest.raises(_lib.Panic, match="Size mismatch in fft, got 5 expected 6"):
        api.ndfft(x, y, handlers.FftHandler(6, _library=L), 1)
    with pytest.raises(_lib.Panic, match="Size mismatch in dct, got 5 expected 4"):
        api.nddct1(np.zeros((3, 5)), np.zeros((3, 5)), handlers.DctHandler(4, _library=L), 1)
    with pytest.raises(_lib.Panic) as e:
        api.ndfft(x, y, handlers.FftHandler(5, _library=L), 2)
    assert e.value.status == _lib.ERR_AXIS
    with pytest.raises(_lib.Panic) as e:
        api.ndfft(x, np.zeros((4, 5), np.complex128), handlers.FftHandler(5, _library=L), 1)
    assert e.value.status == _lib.ERR_SHAPE_MISMATCH
    with pytest.raises(_lib.Panic, match="Size mismatch in fft, got 6 expected 4"):
        api.ndfft_r2c(np.zeros((2, 6)), np.zeros((2, 6), np.complex128), handlers.R2cFftHandler(6, _library=L), 1)
    # no lanes -> nothing runs -> no panic even with the wrong handler length
    api.ndfft(np.zeros((0, 5), np.complex128), np.zeros((0, 5), np.complex128), handlers.FftHandler(6, _library=L), 1)
    with pytest.raises(TypeError):
        api.ndfft(x, y, handlers.DctHandler(5, _library=L), 1)
    with pytest.raises(TypeError):
        api.ndfft(x.astype(np.complex64), y, handlers.FftHandler(5, _library=L), 1)


# ---- sizes: every kernel family --------------------------------------------------------------------
SIZE_SWEEP = [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 13, 16, 17, 25, 27, 32, 49, 60, 64, 74, 96, 121, 128, 169, 210, 256, 264,
              265, 343, 512, 513, 1000, 1024, 2048]


def size_sweep(L, n, rdt=np.float64):
    for name in OPS:
        run_case(L, name, (3, n), 1, rdt, offset=n)


def reference_bench_shapes(L, sizes_fft=(128, 264, 512, 1024), sizes_dct=(129, 265, 513, 1025)):
    """benches/ndrustfft.rs:6-7: n x n f64, axis 0 (strategy ii), fill re = im = flat index."""
    for n in sizes_fft:
        x = synth.bench_fill_complex((n, n)); y = np.zeros_like(x); yo = np.zeros_like(x)
        api.ndfft(x, y, handlers.FftHandler(n, _library=L), 0); orc.ndfft(x, yo, orc.FftHandler(n), 0)
        assert_close(y, yo, 0, 1e-10, f"bench fft2d n={n}")
        xr = np.arange(n * n, dtype=np.float64).reshape(n, n); m = n // 2 + 1
        yr = np.zeros((m, n), np.complex128); yro = np.zeros_like(yr)
        api.ndfft_r2c(xr, yr, handlers.R2cFftHandler(n, _library=L), 0); orc.ndfft_r2c(xr, yro, orc.R2cFftHandler(n), 0)
        assert_close(yr, yro, 0, 1e-10, f"bench rfft2d n={n}")
    for n in sizes_dct:
        xr = np.arange(n * n, dtype=np.float64).reshape(n, n); y = np.zeros_like(xr); yo = np.zeros_like(xr)
        api.nddct1(xr, y, handlers.DctHandler(n, _library=L), 0); orc.nddct1(xr, yo, orc.DctHandler(n), 0)
        assert_close(y, yo, 0, 1e-10, f"bench dct2d n={n}")


def long_strided_lanes(L):
    """Strategy (ii) with lanes too long for an LDS tile of adjacent lanes: the transpose route."""
    cases = (("ndfft", (4096, 24), 0, np.float64, "transpose+pow2_reg"), ("ndifft", (4096, 24), 0, np.float64, "transpose+pow2_reg"),
             ("ndfft_r2c", (8192, 40), 0, np.float32, "transpose+pow2_real"), ("ndifft_r2c", (8192, 20), 0, np.float32, "transpose+pow2_real"),
             ("nddct2", (3, 4096, 17), 1, np.float64, "transpose+pow2_real"), ("ndfft", (3000, 33), 0, np.float64, "transpose+generic_row"),
             ("nddct3", (1, 4000, 16), 1, np.float32, "transpose+generic_row"))      # (16 lanes x 2000 points: below the 2^15-point threshold of the specialised kernels)
    with switches(L, NDFFT_COLSPLIT="0"):       # (the column four-step would take the first four)
        for name, shape, axis, rdt, want in cases:
            assert run_case(L, name, shape, axis, rdt) == want, (name, shape)


def pow2_real_sizes(L, sizes=(64, 128, 256, 512, 1024, 2048, 4096, 8192), dtypes=(np.float64, np.float32)):
    """The register-resident real-op kernels: every op x every supported inner FFT length F = n/2
    (DCT-I: n = F + 1), both norms."""
    for F in sizes:
        for rdt in dtypes:
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3", "nddct4"):
                for norm in ("Default", "None"):
                    assert run_case(L, name, (5, 2 * F), 1, rdt, norm=norm, offset=F) == "pow2_real", (name, F)
            assert run_case(L, "nddct1", (5, F + 1), 1, rdt, offset=F) == "pow2_real", ("nddct1", F)


def long_lanes_padded_views(L):
    """The two-pass long-lane routes (real four-step both directions, fused DCT-IV) on views with an offset and a pitch larger than the lane, odd in elements:
    results against numpy / scipy, and every element outside the output view keeps its sentinel."""
    import scipy.fft
    n = 1 << 16
    rng = np.random.default_rng(5)
    for rdt, tol in ((np.float64, 1e-10), (np.float32, 2e-4)):
        cdt = cdt_of(rdt)
        big = rng.standard_normal((3, n + 11)).astype(rdt); x = big[:, 3:3 + n]
        ybig = np.full((3, n + 7), 7.5, rdt); y = ybig[:, 5:5 + n]
        hd = handlers.DctHandler(n, rdt, _library=L)
        for name, fn, t in (("nddct2", api.nddct2, 2), ("nddct3", api.nddct3, 3), ("nddct4", api.nddct4, 4)):
            ybig[:] = 7.5; fn(x, y, hd, 1)
            assert L.last_path() == "real_four_step", (name, L.last_path())
            ref = scipy.fft.dct(x.astype(np.float64), type=t, axis=1)
            err = np.abs(y - ref).max() / np.abs(ref).max()
            assert err < tol and np.all(ybig[:, :5] == 7.5) and np.all(ybig[:, 5 + n:] == 7.5), (name, err)
        hr = handlers.R2cFftHandler(n, rdt, _library=L)
        cbig = np.full((3, n // 2 + 1 + 5), 1.5 + 2.5j, cdt); c = cbig[:, 2:2 + n // 2 + 1]
        api.ndfft_r2c(x, c, hr, 1)
        assert L.last_path() == "real_four_step", L.last_path()
        ref = np.fft.rfft(x.astype(np.float64), axis=1); err = np.abs(c - ref).max() / np.abs(ref).max()
        assert err < tol and np.all(cbig[:, :2] == 1.5 + 2.5j) and np.all(cbig[:, 2 + n // 2 + 1:] == 1.5 + 2.5j), err
        ybig[:] = 7.5; api.ndifft_r2c(c, y, hr, 1)
        assert L.last_path() == "real_four_step", L.last_path()
        err = np.abs(y - x).max() / np.abs(x).max()
        assert err < tol and np.all(ybig[:, :5] == 7.5) and np.all(ybig[:, 5 + n:] == 7.5), err


def long_lanes_four_step(L, full=True):
    """Lanes longer than one workgroup's LDS: four-step on the row kernels (any op, C2C inverse scaling,
    non-power-of-two splits, strided axis through the transpose route); and the documented refusal."""
    import pytest
    # (round 6: the FORWARD real ops of a smooth lane length divisible by 4 take the real four-step with hiprtc passes -- "real_four_step" -- where hiprtc exists; the CPU emulation has
    #  none and keeps the packed route, "four_step")
    smooth_fwd = ("real_four_step", "four_step")
    cases = [("ndfft", (2, 32768), 1, np.float64, "four_step"), ("ndifft", (2, 32768), 1, np.float64, "four_step"),
             ("ndfft", (3, 6000), 1, np.float64, "four_step"), ("ndfft_r2c", (2, 20000), 1, np.float64, smooth_fwd),
             ("ndifft_r2c", (2, 20000), 1, np.float64, smooth_fwd), ("nddct2", (2, 12000), 1, np.float64, smooth_fwd),
             ("nddct3", (2, 12000), 1, np.float64, smooth_fwd), ("nddct1", (2, 10001), 1, np.float64, smooth_fwd),
             ("nddct4", (2, 12000), 1, np.float64, smooth_fwd), ("ndfft", (2, 65536), 1, np.float32, "four_step"),
             ("ndfft", (20000, 3), 0, np.float64, "transpose+four_step"), ("ndfft_r2c", (2, 9999), 1, np.float64, "four_step"),
             ("nddct2", (2, 9999), 1, np.float32, "four_step"),
             # R2C without its PRE pass / C2R without its POST pass (the real lane addressed as complex), two-pass power-of-two route
             ("ndfft_r2c", (2, 1 << 17), 1, np.float32, "real_four_step"), ("ndifft_r2c", (3, 1 << 17), 1, np.float64, "real_four_step")]
    if full:
        cases += [("ndfft", (2, 1 << 20), 1, np.float64, "four_step"), ("ndfft", (2, 1 << 20), 1, np.float32, "four_step"), ("ndifft", (3, 1 << 19), 1, np.float32, "four_step"), ("ndifft", (1, 1 << 22), 1, np.float32, "four_step"),
                  ("nddct2", (3, 1 << 18), 1, np.float64, "real_four_step"), ("ndfft_r2c", (2, 3 * (1 << 17)), 1, np.float32, smooth_fwd)]
    if full:   # round 6: smooth NON-power-of-two factors run the two four-step passes on hiprtc-specialised kernels (one factor or both); the inverse real ops and DCT-IV through the packed route,
               # the forward real ops (R2C, DCT-II, DCT-I with 2 (n - 1) smooth) on the REAL four-step with hiprtc passes
        cases += [("ndfft", (2, 196608), 1, np.float64, "four_step"), ("ndifft", (3, 163840), 1, np.float32, "four_step"), ("ndfft", (2, 200000), 1, np.float64, "four_step"),
                  ("ndifft", (2, 147456), 1, np.float64, "four_step"), ("ndfft", (5, 100000), 1, np.float32, "four_step"), ("nddct2", (2, 196608), 1, np.float64, "real_four_step"),
                  ("ndfft_r2c", (2, 163840), 1, np.float32, "real_four_step"), ("ndifft_r2c", (2, 200000), 1, np.float64, smooth_fwd), ("nddct1", (2, 147457), 1, np.float64, "real_four_step"),
                  ("nddct4", (2, 196608), 1, np.float32, smooth_fwd), ("nddct3", (2, 120000), 1, np.float64, smooth_fwd),
                  ("nddct4", (3, 163840), 1, np.float64, "real_four_step"), ("nddct4", (2, 200000), 1, np.float64, smooth_fwd), ("nddct4", (2, 524160), 1, np.float32, smooth_fwd),
                  ("ndfft_r2c", (3, 200000), 1, np.float64, "real_four_step"), ("nddct2", (2, 120000), 1, np.float32, "real_four_step"), ("nddct1", (3, 196609), 1, np.float32, "real_four_step"),
                  ("ndfft_r2c", (2, 524160), 1, np.float64, "real_four_step"), ("nddct2", (5, 147456), 1, np.float64, "real_four_step"),
                  # the inverse direction where the second factor has a power-of-two or whole-round recipe (else the packed route)
                  ("ndifft_r2c", (2, 196608), 1, np.float64, "real_four_step"), ("nddct3", (3, 163840), 1, np.float32, smooth_fwd), ("ndifft_r2c", (3, 200000), 1, np.float32, smooth_fwd),
                  ("nddct3", (2, 147456), 1, np.float64, smooth_fwd), ("nddct3", (2, 200000), 1, np.float64, smooth_fwd), ("ndifft_r2c", (2, 524160), 1, np.float64, smooth_fwd),
                  ("ndifft_r2c", (5, 147456), 1, np.float64, "real_four_step"), ("nddct3", (2, 98304), 1, np.float32, smooth_fwd),
                  # factors without a whole-round recipe (7 / 11 / 13, 675 = 5.5.3.3.3): partial rounds in both passes
                  ("ndfft", (2, 524160), 1, np.float64, "four_step"), ("ndifft", (3, 128700), 1, np.float32, "four_step"), ("ndfft", (2, 394875), 1, np.float64, "four_step"),
                  ("nddct2", (2, 240570), 1, np.float32, "four_step")]
    for name, shape, axis, rdt, want in cases:
        got = run_case(L, name, shape, axis, rdt)
        assert got == want or (isinstance(want, tuple) and got in want), (name, shape, got)
    # REAL four-step (round 3): R2C (f64) and DCT-II of power-of-two lanes in two passes -- real FFTs of length N1 over the strided index, row store of the
    # half spectrum, then twiddled complex FFTs of length N2 writing X[k] / conj at the mirrored index (DCT-II: y[k], y[n-k]); staged and lane-fastest
    # kernels for pass 2, the packed route it replaces, and f32 R2C forced through it
    def with_env(env, fn):
        with switches(L, **env):
            return fn()
    for norm in ("Default", "None"):
        for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3"):
            assert run_case(L, name, (2, 1 << 16), 1, np.float64, norm=norm) == "real_four_step", name
            assert run_case(L, name, (3, 1 << 17), 1, np.float32, norm=norm) == "real_four_step", name
    for direct in ("0", "1"):   # pass 2 of the forward direction on the staged / the lane-fastest kernels; the same switch for the complex four-step
        for name, shape, rdt in (("ndfft_r2c", (3, 1 << 16), np.float64), ("nddct2", (2, 1 << 16), np.float64), ("nddct2", (1, 1 << 17), np.float32), ("ndfft_r2c", (2, 1 << 17), np.float32),
                                 ("ndfft", (2, 32768), np.float64), ("ndifft", (3, 65536), np.float32)):
            want = "four_step" if name in ("ndfft", "ndifft") else "real_four_step"
            assert with_env({"NDFFT_FS_DIRECT": direct}, lambda: run_case(L, name, shape, 1, rdt)) == want, (name, shape, direct)
    for name, rdt in (("ndfft_r2c", np.float64), ("nddct2", np.float64), ("nddct2", np.float32), ("ndifft_r2c", np.float64), ("nddct3", np.float32)):   # the packed route stays covered
        assert with_env({"NDFFT_REAL_FOURSTEP": "0"}, lambda: run_case(L, name, (2, 1 << 16 if rdt is np.float64 else 1 << 17), 1, rdt)) == "four_step"
    for name, rdt in (("ndifft_r2c", np.float64), ("nddct3", np.float32)):   # last pass of the inverse direction through dispatch() (general column kernel / column four-step)
        assert with_env({"NDFFT_RFS_C2R_TILE": "0"}, lambda: run_case(L, name, (3, 1 << 16 if rdt is np.float64 else 1 << 17), 1, rdt)) == "real_four_step"
    # a lane count that leaves a partial tile in every pass, an output pitch larger than the lane, DCT types without a real four-step (I, IV) on the same length
    for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3"):
        assert run_case(L, name, (5, 1 << 16), 1, np.float64, offset=3) == "real_four_step", name
    # DCT-IV: the complex four-step of length n/2 with its fold built by pass 1's load and its outputs written by pass 2's store (two passes instead of four)
    for norm in ("Default", "None"):
        assert run_case(L, "nddct4", (3, 1 << 16), 1, np.float64, norm=norm) == "real_four_step"
        assert run_case(L, "nddct4", (2, 1 << 17), 1, np.float32, norm=norm) == "real_four_step"
    assert with_env({"NDFFT_FS_DIRECT": "1"}, lambda: run_case(L, "nddct4", (2, 1 << 16), 1, np.float64)) == "real_four_step"
    assert with_env({"NDFFT_REAL_FOURSTEP": "0"}, lambda: run_case(L, "nddct4", (2, 1 << 16), 1, np.float64)) == "four_step"
    # DCT-I with n - 1 a power of two (round 5): the real four-step on the even extension, 2 (n - 1) = N1 N2; NDFFT_REAL_FOURSTEP=0 keeps the packed route
    for norm in ("Default", "None"):
        assert run_case(L, "nddct1", (2, (1 << 16) + 1), 1, np.float64, norm=norm) == "real_four_step"
    assert run_case(L, "nddct1", (3, (1 << 16) + 1), 1, np.float32, offset=5) == "real_four_step"
    assert with_env({"NDFFT_REAL_FOURSTEP": "0"}, lambda: run_case(L, "nddct1", (2, (1 << 16) + 1), 1, np.float64)) == "four_step"
    if full:
        for a in ("8", "10", "11"):   # other splits n = N1 * N2 (developer knob of the plan)
            for name in ("nddct2", "nddct3", "ndfft_r2c", "ndifft_r2c"):
                assert with_env({"NDFFT_RFS_LOGN1": a}, lambda: run_case(L, name, (2, 1 << 18), 1, np.float64)) == "real_four_step", (name, a)
        assert run_case(L, "ndfft_r2c", (5, 1 << 21), 1, np.float64) == "real_four_step"
        assert run_case(L, "ndifft_r2c", (3, 1 << 20), 1, np.float32) == "real_four_step"
        # f32 at 2^21: the plan's table keeps DCT-II / R2C / C2R on the packed route (measured faster), DCT-III on the real four-step; 2 forces it
        assert run_case(L, "nddct2", (2, 1 << 21), 1, np.float32) == "four_step"
        assert run_case(L, "nddct3", (2, 1 << 21), 1, np.float32) == "real_four_step"
        assert run_case(L, "ndfft_r2c", (2, 1 << 20), 1, np.float32) == "four_step"
        for name in ("nddct2", "ndfft_r2c", "ndifft_r2c"):
            assert with_env({"NDFFT_REAL_FOURSTEP": "2"}, lambda: run_case(L, name, (2, 1 << 21), 1, np.float32)) == "real_four_step", name
        assert run_case(L, "nddct3", (2, 1 << 19), 1, np.float64) == "real_four_step"
        assert run_case(L, "nddct4", (2, 1 << 21), 1, np.float64) == "real_four_step"
        assert run_case(L, "nddct4", (130, 1 << 16), 1, np.float32) == "real_four_step"
        assert run_case(L, "nddct2", (130, 1 << 16), 1, np.float64) == "real_four_step"     # more lanes than one group of XCD runs
        assert run_case(L, "nddct3", (130, 1 << 16), 1, np.float32) == "real_four_step"
    # the power-of-two lengths above took the two-pass form (column load / row store, then twiddled column pass); the
    # three-pass form they replace stays covered, and both directions / norms of the new one on an odd lane count
    assert run_case(L, "ndifft_r2c", (2, 1 << 16), 1, np.float32, norm="None") == "real_four_step"
    for name in ("ndfft", "ndifft"):
        for norm in ("Default", "None"):
            assert run_case(L, name, (3, 32768), 1, np.float64, norm=norm) == "four_step"
            assert run_case(L, name, (5, 65536), 1, np.float32, norm=norm) == "four_step"
    with switches(L, NDFFT_FOURSTEP2="0"):
        assert run_case(L, "ndfft", (2, 32768), 1, np.float64) == "four_step"
        assert run_case(L, "ndifft", (2, 65536), 1, np.float32) == "four_step"
    # long lanes in an arbitrary strided layout (stepped + reversed input view, padded output view): pack -> rows -> unpack
    n = 1 << 15
    big = synth.complex_array((3, 2 * n)); x = big[::-1, ::2]
    ybig = np.full((3, n + 4), 1.5 + 0j); y = ybig[:, 2:n + 2]
    h = handlers.FftHandler(n, _library=L); api.ndfft(x, y, h, 1)
    assert L.last_path() == "pack+four_step", L.last_path()
    assert_close(np.ascontiguousarray(y), np.fft.fft(np.ascontiguousarray(x), axis=1), 1, 1e-10, "packed long lanes")
    assert np.all(ybig[:, :2] == 1.5) and np.all(ybig[:, n + 2:] == 1.5)
    # a huge prime factor with no usable split: Bluestein over global memory (see huge_prime_factors)
    assert run_case(L, "ndfft", (2, 2 * 10007), 1, np.float64) == "blue_global"
    # ... beyond M = 2^21 it is refused, never computed on a CPU
    if full:
        with pytest.raises(_lib.NdfftError) as e:
            h = handlers.FftHandler(1048583, np.float32, _library=L)          # prime; M would be 2^22
            x = np.zeros((1, 1048583), np.complex64); api.ndfft(x, np.zeros_like(x), h, 1)
        assert e.value.status == _lib.ERR_UNSUPPORTED


def pow2_col_sizes(L, sizes=(64, 128, 256, 512, 1024), dtypes=(np.float64, np.float32)):
    """Strategy (ii) on the register engine: a strided axis whose adjacent lanes are contiguous, tile of
    adjacent lanes staged through LDS.  Every op (incl. C2C), 2-D axis 0 and 3-D middle axis, ragged tiles."""
    for F in sizes:
        for rdt in dtypes:
            for name in OPS:
                n = F if name in ("ndfft", "ndifft") else (F + 1 if name == "nddct1" else 2 * F)
                for shape, axis in (((n, 40), 0), ((3, n, 17), 1), ((n, 9), 0)):
                    # (round 3: f32 C2R lanes of 2048 points take the column four-step when the tile can be >= 16 lanes wide; the one-pass tile stays covered)
                    split = name == "ndifft_r2c" and np.dtype(rdt) == np.float32 and n == 2048 and shape[-1] >= 16
                    assert run_case(L, name, shape, axis, rdt, offset=F) == ("col_split" if split else "pow2_col"), (name, shape)
                    if split:
                        with switches(L, NDFFT_COLSPLIT="0"):
                            assert run_case(L, name, shape, axis, rdt, offset=F) == "pow2_col", (name, shape)


def shared_handler_across_threads(L, nthreads=8):
    """The reference shares &handler across rayon workers (lib.rs:192-194): one plan, many host threads."""
    import threading
    h = handlers.FftHandler(256, _library=L); hd = handlers.DctHandler(100, _library=L)
    errs = []

    def work(i):
        try:
            for rep in range(4):
                x = synth.complex_array((16, 256), offset=1000 * i + rep); y = np.zeros_like(x)
                api.ndfft(x, y, h, 1)
                assert_close(y, np.fft.fft(x, axis=1), 1, 1e-10, f"thread {i}")
                xr = synth.real_array((7, 100), offset=77 * i + rep); yr = np.zeros_like(xr); yo = np.zeros_like(xr)
                api.nddct2(xr, yr, hd, 1); orc.nddct2(xr, yo, orc.DctHandler(100), 1)
                assert_close(yr, yo, 1, 1e-10, f"thread {i} dct")
        except Exception as e:  # pragma: no cover
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs


def narrow_xcd_tiles(L):
    """Long strided lanes (inner FFT 2048..8192) with >= 64 adjacent lanes: XCD-aware narrow column tiles,
    incl. ragged tails, 3-D outer dims and every op family."""
    cases = (("ndfft_r2c", (8192, 128), 0, np.float32), ("ndifft_r2c", (8192, 64), 0, np.float32), ("ndfft", (4096, 72), 0, np.float32),
             ("ndfft", (4096, 64), 0, np.float64), ("ndifft", (2048, 200), 0, np.float32),
             ("ndfft", (8192, 64), 0, np.float32), ("ndfft_r2c", (16384, 66), 0, np.float32))
    # (the DCTs left the narrow tiles in round 3: transpose -> rows -> transpose measured faster; their kernels stay compiled and are exercised through NDFFT_NARROW_DCT=1)
    dct_cases = (("nddct2", (2, 4096, 130), 1, np.float64), ("nddct1", (4097, 64), 0, np.float32), ("nddct3", (4096, 1000), 0, np.float64), ("nddct4", (8192, 96), 0, np.float32))
    for name, shape, axis, rdt in dct_cases:
        assert run_case(L, name, shape, axis, rdt).startswith("transpose+"), (name, shape)
    with switches(L, NDFFT_NARROW_DCT="1"):
        for name, shape, axis, rdt in dct_cases:
            assert run_case(L, name, shape, axis, rdt) == "pow2_col_xcd", (name, shape)
    with switches(L, NDFFT_COLSPLIT="0"):       # the column four-step would take the C2C / R2C / C2R cases
        for name, shape, axis, rdt in cases:
            assert run_case(L, name, shape, axis, rdt) == "pow2_col_xcd", (name, shape)
    # round 3: f64 lanes whose OUTPUT rows are complex take ordinary column tiles of 4 lanes (64-byte rows) at inner lengths 1024 / 2048 instead
    # (c128 n = 2048: narrow tiles 216 us -> 157 us; f64 R2C n = 4096: 132 -> 83 us); real output rows keep the 8-lane minimum
    for name, shape, axis, rdt, want in (("ndifft", (2048, 200), 0, np.float64, "pow2_col"), ("ndfft", (3, 2048, 10), 1, np.float64, "pow2_col"), ("ndfft", (1024, 21), 0, np.float64, "pow2_col"),
                                         ("ndfft_r2c", (4096, 70), 0, np.float64, "pow2_col"), ("ndfft_r2c", (2048, 21), 0, np.float64, "pow2_col"),
                                         ("ndifft_r2c", (4096, 64), 0, np.float64, "pow2_col_xcd"), ("nddct2", (4096, 64), 0, np.float64, "transpose+pow2_real")):
        assert run_case(L, name, shape, axis, rdt) == want, (name, shape)


def column_four_step(L):
    """Long strided power-of-two lanes on a dense C-layout block: two passes of wide column tiles
    (C2C n >= 4096, R2C / C2R n >= 8192), incl. 3-D outer dims, ragged inner extents and both norms."""
    cases = (("ndfft", (4096, 48), 0, np.float64, "Default"), ("ndifft", (4096, 40), 0, np.float32, "Default"),
             ("ndifft", (4096, 16), 0, np.float64, "None"), ("ndfft", (8192, 33), 0, np.float32, "Default"),
             ("ndfft", (2, 4096, 24), 1, np.float64, "Default"), ("ndifft", (3, 4096, 17), 1, np.float32, "Default"),
             ("ndfft_r2c", (8192, 64), 0, np.float32, "Default"), ("ndfft_r2c", (8192, 21), 0, np.float64, "Default"),
             ("ndifft_r2c", (8192, 48), 0, np.float64, "Default"), ("ndifft_r2c", (8192, 40), 0, np.float32, "None"),
             ("ndfft_r2c", (2, 8192, 20), 1, np.float32, "Default"), ("ndifft_r2c", (3, 8192, 16), 1, np.float32, "Default"),
             ("ndfft_r2c", (16384, 35), 0, np.float64, "Default"), ("ndifft_r2c", (16384, 32), 0, np.float32, "Default"),
             ("ndfft", (16384, 19), 0, np.float32, "Default"),
             # rows of whole tiles (inner a multiple of 32): the stage kernels' 3-D grid (tile of i, k1, o), with and without an outer dimension (round 6)
             ("ndifft_r2c", (2, 8192, 64), 1, np.float32, "Default"), ("ndifft_r2c", (8192, 96), 0, np.float64, "Default"), ("ndfft", (3, 4096, 32), 1, np.float64, "Default"),
             ("ndfft_r2c", (2, 8192, 32), 1, np.float64, "None"), ("ndifft", (2, 4096, 64), 1, np.float32, "Default"), ("ndfft_r2c", (8192, 128), 0, np.float32, "Default"))
    for name, shape, axis, rdt, norm in cases:
        assert run_case(L, name, shape, axis, rdt, norm=norm) == "col_split", (name, shape)
    # round 3: per-op, per-dtype lower ends -- f32 C2C and C2R from n = 2048 (first factor 32), f32 R2C from 4096; f64 keeps 4096 / 8192
    for name, shape, axis, rdt, want in (("ndfft", (2048, 40), 0, np.float32, "col_split"), ("ndifft", (2048, 33), 0, np.float32, "col_split"), ("ndfft", (2, 2048, 24), 1, np.float32, "col_split"),
                                         ("ndfft_r2c", (4096, 48), 0, np.float32, "col_split"), ("ndifft_r2c", (4096, 40), 0, np.float32, "col_split"),
                                         ("ndifft_r2c", (2048, 24), 0, np.float32, "col_split"), ("ndifft_r2c", (3, 2048, 17), 1, np.float32, "col_split"),
                                         ("ndfft_r2c", (2048, 24), 0, np.float32, "pow2_col"), ("ndfft", (2048, 40), 0, np.float64, "pow2_col"), ("ndfft_r2c", (4096, 48), 0, np.float64, "pow2_col")):
        for norm in ("Default", "None"):
            assert run_case(L, name, shape, axis, rdt, norm=norm) == want, (name, shape)
    # too few adjacent lanes for a wide tile: narrow tiles / transpose route as before
    assert run_case(L, "ndfft", (4096, 8), 0, np.float64) != "col_split"
    # column chunks (Infinity-Cache-resident intermediate): force small chunks so that these shapes split
    with switches(L, NDFFT_CS_CHUNK_MB="1"):
        for name, shape, rdt in (("ndfft_r2c", (8192, 200), np.float32), ("ndifft_r2c", (8192, 136), np.float32),
                                 ("ndfft", (4096, 100), np.float64), ("ndifft", (8192, 150), np.float32)):
            assert run_case(L, name, shape, 0, rdt) == "col_split", (name, shape)


def bluestein_register_kernel(L, sizes=((17, 64), (31, 64), (97, 256), (127, 256)), col_max_M=256):
    """Lengths with a prime factor > 13 on the register-resident Bluestein kernel (blue_kernel.h): every op
    family incl. the odd-n variants, rows and column tiles.  `sizes` = (F, M): inner FFT length F, M = 2^k >= 2F-1.
    (NDFFT_RADER=0: lengths that have a Rader recipe would otherwise go to rader_kernel.h -- see rader_kernel below.)"""
    with switches(L, NDFFT_RADER="0"):
        _bluestein_register_kernel(L, sizes, col_max_M)


def odd_real_lengths(L, sizes=(45,), dct4=False):
    """Odd-n forms of R2C / C2R / DCT-II / DCT-III (/ DCT-IV: inner FFT 2n) with a smooth n on the register kernel (plain_kernel.h), rows and column tiles."""
    for n in sizes:
        rows = (1 << 17) // n + 5
        for rdt in (np.float64, np.float32):
            names = ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3") + (("nddct4",) if dct4 else ())
            for name in names:
                for norm in ("Default", "None"):
                    assert run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=n) in ("plain_real", "regreal_row"), (name, n, rdt)
                # (a tile of 8 lanes must fit LDS: longer lanes go through the transpose route to the row kernel)
                ok_col = ("plain_col", "regreal_col", "transpose+plain_real")
                assert run_case(L, name, (n, rows + 3), 0, rdt, offset=n + 1) in ok_col, (name, n, rdt)
                assert run_case(L, name, (3, n, rows // 2), 1, rdt, offset=n + 2) in ok_col, (name, n, rdt)


def rader_kernel(L, sizes=(31, 62, 97, 103, 306, 511), col_max_F=128, dtypes=(np.float64, np.float32)):
    """Inner FFT lengths F = (cofactor <= 16) x (prime p, p - 1 smooth) on the Rader / Good-Thomas register kernel
    (rader_kernel.h): every op family incl. the odd-n variants, both normalisations, rows and column tiles."""
    for F in sizes:
        rows = (1 << 17) // F + 5
        for rdt in dtypes:
            ok_row = ("rader_reg", "reg_row", "regreal_row") if F <= 96 else ("rader_reg",)
            for name in ("ndfft", "ndifft", "ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3"):       # n = F (odd-n variants when F is odd)
                if F % 2 == 0 and name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3"):
                    continue        # even n: the inner FFT is n / 2, covered below
                for norm in ("Default", "None"):
                    assert run_case(L, name, (rows, F), 1, rdt, norm=norm, offset=F) in ok_row, (name, F, rdt)
            ok2 = ("rader_reg", "regreal_row") if 2 * F <= 96 else ("rader_reg",)
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3", "nddct4"):               # even n = 2F
                assert run_case(L, name, (rows // 2, 2 * F), 1, rdt, offset=F + 1) in ok2, (name, 2 * F, rdt)
            assert run_case(L, "nddct1", (rows, F + 1), 1, rdt, offset=F) in ok_row, ("nddct1", F + 1, rdt)
            if F > col_max_F:
                continue
            ok_col = ("rader_col", "reg_col", "regreal_col") if F <= 96 else ("rader_col",)
            for name, n in (("ndfft", F), ("ndifft", F), ("ndifft_r2c", 2 * F), ("nddct1", F + 1), ("nddct2", 2 * F), ("ndfft_r2c", 2 * F), ("nddct4", 2 * F)):
                assert run_case(L, name, (n, rows + 3), 0, rdt, offset=n) in ok_col, (name, n, rdt)
                assert run_case(L, name, (3, n, rows // 2), 1, rdt, offset=n + 1) in ok_col, (name, n, rdt)


def dct1_power_of_two_lengths(L):
    """nddct1 with n a power of two: F = n - 1 = 255 (15 x 17), 511 (7 x 73), 1023 (33 x 31: cofactor 33 = 11 x 3, allowed in the DCT-I slot only), 2047 (23 x 89) -- the symmetric Rader form
    (rader_kernel.h: SYM), rows and column tiles, both precisions and normalisations."""
    for n in (256, 512, 1024, 2048):       # (2048: F = 2047 = 23 x 89 -- a radix-23 cofactor butterfly, allowed in the DCT-I slot; the column form goes through the transpose route)
        rows = (1 << 17) // n + 3
        for rdt in (np.float64, np.float32):
            for norm in ("Default", "None"):
                assert run_case(L, "nddct1", (rows, n), 1, rdt, norm=norm, offset=n) == "rader_reg", (n, rdt)
            col = ("rader_col",) if n < 2048 else ("rader_col", "transpose+rader_reg")
            assert run_case(L, "nddct1", (n, rows + 5), 0, rdt, offset=n + 1) in col, (n, rdt)
            assert run_case(L, "nddct1", (2, n, rows // 2), 1, rdt, offset=n + 2) in col, (n, rdt)


def _bluestein_register_kernel(L, sizes, col_max_M):
    for F, M in sizes:
        rows = (1 << 17) // M + 5        # (the library may pick a smooth convolution length down to M / 2: stay above its 2^16-point threshold)
        for rdt in (np.float64, np.float32):
            # C2C (n = F), odd-n real ops (inner FFT n = F; DCT-IV odd uses 2n, so it is not in this list)
            for name in ("ndfft", "ndifft", "ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3"):
                for norm in ("Default", "None"):
                    assert run_case(L, name, (rows, F), 1, rdt, norm=norm, offset=F) in (("blue_reg", "reg_row", "regreal_row") if F <= 96 else ("blue_reg",)), (name, F, rdt)
            # even-n real ops: inner FFT n/2 = F;  DCT-I: n - 1 = F
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3", "nddct4"):
                assert run_case(L, name, (rows, 2 * F), 1, rdt, offset=F + 1) in (("blue_reg", "regreal_row") if 2 * F <= 96 else ("blue_reg",)), (name, 2 * F, rdt)
            assert run_case(L, "nddct1", (rows, F + 1), 1, rdt, offset=F) in (("blue_reg", "regreal_row") if F < 96 else ("blue_reg",)), ("nddct1", F + 1, rdt)
            if M > col_max_M:
                continue
            # column tiles (strategy ii)
            for name, n in (("ndfft", F), ("ndifft_r2c", 2 * F), ("nddct1", F + 1), ("nddct2", F), ("ndfft_r2c", F)):
                # (short lanes with enough of them go to the thread-per-lane register kernels first: reg_kernel.h)
                ok = ("blue_col", "reg_col", "regreal_col") if n <= 96 else ("blue_col",)
                assert run_case(L, name, (n, rows + 3), 0, rdt, offset=n) in ok, (name, n, rdt)
                assert run_case(L, name, (3, n, rows // 2), 1, rdt, offset=n + 1) in ok, (name, n, rdt)
    # few lanes: not worth a compile
    assert run_case(L, "ndfft", (3, 97), 1, np.float64) == "generic_row"


def partial_round_configs(L, sizes=(264, 210)):
    """Smooth lengths whose radix list does not divide any E (264 = 11.8.3, 210 = 7.6.5, 840, ...): the C2C row
    kernel with PARTIAL butterfly rounds (pow2_kernel.h: slots / full)."""
    for n in sizes:
        rows = (1 << 17) // n + 7
        for rdt in (np.float64, np.float32):
            for name, norm in (("ndfft", "Default"), ("ndifft", "Default"), ("ndifft", "None")):
                # (lanes short enough for the thread-per-lane register kernel go there first: reg_kernel.h)
                assert run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=n) in (("jit_reg", "reg_row") if n <= 96 else ("jit_reg",)), (name, n, rdt)
        # padded lane pitch
        big = synth.complex_array((rows, n + 3), np.complex128); x = big[:, :n]; y = np.zeros((rows, n + 5), np.complex128)[:, 2:n + 2]
        h = handlers.FftHandler(n, _library=L); api.ndfft(x, y, h, 1)
        assert L.last_path() in (("jit_reg", "reg_row") if n <= 96 else ("jit_reg",))
        assert_close(y, np.fft.fft(x, axis=1), 1, 1e-10, f"partial-round {n} padded")
    # the real-op / column kernel with the same configurations: inner FFT F = n (R2C / DCT of length 2F, DCT-I of F + 1)
    for F in sizes[:2]:
        rows = (1 << 16) // F + 5
        for rdt in (np.float64, np.float32):
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3", "nddct4"):
                assert run_case(L, name, (rows, 2 * F), 1, rdt, offset=F) == "jit_real", (name, F, rdt)
            assert run_case(L, "nddct1", (rows, F + 1), 1, rdt, offset=F) == "jit_real", ("nddct1", F, rdt)
            for name, n in (("ndfft", F), ("ndifft", F), ("ndfft_r2c", 2 * F), ("nddct2", 2 * F), ("nddct1", F + 1)):
                assert run_case(L, name, (n, rows + 3), 0, rdt, offset=n) == "jit_col", (name, n, rdt)


def long_smooth_lanes(L):
    """Smooth non-power-of-two lanes beyond the LDS kernel's reach but within one lane's half exchange
    (n <= 19274 f64 / 32768 f32): one specialised launch instead of the four-step."""
    for n, rdt in ((10000, np.float64), (18000, np.float64), (19200, np.float64), (24576, np.float32), (30000, np.float32)):
        rows = (1 << 17) // n + 9
        for name, norm in (("ndfft", "Default"), ("ndifft", "Default")):
            assert run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=n) == "jit_reg", (name, n, rdt)
    for name, n, rdt in (("nddct2", 12000, np.float64), ("ndfft_r2c", 16000, np.float64), ("ndifft_r2c", 20000, np.float32), ("nddct4", 20000, np.float32)):
        assert run_case(L, name, ((1 << 16) // n + 9, n), 1, rdt, offset=n) == "jit_real", (name, n, rdt)


def huge_prime_factors(L, full=True):
    """Lane lengths whose prime factor is too large for any single-launch Bluestein (FftHandler::new(n) is
    infallible for any n, src/lib.rs:294): Bluestein over global memory around the power-of-two row path."""
    cases = [("ndfft", 4099, np.float64), ("ndifft", 4099, np.float64), ("nddct1", 8192, np.float64), ("ndfft_r2c", 2 * 4099, np.float64),
             ("ndifft_r2c", 2 * 4099, np.float64), ("nddct2", 4099, np.float64)]
    if full:
        cases += [("ndfft", 10007, np.float64), ("ndfft", 20011, np.float32), ("ndifft", 65537, np.float32), ("nddct4", 10007, np.float32),
                  ("nddct3", 2 * 8209, np.float64)]
    for name, n, rdt in cases:
        assert run_case(L, name, (3, n), 1, rdt, offset=n) == "blue_global", (name, n, rdt)
    assert run_case(L, "ndfft", (4099, 20), 0, np.float64) == "transpose+blue_global"


def jit_specialised_sizes(L):
    """Smooth non-power-of-two C2C lanes: the register-resident kernel specialised with hiprtc at first use."""
    for n in (96, 100, 144, 384, 500, 768, 1000, 1296, 1536, 2000, 2187, 3072, 3125):
        rows = max(4, (1 << 17) // n + 3)
        for rdt in (np.float64, np.float32):
            for name, norm in (("ndfft", "Default"), ("ndifft", "Default"), ("ndifft", "None")):
                assert run_case(L, name, (rows, n), 1, rdt, norm=norm, offset=n) == "jit_reg", (name, n, rdt)
    # odd lane pitch / unaligned base -> the scalar-access variant is compiled
    n = 1000
    big = synth.complex_array((140, n + 1), np.complex64); x = big[:, 1:]; y = np.zeros((140, n + 1), np.complex64)[:, 1:]
    h = handlers.FftHandler(n, np.float32, _library=L); api.ndfft(x, y, h, 1)
    assert L.last_path() == "jit_reg"
    assert_close(y, np.fft.fft(x.astype(np.complex128), axis=1), 1, 1e-4, "jit unaligned")
    # small problems are not worth a compile: LDS kernel
    assert run_case(L, "ndfft", (3, 1000), 1, np.float64) == "generic_row"
    # real-data ops with a smooth inner FFT (rows) and every op on column tiles (strategy ii)
    for n, F in ((1000, 500), (192, 96), (2000, 1000), (288, 144), (500, 250)):
        rows = max(8, (1 << 17) // n + 3)
        for rdt in (np.float64, np.float32):
            for name in ("ndfft_r2c", "ndifft_r2c", "nddct2", "nddct3", "nddct4"):
                assert run_case(L, name, (rows, n), 1, rdt, offset=n) == "jit_real", (name, n)
            assert run_case(L, "nddct1", (rows, F + 1), 1, rdt, offset=n) == "jit_real", ("nddct1", F)
    for n in (100, 250, 144):
        for rdt in (np.float64, np.float32):
            for name in OPS:
                m = n if name in ("ndfft", "ndifft") else (n + 1 if name == "nddct1" else 2 * n)
                assert run_case(L, name, (m, 1500), 0, rdt, offset=n) == "jit_col", (name, m)
                assert run_case(L, name, (3, m, 700), 1, rdt, offset=n + 1) == "jit_col", (name, m)
    # round 4: column tiles of short f32 C2C lanes run their own recipe where the rows' re-planned one measured slower (jit.hip: jit_choose_col)
    for n in (98, 99, 156, 220):
        for name in ("ndfft", "ndifft"):
            assert run_case(L, name, (n, 1400), 0, np.float32, offset=n) == "jit_col", (name, n)
            run_case(L, name, (900, n), 1, np.float32, offset=n)                       # rows keep the re-planned recipe (parity only)


def handler_clone_shares_plan(L):
    h = handlers.FftHandler(16, _library=L)
    h2 = h.clone()
    del h
    x = synth.complex_array((2, 16)); y = np.zeros_like(x)
    api.ndfft(x, y, h2, 1)
    assert_close(y, np.fft.fft(x, axis=1), 1, 1e-10, "cloned handler")
