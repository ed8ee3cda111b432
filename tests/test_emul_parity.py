"""CPU-container run of the parity suite on the SAME kernel sources compiled for the host through
tests/emul (fiber emulation of a workgroup).  This debugs index arithmetic, LDS layout and the host
logic without a GPU; it is NOT the parity claim -- that is tests/test_gpu_parity.py on an MI355X."""
import os
import subprocess

import numpy as np
import pytest

import parity_suite as ps
from helpers import GOLDEN_SIZES
from ndrustfft_amd import _lib
from ndrustfft_amd import api as api_mod, handlers as handlers_mod

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul")


os.environ.setdefault("EMUL_DEVICES", "3")     # three fake devices: the multi-device paths run on the CPU container too


@pytest.fixture(scope="module")
def L():
    if os.environ.get("NDFFT_EMUL_LIB"):                    # a side build of the emulation (make -C tests/emul SAN=1: UBSan)
        return _lib.Library(os.path.abspath(os.environ["NDFFT_EMUL_LIB"]))
    subprocess.check_call(["make", "-C", EMUL_DIR, "-s", "-j4"])
    return _lib.Library(os.path.join(EMUL_DIR, "_build", "libndfft_emul.so"))


def test_reference_unit_tests(L, refvec): ps.reference_unit_tests(L, refvec)
def test_reference_examples(L, refvec): ps.reference_examples(L, refvec)
def test_layouts(L): ps.layouts(L)
def test_normalization(L): ps.normalization_modes(L)
def test_panics(L): ps.panics(L)
def test_clone(L): ps.handler_clone_shares_plan(L)
def test_wave_short_lanes(L): ps.wave_short_lanes(L)
def test_tiny_lanes(L): ps.tiny_lanes(L)
def test_reg_lanes(L): ps.reg_lanes(L, sizes=(18, 23, 30, 40), sizes_f32=())
def test_regreal_lanes(L):
    seen = ps.regreal_lanes(L, sizes=(18, 21), sizes_f32=())
    assert {"regreal_row", "regreal_col"} <= seen, seen
def test_tinymat_lanes(L): ps.tinymat_lanes(L)
def test_host_pipeline_pageable(L): ps.host_pipeline_pageable(L, shapes=(("ndfft", (520, 1024), 1, np.float64), ("nddct2", (70, 96, 128), 1, np.float64)))
def test_sharded_exec_three_fake_devices(L):
    assert L.c.ndfft_device_count() == 3
    ps.sharded_exec(L, [0, 1, 2])
    ps.sharded_exec(L, [2, 0])
def test_sharded_device_resident_interleaved_blocks(L):
    """The case round 2's advisor broke: the split dimension is NOT the outermost one in memory (axis = 0 on a C-contiguous array -- the second pass of
    fft2), so every block's address span covers nearly the whole array and interleaves with the other devices' blocks.  Blocks now travel as dense
    images (pack / unpack kernels on the root device); the emulation aborts on a peer copy without peer access.  Repeated: the old failure was a race."""
    ps.dev_sharded_case(L, "ndfft", (64, 300), 0, root=1, ids=[0, 1, 2], repeats=20)
    ps.dev_sharded_case(L, "ndfft", (64, 300), 0, root=0, ids=[2, 1], repeats=5)
    ps.dev_sharded_case(L, "ndfft_r2c", (64, 5, 3), 0, root=2, ids=[0, 1, 2], repeats=5)
    ps.dev_sharded_case(L, "ndifft_r2c", (32, 7, 9), 0, root=1, ids=[0, 1, 2], repeats=3, rdt=np.float32)
    ps.dev_sharded_case(L, "nddct2", (5, 16, 33), 1, root=0, ids=[1, 2, 0], repeats=3)
    ps.dev_sharded_case(L, "nddct1", (9, 6, 4), 0, root=1, ids=[2, 0], repeats=3)


def test_sharded_device_resident_views_with_holes(L):
    """Output (and input) views with holes on the device: stepped, reversed and padded views of a larger allocation; every element outside
    the view keeps its sentinel, with the holes interleaved between the blocks of different devices."""
    ps.dev_sharded_case(L, "ndfft", (9, 16, 6), 1, root=1, ids=[0, 1, 2], out_view=((9, 16, 12), np.s_[:, :, ::2]), repeats=5)
    ps.dev_sharded_case(L, "ndfft", (9, 16, 6), 1, root=0, ids=[1, 2], out_view=((9, 16, 12), np.s_[::-1, :, 1::2]), in_view=((18, 16, 6), np.s_[::2]), repeats=3)
    ps.dev_sharded_case(L, "ndfft", (16, 40), 0, root=2, ids=[0, 1, 2], out_view=((16, 50), np.s_[:, 5:45]), repeats=5)
    ps.dev_sharded_case(L, "nddct3", (12, 8), 1, root=1, ids=[0, 2], out_view=((24, 8), np.s_[1::2, :]), repeats=3)
    ps.dev_sharded_case(L, "ndfft_r2c", (10, 21), 0, root=0, ids=[2, 1, 0], out_view=((6, 64), np.s_[:, 1:64:3]), repeats=3)


def test_sharded_device_resident_chunk_pipeline(L, monkeypatch):
    """Blocks cut into many pipelined chunks (scatter of chunk c+1 beside the transform of chunk c and the gather of chunk c-1; two buffer slots)."""
    monkeypatch.setenv("NDFFT_SHARD_CHUNK_KB", "4"); L.reload_switches()                  # (conftest reloads again after the test)
    ps.dev_sharded_case(L, "ndfft", (61, 64), 1, root=1, ids=[0, 1, 2], repeats=2)          # contiguous spans, uneven chunks
    ps.dev_sharded_case(L, "ndfft", (64, 90), 0, root=0, ids=[1, 2], repeats=2)             # packed, many chunks
    ps.dev_sharded_case(L, "nddct2", (40, 32, 3), 1, root=2, ids=[0, 1], out_view=((40, 32, 6), np.s_[:, :, ::2]), repeats=2)


def test_sharded_fft2_on_three_devices(L, monkeypatch):
    """fft2 / rfft2 with every array resident on one device and both passes split over three (the second pass = the re-shard), also in many small chunks."""
    ps.dev_sharded_fft2(L, (96, 128), root=1, ids=[0, 1, 2])
    ps.dev_sharded_fft2(L, (90, 64), root=0, ids=[2, 1], real=True)
    monkeypatch.setenv("NDFFT_SHARD_CHUNK_KB", "8"); L.reload_switches()
    ps.dev_sharded_fft2(L, (128, 256), root=2, ids=[0, 1, 2])
    ps.dev_sharded_fft2(L, (64, 100), root=0, ids=[0, 1, 2], real=True)


def test_baseline_length_fixtures(L, blvec): ps.baseline_length_fixtures(L, blvec)
def test_infinity_cache_residency_model(L):
    """exec.hip: MallModel through ndfft_last_input_policy (device-resident arrays on the emulation): an unknown input keeps the size rule (plain loads); a re-read
    input is worth plain loads while its reuse distance fits the cache; an output of more than 64 MiB was written with nt stores and is cold (streaming) when it
    becomes an input; a rotation of inputs larger than the cache streams every member; a small output stays resident; ndfft_set_input_hint overrides."""
    import ctypes
    n = 64
    h = handlers_mod.FftHandler(n, _library=L)
    def dev_buf(rows):
        p = ctypes.c_void_p(); L.check(L.c.ndfft_dev_alloc(ctypes.byref(p), rows * n * 16)); return p
    def fft(src, dst, rows, op=_lib.OP_C2C_FWD):
        with ps.switches(L, NDFFT_WAVE="0"):    # n = 64 on the register row kernel (pow2_reg), the one the model serves
            L.check(L.c.ndfft_exec_device(h._plan, op, src, dst, 2, api_mod._i64((rows, n)), api_mod._i64((n, 1)), api_mod._i64((rows, n)), api_mod._i64((n, 1)), 1, _lib.NORM_DEFAULT, 0.0, None))
        assert L.last_path() == "pow2_reg", L.last_path()
        return L.c.ndfft_last_input_policy()
    big = (72 << 20) // (n * 16)                 # 72 MiB arrays: outputs above the 64 MiB line
    a, b, c = dev_buf(big), dev_buf(big), dev_buf(big)
    small = (8 << 20) // (n * 16)
    s1, s2 = dev_buf(small), dev_buf(small)
    huge = (260 << 20) // (n * 16)               # (allocated up front: addresses the model has never seen)
    x1, x2, y1 = dev_buf(huge), dev_buf(huge), dev_buf(huge)
    try:
        assert fft(a, b, big) == 0               # never seen: size rule -> plain
        assert fft(a, b, big) == 0               # read a moment ago: reuse distance 0 -> plain
        assert fft(b, c, big) == 1               # b is an output of > 64 MiB (nt stores): cold
        assert fft(b, c, big) == 0               # ... read again right away: a re-read input is worth making resident
        assert fft(s1, s2, small) == 0 and fft(s2, s1, small) == 0      # small outputs stay in the cache
        L.check(L.c.ndfft_set_input_hint(_lib.INPUT_COLD)); assert fft(a, b, big) == 1
        L.check(L.c.ndfft_set_input_hint(_lib.INPUT_CACHED)); assert fft(b, c, big) == 0      # (b was just written: the model alone would stream it)
        L.check(L.c.ndfft_set_input_hint(_lib.INPUT_AUTO))
        assert fft(b, c, big) == 0               # re-read
        assert fft(c, a, big) == 1               # c: output of the call before
        # reuse distance: a is read, then four other 72 MiB inputs go through; a no longer fits 256 MiB of LRU stack
        assert fft(a, c, big) in (0, 1)
        assert fft(a, c, big) == 0
        others = [dev_buf(big) for _ in range(4)]
        for o in others: assert fft(o, c, big) == 0
        assert fft(a, c, big) == 1, "288 MiB of other inputs since a was last read: streaming loads"
        for o in others: assert fft(o, c, big) == 1, "a rotation larger than the cache streams every member"
        for o in others: L.check(L.c.ndfft_dev_free(o))
        # round 5: a buffer LARGER than the cache (BASELINE configs[2]'s 4097 x 8192 c64 is 64 KiB over 256 MiB): only an immediate re-read keeps the size rule
        assert fft(x1, y1, huge) == 0        # never seen: size rule (<= 384 MiB -> plain)
        assert fft(x1, y1, huge) == 0        # read a moment ago: the size rule again
        assert fft(x2, y1, huge) == 0        # never seen
        assert fft(x1, y1, huge) == 1, "larger than the cache and another input was read since: streaming loads"
        assert fft(x2, y1, huge) == 1
    finally:
        L.c.ndfft_set_input_hint(_lib.INPUT_AUTO)
        for p in (a, b, c, s1, s2, x1, x2, y1): L.check(L.c.ndfft_dev_free(p))


def test_host_registration_cache_lru(L):
    """ndfft_host_reg_cache on the CPU container (the emulation tracks hipHostRegister ranges): nothing is registered while the cache is off or on an
    array's first use; the second use registers input and output and the call runs the pinned pipeline with identical results; a sub-view of a
    registered array is served by its registration; the byte budget evicts the least recently used arrays; forget / switching off drop everything."""
    import ctypes
    import synth
    from ndrustfft_amd import api, handlers
    from oracle import oracle_ctypes as orc
    reg = lambda: L.c.emul_host_registered_bytes()
    L.c.emul_host_registered_bytes.restype = ctypes.c_size_t
    n = 1024; h = handlers.FftHandler(n, _library=L); o = orc.FftHandler(n)
    def pair(k):
        x = synth.complex_array((520, n), offset=1000 * k); y = np.zeros_like(x); yo = np.zeros_like(x); orc.ndfft(x, yo, o, 1)
        return x, y, yo
    x0, y0, yo0 = pair(0)
    api.ndfft(x0, y0, h, 1); api.ndfft(x0, y0, h, 1)
    assert reg() == 0, "cache is off by default"
    L.check(L.c.ndfft_host_reg_cache(3 * x0.nbytes))                      # room for three arrays
    try:
        api.ndfft(x0, y0, h, 1); assert reg() == 0                        # first sighting
        y0[...] = 0; api.ndfft(x0, y0, h, 1); assert reg() == 2 * x0.nbytes   # second: both registered, pinned pipeline
        assert np.abs(y0 - yo0).max() <= 1e-10 * np.abs(yo0).max()
        y0[...] = 0; api.ndfft(x0[:260], y0[:260], h, 1); assert reg() == 2 * x0.nbytes
        assert np.abs(y0[:260] - yo0[:260]).max() <= 1e-10 * np.abs(yo0).max()
        x1, y1, yo1 = pair(1)
        for _ in range(3): api.ndfft(x1, y1, h, 1)
        assert np.abs(y1 - yo1).max() <= 1e-10 * np.abs(yo1).max()
        assert reg() <= 3 * x0.nbytes, "the byte budget evicts the least recently used registrations"
        assert L.c.ndfft_host_forget(ctypes.c_void_p(x1.ctypes.data)) == 0 and L.c.ndfft_host_forget(ctypes.c_void_p(y1.ctypes.data)) == 0
        before = reg()
        api.ndfft(x1, y1, h, 1); assert reg() == before                   # forgotten: back to a first sighting
        assert np.abs(y1 - yo1).max() <= 1e-10 * np.abs(yo1).max()
    finally:
        L.check(L.c.ndfft_host_reg_cache(0))
    assert reg() == 0


def test_host_registration_cache_keeps_hot_arrays_and_retries_stale_ones(L):
    """Round 4 (advisor): steady-state calls go through the cache BEFORE is_pinned, so (1) a hot pair's LRU stamp stays fresh -- with room for two
    pairs, pair A used again and again survives the registration of pair C, the idle pair B is the one evicted; (2) a copy that fails on a
    cache-owned registration (injected: what a stale registration does on the MI355X) makes the call forget the ranges and finish through the
    bounce buffers, with the right answer."""
    import ctypes
    import synth
    from ndrustfft_amd import api, handlers
    from oracle import oracle_ctypes as orc
    L.c.emul_host_registered_bytes.restype = ctypes.c_size_t
    isreg = lambda a: bool(L.c.emul_is_registered(ctypes.c_void_p(a.ctypes.data)))
    n = 1024; h = handlers.FftHandler(n, _library=L); o = orc.FftHandler(n)
    def pair(k):
        x = synth.complex_array((520, n), offset=1000 * k); y = np.zeros_like(x); yo = np.zeros_like(x); orc.ndfft(x, yo, o, 1)
        return x, y, yo
    (xa, ya, yoa), (xb, yb, yob), (xc, yc, yoc) = pair(0), pair(1), pair(2)
    L.check(L.c.ndfft_host_reg_cache(4 * xa.nbytes + 4096))                # room for two pairs
    try:
        for _ in range(2): api.ndfft(xa, ya, h, 1)
        for _ in range(2): api.ndfft(xb, yb, h, 1)
        assert isreg(xa) and isreg(ya) and isreg(xb) and isreg(yb)
        for _ in range(10):                                                # A is hot: every call must refresh its stamp
            ya[...] = 0; api.ndfft(xa, ya, h, 1)
            assert np.abs(ya - yoa).max() <= 1e-10 * np.abs(yoa).max()
        for _ in range(2): api.ndfft(xc, yc, h, 1)                         # registering C needs room: the idle pair B goes
        assert isreg(xc) and isreg(yc)
        assert isreg(xa) and isreg(ya), "the hot pair was evicted"
        assert not isreg(xb) and not isreg(yb)
        assert L.c.emul_host_registered_bytes() <= 4 * xa.nbytes + 4096
        # a stale registration: the first async copy on a registered range fails; the call recovers through the bounce buffers
        L.c.emul_fail_registered_copies(1)
        ya[...] = 0; api.ndfft(xa, ya, h, 1)
        L.c.emul_fail_registered_copies(0)
        assert np.abs(ya - yoa).max() <= 1e-10 * np.abs(yoa).max()
        assert not isreg(xa) and not isreg(ya), "the failing ranges were not forgotten"
        ya[...] = 0; api.ndfft(xa, ya, h, 1)                               # and the pair is usable (and registrable) again
        assert np.abs(ya - yoa).max() <= 1e-10 * np.abs(yoa).max()
    finally:
        L.c.emul_fail_registered_copies(0)
        L.check(L.c.ndfft_host_reg_cache(0))
    assert L.c.emul_host_registered_bytes() == 0


def test_interleaved_mut_views(L): ps.interleaved_mut_views_two_threads(L, rounds=1)
def test_long_strided_lanes(L): ps.long_strided_lanes(L)
def test_narrow_xcd_tiles(L): ps.narrow_xcd_tiles(L)
def test_column_four_step(L): ps.column_four_step(L)
def test_huge_prime_factors(L): ps.huge_prime_factors(L, full=False)
def test_fuzz(L): ps.fuzz(L, seed=11, count=120, max_points=1 << 13)
def test_fuzz_streaming_loads(L):
    L.check(L.c.ndfft_set_input_hint(_lib.INPUT_COLD))
    try:
        ps.fuzz(L, seed=17, count=60, max_points=1 << 13)
    finally:
        L.check(L.c.ndfft_set_input_hint(_lib.INPUT_AUTO))
def test_bluestein_register_kernel(L): ps.bluestein_register_kernel(L)
def test_bluestein_smooth_length_partial_rounds(L):
    """F = 263 on the smooth convolution length 550 = 11.10.5 (55 threads, partial first round) instead of 1024: blue_kernel.h's partial-round guards and the
    long-double DFT of the bhat table for a non-power-of-two M, on the CPU."""
    ps.bluestein_register_kernel(L, sizes=((263, 1024),), col_max_M=1024)
def test_partial_round_configs(L): ps.partial_round_configs(L)
def test_rader_kernel(L): ps.rader_kernel(L)
def test_odd_real_lengths(L): ps.odd_real_lengths(L, dct4=True)
def test_long_lanes_four_step(L): ps.long_lanes_four_step(L, full=False)
def test_long_lanes_padded_views(L): ps.long_lanes_padded_views(L)


@pytest.mark.parametrize("dt", ["f64", "f32"])
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_golden(L, npvec, dt, n): ps.golden_vectors(L, npvec, dt, n)


@pytest.mark.parametrize("n", [n for n in ps.SIZE_SWEEP if n <= 1024])
def test_sizes_f64(L, n): ps.size_sweep(L, n, np.float64)


@pytest.mark.parametrize("n", [7, 64, 100, 264, 512])
def test_sizes_f32(L, n): ps.size_sweep(L, n, np.float32)


def test_pow2_tuned_sizes(L):
    for n, rdt in ((2048, np.float64), (4096, np.float64), (8192, np.float32), (4096, np.float32)):
        for name in ("ndfft", "ndifft"):
            assert ps.run_case(L, name, (2, n), 1, rdt, offset=n) == "pow2_reg"


def test_reference_bench_shapes_small(L):
    ps.reference_bench_shapes(L, sizes_fft=(128, 264), sizes_dct=(129, 265))


def test_pow2_real_sizes(L):
    ps.pow2_real_sizes(L, sizes=(64, 128, 256, 512, 1024, 4096), dtypes=(np.float64,))
    ps.pow2_real_sizes(L, sizes=(64, 2048, 8192), dtypes=(np.float32,))


def test_pow2_col_sizes(L):
    ps.pow2_col_sizes(L, sizes=(64, 256), dtypes=(np.float64,))
    ps.pow2_col_sizes(L, sizes=(128, 1024), dtypes=(np.float32,))


def test_alternating_devices_on_one_thread(L):
    """one host thread alternates ndfft_set_device between calls.  Workspaces (staging, scratch) are kept
    per device: the emulation aborts if a buffer allocated on one device is used while another is current."""
    import ctypes
    import synth
    from ndrustfft_amd import api, handlers
    from oracle import oracle_ctypes as orc
    x = synth.real_array((4096, 24), np.float32); yo = np.zeros((2049, 24), np.complex64)
    orc.ndfft_r2c(x, yo, orc.R2cFftHandler(4096, np.float32), 0)
    h = handlers.R2cFftHandler(4096, np.float32, _library=L)          # axis 0, long lanes: scratch + staging in play
    try:
        for rep in range(2):
            for dev in (0, 1, 2, 1, 0):
                assert L.c.ndfft_set_device(dev) == 0
                y = np.zeros((2049, 24), np.complex64)
                api.ndfft_r2c(x, y, h, 0)
                assert np.abs(y - yo).max() <= 1e-4 * np.abs(yo).max()
                # device-resident call on the null stream of that device
                nb_in, nb_out = x.nbytes, y.nbytes
                din, dout = ctypes.c_void_p(), ctypes.c_void_p()
                L.check(L.c.ndfft_dev_alloc(ctypes.byref(din), nb_in)); L.check(L.c.ndfft_dev_alloc(ctypes.byref(dout), nb_out))
                L.check(L.c.ndfft_dev_upload(din, ctypes.c_void_p(x.ctypes.data), nb_in))
                L.check(L.c.ndfft_exec_device(h._plan, _lib.OP_R2C, din, dout, 2, api._i64(x.shape), api._i64((24, 1)), api._i64(y.shape),
                                              api._i64((24, 1)), 0, _lib.NORM_DEFAULT, 0.0, None))
                y2 = np.zeros_like(y)
                L.check(L.c.ndfft_dev_sync(None)); L.check(L.c.ndfft_dev_download(ctypes.c_void_p(y2.ctypes.data), dout, nb_out))
                assert np.array_equal(y2, y)
                L.check(L.c.ndfft_dev_free(din)); L.check(L.c.ndfft_dev_free(dout))
        assert L.c.ndfft_release_workspace() == 0
    finally:
        L.c.ndfft_set_device(0)


def test_sharded_device_resident_fake_devices(L):
    """ndfft_exec_sharded_device with the array resident on fake device 1 and blocks on devices 0, 1, 2: the emulation's
    hipMemcpyPeerAsync checks that every pointer is on the device it is claimed to be on."""
    import ctypes
    import synth
    from ndrustfft_amd import api, handlers
    from oracle import oracle_ctypes as orc
    x = synth.complex_array((7, 64, 3)); yo = np.zeros_like(x)
    orc.ndfft(x, yo, orc.FftHandler(64), 1)
    h = handlers.FftHandler(64, _library=L)
    try:
        assert L.c.ndfft_set_device(1) == 0
        din, dout = ctypes.c_void_p(), ctypes.c_void_p()
        L.check(L.c.ndfft_dev_alloc(ctypes.byref(din), x.nbytes)); L.check(L.c.ndfft_dev_alloc(ctypes.byref(dout), x.nbytes))
        L.check(L.c.ndfft_dev_upload(din, ctypes.c_void_p(x.ctypes.data), x.nbytes))
        ids = (ctypes.c_int * 3)(0, 1, 2)
        L.check(L.c.ndfft_exec_sharded_device(h._plan, _lib.OP_C2C_FWD, din, dout, 3, api._i64(x.shape), api._i64((192, 3, 1)), api._i64(x.shape),
                                              api._i64((192, 3, 1)), 1, _lib.NORM_DEFAULT, 0.0, 3, ids, None))
        y = np.zeros_like(x)
        L.check(L.c.ndfft_dev_download(ctypes.c_void_p(y.ctypes.data), dout, x.nbytes))
        assert np.abs(y - yo).max() <= 1e-10 * np.abs(yo).max()
        L.check(L.c.ndfft_dev_free(din)); L.check(L.c.ndfft_dev_free(dout))
    finally:
        L.c.ndfft_set_device(0)
