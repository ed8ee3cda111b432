"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module (see oracle/ndfft_oracle.h).  The product package ``ndrustfft_amd`` never does.

The wrappers keep the reference's call shape ``nd*(input, output, handler, axis)``
(/root/reference/src/lib.rs:105-110) on numpy arrays of any layout.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libndfft_oracle.so")

F32, F64 = 0, 1
NORM_NONE, NORM_DEFAULT, NORM_CUSTOM = 0, 1, 2
H_FFT, H_R2C, H_DCT = 0, 1, 2
NDFFT, NDIFFT, NDFFT_R2C, NDIFFT_R2C, NDDCT1, NDDCT2, NDDCT3, NDDCT4 = range(8)
OK, PANIC_SIZE, PANIC_AXIS, PANIC_ZIP, BAD_ARG = range(5)

CUSTOM_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t)


class OraclePanic(RuntimeError):
    """A panic the reference would raise, restated (message text matches src/lib.rs)."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def build(force=False):
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("ndfft_oracle.c", "oracle_lane.inc", "ndfft_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def use_fast_build():
    """bench.py's cpu_baseline leg only: (re)bind this module to the -O3 -march=native build of the same sources, built here and now for this host
    (oracle/Makefile: fast).  Returns the compiler flags on success, None if it cannot be built or loaded (the -O2 build stays bound)."""
    global _SO, _lib
    fast = os.path.join(_HERE, "_build", "libndfft_oracle_fast.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "fast"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        ctypes.CDLL(fast)
    except Exception:
        return None
    _SO, _lib = fast, None
    return "-O3 -march=native -funroll-loops"


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_handler_new.restype = ctypes.c_void_p
        L.orc_handler_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        L.orc_handler_normalization.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_handler_free.argtypes = [ctypes.c_void_p]
        L.orc_nd.restype = ctypes.c_int
        L.orc_nd.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        L.orc_truth_dft.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        L.orc_truth_dct.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.orc_last_strategy.restype = ctypes.c_int
        L.orc_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _dtype_code(real_dtype):
    real_dtype = np.dtype(real_dtype)
    if real_dtype == np.float32:
        return F32
    if real_dtype == np.float64:
        return F64
    raise TypeError("T must be f32 or f64 (src/lib.rs:111)")


class _Handler:
    kind = None

    def __init__(self, n, dtype=np.float64):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self._h = lib().orc_handler_new(self.kind, _dtype_code(dtype), self.n)
        self._cb = None

    def normalization(self, mode, fn=None):
        """Builder, returns self.  mode in {NORM_NONE, NORM_DEFAULT, NORM_CUSTOM};
        fn(view) mutates a 1-D numpy view of the lane in place (Custom(fn(&mut [T])))."""
        cb = None
        if mode == NORM_CUSTOM:
            elem = self._norm_elem_dtype()

            def tramp(ptr, length):
                buf = (ctypes.c_char * (length * elem.itemsize)).from_address(ptr)
                fn(np.frombuffer(buf, dtype=elem, count=length))

            cb = CUSTOM_FN(tramp)
        self._cb = cb
        lib().orc_handler_normalization(self._h, mode, ctypes.cast(cb, ctypes.c_void_p) if cb else None)
        return self

    def _norm_elem_dtype(self):
        c = np.complex64 if self.dtype == np.float32 else np.complex128
        return np.dtype(c)

    def __del__(self):
        try:
            lib().orc_handler_free(self._h)
        except Exception:
            pass


class FftHandler(_Handler):
    kind = H_FFT


class R2cFftHandler(_Handler):
    kind = H_R2C


class DctHandler(_Handler):
    kind = H_DCT

    def _norm_elem_dtype(self):
        return self.dtype


def _arr_i64(v):
    return (ctypes.c_int64 * len(v))(*[int(x) for x in v])


def _call(func, par, inp, out, handler, axis):
    assert isinstance(inp, np.ndarray) and isinstance(out, np.ndarray)
    if inp.ndim != out.ndim:
        raise TypeError("input and output must have the same dimensionality D")
    if axis < 0:
        raise OverflowError("axis: usize")
    sin = [s // inp.itemsize for s in inp.strides]
    sout = [s // out.itemsize for s in out.strides]
    err = ctypes.create_string_buffer(256)
    rc = lib().orc_nd(func, int(par), inp.ctypes.data, out.ctypes.data, inp.ndim,
                      _arr_i64(inp.shape), _arr_i64(sin), _arr_i64(out.shape), _arr_i64(sout),
                      handler._h, axis, err, 256)
    if rc != OK:
        raise OraclePanic(rc, err.value.decode() or f"oracle error {rc}")


def ndfft(i, o, h, axis): _call(NDFFT, 0, i, o, h, axis)
def ndifft(i, o, h, axis): _call(NDIFFT, 0, i, o, h, axis)
def ndfft_r2c(i, o, h, axis): _call(NDFFT_R2C, 0, i, o, h, axis)
def ndifft_r2c(i, o, h, axis): _call(NDIFFT_R2C, 0, i, o, h, axis)
def nddct1(i, o, h, axis): _call(NDDCT1, 0, i, o, h, axis)
def nddct2(i, o, h, axis): _call(NDDCT2, 0, i, o, h, axis)
def nddct3(i, o, h, axis): _call(NDDCT3, 0, i, o, h, axis)
def nddct4(i, o, h, axis): _call(NDDCT4, 0, i, o, h, axis)
def ndfft_par(i, o, h, axis): _call(NDFFT, 1, i, o, h, axis)
def ndifft_par(i, o, h, axis): _call(NDIFFT, 1, i, o, h, axis)
def ndfft_r2c_par(i, o, h, axis): _call(NDFFT_R2C, 1, i, o, h, axis)
def ndifft_r2c_par(i, o, h, axis): _call(NDIFFT_R2C, 1, i, o, h, axis)
def nddct1_par(i, o, h, axis): _call(NDDCT1, 1, i, o, h, axis)
def nddct2_par(i, o, h, axis): _call(NDDCT2, 1, i, o, h, axis)
def nddct3_par(i, o, h, axis): _call(NDDCT3, 1, i, o, h, axis)
def nddct4_par(i, o, h, axis): _call(NDDCT4, 1, i, o, h, axis)


def last_strategy():
    return lib().orc_last_strategy()


def num_threads():
    return lib().orc_num_threads()


def truth_dft(x, sign=-1):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.empty_like(x)
    lib().orc_truth_dft(x.ctypes.data, y.ctypes.data, x.size, sign)
    return y


def truth_dct(kind, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().orc_truth_dct(kind, x.ctypes.data, y.ctypes.data, x.size)
    return y
