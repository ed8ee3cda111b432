"""FftHandler / R2cFftHandler / DctHandler / Normalization -- host-side mirror of
/root/reference/src/lib.rs:89-98, 269-311, 451-495, 640-686 over the C ABI."""
import ctypes

import numpy as np

from . import _lib


class Normalization:
    """enum Normalization<T> { None, Default, Custom(fn(&mut [T])) }  (lib.rs:89-98)."""
    NONE = "None"
    DEFAULT = "Default"

    def __init__(self, kind, fn=None):
        self.kind, self.fn = kind, fn

    @staticmethod
    def none():
        return Normalization(Normalization.NONE)

    @staticmethod
    def default():
        return Normalization(Normalization.DEFAULT)

    @staticmethod
    def custom(fn):
        """fn(lane) mutates one 1-D numpy lane in place -- a host function, as in the reference."""
        return Normalization("Custom", fn)


def _dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return _lib.F32
    if dtype == np.float64:
        return _lib.F64
    raise TypeError("T must be f32 or f64 (FftNum, lib.rs:111)")


class _Handler:
    KIND = None

    def __init__(self, n, dtype=np.float64, *, _library=None, _share=None):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.norm = Normalization.default()          # lib.rs:302, 486, 677
        self._L = _library or _lib.default()
        if _share is not None:                        # Clone = Arc bump (lib.rs:269, 451, 640)
            self._plan = _share
            self._L.check(self._L.c.ndfft_plan_retain(self._plan))
        else:
            p = ctypes.c_void_p()
            self._L.check(self._L.c.ndfft_plan_create(self.KIND, _dtype_code(dtype), self.n, ctypes.byref(p)))
            self._plan = p

    def normalization(self, norm):
        """Builder: consumes and returns the handler (lib.rs:308-311, 492-495, 683-686)."""
        if norm is None:
            norm = Normalization.none()
        self.norm = norm
        return self

    def clone(self):
        h = type(self)(self.n, self.dtype, _library=self._L, _share=self._plan)
        h.norm = self.norm
        return h

    @property
    def real_dtype(self):
        return self.dtype

    @property
    def complex_dtype(self):
        return np.dtype(np.complex64 if self.dtype == np.float32 else np.complex128)

    def __del__(self):
        try:
            self._L.c.ndfft_plan_destroy(self._plan)
        except Exception:
            pass


class FftHandler(_Handler):
    """FftHandler<T>::new(n)  (lib.rs:294-304)."""
    KIND = _lib.KIND_C2C


class R2cFftHandler(_Handler):
    """R2cFftHandler<T>::new(n), m = n/2 + 1  (lib.rs:477-488)."""
    KIND = _lib.KIND_R2C

    @property
    def m(self):
        return self.n // 2 + 1


class DctHandler(_Handler):
    """DctHandler<T>::new(n): plans DCT-I..IV eagerly  (lib.rs:665-679)."""
    KIND = _lib.KIND_DCT
