"""ctypes binding of libndfft_mi355x.so (the C ABI of include/ndfft_mi355x.h).

There is NO CPU fallback: if the in-tree HIP library is missing or no MI355X is visible the calls
raise.  `Library(path)` exists so that tests can bind a differently built copy of the same sources;
the package itself only ever loads the in-tree gfx950 build.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NDFFT_MI355X_LIB") or os.path.join(_HERE, "csrc", "libndfft_mi355x.so")   # (the variable: developer A/B builds)

OK, ERR_INVALID_ARG, ERR_SIZE_MISMATCH, ERR_SHAPE_MISMATCH, ERR_AXIS, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC = range(9)
F32, F64 = 0, 1
KIND_C2C, KIND_R2C, KIND_DCT = 0, 1, 2
OP_C2C_FWD, OP_C2C_INV, OP_R2C, OP_C2R, OP_DCT1, OP_DCT2, OP_DCT3, OP_DCT4 = range(8)
NORM_NONE, NORM_DEFAULT, NORM_SCALE = 0, 1, 2
INPUT_AUTO, INPUT_CACHED, INPUT_COLD = 0, 1, 2

# every symbol include/ndfft_mi355x.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "ndfft_abi_version", "ndfft_abi_minor", "ndfft_last_error", "ndfft_device_count", "ndfft_set_device",
    "ndfft_plan_create", "ndfft_plan_retain", "ndfft_plan_destroy", "ndfft_plan_n", "ndfft_plan_kind",
    "ndfft_plan_dtype", "ndfft_plan_lane_len_in", "ndfft_plan_lane_len_out",
    "ndfft_exec", "ndfft_exec_device", "ndfft_exec_sharded", "ndfft_exec_sharded_device", "ndfft_last_path", "ndfft_explain_plan",
    "ndfft_dev_alloc", "ndfft_dev_free", "ndfft_dev_upload", "ndfft_dev_download", "ndfft_dev_sync",
    "ndfft_release_workspace", "ndfft_host_alloc", "ndfft_host_free", "ndfft_set_input_hint", "ndfft_host_forget", "ndfft_host_reg_cache", "ndfft_last_input_policy",
    "ndfft_documented_switches", "ndfft_reload_switches", "ndfft_jit_prebuild",
]


_loaded = []          # every Library this process has opened (tests/conftest.py reloads their switches between tests)


class NdfftError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class Panic(NdfftError):
    """A condition on which the reference panics (size / axis / Zip mismatch); message text matches."""


class Library:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: build it with `make -C ndrustfft_amd/csrc` (or __graft_entry__.build()). "
                "ndrustfft_amd has no CPU fallback.")
        self.path = path
        L = self.c = ctypes.CDLL(path)
        _loaded.append(self)
        vp, i32, sz, i64p, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_double
        L.ndfft_abi_version.restype = i32
        L.ndfft_abi_minor.restype = i32
        L.ndfft_last_error.restype = ctypes.c_char_p
        L.ndfft_last_path.restype = ctypes.c_char_p
        L.ndfft_explain_plan.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        L.ndfft_explain_plan.restype = ctypes.c_int
        L.ndfft_device_count.restype = i32
        L.ndfft_set_device.argtypes = [i32]
        L.ndfft_plan_create.argtypes = [i32, i32, sz, ctypes.POINTER(vp)]
        L.ndfft_plan_retain.argtypes = [vp]
        L.ndfft_plan_destroy.argtypes = [vp]
        for f in (L.ndfft_plan_n,):
            f.restype = sz; f.argtypes = [vp]
        for f in (L.ndfft_plan_kind, L.ndfft_plan_dtype):
            f.restype = i32; f.argtypes = [vp]
        for f in (L.ndfft_plan_lane_len_in, L.ndfft_plan_lane_len_out):
            f.restype = sz; f.argtypes = [vp, i32]
        L.ndfft_exec.argtypes = [vp, i32, vp, vp, i32, i64p, i64p, i64p, i64p, i32, i32, dbl]
        L.ndfft_exec_device.argtypes = [vp, i32, vp, vp, i32, i64p, i64p, i64p, i64p, i32, i32, dbl, vp]
        ip = ctypes.POINTER(ctypes.c_int)
        L.ndfft_exec_sharded.argtypes = [vp, i32, vp, vp, i32, i64p, i64p, i64p, i64p, i32, i32, dbl, i32, ip]
        L.ndfft_exec_sharded_device.argtypes = [vp, i32, vp, vp, i32, i64p, i64p, i64p, i64p, i32, i32, dbl, i32, ip, vp]
        L.ndfft_dev_alloc.argtypes = [ctypes.POINTER(vp), sz]
        L.ndfft_dev_free.argtypes = [vp]
        L.ndfft_dev_upload.argtypes = [vp, vp, sz]
        L.ndfft_dev_download.argtypes = [vp, vp, sz]
        L.ndfft_dev_sync.argtypes = [vp]
        L.ndfft_release_workspace.argtypes = []; L.ndfft_release_workspace.restype = ctypes.c_int
        L.ndfft_host_alloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]; L.ndfft_host_alloc.restype = ctypes.c_int
        L.ndfft_host_free.argtypes = [vp]; L.ndfft_host_free.restype = ctypes.c_int
        L.ndfft_set_input_hint.argtypes = [i32]; L.ndfft_set_input_hint.restype = i32
        L.ndfft_host_forget.argtypes = [vp]; L.ndfft_host_forget.restype = i32
        L.ndfft_host_reg_cache.argtypes = [sz]; L.ndfft_host_reg_cache.restype = i32
        L.ndfft_last_input_policy.argtypes = []; L.ndfft_last_input_policy.restype = i32
        L.ndfft_documented_switches.argtypes = [ctypes.c_char_p, sz]; L.ndfft_documented_switches.restype = i32
        L.ndfft_reload_switches.argtypes = []; L.ndfft_reload_switches.restype = i32
        ip = ctypes.POINTER(ctypes.c_int)
        try:
            L.ndfft_jit_prebuild.argtypes = [ctypes.c_char_p, ctypes.c_char_p, i32, i32, ip, ip, ip]; L.ndfft_jit_prebuild.restype = i32
        except AttributeError:      # a side build older than ABI minor 3 loaded through NDFFT_MI355X_LIB (A-B runs against earlier rounds); tests/test_abi.py checks the product's exports
            if not os.environ.get("NDFFT_MI355X_LIB"):
                raise

    def check(self, status):
        if status == OK:
            return
        msg = self.c.ndfft_last_error().decode() or f"ndfft status {status}"
        if status in (ERR_SIZE_MISMATCH, ERR_SHAPE_MISMATCH, ERR_AXIS):
            raise Panic(status, msg)
        raise NdfftError(status, msg)

    def last_path(self):
        return self.c.ndfft_last_path().decode()

    def reload_switches(self):
        """Re-read the NDFFT_* environment switches (test hook: the library parses them once; see include/ndfft_mi355x.h)."""
        self.check(self.c.ndfft_reload_switches())

    def jit_prebuild(self, manifest, out_dir, first=0, stride=1):
        """Compile manifest entries first, first + stride, ... into out_dir (no GPU needed); returns (built, present, failed)."""
        b, p, f = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self.check(self.c.ndfft_jit_prebuild(os.fsencode(manifest), os.fsencode(out_dir), first, stride, ctypes.byref(b), ctypes.byref(p), ctypes.byref(f)))
        return b.value, p.value, f.value

    def documented_switches(self):
        buf = ctypes.create_string_buffer(4096)
        self.c.ndfft_documented_switches(buf, len(buf))
        return buf.value.decode().split()

    def last_input_policy(self):
        """Load policy of the last device exec on this thread: 0 default, 1 streaming loads, -1 fixed by the kernel (diagnostic)."""
        return int(self.c.ndfft_last_input_policy())

    def explain_plan(self, kind, dtype, n):
        """Text description of the recipes a handler of (kind, dtype, n) would use (diagnostic; needs no GPU)."""
        buf = ctypes.create_string_buffer(4096)
        need = self.c.ndfft_explain_plan(kind, dtype, n, buf, len(buf))
        if need < 0:
            raise NdfftError(-need, self.c.ndfft_last_error().decode())
        return buf.value.decode()


_default = None


def default():
    """The in-tree gfx950 library; raises if it has not been built."""
    global _default
    if _default is None:
        _default = Library()
    return _default
