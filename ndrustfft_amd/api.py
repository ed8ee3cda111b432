"""The sixteen nd* functions of ndrustfft (lib.rs:350-421, 543-611, 753-844) over the C ABI.

Call shape is the reference's: ``ndfft(input, output, handler, axis)``; `output` is written in
place.  Arrays are numpy arrays of ANY layout (host path, ndfft_exec) or torch tensors on an
MI355X (device path, ndfft_exec_device, asynchronous on torch's current stream).  The ``_par``
names exist for drop-in compatibility and dispatch to the same batched device call: on the GPU
every lane is always processed in parallel.
"""
import ctypes

import numpy as np

from . import _lib
from .handlers import DctHandler, FftHandler, Normalization, R2cFftHandler

try:  # torch is optional plumbing for device-resident arrays
    import torch
except Exception:  # pragma: no cover
    torch = None

_SPEC = {
    # op: (handler type, input is complex, output is complex)
    _lib.OP_C2C_FWD: (FftHandler, True, True),
    _lib.OP_C2C_INV: (FftHandler, True, True),
    _lib.OP_R2C: (R2cFftHandler, False, True),
    _lib.OP_C2R: (R2cFftHandler, True, False),
    _lib.OP_DCT1: (DctHandler, False, False),
    _lib.OP_DCT2: (DctHandler, False, False),
    _lib.OP_DCT3: (DctHandler, False, False),
    _lib.OP_DCT4: (DctHandler, False, False),
}


_PAR_DEVICES = None


def set_par_devices(device_ids):
    """Which GPUs the `_par` functions spread one call over (None / one id = the current device only).

    The reference's `_par` twins hand the independent lanes to rayon's worker threads (create_transform_par!,
    lib.rs:169-238); here the workers are GPUs: the array is cut in contiguous blocks along its outermost
    non-transform dimension, one per device, no collective (ndfft_exec_sharded / ndfft_exec_sharded_device)."""
    global _PAR_DEVICES
    ids = None if device_ids is None else [int(d) for d in device_ids]
    _PAR_DEVICES = ids if ids and len(ids) >= 1 else None


def par_devices():
    return None if _PAR_DEVICES is None else list(_PAR_DEVICES)


def _i64(v):
    return (ctypes.c_int64 * max(len(v), 1))(*[int(x) for x in v])


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _np_dtype_of(x):
    if _is_torch(x):
        return np.dtype(str(x.dtype).replace("torch.", ""))
    return x.dtype


def _apply_custom_host(op, inp, handler, axis):
    """Normalization::Custom is a host fn pointer: run it on the host at the reference's
    application point.  C2R and DCT apply it BEFORE the transform, to a copy of each input lane
    (lib.rs:511-515, 692-696); returns the array to feed the device with NORM_NONE."""
    work = np.array(inp, copy=True, order="C")
    lanes = np.moveaxis(work, axis, -1)
    for idx in np.ndindex(lanes.shape[:-1]):
        lane = np.ascontiguousarray(lanes[idx])
        handler.norm.fn(lane)
        lanes[idx] = lane
    return work


def _transform(op, inp, out, handler, axis, par=False):
    htype, in_c, out_c = _SPEC[op]
    if not isinstance(handler, htype):
        raise TypeError(f"handler must be a {htype.__name__}")
    if isinstance(axis, bool) or not isinstance(axis, (int, np.integer)) or axis < 0:
        raise TypeError("axis: usize")
    L = handler._L
    dev = _is_torch(inp) or _is_torch(out)
    if dev and not (_is_torch(inp) and _is_torch(out) and inp.is_cuda and out.is_cuda):
        raise TypeError("device path needs both arrays as torch tensors on the GPU")
    want_in = handler.complex_dtype if in_c else handler.real_dtype
    want_out = handler.complex_dtype if out_c else handler.real_dtype
    if _np_dtype_of(inp) != want_in or _np_dtype_of(out) != want_out:
        raise TypeError(f"element types must be {want_in} -> {want_out} for this handler")
    if inp.ndim != out.ndim:
        raise TypeError("input and output must have the same dimensionality D")

    norm = handler.norm
    mode, scale = _lib.NORM_DEFAULT, 0.0
    post_custom = False
    if norm.kind == Normalization.NONE:
        mode = _lib.NORM_NONE
    elif norm.kind == "Custom":
        mode = _lib.NORM_NONE
        if op in (_lib.OP_C2C_FWD, _lib.OP_R2C):
            pass                                      # forward C2C / R2C never look at it (lib.rs:313-318, 497-503)
        elif op == _lib.OP_C2C_INV:
            post_custom = True                        # after, on the output lane (lib.rs:326-330)
        else:
            if dev:
                inp = torch.from_numpy(_apply_custom_host(op, inp.cpu().numpy(), handler, axis)).to(out.device)
            else:
                inp = _apply_custom_host(op, inp, handler, axis)

    if dev:
        if inp.device != out.device:
            raise ValueError(f"input is on {inp.device} but output is on {out.device}: both arrays must live on the same GPU")
        # torch's lazy conj / neg bits are not in the bytes data_ptr() points at: materialise them for the input,
        # refuse them on the output (writing through such a view would store the wrong sign)
        inp = inp.resolve_conj().resolve_neg()
        if out.is_conj() or out.is_neg():
            raise ValueError("output tensor has a lazy conj/neg bit set: pass a plain tensor (out.resolve_conj())")
        shape_in, shape_out = list(inp.shape), list(out.shape)
        sin, sout = list(inp.stride()), list(out.stride())
        # the C side picks twiddle tables, JIT modules and workspaces by the CURRENT device: make it the tensors' one
        with torch.cuda.device(out.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)
            if par and _PAR_DEVICES and len(_PAR_DEVICES) > 1:
                ids = (ctypes.c_int * len(_PAR_DEVICES))(*_PAR_DEVICES)
                st = L.c.ndfft_exec_sharded_device(handler._plan, op, ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                   inp.ndim, _i64(shape_in), _i64(sin), _i64(shape_out), _i64(sout), int(axis), mode,
                                                   scale, len(_PAR_DEVICES), ids, stream)
            else:
                st = L.c.ndfft_exec_device(handler._plan, op, ctypes.c_void_p(inp.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                       inp.ndim, _i64(shape_in), _i64(sin), _i64(shape_out), _i64(sout), int(axis), mode,
                                       scale, stream)
    else:
        if not out.flags.writeable:
            raise ValueError("output must be writeable (&mut)")
        sin = [s // inp.itemsize for s in inp.strides]
        sout = [s // out.itemsize for s in out.strides]
        if par and _PAR_DEVICES and len(_PAR_DEVICES) > 1:
            ids = (ctypes.c_int * len(_PAR_DEVICES))(*_PAR_DEVICES)
            st = L.c.ndfft_exec_sharded(handler._plan, op, ctypes.c_void_p(inp.ctypes.data), ctypes.c_void_p(out.ctypes.data), inp.ndim,
                                        _i64(inp.shape), _i64(sin), _i64(out.shape), _i64(sout), int(axis), mode, scale,
                                        len(_PAR_DEVICES), ids)
        else:
            st = L.c.ndfft_exec(handler._plan, op, ctypes.c_void_p(inp.ctypes.data), ctypes.c_void_p(out.ctypes.data), inp.ndim,
                                _i64(inp.shape), _i64(sin), _i64(out.shape), _i64(sout), int(axis), mode, scale)
    L.check(st)
    if post_custom:
        host = out.cpu().numpy() if dev else out
        lanes = np.moveaxis(host, axis, -1)
        for idx in np.ndindex(lanes.shape[:-1]):
            lane = np.ascontiguousarray(lanes[idx])
            norm.fn(lane)
            lanes[idx] = lane
        if dev:
            out.copy_(torch.from_numpy(host))


def ndfft(input, output, handler, axis): _transform(_lib.OP_C2C_FWD, input, output, handler, axis)
def ndifft(input, output, handler, axis): _transform(_lib.OP_C2C_INV, input, output, handler, axis)
def ndfft_r2c(input, output, handler, axis): _transform(_lib.OP_R2C, input, output, handler, axis)
def ndifft_r2c(input, output, handler, axis): _transform(_lib.OP_C2R, input, output, handler, axis)
def nddct1(input, output, handler, axis): _transform(_lib.OP_DCT1, input, output, handler, axis)
def nddct2(input, output, handler, axis): _transform(_lib.OP_DCT2, input, output, handler, axis)
def nddct3(input, output, handler, axis): _transform(_lib.OP_DCT3, input, output, handler, axis)
def nddct4(input, output, handler, axis): _transform(_lib.OP_DCT4, input, output, handler, axis)


# #[cfg(feature = "parallel")] twins (lib.rs:374-421, 589-611, 777-844).  On one GPU every lane is processed in
# parallel anyway, so they are the same batched call; with set_par_devices([...]) they spread the call over several GPUs.
def ndfft_par(input, output, handler, axis): _transform(_lib.OP_C2C_FWD, input, output, handler, axis, par=True)
def ndifft_par(input, output, handler, axis): _transform(_lib.OP_C2C_INV, input, output, handler, axis, par=True)
def ndfft_r2c_par(input, output, handler, axis): _transform(_lib.OP_R2C, input, output, handler, axis, par=True)
def ndifft_r2c_par(input, output, handler, axis): _transform(_lib.OP_C2R, input, output, handler, axis, par=True)
def nddct1_par(input, output, handler, axis): _transform(_lib.OP_DCT1, input, output, handler, axis, par=True)
def nddct2_par(input, output, handler, axis): _transform(_lib.OP_DCT2, input, output, handler, axis, par=True)
def nddct3_par(input, output, handler, axis): _transform(_lib.OP_DCT3, input, output, handler, axis, par=True)
def nddct4_par(input, output, handler, axis): _transform(_lib.OP_DCT4, input, output, handler, axis, par=True)


def pinned_empty(shape, dtype, _library=None):
    """A C-layout numpy array in page-locked host memory from ndfft_host_alloc: ndfft_exec on such arrays overlaps
    upload, transform and download (include/ndfft_mi355x.h).  The memory is released when the array (and every
    view of it) is garbage-collected."""
    import weakref
    L = _library or _lib.default()
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    p = ctypes.c_void_p()
    L.check(L.c.ndfft_host_alloc(ctypes.byref(p), max(n * dt.itemsize, 1)))
    buf = (ctypes.c_char * max(n * dt.itemsize, 1)).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, L.c.ndfft_host_free, p)
    return arr
