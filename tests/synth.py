"""Synthetic inputs named by SURVEY.md section 8d: U[-1,1) from a splitmix64 stream.

value(k) = (splitmix64_k(seed) >> 11) * 2**-52 - 1, with k = flat_index (real arrays) or
2*flat_index (+1 for the imaginary part) for complex arrays; seed = 20241008.
Shared by tests/, bench.py and tests/golden/make_golden.py so every party sees the same numbers.
"""
import numpy as np

SEED = 20241008
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, k):
    """k-th output (k = 0, 1, ...) of the splitmix64 stream started at `seed`; vectorised over k."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (np.asarray(k, dtype=np.uint64) + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(seed, k):
    return (splitmix64(seed, k) >> np.uint64(11)).astype(np.float64) * 2.0 ** -52 - 1.0


def real_array(shape, dtype=np.float64, seed=SEED, offset=0):
    n = int(np.prod(shape))
    return uniform(seed, np.arange(offset, offset + n, dtype=np.uint64)).astype(dtype).reshape(shape)


def complex_array(shape, dtype=np.complex128, seed=SEED, offset=0):
    n = int(np.prod(shape))
    u = uniform(seed, np.arange(2 * offset, 2 * (offset + n), dtype=np.uint64))
    rdt = np.float32 if np.dtype(dtype) == np.complex64 else np.float64
    out = np.empty(n, dtype=dtype)
    out.real = u[0::2].astype(rdt)
    out.imag = u[1::2].astype(rdt)
    return out.reshape(shape)


def bench_fill_complex(shape, dtype=np.complex128):
    """The reference benches' fill: re = im = flat index (benches/ndrustfft.rs:15-18)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float64)
    return (i + 1j * i).astype(dtype).reshape(shape)


# ---- the same stream generated with torch (on the GPU for bench.py's large arrays: 65536 x 4096 c128 = 4 GiB) ----
def _s64(c):
    """uint64 constant as the int64 with the same bits (torch has no uint64 arithmetic; int64 wraps)."""
    c = int(c) & 0xFFFFFFFFFFFFFFFF
    return c - (1 << 64) if c >= (1 << 63) else c


def _lshr(z, s):
    return (z >> s) & ((1 << (64 - s)) - 1)


def uniform_torch(k, seed=SEED):
    """uniform(seed, k) for an int64 torch tensor k (any device); bit-identical to the numpy version."""
    z = (k + 1) * _s64(_GAMMA) + _s64(seed)
    z = (z ^ _lshr(z, 30)) * _s64(_M1)
    z = (z ^ _lshr(z, 27)) * _s64(_M2)
    z = z ^ _lshr(z, 31)
    return _lshr(z, 11).double() * 2.0 ** -52 - 1.0


def complex_array_torch(shape, device, offset=0, seed=SEED, chunk_rows=4096):
    """complex_array(shape, complex128, seed, offset) built on `device`, a block of rows at a time."""
    import torch
    rows = int(shape[0]); inner = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    out = torch.empty((rows, inner, 2), dtype=torch.float64, device=device)
    for r0 in range(0, rows, chunk_rows):
        r1 = min(rows, r0 + chunk_rows)
        k = torch.arange(2 * (offset + r0 * inner), 2 * (offset + r1 * inner), dtype=torch.int64, device=device)
        out[r0:r1] = uniform_torch(k, seed).view(r1 - r0, inner, 2)
    return torch.view_as_complex(out).view(*shape)
