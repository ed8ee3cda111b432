"""Error metrics named by SURVEY.md section 8c: per-lane relative L2 and global max-abs/max-abs."""
import numpy as np

# tolerances stated by BASELINE.json north_star
TOL = {np.dtype(np.float64): 1e-10, np.dtype(np.float32): 1e-4,
       np.dtype(np.complex128): 1e-10, np.dtype(np.complex64): 1e-4}

GOLDEN_SIZES = list(range(1, 18)) + [24, 30, 37, 64, 100, 128, 129, 264, 265, 512, 513]


def rel_global(got, ref):
    got = np.asarray(got); ref = np.asarray(ref)
    d = np.abs(got.astype(np.complex128) - ref.astype(np.complex128)).max() if got.size else 0.0
    s = np.abs(ref).max() if ref.size else 0.0
    return d / s if s > 0 else d


def rel_lane_l2(got, ref, axis):
    got = np.asarray(got).astype(np.complex128); ref = np.asarray(ref).astype(np.complex128)
    num = np.sqrt((np.abs(got - ref) ** 2).sum(axis=axis))
    den = np.sqrt((np.abs(ref) ** 2).sum(axis=axis))
    den = np.where(den > 0, den, 1.0)
    return (num / den).max() if num.size else 0.0


def assert_close(got, ref, axis, tol, what=""):
    g = rel_global(got, ref); l = rel_lane_l2(got, ref, axis)
    assert g <= tol and l <= tol, f"{what}: global rel {g:.3e}, lane L2 rel {l:.3e} > {tol:.1e}"


def cdt_of(rdt):
    return np.complex64 if np.dtype(rdt) == np.float32 else np.complex128
