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

EMUL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul")


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-C", EMUL_DIR, "-s", "-j4"])
    return _lib.Library(os.path.join(EMUL_DIR, "_build", "libndfft_emul.so"))


def test_reference_unit_tests(L, refvec): ps.reference_unit_tests(L, refvec)
def test_reference_examples(L, refvec): ps.reference_examples(L, refvec)
def test_layouts(L): ps.layouts(L)
def test_normalization(L): ps.normalization_modes(L)
def test_panics(L): ps.panics(L)
def test_clone(L): ps.handler_clone_shares_plan(L)
def test_interleaved_mut_views(L): ps.interleaved_mut_views_two_threads(L, rounds=1)
def test_long_strided_lanes(L): ps.long_strided_lanes(L)
def test_narrow_xcd_tiles(L): ps.narrow_xcd_tiles(L)
def test_column_four_step(L): ps.column_four_step(L)
def test_huge_prime_factors(L): ps.huge_prime_factors(L, full=False)
def test_fuzz(L): ps.fuzz(L, seed=11, count=120, max_points=1 << 13)
def test_bluestein_register_kernel(L): ps.bluestein_register_kernel(L)
def test_partial_round_configs(L): ps.partial_round_configs(L)
def test_long_lanes_four_step(L): ps.long_lanes_four_step(L, full=False)


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
