"""ndrustfft_amd -- MI355X-native engine behind ndrustfft's axis-transform API.

Host-side mirror (Python flavour) of the reference's public surface; the compute lives in
hand-written HIP behind the C ABI of include/ndfft_mi355x.h.  No CPU fallback.
"""
from ._lib import NdfftError, Panic  # noqa: F401
from .api import (nddct1, nddct1_par, nddct2, nddct2_par, nddct3, nddct3_par, nddct4, nddct4_par,  # noqa: F401
                  ndfft, ndfft_par, ndfft_r2c, ndfft_r2c_par, ndifft, ndifft_par, ndifft_r2c, ndifft_r2c_par, par_devices, pinned_empty, set_par_devices)
from .handlers import DctHandler, FftHandler, Normalization, R2cFftHandler  # noqa: F401

__all__ = [
    "ndfft", "ndifft", "ndfft_r2c", "ndifft_r2c", "nddct1", "nddct2", "nddct3", "nddct4",
    "ndfft_par", "ndifft_par", "ndfft_r2c_par", "ndifft_r2c_par", "nddct1_par", "nddct2_par", "nddct3_par",
    "nddct4_par", "pinned_empty", "set_par_devices", "par_devices", "FftHandler", "R2cFftHandler", "DctHandler", "Normalization", "NdfftError", "Panic",
]
