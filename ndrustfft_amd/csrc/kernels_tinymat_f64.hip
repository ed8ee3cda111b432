// kernels_tinymat_f64.hip -- thread-per-lane real-op kernels, f64 (tinymat_kernel.h)
#define NDFFT_TM_T double
#define NDFFT_TM_NAME launch_tinymat_f64
#include "kernels_tinymat.inc"
