// kernels_tinymat_f32.hip -- thread-per-lane real-op kernels, f32 (tinymat_kernel.h)
#define NDFFT_TM_T float
#define NDFFT_TM_NAME launch_tinymat_f32
#include "kernels_tinymat.inc"
