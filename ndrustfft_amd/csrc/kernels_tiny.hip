// kernels_tiny.hip -- launcher of the thread-per-lane kernel for very short C2C lanes (tiny_kernel.h).
#include "tiny_kernel.h"

namespace ndfft {

bool tiny_supported(int n) { return (n >= 2 && n <= 13) || n == 16; }

template <typename T, int N, bool STAGE> static int launch_tiny_one(const TinyArgs &a, hipStream_t s) {
    using K = TinyFft<T, N, STAGE>;
    if constexpr (K::LDS_BYTES > 64 * 1024) { NDFFT_ENSURE_LDS_ATTR((k_tiny<K>)); }
    const int64_t nblk = (a.nlanes + K::THREADS - 1) / K::THREADS;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL(k_tiny<K>, dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template <typename T, int N> static int launch_tiny_n(bool stage, const TinyArgs &a, hipStream_t s) {
    return stage ? launch_tiny_one<T, N, true>(a, s) : launch_tiny_one<T, N, false>(a, s);
}

// stage: the lanes are dense and contiguous (elem stride 1, lane stride n, one batch dim): staged through LDS
int launch_tiny(int dtype, int n, bool stage, const TinyArgs &a, hipStream_t s) {
    switch (n) {
#define NDFFT_TINY_CASE(N_) case N_: return dtype == NDFFT_F32 ? launch_tiny_n<float, N_>(stage, a, s) : launch_tiny_n<double, N_>(stage, a, s);
        NDFFT_TINY_CASE(2) NDFFT_TINY_CASE(3) NDFFT_TINY_CASE(4) NDFFT_TINY_CASE(5) NDFFT_TINY_CASE(6) NDFFT_TINY_CASE(7) NDFFT_TINY_CASE(8)
        NDFFT_TINY_CASE(9) NDFFT_TINY_CASE(10) NDFFT_TINY_CASE(11) NDFFT_TINY_CASE(12) NDFFT_TINY_CASE(13) NDFFT_TINY_CASE(16)
#undef NDFFT_TINY_CASE
        default: return fail(NDFFT_ERR_UNSUPPORTED, "tiny kernel: unsupported n");
    }
}

}  // namespace ndfft
