// kernels_wave.hip -- launcher of the LDS-free wavefront kernel for short dense C2C lanes (wave_kernel.h).
#include "wave_kernel.h"

namespace ndfft {

bool wave_supported(int n) { return n >= 2 && n <= 64 && (n & (n - 1)) == 0; }

template <typename T, int LOGN> static int launch_wave_one(const WaveArgs &a, hipStream_t s) {
    using K = WaveFft<T, LOGN>;
    const int64_t per_block = (int64_t)K::CHUNK * (K::THREADS / 64);
    const int64_t nblk = (a.total + per_block - 1) / per_block;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    WaveArgs b = a;
    b.xcd_chunk = xcd_chunk_for((size_t)per_block * sizeof(cpx<T>), nblk);
    hipLaunchKernelGGL(k_wave<K>, dim3((unsigned)nblk), dim3(K::THREADS), 0, s, b);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

int launch_wave(int dtype, int n, const WaveArgs &a, hipStream_t s) {
    int logn = 0; while ((1 << logn) < n) ++logn;
    switch (logn) {
#define NDFFT_WAVE_CASE(L) case L: return dtype == NDFFT_F32 ? launch_wave_one<float, L>(a, s) : launch_wave_one<double, L>(a, s);
        NDFFT_WAVE_CASE(1) NDFFT_WAVE_CASE(2) NDFFT_WAVE_CASE(3) NDFFT_WAVE_CASE(4) NDFFT_WAVE_CASE(5) NDFFT_WAVE_CASE(6)
#undef NDFFT_WAVE_CASE
        default: return fail(NDFFT_ERR_UNSUPPORTED, "wave kernel: unsupported n");
    }
}

}  // namespace ndfft
