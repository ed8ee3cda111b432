// kernels_pow2_real.hip -- instantiations + launcher of the register-resident real-op kernels
// (see pow2_real.h).  F = inner complex FFT length.
#include "pow2_real.h"

namespace ndfft {

// F, threads per lane (E = F/TPL = 8 complex per thread), radices
#define NDFFT_REAL_CONFIGS(X) \
    X(64, 8, 8, 8)            \
    X(128, 16, 8, 4, 4)       \
    X(256, 32, 8, 8, 4)       \
    X(512, 64, 8, 8, 8)       \
    X(1024, 128, 8, 8, 4, 4)  \
    X(2048, 256, 8, 8, 8, 4)  \
    X(4096, 512, 8, 8, 8, 8)  \
    X(8192, 1024, 8, 8, 8, 4, 4)

template <int F> struct RealCfg;
#define NDFFT_DEF_RCFG(F_, TPL_, ...)               \
    template <> struct RealCfg<F_> {                \
        static constexpr int TPL = TPL_;            \
        using RL = RadixList<__VA_ARGS__>;          \
    };
NDFFT_REAL_CONFIGS(NDFFT_DEF_RCFG)

bool pow2_real_supported(int F) {
    switch (F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: return true;
        NDFFT_REAL_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return false;
    }
}

void pow2_real_build_twiddles(int F, HostTable &out) {
    switch (F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: build_tw<RealCfg<F_>::RL>(out); break;
        NDFFT_REAL_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: break;
    }
}

template <typename K, typename T> static int launch_k(const RealArgs<T> &a, int lpb, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_pow2_real<K, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL((k_pow2_real<K, T>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// threads of a COL workgroup: aim at 32 adjacent lanes per tile row, at most 1024 threads
static constexpr int col_threads(int tpl) { return tpl * 32 > 1024 ? 1024 : (tpl * 32 < 256 ? 256 : tpl * 32); }
template <typename T, int F> struct ColGeom {
    static constexpr int TPL = RealCfg<F>::TPL;
    static constexpr int LPB = col_threads(TPL) / TPL;
    static constexpr size_t LDS = (size_t)LPB * (((F + (F >> 4) + 2) | 1)) * 2 * sizeof(T);
    static constexpr bool OK = LPB >= 8 && LDS <= 160 * 1024;
};

template <typename T, int F, int OP> static int launch_real_one(const RealArgs<T> &a, bool col, hipStream_t s) {
    constexpr int TPL = RealCfg<F>::TPL;
    if (!col) {
        if constexpr (OP == G_C2C_FWD || OP == G_C2C_INV) return fail(NDFFT_ERR_INVALID_ARG, "row C2C goes through k_pow2");
        else {
            constexpr int LPB = TPL >= 256 ? 1 : 256 / TPL;
            return launch_k<RealPow2Kernel<T, F, TPL, LPB, typename RealCfg<F>::RL, OP, false>, T>(a, LPB, s);
        }
    }
    if constexpr (ColGeom<T, F>::OK) {
        constexpr int LPB = ColGeom<T, F>::LPB;
        return launch_k<RealPow2Kernel<T, F, TPL, LPB, typename RealCfg<F>::RL, OP, true>, T>(a, LPB, s);
    } else {
        return fail(NDFFT_ERR_UNSUPPORTED, "pow2 real kernel: no column tile for this F");
    }
}

template <typename T, int F> static int launch_real_F(int op, const RealArgs<T> &a, bool col, hipStream_t s) {
    switch (op) {
        case G_C2C_FWD: return launch_real_one<T, F, G_C2C_FWD>(a, col, s);
        case G_C2C_INV: return launch_real_one<T, F, G_C2C_INV>(a, col, s);
        case G_R2C_EVEN: return launch_real_one<T, F, G_R2C_EVEN>(a, col, s);
        case G_C2R_EVEN: return launch_real_one<T, F, G_C2R_EVEN>(a, col, s);
        case G_DCT1: return launch_real_one<T, F, G_DCT1>(a, col, s);
        case G_DCT2_EVEN: return launch_real_one<T, F, G_DCT2_EVEN>(a, col, s);
        case G_DCT3_EVEN: return launch_real_one<T, F, G_DCT3_EVEN>(a, col, s);
        case G_DCT4_EVEN: return launch_real_one<T, F, G_DCT4_EVEN>(a, col, s);
        default: return fail(NDFFT_ERR_INVALID_ARG, "pow2 real kernel: bad op");
    }
}

// lanes per column tile for inner FFT length F (0 = no column kernel)
template <typename T> int pow2_real_col_lanes(int F) {
    switch (F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: return ColGeom<T, F_>::OK ? ColGeom<T, F_>::LPB : 0;
        NDFFT_REAL_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return 0;
    }
}
template int pow2_real_col_lanes<float>(int);
template int pow2_real_col_lanes<double>(int);

template <typename T> int launch_pow2_real(int op, const RealArgs<T> &a, bool col, hipStream_t s) {
    switch (a.F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: return launch_real_F<T, F_>(op, a, col, s);
        NDFFT_REAL_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return fail(NDFFT_ERR_UNSUPPORTED, "pow2 real kernel: unsupported F");
    }
}
template int launch_pow2_real<float>(int, const RealArgs<float> &, bool, hipStream_t);
template int launch_pow2_real<double>(int, const RealArgs<double> &, bool, hipStream_t);

}  // namespace ndfft
