// kernels_fourstep_real.hip -- the passes of the REAL row four-step for long contiguous real-data lanes (exec.hip: real_fourstep).
// n = N1 * N2 real points per lane, x[n1 N2 + n2]:
//   stage 1  real FFTs of length N1 over n1 (stride N2, adjacent n2 contiguous: column LOAD; DCT-II: through Makhoul's permutation),
//            half spectrum k1 = 0..N1/2 stored TRANSPOSED as s[n2][k1] (every lane one contiguous run: ROW store)      -- R2C ROWOUT kernels
//   stage 2  s[n2][k1]: complex FFTs of length N2 over n2 (stride N1/2 + 1, adjacent k1 contiguous), twiddle W_n^(n2 k1) on load;
//            X[k1 + N1 k2] for k <= n/2, the other half of every lane conjugated into its mirrored place                -- CS = 5 kernels
//   stage 3  the same with the DCT-II post-twiddle fused: two real outputs per spectrum element                          -- CS = 6 kernels
// and the inverse direction (C2R, DCT-III), whose LAST pass is the ordinary column C2R kernel of length N1 (kernels_pow2_real.hip):
//   stage 4  X[k1 + N1 k2], k1 = 0..N1/2 (Hermitian gather): inverse FFTs of length N2 over k2, times W_n^(-n2 k1), row store s[k1][n2]   -- col_direct.h mode 7
//   stage 5  the same with DCT-III's pre-twiddle V[k] built from x[k], x[n-k] on load                                                       -- col_direct.h mode 8
// and the second pass of the fused DCT-IV four-step (exec.hip: dct4_fourstep; its first pass is kernels_fourstep.hip's with makhoul = 2):
//   stage 6  twiddled complex FFTs of length F2, y[2k] = Re(Z[k] c_k), y[n-1-2k] = -Im(Z[k] c_k)                                           -- col_direct.h mode 9
// Replaces (packed complex four-step of length n/2 = two passes) + (split pass) [+ (Makhoul pass)]: 3 (R2C) or 4 (DCT-II) passes
// over global memory become 2.  Reference semantics: R2cFftHandler / DctHandler accept any n (src/lib.rs:477, 665).
#include "col_direct.h"
#include <cstdlib>

namespace ndfft {

// the E = 8 configurations of kernels_pow2_real.hip (same tables as kernels_fourstep.hip)
template <int F> struct RfsCfg;
#define NDFFT_RFS(F_, TPL_, ...) template <> struct RfsCfg<F_> { static constexpr int TPL = TPL_; using RL = RadixList<__VA_ARGS__>; };
NDFFT_RFS(64, 8, 8, 8)
NDFFT_RFS(128, 16, 8, 4, 4)
NDFFT_RFS(256, 32, 8, 8, 4)
NDFFT_RFS(512, 64, 8, 8, 8)
NDFFT_RFS(1024, 128, 8, 8, 4, 4)

// adjacent lanes per tile.  Stage 1 reads REAL rows: 16 doubles / 32 floats = 128 bytes (compile-time knobs, A/B in profiles/);
// stages 2 / 3 read complex rows like the complex four-step (8 c128 / 16 c64)
#ifndef NDFFT_RFS_LANES1_F64
#define NDFFT_RFS_LANES1_F64 16
#endif
#ifndef NDFFT_RFS_LANES1_F32
#define NDFFT_RFS_LANES1_F32 32
#endif
#ifndef NDFFT_RFS_LANES2_F64
#define NDFFT_RFS_LANES2_F64 8
#endif
#ifndef NDFFT_RFS_LANES2_F32
#define NDFFT_RFS_LANES2_F32 16
#endif
template <typename T, int F, int STAGE> struct RfsGeom {
    static constexpr int TPL = RfsCfg<F>::TPL;
    static constexpr int WANT = STAGE == 1 ? (sizeof(T) == 8 ? NDFFT_RFS_LANES1_F64 : NDFFT_RFS_LANES1_F32) : (sizeof(T) == 8 ? NDFFT_RFS_LANES2_F64 : NDFFT_RFS_LANES2_F32);
    // at most 1024 threads, at least 256; tiles above 80 KiB of LDS (one workgroup per CU) are halved down to 8 lanes
    static constexpr size_t LANE_BYTES = (size_t)((F + (F >> 4) + 2) | 1) * 2 * sizeof(T);
    static constexpr int L0 = WANT * TPL > 1024 ? 1024 / TPL : WANT;
    static constexpr int L1 = (L0 > 8 && L0 * LANE_BYTES > 80 * 1024) ? L0 / 2 : L0;
    static constexpr int L2 = (L1 > 8 && L1 * LANE_BYTES > 80 * 1024) ? L1 / 2 : L1;
    static constexpr int LPB = TPL * L2 < 256 ? 256 / TPL : L2;
    static_assert(TPL * LPB <= 1024, "workgroup too large");
};

// Entry point of the staged kernels of this file with a floor on waves per SIMD (= a cap on VGPRs).  f32: 8 waves = 64 VGPRs, which these kernels fit without
// scratch (they take 80-92 otherwise: ONE 1024-thread workgroup per CU) -- 64 x 262144 f32 nddct2 100 -> 85 us, ndfft_r2c 91 -> 78 us; f64 spills 20-224 bytes per lane
// under the same floor (ndifft_r2c 118 -> 205 us) and keeps the compiler's choice (profiles/r06/r06zu_*)
template <typename K, typename T, int MW> __global__ __launch_bounds__(K::THREADS, MW) void k_rfs_staged(const RealArgs<T> a) { K::run(a); }
template <typename T> struct RfsStagedWaves { static constexpr int value = sizeof(T) == 4 ? 8 : 1; };

template <typename T, int F, int STAGE> static int launch_rfs(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = RfsGeom<T, F, STAGE>::LPB;
    using K = typename cond_type<STAGE == 1,
                                 RealPow2Kernel<T, F, RfsCfg<F>::TPL, LPB, typename RfsCfg<F>::RL, G_R2C_EVEN, true, false, 0, true>,
                                 RealPow2Kernel<T, F, RfsCfg<F>::TPL, LPB, typename RfsCfg<F>::RL, G_C2C_FWD, true, false, STAGE == 2 ? 5 : 6, false>>::type;
    static_assert(K::LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
    NDFFT_ENSURE_LDS_ATTR((k_rfs_staged<K, T, RfsStagedWaves<T>::value>));
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    RealArgs<T> b = a;
    real_args_set_inner_shift(b, LPB);
    hipLaunchKernelGGL((k_rfs_staged<K, T, RfsStagedWaves<T>::value>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, b);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// stages 2 / 3 on the lane-fastest register kernels of col_direct.h: f64 by default (64 x 262144 nddct2 187 -> 155 us with runs of tiles per XCD
// and plain stores at shared lines), f32 stays on the staged column kernels above (120 vs 127 us); NDFFT_FS_DIRECT=0 / 1 forces one form
template <typename T> static bool rfs_direct() {
    const int f = sw().fs_direct;                    // NDFFT_FS_DIRECT
    return f >= 0 ? f == 1 : sizeof(T) == 8;
}
template <typename T, int F, int STAGE> static int launch_rfsd(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = RfsGeom<T, F, 2>::LPB;
    using K = ColDirectKernel<T, F, RfsCfg<F>::TPL, LPB, typename RfsCfg<F>::RL, G_C2C_FWD, STAGE + 3>;   // stages 2..6 = modes 5..9
    static_assert(K::LDS_BYTES <= 160 * 1024, "a workgroup's LDS");
    NDFFT_ENSURE_LDS_ATTR((k_col_direct<K, T>));          // RfsGeom<double, 1024, 2>: 69,696 B, above the 64 KiB a launch may ask for without the opt-in
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL((k_col_direct<K, T>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// last pass of the inverse direction: the column C2R kernel of length N1 = 2 F on tiles of 128-byte real rows (16 f64 / 32 f32 lanes, halved above 80 KiB) --
// the general column kernels (kernels_pow2_real.hip) use 32-lane tiles, 140 KiB = one workgroup per CU at F = 256 f64
template <typename T, int F> static int launch_rfs_c2r(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = RfsGeom<T, F, 1>::LPB;
    using K = RealPow2Kernel<T, F, RfsCfg<F>::TPL, LPB, typename RfsCfg<F>::RL, G_C2R_EVEN, true, false, 0, false>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
    NDFFT_ENSURE_LDS_ATTR((k_rfs_staged<K, T, RfsStagedWaves<T>::value>));
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    RealArgs<T> b = a;
    real_args_set_inner_shift(b, LPB);
    hipLaunchKernelGGL((k_rfs_staged<K, T, RfsStagedWaves<T>::value>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, b);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// N1 = real length of stage 1 (inner complex FFT N1 / 2), N2 = complex length of stage 2
bool fourstep_real_supported(int N1, int N2) {
    auto ok = [](int F) { return F == 64 || F == 128 || F == 256 || F == 512 || F == 1024; };
    return N1 % 2 == 0 && ok(N1 / 2) && ok(N2);
}

// stages 1, 7: F = N1 / 2; the others: F = N2
template <typename T> int launch_fourstep_real(int stage, int F, const RealArgs<T> &a, hipStream_t s) {
#define NDFFT_RFS_CASE(F_)                                       \
    case F_:                                                     \
        if (stage == 7) return launch_rfs_c2r<T, F_>(a, s);      \
        if (stage == 4) return launch_rfsd<T, F_, 4>(a, s);      \
        if (stage == 6) return launch_rfsd<T, F_, 6>(a, s);      \
        if (stage == 5) return launch_rfsd<T, F_, 5>(a, s);      \
        if (stage == 1) return launch_rfs<T, F_, 1>(a, s);       \
        if (rfs_direct<T>()) return stage == 2 ? launch_rfsd<T, F_, 2>(a, s) : launch_rfsd<T, F_, 3>(a, s); \
        if (stage == 2) return launch_rfs<T, F_, 2>(a, s);       \
        return launch_rfs<T, F_, 3>(a, s);
    switch (F) {
        NDFFT_RFS_CASE(64) NDFFT_RFS_CASE(128) NDFFT_RFS_CASE(256) NDFFT_RFS_CASE(512) NDFFT_RFS_CASE(1024)
        default: return fail(NDFFT_ERR_UNSUPPORTED, "real four-step: unsupported factor");
    }
#undef NDFFT_RFS_CASE
}
template int launch_fourstep_real<float>(int, int, const RealArgs<float> &, hipStream_t);
template int launch_fourstep_real<double>(int, int, const RealArgs<double> &, hipStream_t);

}  // namespace ndfft
