// kernels_fourstep.hip -- the two passes of the ROW four-step for lanes longer than one workgroup (exec.hip: big_fft),
// both on the register-resident column kernels of pow2_real.h, no transpose launch:
//   pass 1  x[n1 F2 + n2]: FFTs of length F1 over n1 (stride F2, adjacent n2 contiguous: column LOAD), stored
//           TRANSPOSED as s[n2][k1] (every lane one contiguous run: ROW store)                      -- ROWOUT kernels
//   pass 2  s[n2][k1]: FFTs of length F2 over n2 (stride F1, adjacent k1 contiguous), twiddle W_F^(n2 k1) on load,
//           stored at k1 + F1 k2 = natural order (column store)                                     -- CS = 4 kernels
// Replaces transpose -> row FFT -> twiddle -> transpose -> row FFT -> transpose (six passes) and the three-pass form.
#include "col_direct.h"
#include <cstdlib>

namespace ndfft {

// the E = 8 configurations of kernels_pow2_real.hip (per-pass twiddles: the C2C plans' twp_col tables)
template <int F> struct FsCfg;
#define NDFFT_FS(F_, TPL_, ...) template <> struct FsCfg<F_> { static constexpr int TPL = TPL_; using RL = RadixList<__VA_ARGS__>; };
NDFFT_FS(64, 8, 8, 8)
NDFFT_FS(128, 16, 8, 4, 4)
NDFFT_FS(256, 32, 8, 8, 4)
NDFFT_FS(512, 64, 8, 8, 8)
NDFFT_FS(1024, 128, 8, 8, 4, 4)

// adjacent lanes per tile (compile-time knobs, A/B in profiles/)
#ifndef NDFFT_FS_LANES_F64
#define NDFFT_FS_LANES_F64 8
#endif
#ifndef NDFFT_FS_LANES_F32
#define NDFFT_FS_LANES_F32 16
#endif
template <typename T, int F> struct FsGeom {
    static constexpr int TPL = FsCfg<F>::TPL;
    // adjacent lanes per tile: 128-byte rows (8 complex f64 / 16 complex f32), at least 256 threads.  Narrower than the
    // 32-lane tiles of the general column kernels on purpose: F = 256 f64 then takes 35 KiB of LDS instead of 140 KiB,
    // four workgroups per CU instead of one (256 x 65536 c128: 251 -> see DESIGN.md section 3.5)
    // F = 1024 c128: 4 lanes (64-byte rows, 70 KiB = two workgroups per CU) since the half-line tiles run in XCD runs (pow2_real.h): 32 x 2^20 c128
    // 591 -> 569 us (profiles/r06/r06u_*; without the runs the same tiles lost 15 %: profiles/r05/r05g_*)
#ifndef NDFFT_FS_LANES_F64_1024
#define NDFFT_FS_LANES_F64_1024 4
#endif
#ifndef NDFFT_FS_LANES_F32_1024
#define NDFFT_FS_LANES_F32_1024 NDFFT_FS_LANES_F32
#endif
    static constexpr int MINL = F == 1024 ? (sizeof(T) == 8 ? NDFFT_FS_LANES_F64_1024 : NDFFT_FS_LANES_F32_1024) : (sizeof(T) == 8 ? NDFFT_FS_LANES_F64 : NDFFT_FS_LANES_F32);
    static constexpr int LPB = TPL * MINL < 256 ? 256 / TPL : (TPL * MINL > 1024 ? 1024 / TPL : MINL);
    static_assert(TPL * LPB <= 1024, "workgroup too large");
};

// Entry point with a floor on waves per SIMD (= a cap on VGPRs).  c64: 8 waves = 64 VGPRs, which these kernels fit (at most 28 bytes of scratch in the forward
// transposing pass); without it they take ~80 and a 1024-thread tile is ONE workgroup per CU: 32 x 2^20 c64 375 -> 302 us (0.18 -> 0.22; profiles/r06/r06zu_*).
// c128 runs on the lane-fastest kernels below, which have their own floor (col_direct.h).
#ifndef NDFFT_FS_STAGED_MIN_WAVES_F32
#define NDFFT_FS_STAGED_MIN_WAVES_F32 8
#endif
template <typename K, typename T, int MW> __global__ __launch_bounds__(K::THREADS, MW) void k_fs_staged(const RealArgs<T> a) { K::run(a); }
template <typename T> struct FsStagedWaves { static constexpr int value = sizeof(T) == 4 ? NDFFT_FS_STAGED_MIN_WAVES_F32 : 1; };

template <typename T, int F, int OP, int CS, bool ROWOUT> static int launch_fs(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = FsGeom<T, F>::LPB;
    using K = RealPow2Kernel<T, F, FsCfg<F>::TPL, LPB, typename FsCfg<F>::RL, OP, true, false, CS, ROWOUT>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
    NDFFT_ENSURE_LDS_ATTR((k_fs_staged<K, T, FsStagedWaves<T>::value>));
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    RealArgs<T> b = a;
    real_args_set_inner_shift(b, LPB);
    hipLaunchKernelGGL((k_fs_staged<K, T, FsStagedWaves<T>::value>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, b);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// the lane-fastest register kernels of col_direct.h for the same two passes.  c128: faster since their VGPR cap (two workgroups per CU) -- 256 x 65536 218 -> 204 us,
// 16 x 2^20 282 -> 261 us (profiles/r06/r06zv_*); c64: slower (32 x 2^20 372 -> 542 us: a wavefront of 8-16 lanes writes 32-64-byte runs in the transposing pass) and stays on
// the staged kernels.  NDFFT_FS_DIRECT=0 / 1 forces one form (read per call: the parity tests switch it).
template <typename T> static bool fs_direct() {
    const int f = sw().fs_direct;                    // NDFFT_FS_DIRECT
    return f >= 0 ? f == 1 : sizeof(T) == 8;
}
template <typename T, int F, int OP, int MODE> static int launch_fsd(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = FsGeom<T, F>::LPB;
    using K = ColDirectKernel<T, F, FsCfg<F>::TPL, LPB, typename FsCfg<F>::RL, OP, MODE>;
    static_assert(K::LDS_BYTES <= 160 * 1024, "a workgroup's LDS");
    NDFFT_ENSURE_LDS_ATTR((k_col_direct<K, T>));
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL((k_col_direct<K, T>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// (round 6: c64 with ONLY the second pass on the lane-fastest kernel -- its column store writes the same 64-byte rows as the staged form -- measured 32 x 2^20 c64 259 -> 363 us: not kept)
bool fourstep_supported(int F) { return F == 64 || F == 128 || F == 256 || F == 512 || F == 1024; }

// The WIDE recipe of the 1024-point passes (round 6): 64 threads x 16 elements per lane, radix 16.8.8, on the lane-fastest kernel.  With E = 8 a 1024-point lane needs 128 threads,
// and 1024 threads hold 8 lanes -- for c128 the tiles were cut to 4 lanes (64-byte rows) to keep two workgroups per CU, for c64 8 lanes ARE 64-byte rows.  With E = 16 the same
// threads hold twice the lanes: c128 8 lanes x 64 threads (128-byte rows, 70 KiB of half-exchange LDS), c64 would be 16 lanes x 64 threads = 1024 threads (see fourstep_wide: not built).
using WideRL = RadixList<16, 8, 8>;
static constexpr int kWideTPL = 64;
void fourstep_build_wide_twiddles(int F, HostTable &out) { if (F == 1024) build_tw<WideRL>(out); }
bool fourstep_wide(int dtype, int pass, int F) {
    if (F != 1024) return false;
    const long k = NDFFT_DEV_INT("NDFFT_FS_WIDE", 1);          // developer build: 0 = off (A/B)
    if (!k || sw().fs_direct == 0) return false;
    // (c64: 16 lanes x 64 threads = 1024 threads would need <= 64 VGPRs for two workgroups per CU -- the E = 16 recipe spills ~4 KiB per thread under that cap: f64 only)
    return dtype == NDFFT_F64;
}
template <typename K, typename T, int MW> __global__ __launch_bounds__(K::THREADS, MW) void k_col_direct_w(const RealArgs<T> a) { K::run(a); }
template <typename T, int OP, int MODE> static int launch_fsd_wide(const RealArgs<T> &a, hipStream_t s) {
    constexpr int LPB = 8;                                       // 128-byte rows of c128
    using K = ColDirectKernel<T, 1024, kWideTPL, LPB, WideRL, OP, MODE>;
    constexpr int MW = 4;                                        // 512 threads, <= 128 VGPRs: two workgroups per CU
    static_assert(K::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    NDFFT_ENSURE_LDS_ATTR((k_col_direct_w<K, T, MW>));
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL((k_col_direct_w<K, T, MW>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// pass = 1: column load / row store; pass = 2: twiddle by the inner index on load, column store
template <typename T> int launch_fourstep(int pass, int F, bool inverse, const RealArgs<T> &a, hipStream_t s) {
    if constexpr (sizeof(T) == 8) if (a.wide && F == 1024 && !a.makhoul) {
        if (pass == 1) return inverse ? launch_fsd_wide<T, G_C2C_INV, 0>(a, s) : launch_fsd_wide<T, G_C2C_FWD, 0>(a, s);
        return inverse ? launch_fsd_wide<T, G_C2C_INV, 4>(a, s) : launch_fsd_wide<T, G_C2C_FWD, 4>(a, s);
    }
#define NDFFT_FS_CASE(F_)                                                                                              \
    case F_:                                                                                                           \
        if (fs_direct<T>() && !a.makhoul) {   /* (the fused DCT-IV first pass exists in the staged form only) */                \
            if (pass == 1) return inverse ? launch_fsd<T, F_, G_C2C_INV, 0>(a, s) : launch_fsd<T, F_, G_C2C_FWD, 0>(a, s); \
            return inverse ? launch_fsd<T, F_, G_C2C_INV, 4>(a, s) : launch_fsd<T, F_, G_C2C_FWD, 4>(a, s);            \
        }                                                                                                              \
        if (pass == 1) return inverse ? launch_fs<T, F_, G_C2C_INV, 0, true>(a, s) : launch_fs<T, F_, G_C2C_FWD, 0, true>(a, s); \
        return inverse ? launch_fs<T, F_, G_C2C_INV, 4, false>(a, s) : launch_fs<T, F_, G_C2C_FWD, 4, false>(a, s);
    switch (F) {
        NDFFT_FS_CASE(64) NDFFT_FS_CASE(128) NDFFT_FS_CASE(256) NDFFT_FS_CASE(512) NDFFT_FS_CASE(1024)
        default: return fail(NDFFT_ERR_UNSUPPORTED, "row four-step: unsupported factor");
    }
#undef NDFFT_FS_CASE
}
template int launch_fourstep<float>(int, int, bool, const RealArgs<float> &, hipStream_t);
template int launch_fourstep<double>(int, int, bool, const RealArgs<double> &, hipStream_t);

}  // namespace ndfft
