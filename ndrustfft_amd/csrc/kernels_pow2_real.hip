// kernels_pow2_real.hip -- instantiations + launcher of the register-resident real-op kernels
// (see pow2_real.h).  F = inner complex FFT length.
#include <algorithm>
#include <cstdlib>

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

// threads per lane + radix list of the E = 8 configuration for inner length F (also used to specialise
// blue_kernel.h for M = F with hiprtc)
bool pow2_real_config(int F, JitCfg &cfg) {
    switch (F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: cfg.n = F_; cfg.tpl = TPL_; cfg.e = F_ / TPL_; cfg.radix = {__VA_ARGS__}; return true;
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

// Entry point with a floor on waves per SIMD (= a cap on VGPRs) for the f32 WIDE COLUMN kernels whose workgroup is 512 or 1024 threads (from 512: nddct2 n = 128 31.5 -> 28.3 us, the rest
// unchanged, no scratch either): left alone
// they take 80-92 VGPRs = ONE workgroup per CU; a floor of 8 waves caps them at 64, which they fit without scratch.  Transform along axis 0 of 2^24-point f32 arrays
// (profiles/r06/r06zt_*): nddct2 n = 512 / 1024 45.5 / 49 -> 37.3 / 39.5 us, ndfft_r2c n = 512 / 1024 / 2048 42.5 / 44.4 / 49.7 -> 32.9 / 33.7 / 41 us; C2C and C2R unchanged.
// Not for: f64 (spills under any floor), the narrow XCD tiles (12-220 bytes of scratch at 64 VGPRs), DCT-III column tiles (its V[k] registers: 12 bytes).
#ifndef NDFFT_REAL_F32_FLOOR_THREADS
#define NDFFT_REAL_F32_FLOOR_THREADS 512
#endif
template <typename K, typename T, int MW> __global__ __launch_bounds__(K::THREADS, MW) void k_real_aot(const RealArgs<T> a) { K::run(a); }
template <typename K> struct IsWideCol { static constexpr bool value = false; };
template <typename T, int F, int TPL, int LPB, typename RL, int OP> struct IsWideCol<RealPow2Kernel<T, F, TPL, LPB, RL, OP, true, false, 0, false>> { static constexpr bool value = OP != G_DCT3_EVEN; };
// ... and for the f32 ROW kernels of 512 / 1024 threads (F = 4096 / 8192: 66 VGPRs left alone = 7 waves per SIMD, i.e. three 512-thread or ONE 1024-thread workgroup per CU
// where the LDS allows four / two): ndfft_r2c rows n = 16384 46.5 -> 34 us (0.36 -> 0.49) although that kernel keeps 12 bytes of scratch at 64 VGPRs, 8192 x 8192 f32 108.7 -> 101.9 us,
// n = 8192 30.5 -> 29.2 us (profiles/r06/r06zs_*)
#ifndef NDFFT_REAL_F32_ROW_FLOOR_THREADS
#define NDFFT_REAL_F32_ROW_FLOOR_THREADS 512
#endif
template <typename K> struct IsRowKernel { static constexpr bool value = false; };
template <typename T, int F, int TPL, int LPB, typename RL, int OP> struct IsRowKernel<RealPow2Kernel<T, F, TPL, LPB, RL, OP, false, false, 0, false>> { static constexpr bool value = true; };
template <typename K, typename T> struct RealAotWaves {
    static constexpr int value = (sizeof(T) == 4 && ((K::THREADS >= NDFFT_REAL_F32_FLOOR_THREADS && IsWideCol<K>::value) || (K::THREADS >= NDFFT_REAL_F32_ROW_FLOOR_THREADS && IsRowKernel<K>::value))) ? 8 : 1;
};

template <typename K, typename T> static int launch_k(const RealArgs<T> &a, int lpb, hipStream_t s) {
    NDFFT_ENSURE_LDS_ATTR((k_real_aot<K, T, RealAotWaves<K, T>::value>));
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    RealArgs<T> b = a;
    if (b.xcd_chunk < 0) {   // -1: the caller leaves the choice to the launcher
        if (a.inner > 1) {
            // column tiles: consecutive tiles are adjacent pieces of the SAME rows; 4 per XCD run (1 KiB of every row) measured
            // +3 % on cfg4' warm, +1 % cold, 16 / 64 nothing (profiles/r02n_xcd_map_real_kernels.txt); row kernels: neutral at n = 512
            const int col = (int)NDFFT_DEV_INT("NDFFT_XCD_CHUNK_COL", 4);
            b.xcd_chunk = nblk >= 16 * (int64_t)std::max(col, 1) ? col : 0;
        } else {
            const size_t esz = sizeof(T) * (K::IN_CPLX ? 2 : 1);
            b.xcd_chunk = xcd_chunk_for((size_t)lpb * (size_t)a.n_in * esz, nblk);
        }
    }
    if (a.inner > 1) real_args_set_inner_shift(b, lpb);
    hipLaunchKernelGGL((k_real_aot<K, T, RealAotWaves<K, T>::value>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, b);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// developer macro (variant builds only): 1 = the real-lane tiles of F = 1024 (nddct1 n = 1025, 8 lanes x 1024 threads) may narrow to 4 lanes too.  Measured by row pitch
// (profiles/r09/r09x_dct1_1025_narrow_tiles_by_pitch.txt, tools/probes/col_pitch_probe.py, HIP-graph replay): 1025 x W f64, 8 lanes: 10.8 us flat from W = 512 to 1088
// at pitches that are multiples of 64 bytes, 13.4 us at the reference's W = 1025 (pitch 8200: every 64-byte row segment straddles two 64-byte chunks); 4 lanes:
// 8.7 us at W = 512 (128 tiles) rising to 11.6 us at W = 1024 (256 tiles) and 18.0 us at W = 1025 (32-byte segments, misaligned).  So: not kept.
#ifndef NDFFT_SMALL_GRID_REAL_1024
#define NDFFT_SMALL_GRID_REAL_1024 0
#endif
#ifndef NDFFT_COL_LANES_F32
#define NDFFT_COL_LANES_F32 32
#endif
#ifndef NDFFT_SMALL_GRID_FLAGS
#define NDFFT_SMALL_GRID_FLAGS 32
#endif
#ifndef NDFFT_COL_LANES_F64
#define NDFFT_COL_LANES_F64 32
#endif
// threads of a COL workgroup: aim at 32 adjacent lanes per tile row, at most 1024 threads
static constexpr int col_threads(int tpl, int lanes) { return tpl * lanes > 1024 ? 1024 : (tpl * lanes < 256 ? 256 : tpl * lanes); }
// Rows of a column tile should be 256 bytes wide: tools/tilecopy.hip (profiles/r02t_tilecopy.txt) copies 128-row tiles whose
// rows are 2 MiB apart -- the column four-step's first stage on cfg3-A -- at 0.56-0.62 of 8 TB/s with 128-byte rows, 0.71-0.74
// with 256-byte rows, 0.68-0.71 with 512.  f32 REAL data (4 bytes per lane) therefore takes 64 lanes per tile where the
// tile still fits (REAL = the op reads or writes real lanes); everything else keeps 32 lanes (256 B for c64 / f64, 512 B for c128).
// KIND: 0 = C2C (complex rows on both sides), 1 = R2C (real in, complex out), 2 = C2R (complex in, real out), 3 = DCT (both sides real)
template <typename T, int F, int KIND = 0> struct ColGeom {
    static constexpr int TPL = RealCfg<F>::TPL;
    static constexpr int WANT = (sizeof(T) == 4 && KIND > 0) ? 2 * NDFFT_COL_LANES_F32 : (sizeof(T) == 4 ? NDFFT_COL_LANES_F32 : NDFFT_COL_LANES_F64);
    static constexpr int LPB0 = col_threads(TPL, WANT) / TPL;
    static constexpr size_t LANE_BYTES = (size_t)(((F + (F >> 4) + 2) | 1)) * 2 * sizeof(T);
    // A tile above 80 KiB is ONE 1024-thread workgroup per CU: nothing runs while it loads or stores.  f64 tiles are halved:
    //   F = 256 (32 -> 16 lanes): the strided axes of cfg4 89.5 / 91.8 -> 87.1 us, the last pass of the inverse real four-step -15 % (profiles/r06/r06s_*, r06t_*)
    //   F = 512 (16 -> 8 lanes):  ndfft axis 0 of 512 x 32768 c128 117 -> 87 us (0.57 -> 0.77), nddct2 / ndfft_r2c axis 0 of 1024 x 16384 f64 71 / 64 -> 57 / 50 us
    //                             although their real rows are 64 bytes then (profiles/r06/r06v_*)
    //   F = 1024 (8 -> 4 lanes) and F = 2048 (4 lanes, no column tile before): only where the OUTPUT rows are complex (64 bytes): ndfft axis 0 of 1024 x 16384 c128
    //                             128 -> 120 us, of 2048 x 8192 c128 216 (narrow XCD tiles) -> 157 us, ndfft_r2c f64 n = 2048 / 4096 68 / 132 -> 61 / 83 us;
    //                             32-byte output rows are ruinous (ndifft_r2c n = 2048 74 -> 241 us, nddct2 82 -> 200 us): not for KIND >= 2   (profiles/r06/r06w_*)
    // f32 tiles (F = 256 real ops: 64 lanes, 140 KiB) measured the same halved or not and stay.
    static constexpr int HALVE_FROM = sizeof(T) == 8 ? (KIND >= 2 ? 16 : 8) : 1000;
    static constexpr int LPB = (LPB0 >= HALVE_FROM && LPB0 * LANE_BYTES > 80 * 1024) ? LPB0 / 2 : LPB0;
    static constexpr size_t LDS = (size_t)LPB * LANE_BYTES;
#ifndef NDFFT_COL_MIN_LANES
    static constexpr int MIN_LANES = (sizeof(T) == 8 && KIND <= 1) ? 4 : 8;
#else
    static constexpr int MIN_LANES = NDFFT_COL_MIN_LANES;      // side builds only (tools/probes)
#endif
    static constexpr bool OK = LPB >= MIN_LANES && LDS <= 160 * 1024;
};

// narrow (XCD-aware) column tiles for long lanes: 1024 threads, LPB = 2 or 4 lanes
template <typename T, int F> struct NarrowCfg { static constexpr bool OK = false; };
#define NDFFT_NARROW(T_, F_, TPL_, LPB_, ...)                                   \
    template <> struct NarrowCfg<T_, F_> {                                     \
        static constexpr bool OK = true;                                       \
        static constexpr int TPL = TPL_, LPB = LPB_;                           \
        using RL = RadixList<__VA_ARGS__>;                                     \
    };
NDFFT_NARROW(float, 2048, 256, 4, 8, 8, 8, 4)
NDFFT_NARROW(float, 4096, 256, 4, 16, 16, 16)
NDFFT_NARROW(float, 8192, 512, 2, 16, 16, 8, 4)
NDFFT_NARROW(double, 2048, 256, 4, 8, 8, 8, 4)
NDFFT_NARROW(double, 4096, 512, 2, 8, 8, 8, 8)

template <typename T> int pow2_real_narrow_lanes(int F) {
    switch (F) {
        case 2048: return NarrowCfg<T, 2048>::OK ? NarrowCfg<T, 2048>::LPB : 0;
        case 4096: return NarrowCfg<T, 4096>::OK ? NarrowCfg<T, 4096>::LPB : 0;
        case 8192: if constexpr (NarrowCfg<T, 8192>::OK) return NarrowCfg<T, 8192>::LPB; else return 0;
        default: return 0;
    }
}
template int pow2_real_narrow_lanes<float>(int);
template int pow2_real_narrow_lanes<double>(int);

template <typename T, int F> static void narrow_tw(HostTable &out) {
    if constexpr (NarrowCfg<T, F>::OK) build_tw<typename NarrowCfg<T, F>::RL>(out);
}
void pow2_real_build_narrow_twiddles(int dtype, int F, HostTable &out) {
    switch (F) {
        case 2048: if (dtype == NDFFT_F32) narrow_tw<float, 2048>(out); else narrow_tw<double, 2048>(out); break;
        case 4096: if (dtype == NDFFT_F32) narrow_tw<float, 4096>(out); else narrow_tw<double, 4096>(out); break;
        case 8192: if (dtype == NDFFT_F32) narrow_tw<float, 8192>(out); else narrow_tw<double, 8192>(out); break;
        default: break;
    }
}

template <typename T, int F, int OP> static int launch_narrow_one(const RealArgs<T> &a, hipStream_t s) {
    if constexpr (NarrowCfg<T, F>::OK) {
        using C = NarrowCfg<T, F>;
        return launch_k<RealPow2Kernel<T, F, C::TPL, C::LPB, typename C::RL, OP, true, true>, T>(a, C::LPB, s);
    } else {
        return fail(NDFFT_ERR_UNSUPPORTED, "no narrow column kernel for this F");
    }
}
template <typename T, int F> static int launch_narrow_F(int op, const RealArgs<T> &a, hipStream_t s) {
    switch (op) {
        case G_C2C_FWD: return launch_narrow_one<T, F, G_C2C_FWD>(a, s);
        case G_C2C_INV: return launch_narrow_one<T, F, G_C2C_INV>(a, s);
        case G_R2C_EVEN: return launch_narrow_one<T, F, G_R2C_EVEN>(a, s);
        case G_C2R_EVEN: return launch_narrow_one<T, F, G_C2R_EVEN>(a, s);
        case G_DCT1: return launch_narrow_one<T, F, G_DCT1>(a, s);
        case G_DCT2_EVEN: return launch_narrow_one<T, F, G_DCT2_EVEN>(a, s);
        case G_DCT3_EVEN: return launch_narrow_one<T, F, G_DCT3_EVEN>(a, s);
        case G_DCT4_EVEN: return launch_narrow_one<T, F, G_DCT4_EVEN>(a, s);
        default: return fail(NDFFT_ERR_INVALID_ARG, "narrow kernel: bad op");
    }
}
template <typename T> int launch_pow2_real_narrow(int op, const RealArgs<T> &a, hipStream_t s) {
    switch (a.F) {
        case 2048: return launch_narrow_F<T, 2048>(op, a, s);
        case 4096: return launch_narrow_F<T, 4096>(op, a, s);
        case 8192: return launch_narrow_F<T, 8192>(op, a, s);
        default: return fail(NDFFT_ERR_UNSUPPORTED, "narrow kernel: unsupported F");
    }
}
template int launch_pow2_real_narrow<float>(int, const RealArgs<float> &, hipStream_t);
template int launch_pow2_real_narrow<double>(int, const RealArgs<double> &, hipStream_t);

// threads per row workgroup (whole lanes): a documented compile-time knob, see DESIGN.md section 6
#ifndef NDFFT_REAL_ROW_THREADS
#define NDFFT_REAL_ROW_THREADS 256
#endif
template <typename T, int F, int OP> static int launch_real_one(const RealArgs<T> &a, bool col, hipStream_t s) {
    constexpr int TPL = RealCfg<F>::TPL;
    if (!col) {
        if constexpr (OP == G_C2C_FWD || OP == G_C2C_INV) return fail(NDFFT_ERR_INVALID_ARG, "row C2C goes through k_pow2");
        else {
            // one-wave workgroups for f64 and for short f32 lanes (alternating A-B-A-B runs, profiles/r04/r04s_abab_pow2real_*.txt: nddct2 f64 n = 128..1024 +2.5-6 %,
            // cfg4 nddct2 81.4 -> 78.8 us; ndfft_r2c f32 n = 128 / 256 +3-5 %, n >= 512 no gain: they keep 256 threads)
            constexpr int THR = (NDFFT_REAL_ROW_THREADS == 256 && (sizeof(T) == 8 || F <= 128)) ? 64 : NDFFT_REAL_ROW_THREADS;
            constexpr int LPB = TPL >= THR ? 1 : THR / TPL;
            // ... but TWO waves when the f64 input comes from HBM (stream_in, set by the residency model): A-B-A-B on cfg4 with three rotating pairs
            // nddct2 95.6 -> 93.7 us, nddct3 98.2 -> 97.1 us, nddct4 98.5 -> 95.7 us (profiles/r08/r08b_cfg4_row_threads_cold_abab.txt; 256 threads: no gain)
            if constexpr (sizeof(T) == 8 && THR == 64 && TPL <= 64 && F >= 128 && NDFFT_REAL_ROW_THREADS == 256) {
                constexpr int LPB2 = 128 / TPL;
                if (a.stream_in) return launch_k<RealPow2Kernel<T, F, TPL, LPB2, typename RealCfg<F>::RL, OP, false>, T>(a, LPB2, s);
            }
            return launch_k<RealPow2Kernel<T, F, TPL, LPB, typename RealCfg<F>::RL, OP, false>, T>(a, LPB, s);
        }
    }
    constexpr int KIND = (OP == G_C2C_FWD || OP == G_C2C_INV) ? 0 : (OP == G_R2C_EVEN ? 1 : (OP == G_C2R_EVEN ? 2 : 3));   // which sides are real lanes (cfg3-A 210 -> 202 us, cfg3-A' 259 -> 235 us)
    if constexpr (ColGeom<T, F, KIND>::OK) {
        constexpr int LPB = ColGeom<T, F, KIND>::LPB;
        // Small grids (the reference's own bench shapes, benches/ndrustfft.rs:6-7: n x n arrays with n = 512 ... 1025 along axis 0): with the tile widths above
        // the whole call is 33 ... 128 workgroups on 256 CUs, and what a call costs is the latency of ONE tile.  Narrower tiles (down to 4 f64 lanes = 32 / 64-byte
        // rows -- ruinous for arrays that have to come from HBM, irrelevant for a few MiB) spread the lanes over more CUs: replayed from a HIP graph ndfft
        // n = 512 6.15 -> 4.98 us, ndfft_r2c n = 512 / 1024 7.06 / 7.24 -> 5.43 / 6.44 us, nddct1 n = 513 7.34 -> 6.43 us; nddct1 n = 1025 (F = 1024, real rows)
        // measured no better (14.2 -> 14.9 us) and keeps its tile (profiles/r08/r08h_small_shapes_col_lanes.txt).  Caller's column tiles only (not the stages of the
        // four-step routes, which set keep_out / makhoul / stream_in themselves).
        // The small-grid tiles run the LATENCY form of the passes (pow2_kernel.h FLAGS 32: next pass's twiddles loaded before the exchange, LDS-only barriers): SMALL_FLAGS.
        if constexpr (sizeof(T) == 8 && F >= 256) {
            constexpr int64_t kCus = 256;
            constexpr int SMALL_FLAGS = NDFFT_SMALL_GRID_FLAGS;
            // (not for an input the residency model marks HBM-sourced: 32 / 64-byte rows are ruinous there, and a call of few tiles whose array is not cache-resident is rare)
            if (!a.keep_out && !a.makhoul && !a.stream_in) {
                if constexpr (!(F >= 1024 && KIND >= 2) || NDFFT_SMALL_GRID_REAL_1024) {
                    if constexpr (LPB / 4 >= 4) {
                        if ((a.nlanes + LPB / 2 - 1) / (LPB / 2) < kCus)
                            return launch_k<RealPow2Kernel<T, F, TPL, LPB / 4, typename RealCfg<F>::RL, OP, true, false, 0, false, SMALL_FLAGS>, T>(a, LPB / 4, s);
                    }
                    if constexpr (LPB / 2 >= 4) {
                        if ((a.nlanes + LPB - 1) / LPB < kCus)
                            return launch_k<RealPow2Kernel<T, F, TPL, LPB / 2, typename RealCfg<F>::RL, OP, true, false, 0, false, SMALL_FLAGS>, T>(a, LPB / 2, s);
                    }
                }
                // tiles that cannot shrink (ndfft n = 1024: 4 lanes already; nddct1 n = 1025 keeps 8): the same tile in the latency form when the call is at most two tiles per CU
                if constexpr (F >= 1024 && SMALL_FLAGS != 0) {
                    if ((a.nlanes + LPB - 1) / LPB <= 2 * kCus)
                        return launch_k<RealPow2Kernel<T, F, TPL, LPB, typename RealCfg<F>::RL, OP, true, false, 0, false, SMALL_FLAGS>, T>(a, LPB, s);
                }
            }
        }
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

// lanes per column tile for inner FFT length F (0 = no column kernel); kind: 0 = C2C, 1 = R2C, 2 = C2R, 3 = DCT
template <typename T> int pow2_real_col_lanes(int F, int kind) {
    switch (F) {
#define NDFFT_CASE(F_, TPL_, ...) case F_: return kind == 0 ? (ColGeom<T, F_, 0>::OK ? ColGeom<T, F_, 0>::LPB : 0) : kind == 1 ? (ColGeom<T, F_, 1>::OK ? ColGeom<T, F_, 1>::LPB : 0) \
                                                       : kind == 2 ? (ColGeom<T, F_, 2>::OK ? ColGeom<T, F_, 2>::LPB : 0) : (ColGeom<T, F_, 3>::OK ? ColGeom<T, F_, 3>::LPB : 0);
        NDFFT_REAL_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return 0;
    }
}
template int pow2_real_col_lanes<float>(int, int);
template int pow2_real_col_lanes<double>(int, int);

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
