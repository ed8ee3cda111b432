// kernels_pow2.hip -- the north-star kernel: batched C2C FFT along the contiguous axis,
// power-of-two n, ONE read and ONE write of HBM per point (BASELINE cfg2 / cfg5 / cfg3-B).
//
// Replaces, per lane, FftHandler::fft_lane / ifft_lane (src/lib.rs:313-331) together with the
// strategy-(i) row loop that calls it (src/lib.rs:117-124, par 187-194).
//
// Design (CDNA4):
//  * A lane lives in REGISTERS: TPL threads x E = n/TPL complex elements each.  The first radix
//    pass reads global memory directly in the Stockham input pattern x[j + r n/R] -- consecutive
//    threads, consecutive 16-byte elements, 1 KiB per wave instruction -- and the last pass writes
//    y[j + r n/R] the same way, so there is no staging copy on either side.
//  * Between passes the lane is exchanged through LDS at its natural Stockham index, padded by one
//    element every 16 (phi(p) = p + p/16) so the stride-R writes of early passes and the unit
//    stride reads of the next pass are both bank-conflict free.
//  * HALF exchange: real parts, then imaginary parts, through ONE real-sized buffer.  n=4096 f64
//    needs 34 KiB instead of 68 KiB, i.e. 4 workgroups per CU instead of 2 -- the occupancy that
//    keeps enough HBM requests in flight while other workgroups are in their butterfly phase.
//  * Twiddles come from per-pass tables laid out [r-1][k] so that a wave reads them coalesced;
//    they are built on the host in long double (plan.hip) and stay L2-resident (<= 64 KiB).
//  * Streaming cache policy: non-temporal stores (and loads, for inputs larger than the Infinity
//    Cache) -- see launch_pow2.
//  * No MFMA: ~1.9 flop/byte, the kernel is HBM-bound by design.
#include <algorithm>
#include <cstdlib>

#include "pow2_kernel.h"

namespace ndfft {

// N, threads-per-lane, radices.  E = N / TPL must be a multiple of every radix.
// f64 (16-byte elements, one element per global access)
#define NDFFT_POW2_CONFIGS_F64(X) \
    X(64, 8, 8, 8)            \
    X(128, 8, 16, 8)          \
    X(256, 16, 16, 16)        \
    X(512, 64, 8, 8, 8)       \
    X(1024, 64, 16, 8, 8)     \
    X(2048, 128, 16, 16, 8)   \
    X(4096, 512, 8, 8, 8, 8)  \
    X(8192, 512, 8, 8, 8, 16) \
    X(16384, 1024, 16, 16, 16, 4)
// f32 (8-byte elements): first and last radix <= E/2 so that two adjacent elements (16 B) move per
// global access (VEC = 2)
#define NDFFT_POW2_CONFIGS_F32(X) \
    X(64, 8, 8, 8)            \
    X(128, 8, 8, 2, 8)        \
    X(256, 16, 8, 4, 8)       \
    X(512, 32, 8, 8, 8)       \
    X(1024, 64, 8, 16, 8)     \
    X(2048, 128, 8, 4, 8, 8)  \
    X(4096, 256, 8, 8, 8, 8)  \
    X(8192, 512, 8, 16, 8, 8) \
    X(16384, 512, 16, 16, 8, 8)

// threads per workgroup (whole lanes): a documented compile-time knob
#ifndef NDFFT_POW2_ROW_THREADS
#define NDFFT_POW2_ROW_THREADS 256
#endif
static constexpr int lpb_for(int tpl) { return tpl >= NDFFT_POW2_ROW_THREADS ? 1 : NDFFT_POW2_ROW_THREADS / tpl; }

template <typename T, int N> struct Pow2Cfg;
#define NDFFT_DEF_CFG64(N_, TPL_, ...) \
    template <> struct Pow2Cfg<double, N_> { static constexpr int TPL = TPL_; using RL = RadixList<__VA_ARGS__>; };
#define NDFFT_DEF_CFG32(N_, TPL_, ...) \
    template <> struct Pow2Cfg<float, N_> { static constexpr int TPL = TPL_; using RL = RadixList<__VA_ARGS__>; };
NDFFT_POW2_CONFIGS_F64(NDFFT_DEF_CFG64)
NDFFT_POW2_CONFIGS_F32(NDFFT_DEF_CFG32)

// LDS exchange: HALF (real parts, then imaginary parts, through one real-sized buffer) halves the footprint and
// wins for f64 (occupancy).  f32 from n = 2048 up exchanges whole complex elements with 64-bit LDS accesses: half
// the LDS instructions, and the footprint is that of the f64 half exchange anyway (measured after the packed-math
// change, tools/kbench f32_halffull: 2048 83 -> 78 us, 4096 86 -> 80, 8192 93 -> 89, 16384 146 -> 126)
template <typename T, int N> struct Pow2Half { static constexpr bool value = true; };
template <> struct Pow2Half<float, 2048> { static constexpr bool value = false; };
template <> struct Pow2Half<float, 4096> { static constexpr bool value = false; };
template <> struct Pow2Half<float, 8192> { static constexpr bool value = false; };
template <> struct Pow2Half<float, 16384> { static constexpr bool value = false; };

bool pow2_supported(int dtype, int n) {
    (void)dtype;
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...) case N_: return true;
        NDFFT_POW2_CONFIGS_F64(NDFFT_CASE)
#undef NDFFT_CASE
        default: return false;
    }
}

void pow2_build_twiddles(int dtype, int n, HostTable &out) {
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...) \
    case N_: if (dtype == NDFFT_F32) build_tw<Pow2Cfg<float, N_>::RL>(out); else build_tw<Pow2Cfg<double, N_>::RL>(out); break;
        NDFFT_POW2_CONFIGS_F64(NDFFT_CASE)
#undef NDFFT_CASE
        default: break;
    }
}

// Lane blocks per XCD chunk of the workgroup -> lane map (device_common.h: xcd_block): every XCD streams contiguous
// runs of ~512 KiB instead of every eighth lane block.  Measured with tools/kbench f64_map on 4096-point c128 lanes
// (profiles/r02c_*): cache-cold 2 GiB arrays 5.56 -> 6.09 TB/s (6.24 with streaming loads), Infinity-Cache-warm
// 4096 x 4096 6.58 -> 6.85 TB/s; runs of 128 KiB .. 8 MiB are within 2 % of each other, one eighth of the array per
// XCD is worse when cold.  NDFFT_XCD_CHUNK_KB overrides the run length (0 = identity map).
int xcd_chunk_for(size_t block_bytes, int64_t nblk) {
    const long kb = NDFFT_DEV_INT("NDFFT_XCD_CHUNK_KB", 512);
    if (kb <= 0 || block_bytes == 0) return 0;
    if (block_bytes >= ((size_t)256 << 10)) return 0;   // one workgroup already streams >= 256 KiB (n = 16384): the map only cost there (0.54 -> 0.50)
    int64_t c = (int64_t)((size_t)kb * 1024 / block_bytes);
    if (c < 1) c = 1;
    if (nblk < 16 * c) return 0;              // fewer than two whole groups: nothing to gain
    return (int)std::min<int64_t>(c, 1 << 20);
}
// Inputs well beyond the 256 MiB Infinity Cache (> 384 MiB) cannot be resident in it: they are loaded with the streaming (nt)
// policy (cache-cold 2 GiB arrays: +3 % alone, +8..12 % together with the XCD map).  Smaller inputs keep the default
// policy: if their producer left them in the Infinity Cache plain loads are up to 15 % faster (4096 x 4096 c128:
// 0.86 vs 0.74 of the roofline), if not they cost 6 % (0.70 vs 0.74) -- the asymmetric bet; a 268 MB input that the bench loop
// re-reads (cfg3-B) still ran 88 us with plain loads vs 99 us streaming, hence the margin above 256 MiB.  NDFFT_STREAM_LOADS=0/1 forces it.
bool stream_loads_for(size_t in_bytes) {
    const int force = sw().stream_loads;             // NDFFT_STREAM_LOADS
    if (force >= 0) return force != 0;
    return in_bytes > ((size_t)384 << 20);
}

// Position-split exchange (pow2_kernel.h: PSPLIT = 2) for the lengths whose lane fills a CU's LDS: two workgroups per CU instead of one.
// NDFFT_PSPLIT=1 / 2 forces it off / on for n = 8192 and 16384 (A/B runs); default: the measured choice below.
// Measured A-B-A-B on 2^24 points (profiles/r05/r05d_psplit_abab.txt): c64 n = 16384 65.5 -> 56.5 us (0.51 -> 0.59 of 8 TB/s), c64 n = 8192 51.8 -> 48.0 us
// (0.65 -> 0.70); c128 n = 16384 unchanged (126 VGPRs x 1024 threads: registers, not LDS, keep it at one workgroup per CU), c128 n = 8192 -2 %.
template <typename T, int N> struct Pow2PSplit { static constexpr int value = (sizeof(T) == 4 && N >= 8192) ? 2 : 1; };
static int psplit_override() { return (int)NDFFT_DEV_INT("NDFFT_PSPLIT", 0); }

template <typename T, int N, int NT, int VEC, int FL, int PS> static int launch_inst_ps(const Pow2Args &a0, hipStream_t s) {
    constexpr int TPL = Pow2Cfg<T, N>::TPL, LPB = lpb_for(TPL);
    // FLAGS bit 4 from n = 4096: late passes whose twiddle table exceeds 32 KiB load W^k, W^2k, W^4k (, W^8k) and build
    // the other powers (2-4 % on the instruction-bound long kernels; two extra roundings on those twiddles)
    using K = Pow2Kernel<T, N, TPL, LPB, Pow2Half<T, N>::value, typename Pow2Cfg<T, N>::RL, FL | (N >= 4096 ? 16 : 0), 1, NT, VEC, PS>;
    NDFFT_ENSURE_LDS_ATTR((k_pow2<K>));
    const int64_t nblk = (a0.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    Pow2Args a = a0;
    a.xcd_chunk = xcd_chunk_for((size_t)LPB * N * sizeof(cpx<T>), nblk);
    // developer knob (A/B only): extra dynamic LDS per workgroup = fewer resident workgroups per CU
    const size_t pad = (size_t)NDFFT_DEV_INT("NDFFT_POW2_LDS_PAD_KB", 0) << 10;
    hipLaunchKernelGGL(k_pow2<K>, dim3((unsigned)nblk), dim3(K::THREADS), std::min(K::LDS_BYTES + pad, (size_t)160 * 1024), s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template <typename T, int N, int NT, int VEC, int FL = 0> static int launch_inst(const Pow2Args &a, hipStream_t s) {
    if constexpr (N >= 8192) {   // both forms exist for the two longest lengths (shorter ones: within +-2 % either way, profiles/r05/r05k_psplit_1024_4096.txt)
        const int ov = psplit_override();
        const int ps = ov == 1 || ov == 2 ? ov : Pow2PSplit<T, N>::value;
        return ps == 2 ? launch_inst_ps<T, N, NT, VEC, FL, 2>(a, s) : launch_inst_ps<T, N, NT, VEC, FL, 1>(a, s);
    } else {
        return launch_inst_ps<T, N, NT, VEC, FL, 1>(a, s);
    }
}
template <typename T, int N, int NT, int VEC, int FL = 0> static int launch_one(const Pow2Args &a, hipStream_t s) {
    static_assert(NT == 1, "callers name the store policy; the load policy is chosen here");
    const bool nt_in = a.stream_in >= 0 ? a.stream_in != 0 : stream_loads_for((size_t)a.nlanes * N * sizeof(cpx<T>));
    if (nt_in) return launch_inst<T, N, 3, VEC, FL>(a, s);
    return launch_inst<T, N, 1, VEC, FL>(a, s);
}

// f32: 16-byte (two-element) accesses where the configuration allows them (an even number of butterflies per
// thread in the first and last pass) and the lanes are 16-byte aligned; n = 64 runs 8 threads x 8 elements with
// 8-byte accesses (measured 92 us vs 106 us for 4 x 16 with 16-byte accesses on 2^25 points)
template <int N> static constexpr bool f32_can_vec2() {
    using RL = typename Pow2Cfg<float, N>::RL;
    constexpr int E = N / Pow2Cfg<float, N>::TPL;
    return (E / RL::at(0)) % 2 == 0 && (E / RL::at(RL::NP - 1)) % 2 == 0;
}
template <int N> static int launch_f32(bool vec_ok, bool fused, const Pow2Args &a, hipStream_t s) {
    if constexpr (f32_can_vec2<N>()) {
        if (vec_ok) return fused ? launch_one<float, N, 1, 2, 8>(a, s) : launch_one<float, N, 1, 2>(a, s);
    }
    return fused ? launch_one<float, N, 1, 1, 8>(a, s) : launch_one<float, N, 1, 1>(a, s);
}

// Cache policy (measured, tools/kbench.hip f64_4096 sweep over 4096..16384 lanes): the output is
// always stored non-temporally (4096x4096 c128: 84 us vs 105 us with plain stores); the input is
// loaded with the default policy at every size -- non-temporal loads cost 9 % while the input still
// fits the 256 MiB Infinity Cache and are within +-2 % of plain loads beyond it.
// The fused four-step twiddle (Pow2Args::twlo) is its own instantiation (FLAGS bit 3): as a
// run-time branch it cost the plain kernel 23-70 VGPRs (n=8192 f64: 122 -> 194, one block per CU).
int launch_pow2(int dtype, int n, const Pow2Args &a, hipStream_t s) {
    // 16-byte accesses for f32 need even pitches and 16-byte aligned bases
    const bool vec_ok = a.pitch_in % 2 == 0 && a.pitch_out % 2 == 0 && ((uintptr_t)a.in % 16) == 0 && ((uintptr_t)a.out % 16) == 0;
    const bool fused = a.twlo != nullptr;
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...)                                                               \
    case N_:                                                                                    \
        if (dtype == NDFFT_F64) return fused ? launch_one<double, N_, 1, 1, 8>(a, s) : launch_one<double, N_, 1, 1>(a, s); \
        return launch_f32<N_>(vec_ok, fused, a, s);
        NDFFT_POW2_CONFIGS_F64(NDFFT_CASE)
#undef NDFFT_CASE
        default: return fail(NDFFT_ERR_UNSUPPORTED, "pow2 kernel: unsupported n");
    }
}

}  // namespace ndfft
