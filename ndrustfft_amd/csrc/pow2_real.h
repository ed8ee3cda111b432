// pow2_real.h -- register-resident kernel for the real-data transforms whose inner complex FFT has a
// power-of-two length F: R2C / C2R (even n = 2F), DCT-II / DCT-III (even n = 2F, Makhoul through a
// real FFT), DCT-IV (even n = 2F) and DCT-I (n = F + 1, even extension of length 2F).
//
// Per lane (contiguous in memory), one pass over HBM:
//   stage   global -> LDS raw lane, coalesced
//   PRE     LDS raw -> registers, already in the first radix pass's input pattern (realops.h)
//   FFT     the same register Stockham passes + padded LDS exchange as the C2C kernel (pow2_kernel.h)
//   Z       registers -> LDS in natural order
//   POST    LDS gather (split / post-twiddle / un-permute, realops.h) -> global, coalesced, non-temporal
// Replaces R2cFftHandler::fft_r2c_lane / ifft_r2c_lane (src/lib.rs:497-523) and
// DctHandler::dct1..4_lane (src/lib.rs:688-734) together with the strategy-(i) row loop.
#pragma once
#include "pow2_kernel.h"
#include "realops.h"

#ifndef NDFFT_COL_REAL_U
#define NDFFT_COL_REAL_U 8
#endif
#ifndef NDFFT_ROW_NT_MIN_F
#define NDFFT_ROW_NT_MIN_F 1024
#endif

namespace ndfft {

template <bool C, typename A, typename B> struct cond_type { typedef A type; };
template <typename A, typename B> struct cond_type<false, A, B> { typedef B type; };

// Column tiles (COL kernels), POST tables through LDS (round 6).  In the column store map (thread = element-row j0 x lane cl, lanes fastest) the split twiddles aux1[k] and
// DCT-II's aux2[k] depend on the element only: all LPB lanes of a row ask for the SAME table entry -- 5 (R2C, DCT-I) or 15 (DCT-II) wave-level gathers per thread with one or
// two distinct addresses each, beside 8-16 loads of data (the finding of the column four-step's stage kernels, profiles/r09/r09d_cs3_ablation.txt).  The workgroup loads the
// F/2 + 1 (+ F + 1) entries once, coalesced, and everybody reads LDS.  Only where the extra bytes do not cost a resident workgroup; host (jit.hip) and device use this one rule.
__host__ __device__ constexpr size_t col_post_table_entries(int F, int op) {
    return (op == G_R2C_EVEN || op == G_DCT1) ? (size_t)(F / 2 + 1) : op == G_DCT2_EVEN ? (size_t)(F / 2 + 1) + (size_t)(F + 1) : 0;
}
__host__ __device__ constexpr size_t col_post_table_bytes(int F, size_t cpx_bytes, int op, size_t lanes_bytes) {
#ifdef NDFFT_NO_COL_POST_TABLE      // (A-B builds: define it for the kernels_*.hip units only -- the hiprtc kernels always carry the table, and jit.hip's host-side LDS size must match THEM)
    return 0;
#else
    return (col_post_table_entries(F, op) && lanes_bytes &&
            (size_t)(160 * 1024) / (lanes_bytes + col_post_table_entries(F, op) * cpx_bytes) == (size_t)(160 * 1024) / lanes_bytes)
               ? col_post_table_entries(F, op) * cpx_bytes : 0;
#endif
}

template <typename T> struct RealArgs {
    const void *in; void *out;
    int64_t nlanes, pitch_in, pitch_out;   // row layout: pitches in elements of the in / out element type
    int32_t n, F, n_in, n_out;
    T scale;
    const cpx<T> *aux1, *aux2, *twp;
    // column layout (COL kernels): lane L = (o, i), o = L / inner, i = L % inner;
    // element j of lane L lives at o*outer_* + i + j*elem_*   (adjacent lanes are adjacent in memory)
    int64_t inner, outer_in, outer_out, elem_in, elem_out;
    int32_t vec_in;   // row layout: every lane base is 16-byte aligned -> stage with 16-byte loads
    int32_t vec_out;  // row layout: output lanes 16-byte aligned -> stage the outputs in LDS and store 16 bytes per lane
    int32_t xcd_remap;   // narrow tiles: 1 = XCD-aware blockIdx -> tile map (0 only for A/B measurements)
    // column four-step (CS kernels, see below): lane L = (o, k1, i) with (o, k1) = divmod(L / inner, cs_k1n)
    const cpx<T> *cs_twlo, *cs_twhi;     // W_N^m = cs_twhi[m >> cs_logB] * cs_twlo[m & (2^cs_logB - 1)], N = cs_n
    int32_t cs_logB, cs_k1n, cs_f1, cs_n;
    int32_t cs_grid3 = 0;                // CS = 1..3: the grid is (inner / LPB, K1, O) -- blockIdx IS (tile of i, k1, o), no index is divided (inner % LPB == 0, O <= 65535)
    int64_t cs_outer_in, cs_outer_out;   // stride of o on the side that is NOT the dense scratch array
    int64_t cs_pitch;                    // row pitch of that side (= inner unless the block is processed in column chunks)
    const cpx<T> *chirp, *bhat;          // Bluestein kernels (blue_kernel.h): e^{-i pi j^2/F}, FFT_M(conj chirp)/M
    int32_t keep_out;                    // COL kernels: 1 = plain (cache-allocating) stores instead of non-temporal ones: the
                                         // output is an intermediate that the next launch re-reads from the Infinity Cache
    int32_t chunk_out = 0;               // row R2C: output lanes are dense (pitch == F + 1) and the array is 16-byte aligned: the workgroup's
                                         // output lanes form ONE contiguous chunk, staged in LDS and stored with coalesced 16-byte accesses
    int32_t xcd_chunk = 0;               // non-XCD kernels: XCD-aware workgroup -> tile map (device_common.h: xcd_block), 0 = identity
    int32_t inner_shift = -1;            // column tiles: log2(inner) when inner is a power of two that is a multiple of the tile width (set by the launcher): lane -> (o, i) by shift / mask
    int32_t stream_in = 0;               // 1 = streaming (nt) loads of the input.  COL kernels: it is read once and must not push the intermediate of a
                                         // two-stage route out of the Infinity Cache; row kernels (16-byte staging loads): the input comes from HBM
    const cpx<T> *twp_rev = nullptr;     // Bluestein / Rader kernels: per-pass twiddles of the SAME radix list taken back to front (second FFT of the convolution)
    int32_t makhoul = 0;                 // ROWOUT kernels with real input (first pass of the REAL four-step, exec.hip: real_fourstep): 1 = the lane is read through
                                         // Makhoul's permutation v[m] = x[2m] (m < n/2), v[m] = x[2(n-1-m)+1] otherwise (DCT-II, n = this->n * inner);
                                         // column C2R kernels (last pass of the inverse real four-step, DCT-III): 1 = the outputs are written through its inverse;
                                         // ROWOUT C2C kernels: 2 = first pass of the fused DCT-IV four-step (the load builds z from the real lane)
    const cpx<T> *fc1 = nullptr, *fc2 = nullptr;   // real four-step, DCT-II / DCT-III: aux2[k1 + N1 r] = fc1[k1] fc2[r] (engine.h: rfs_c1 / rfs_c2); null = read aux2
    int32_t wide = 0;                    // four-step passes of 1024 points: 1 = twp holds the twiddles of the E = 16 recipe 16.8.8 (kernels_fourstep.hip)
    const int32_t *rader_tab = nullptr;  // Rader kernels (rader_kernel.h): g^i mod P (i < P - 1), then g^-i mod P; bhat = FFT_(P-1)(W_P^(g^-q)) / (P - 1),
                                         // twp / twp_rev = per-pass twiddles of FFT_(P-1) with the radix list front to back / back to front
};

// launcher side of RealArgs::inner_shift (see RealPow2Kernel::lane_split): lpb = lanes per column tile
#ifndef NDFFT_NO_INNER_SHIFT
template <typename T> __host__ __device__ inline void real_args_set_inner_shift(RealArgs<T> &b, int lpb) {
    b.inner_shift = -1;
    if (lpb > 0 && (lpb & (lpb - 1)) == 0 && b.inner >= lpb && (b.inner & (b.inner - 1)) == 0) {
        int s = 0;
        while (((int64_t)1 << s) < b.inner) ++s;
        b.inner_shift = s;
    }
}
#else
template <typename T> __host__ __device__ inline void real_args_set_inner_shift(RealArgs<T> &b, int) { b.inner_shift = -1; }
#endif

// RL back to front: the second FFT of a convolution (Bluestein, Rader) runs the passes in reverse order, so that the register pattern it
// starts from (t + q TPL + r M / R_last) IS the pattern the first one ends in -- the pointwise product never goes through LDS
template <int... I> struct RevSeq {};
template <int N, int... I> struct RevMakeSeq : RevMakeSeq<N - 1, N - 1, I...> {};
template <int... I> struct RevMakeSeq<0, I...> { typedef RevSeq<I...> type; };
template <typename RL, typename S> struct RevImpl;
template <typename RL, int... I> struct RevImpl<RL, RevSeq<I...>> { typedef RadixList<RL::at(RL::NP - 1 - I)...> type; };
template <typename RL> using RadixReversed = typename RevImpl<RL, typename RevMakeSeq<RL::NP>::type>::type;

// Tiles whose lines interleave with their MIRROR tile (Makhoul's permutation, DCT-IV's even / odd halves: one tile uses the even elements of a line, the
// tile of the mirrored lanes the odd ones): the two run back to back on one XCD (blocks b and b + 8), so the shared lines are fetched / merged once in
// that XCD's L2.  tpo = tiles per outer index, a multiple of 16; otherwise the identity.  Placement only: any bijection gives the same results.
__device__ __forceinline__ int64_t mirror_pair_tile(int64_t tile, int64_t tpo) {
    if (tpo <= 0 || tpo % 16) return tile;
    const int64_t o_ = tile / tpo, u = tile % tpo, g = u >> 4, w = u & 15, p = 8 * g + (w & 7);
    return o_ * tpo + (w < 8 ? p : tpo - 1 - p);
}

struct ZiNone { static __device__ __forceinline__ int map(int p) { return p; } };
struct ZiPhi { static __device__ __forceinline__ int map(int p) { return p + (p >> 4); } };

// OP: G_R2C_EVEN, G_C2R_EVEN, G_DCT1, G_DCT2_EVEN, G_DCT3_EVEN, G_DCT4_EVEN, and (COL kernels) G_C2C_FWD / G_C2C_INV.
// COL = false: lanes contiguous in memory (strategy i).  COL = true: the transform axis is strided and
// ADJACENT LANES are contiguous (strategy ii on a C-layout array): the workgroup stages an LDS tile of
// LPB lanes x n elements with lanes fastest -- the fused, LDS-padded transpose that replaces the
// reference's per-lane x.to_vec() / y.assign() (src/lib.rs:133-134) -- and stores the same way.
// XCD = true (only with COL): "narrow" column tiles for LONG strided lanes (F >= 2048), LPB = 2 or 4 lanes
// = 8-16 bytes per row of the tile.  One 128-byte line is then shared by 8 neighbouring tiles; the
// blockIdx -> tile map puts those 8 tiles on ONE XCD, back to back in dispatch order, so that the line
// is fetched from HBM once and the other seven reads hit that XCD's L2 (workgroups are dealt round-robin
// over the 8 XCDs: blocks b and b+8 share one -- MI355X_MICROARCH.md).  Placement only affects speed.
//
// CS != 0 (COL, C2C ops only): the twiddled stage of a COLUMN FOUR-STEP.  A long strided lane of length
// N = F1 * F (F = this kernel's length, the inner factor) is transformed in two passes of wide column
// tiles instead of one pass of narrow ones: stage A runs the ordinary column kernel of length F1 over
// a = row / F (rows b, b + F, ...), stage B is this kernel over b for every k1, with the twiddle
// W_N^(b k1) fused into its load and the output row k1 + F1 k2 (no transpose anywhere: both stages see
// >= 128-byte row segments).  The intermediate lives in a dense scratch array [o][k1][b][i].
//   CS = 1  C2C: twiddle on load (conjugated for the inverse), rows k1 + F1 k2 on store
//   CS = 2  second stage of R2C (OP = C2C_FWD): k1 = 0..F1/2 only (stage A was a real FFT); every row of
//           the half spectrum is written once, either as Z[k1 + F1 k2] or as conj at the mirrored row
//           N - (k1 + F1 k2) = (F1 - k1) + F1 (F - 1 - k2)
//   CS = 3  first stage of C2R (OP = C2C_INV): the Hermitian gather of CS = 2 on load (imaginary parts of
//           DC and Nyquist dropped, src/lib.rs:514-518), conj twiddle on store into the scratch array
//   CS = 4  second pass of the ROW four-step (exec.hip: big_fft): lanes (L, k1) with k1 the INNER (contiguous) index;
//           twiddle W_N^(j k1) on load, ordinary column store (row k2, adjacent k1 contiguous = natural order k1 + F1 k2)
//   CS = 5  second pass of the REAL row four-step (exec.hip: real_fourstep), R2C: lanes (L, k1), k1 = 0..N1/2 the inner index, loaded like
//           CS = 4; every element of the half spectrum X[0..n/2] is written once, as Z[k1 + N1 k2] or as its conjugate at the mirrored index
//           (the rule of CS = 2), adjacent k1 contiguous
//   CS = 6  the same for DCT-II: y[k] = Re(X[k] c_k) scale, y[n-k] = -Im(X[k] c_k) scale (c_k = aux2[k], src/lib.rs:700-710 through Makhoul)
// ROWOUT (COL, C2C or R2C): column load, ROW store -- the tile is read with lanes fastest and every lane is written as one
// contiguous run (pitch_out): the transposing first pass of the row four-step (R2C: of the real four-step, n_out = F + 1 per lane).
// FFLAGS: flags of the inner Pow2Kernel (32 = the LATENCY form of its passes, for calls of few tiles).
template <typename T, int F, int TPL, int LPB, typename RL, int OP, bool COL = false, bool XCD = false, int CS = 0, bool ROWOUT = false, int FFLAGS = 0> struct RealPow2Kernel {
    static_assert(CS == 0 || (COL && !XCD && (OP == G_C2C_FWD || OP == G_C2C_INV)), "CS kernels are column C2C kernels");
    static_assert(!ROWOUT || (COL && !XCD && CS == 0 && (OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_EVEN)), "ROWOUT is a column-load C2C / R2C kernel");
    static_assert(CS < 5 || OP == G_C2C_FWD, "CS = 5 / 6 are forward kernels");
    // strided staging loop with U independent global loads in flight per thread before the first LDS store
    // (a plain `for (j) dst[j] = in[j * stride]` leaves one or two loads outstanding: latency-bound)
    // (round 5: loading a one- or two-element remainder -- lanes of 2^k + 1 points, the DCT-I bench sizes -- BEFORE the full batches, so that it costs no round trip
    //  of its own, gained 2-7 % on the reference's n x n DCT-I bench shapes and cost the register-capped f32 kernels 6-19 %: 32 x 2^20 c64 281 -> 334 us, ndfft_r2c f32
    //  rows n = 100 .. 512 +6-17 % -- profiles/r08/r08n_configs_compare.txt; not kept)
    // The remainder after the batches of U goes in batches of U / 2, U / 4, ... (round 6).  Before, it ran one load at a time -- `global_load; s_waitcnt vmcnt(0)` in a
    // loop, found in the ISA --, and f32 rows have (2 F / 4) / TPL = E / 2 = 4 sixteen-byte loads per thread, FEWER than one batch of 8: four dependent round trips to
    // memory per lane; every real f32 row kernel sat at 0.43-0.59 of the roofline beside 0.70 for the C2C ones (ndfft_r2c f32 rows n = 4096: 35.6 -> 31.6 us,
    // 8192 x 8192: 125.4 -> 117.0 us, profiles/r09/r09y_row_tail_staging_abab.txt).  The same loop serves the other shared-memory kernels (plain / Rader / Bluestein).
    // Not in the column tiles of power-of-two F (their trip counts are whole batches; the dead remainder code cost the register-capped f32 first passes of the
    // row four-step 12 more bytes of scratch and the f64 ones 14 VGPRs).
    static constexpr bool STAGE_TAIL = !COL || (F & (F - 1)) != 0;
    template <int STEP, int U = 8, typename LD, typename ST> static __device__ __forceinline__ void stage_loop(int j0, int n, LD ld, ST st) {
        int j = j0;
        for (; j + (U - 1) * STEP < n; j += U * STEP) {
            decltype(ld(0)) tmp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tmp[u] = ld(j + u * STEP);
#pragma unroll
            for (int u = 0; u < U; ++u) st(j + u * STEP, tmp[u]);
        }
        if constexpr (STAGE_TAIL && U >= 4) stage_loop<STEP, U / 2>(j, n, ld, st);
        else for (; j < n; j += STEP) st(j, ld(j));
    }
    static __device__ __forceinline__ cpx<T> cs_tw(const RealArgs<T> &a, int m) {
        return cmul(a.cs_twhi[m >> a.cs_logB], a.cs_twlo[m & ((1 << a.cs_logB) - 1)]);
    }
    // the same with a compile-time trip count (the CS stage kernels: n_in == F, a multiple of STEP): no loop, no remainder code
    template <int STEP, int N, typename LD, typename ST> static __device__ __forceinline__ void stage_fixed(int j0, LD ld, ST st) {
        static_assert(N % STEP == 0, "stage_fixed: whole rounds only");
        constexpr int NIT = N / STEP, U = NIT < 8 ? NIT : 8;
        static_assert(NIT % U == 0, "stage_fixed: whole batches only");
#pragma unroll
        for (int b = 0; b < NIT; b += U) {
            decltype(ld(0)) tmp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tmp[u] = ld(j0 + (b + u) * STEP);
#pragma unroll
            for (int u = 0; u < U; ++u) st(j0 + (b + u) * STEP, tmp[u]);
        }
    }
    // CS = 1..3: which (o, k1, i) a thread's column is.  cs_grid3: read off blockIdx (round 6: the 64-bit divisions of the flat form and the
    // table loads inside the store loop made the C2R first stage VALU- / latency-bound: 64.5 us per chunk against 46 us for a copy of its shape)
    struct CsPos { int64_t o, ii; int k1; bool live; };
    static __device__ __forceinline__ CsPos cs_pos(const RealArgs<T> &a, int cl) {
        CsPos p;
        if (a.cs_grid3) { p.ii = (int64_t)blockIdx.x * LPB + cl; p.k1 = (int)blockIdx.y; p.o = (int64_t)blockIdx.z; p.live = true; }
        else {
            const int64_t L = (int64_t)blockIdx.x * LPB + cl;
            p.live = L < a.nlanes;
            const int64_t ok = (p.live ? L : 0) / a.inner;
            p.ii = (p.live ? L : 0) % a.inner; p.k1 = (int)(ok % a.cs_k1n); p.o = ok / a.cs_k1n;
        }
        return p;
    }
    static constexpr int THREADS = TPL * LPB;
    // lane L = lane0 + cl of a column tile -> (outer index o, inner index i).  The general form divides a 64-bit lane index per thread (about 60 VALU instructions of
    // the ~650 a column-tile wave issues); with inner a power of two that whole tiles divide, o is the TILE's (a scalar shift) and i a mask and an add.
    static __device__ __forceinline__ void lane_split(const RealArgs<T> &a, int64_t lane0, int cl, int64_t &o, int64_t &i) {
        if (a.inner_shift >= 0) { o = lane0 >> a.inner_shift; i = (lane0 & (((int64_t)1 << a.inner_shift) - 1)) + cl; }
        else { const int64_t L = lane0 + cl; o = L / a.inner; i = L % a.inner; }
    }
    // CS = 1..3, the twiddles W_N^(j k1), j < F, of one tile.  With the 3-D grid k1 is the WORKGROUP's: threads 0 .. F-1 load the two table entries of one
    // twiddle each (in front of the staging loads), multiply and leave the F products in LDS behind the lane regions; everybody reads them from there (CS = 1 / 2
    // in PRE, CS = 3 in the store loop).  Before round 6 every thread loaded its own 2 x 8 entries -- 32 lanes asking for the same address, 64 wave-level loads per
    // tile against 32 for the data: the C2R first stage measured 61.5 us per chunk with them and 51.8 without (profiles/r09/r09d_cs3_ablation.txt).
    // Flat grid (rows of i that are not whole tiles: k1 differs between the lanes of a tile): per-thread table loads as before.
    template <int NQ> static __device__ __forceinline__ void cs3_twiddles(const RealArgs<T> &a, int k1, int j0, cpx<T> (&hi)[NQ], cpx<T> (&lo)[NQ]) {
        const int mask = (1 << a.cs_logB) - 1;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int m = (j0 + i * (THREADS / LPB)) * k1;
            hi[i] = a.cs_twhi[m >> a.cs_logB]; lo[i] = a.cs_twlo[m & mask];
        }
    }
    static constexpr bool CSK = CS >= 1 && CS <= 3;
    // complex elements per lane: padded Z, or F+1 raw complex.  COL: odd, so adjacent lanes spread over the
    // banks; row: even, so every lane base stays 16-byte aligned for the vector staging stores.
    static constexpr int LANE_LDS = COL ? ((F + (F >> 4) + 2) | 1) : ((F + (F >> 4) + 3) & ~1);
    static constexpr size_t LANES_LDS_BYTES = (size_t)LPB * LANE_LDS * 2 * sizeof(T);
    // CS stage kernels: + the tile's F twiddles; CS >= 4 (second pass of the row four-steps): + its E x LPB step twiddles (see the PRE fold)
    // CS >= 4: elements per thread of the FIRST pass, whole or partial rounds: ceil(F / (R0 TPL)) butterflies of radix R0 (= F / TPL with whole rounds)
    static constexpr int CS4_E0 = ((F / RL::at(0) + TPL - 1) / TPL) * RL::at(0);
    static constexpr size_t TBL_BYTES = (COL && !XCD && CS == 0 && !ROWOUT) ? col_post_table_bytes(F, 2 * sizeof(T), OP, LANES_LDS_BYTES) : 0;   // col_post_table_bytes above
    static constexpr size_t LDS_BYTES = LANES_LDS_BYTES + (CSK ? (size_t)F * 2 * sizeof(T) : CS >= 4 ? (size_t)CS4_E0 * LPB * 2 * sizeof(T) : TBL_BYTES);
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN || OP == G_C2C_FWD || OP == G_C2C_INV;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN || OP == G_C2C_FWD || OP == G_C2C_INV;
    using FFT = Pow2Kernel<T, F, TPL, LPB, false, RL, FFLAGS, 1, 0>;
    static constexpr int E = FFT::E;       // = F / TPL unless some pass has a partial last round (pow2_kernel.h)
    // row layout R2C / C2R: the PRE fold reads unit-stride complex elements (ascending, and for C2R also
    // descending), so it loads global memory directly and the LDS staging pass and its barrier are skipped
    // (f64 only: 16-byte elements; for f32 the 16-byte vector staging loads measure faster than 8-byte direct ones)
    // (round 6, after the staging loop's remainder was fixed -- stage_loop below: direct 8-byte loads for f32 R2C rows measured +2 ... +4 % against the staged form, profiles/r09/r09y_direct32_abab.txt)
    static constexpr bool DIRECT_IN = !COL && sizeof(T) == 8 && (OP == G_R2C_EVEN || OP == G_C2R_EVEN);
    // ops whose POST is the real-FFT split: outputs k and F-k share one pair of LDS reads and one twiddle
    static constexpr bool PAIR = OP == G_R2C_EVEN || OP == G_DCT1 || OP == G_DCT2_EVEN;
    // R2C rows are F + 1 complex long: short lanes start and end mid-line, and non-temporal stores of partial lines
    // cost more than they save (262144x128 f32: 132 us non-temporal, 50 us plain; from F = 1024 up non-temporal wins) -- let L2 merge them
    static constexpr bool ROW_NT = F >= NDFFT_ROW_NT_MIN_F;

    // every output derived from the spectrum pair (k, F-k), in four fixed slots (q < 0: slot unused)
    template <typename OT> struct PairOut { OT v[4]; int q[4]; };
    template <typename OT> static __device__ __forceinline__ PairOut<OT> post_pair(const RealArgs<T> &a, const cpx<T> *res, int k) { return post_pair<OT>(a, res, k, a.aux1[k]); }
    template <typename OT> static __device__ __forceinline__ PairOut<OT> post_pair(const RealArgs<T> &a, const cpx<T> *res, int k, cpx<T> wk) { return post_pair<OT>(a, res, k, wk, a.aux2); }
    template <typename OT> static __device__ __forceinline__ PairOut<OT> post_pair(const RealArgs<T> &a, const cpx<T> *res, int k, cpx<T> wk, const cpx<T> *aux2) {
        PairOut<OT> r;
        cpx<T> xk, xf;
        r2c_split_pair<T, ZiPhi>(res, k, F, wk, xk, xf);
        const int kf = F - k;
        r.q[0] = k; r.q[1] = -1; r.q[2] = kf != k ? kf : -1; r.q[3] = -1;
        if constexpr (OP == G_R2C_EVEN) {
            r.v[0] = xk; r.v[2] = xf; r.v[1] = xk; r.v[3] = xk;
        } else if constexpr (OP == G_DCT1) {
            r.v[0] = (T)0.5 * xk.x; r.v[2] = (T)0.5 * xf.x; r.v[1] = 0; r.v[3] = 0;
        } else {   // G_DCT2_EVEN: y[k] = Re(X[k] c_k), y[n-k] = -Im(X[k] c_k)
            const int n = 2 * F;
            const cpx<T> tk = cmul(xk, aux2[k]), tf = cmul(xf, aux2[kf]);
            r.v[0] = tk.x; r.v[1] = -tk.y; r.v[2] = tf.x; r.v[3] = -tf.y;
            r.q[1] = k > 0 ? n - k : -1;
            r.q[3] = (kf != k && kf < F) ? n - kf : -1;
        }
        return r;
    }

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        int64_t tile = blockIdx.x;
        // (not in the CS / ROWOUT stage kernels: they never use the map, and the extra scalar code changed the register
        //  allocation of the CS = 3 kernel from 85 to 76 VGPRs -- fewer staging loads in flight, 67 -> 76 us per launch)
        if constexpr (!XCD && CS == 0 && !ROWOUT) tile = xcd_block(blockIdx.x, gridDim.x, a.xcd_chunk);
        // passes of the row four-step whose tile rows are HALF a line (F = 1024 f32: 8 lanes x 8 bytes): runs of consecutive tiles per XCD, so that the
        // two tiles sharing every line meet in one L2 (only these instantiations: the map costs registers in the others, see above)
        if constexpr ((CS == 4 || (ROWOUT && OP != G_R2C_EVEN)) && LPB * 2 * sizeof(T) < 128) tile = xcd_block(blockIdx.x, gridDim.x, a.xcd_chunk);
        if constexpr (COL && !XCD && CS == 0 && (ROWOUT ? (OP == G_R2C_EVEN || OP == G_C2C_FWD) : OP == G_C2R_EVEN)) {
            // Makhoul's permutation / DCT-IV's fold: see mirror_pair_tile (inner / LPB tiles per lane)
            if (a.makhoul && a.inner % LPB == 0) tile = mirror_pair_tile(tile, a.inner / LPB);
        }
        const int64_t lane0 = tile * LPB;
        const int64_t lane = lane0 + ll;
        const bool live = lane < a.nlanes;
        char *lds = smem + (size_t)ll * LANE_LDS * 2 * sizeof(T);
        // ---- stage the raw lane(s) ----
        constexpr int CSNQ = CSK ? F / (THREADS / LPB) : 1;   // outputs per thread of a CS stage kernel
        cpx<T> *cs_twl = (cpx<T> *)(smem + LANES_LDS_BYTES);   // CS stage kernels, 3-D grid: W_N^(j k1), j < F (see cs3_twiddles); CS >= 4: the step twiddles [e][lane]
        constexpr int CS4_NS = CS >= 4 ? (CS4_E0 + TPL - 1) / TPL : 1;
        cpx<T> cs4_bhi = mk<T>((T)1, (T)0), cs4_blo = cs4_bhi, cs4_shi[CS4_NS], cs4_slo[CS4_NS];
        constexpr int TBL_ENTRIES = (int)(TBL_BYTES / (2 * sizeof(T))), TBL_PER_THREAD = TBL_BYTES ? (TBL_ENTRIES + THREADS - 1) / THREADS : 1;
        cpx<T> tbl_v[TBL_PER_THREAD];
        if constexpr (CSK) {
            static_assert(F % (THREADS / LPB) == 0 && THREADS >= F, "CS stage kernels: whole staging rounds, one thread per twiddle");
            const int cl = threadIdx.x % LPB, j0 = threadIdx.x / LPB;
            const CsPos p = cs_pos(a, cl);
            constexpr int STEP = THREADS / LPB;
            char *dst = smem + (size_t)cl * LANE_LDS * 2 * sizeof(T);
            const bool g3 = a.cs_grid3 != 0;
            cpx<T> tw_hi = mk<T>((T)1, (T)0), tw_lo = tw_hi;
            if (g3 && threadIdx.x < F) {
                const int m = (int)threadIdx.x * p.k1;
                tw_hi = a.cs_twhi[m >> a.cs_logB]; tw_lo = a.cs_twlo[m & ((1 << a.cs_logB) - 1)];
            }
            if (p.live) {
                if constexpr (CS == 3) {
                    const cpx<T> *in = (const cpx<T> *)a.in + p.o * a.cs_outer_in + p.ii;
                    const int k1 = p.k1, f1 = a.cs_f1, nn = a.cs_n;
                    auto put = [&](int j, cpx<T> v) {
                        const int row = k1 + f1 * j;
                        if (2 * row > nn) v.y = -v.y;
                        if (row == 0 || 2 * row == nn) v.y = 0;
                        ((cpx<T> *)dst)[j] = v;
                    };
                    // (stream_in: the caller's array is read once and must not push the intermediate this stage writes out of the Infinity Cache)
                    if (a.stream_in) stage_fixed<STEP, F>(j0, [&](int j) { const int row = k1 + f1 * j; return gload<T, true>(in + (int64_t)(2 * row > nn ? nn - row : row) * a.elem_in); }, put);
                    else stage_fixed<STEP, F>(j0, [&](int j) { const int row = k1 + f1 * j; return in[(int64_t)(2 * row > nn ? nn - row : row) * a.elem_in]; }, put);
                } else {
                    const cpx<T> *in = (const cpx<T> *)a.in + (p.o * a.cs_k1n + p.k1) * a.outer_in + p.ii;
                    if (g3) {   // plain copy: the twiddle is applied in PRE, from LDS
                        stage_fixed<STEP, F>(j0, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, cpx<T> v) { ((cpx<T> *)dst)[j] = v; });
                    } else {
                        const int k1 = p.k1;
                        struct VW { cpx<T> v, w; };
                        stage_fixed<STEP, F>(j0,
                            [&](int j) { VW r; r.v = in[(int64_t)j * a.elem_in]; r.w = cs_tw(a, j * k1); return r; },
                            [&](int j, VW r) { if constexpr (OP == G_C2C_INV) r.w = cconj(r.w); ((cpx<T> *)dst)[j] = cmul(r.v, r.w); });
                    }
                }
            }
            if (g3 && threadIdx.x < F) cs_twl[threadIdx.x] = cmul(tw_hi, tw_lo);
        } else if constexpr (COL) {
            if constexpr (CS >= 4) {
                // The four-step twiddle W_N^(i k1) of element i = t + d (d = q TPL + r F/R0: E values) of this thread's lane (k1 = its inner index) is W^(t k1) x W^(d k1):
                // one base per thread, E x LPB steps per tile (threads t < E load one each and leave the product in LDS); the table entries are loaded here, in front of
                // the staging loads.  Before round 6 the staging loop gathered two table entries per ELEMENT (16 scattered loads per thread beside 8 loads of data).
                constexpr int R0 = RL::at(0), NB0 = F / R0;
                const int mask = (1 << a.cs_logB) - 1;
                int64_t k1o, k1i; lane_split(a, lane0, ll, k1o, k1i);
                const int k1 = lane < a.nlanes ? (int)k1i : 0;
                const int mb = t * k1;
                cs4_bhi = a.cs_twhi[mb >> a.cs_logB]; cs4_blo = a.cs_twlo[mb & mask];
                // (thread row t owns the steps e = t, t + TPL, ...: one where TPL >= E -- every ahead-of-time recipe --, more for the hiprtc recipes with E > TPL)
#pragma unroll
                for (int i = 0; i < CS4_NS; ++i) {
                    const int e = t + i * TPL;
                    if (e < CS4_E0) { const int m = ((e / R0) * TPL + (e % R0) * NB0) * k1; cs4_shi[i] = a.cs_twhi[m >> a.cs_logB]; cs4_slo[i] = a.cs_twlo[m & mask]; }
                }
            }
            if constexpr (TBL_BYTES != 0) {   // the POST tables of this tile, issued in front of the staging loads (see col_post_table_bytes)
                constexpr int N1 = F / 2 + 1;
#pragma unroll
                for (int i = 0; i < TBL_PER_THREAD; ++i) {
                    const int idx = (int)threadIdx.x + i * THREADS;
                    if (idx < N1) tbl_v[i] = a.aux1[idx];
                    else if (idx < TBL_ENTRIES) tbl_v[i] = a.aux2[idx - N1];
                }
            }
            // thread -> (lane cl = tid % LPB fastest, element j = tid / LPB)
            const int cl = threadIdx.x % LPB, j0 = threadIdx.x / LPB;
            const int64_t L = lane0 + cl;
            // (CS = 5 / 6: the inner index k1 runs over a pitch padded to whole 128-byte lines; lanes k1 > N1/2 are padding)
            int64_t Lo, Li; lane_split(a, lane0, cl, Lo, Li);
            if (L < a.nlanes && (CS < 5 || 2 * Li <= a.cs_f1)) {
                const int64_t base = Lo * a.outer_in + Li;
                char *dst = smem + (size_t)cl * LANE_LDS * 2 * sizeof(T);
                constexpr int STEP = THREADS / LPB;
                constexpr int UR = NDFFT_COL_REAL_U;   // loads in flight per thread of a REAL column tile (2 F / TPL = 16 elements per thread with E = 8)
                if constexpr (CS >= 4) {   // plain copy: the twiddle W_N^(j k1) is applied in PRE (base x step, see there)
                    const cpx<T> *in = (const cpx<T> *)a.in + base;
                    stage_loop<STEP>(j0, a.n_in, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, cpx<T> v) { ((cpx<T> *)dst)[j] = v; });
                } else if constexpr (IN_CPLX) {
                    const cpx<T> *in = (const cpx<T> *)a.in + base;
                    bool folded = false;
                    if constexpr (ROWOUT && OP == G_C2C_FWD) {
                        if (a.makhoul == 2) {   // first pass of the fused DCT-IV four-step: element j of lane (o, n2) is z[jj] = (x[2 jj] + i x[n-1-2 jj]) s e^(-i pi (4 jj + 1) / 4n),
                                                // jj = j inner + n2, built from the REAL lane o of n = 2 this->n inner points (realops.h: G_DCT4_EVEN)
                            const T *lane_o = (const T *)a.in + Lo * a.outer_in;
                            const int64_t m0 = Li, nn = 2 * (int64_t)a.n * a.inner;
                            struct XW { T x0, x1; cpx<T> w; };
                            if (a.stream_in) stage_loop<STEP>(j0, a.n_in,
                                [&](int j) { const int64_t jj = (int64_t)j * a.inner + m0; XW r; r.x0 = __builtin_nontemporal_load(lane_o + 2 * jj); r.x1 = __builtin_nontemporal_load(lane_o + (nn - 1 - 2 * jj)); r.w = a.aux1[jj]; return r; },
                                [&](int j, XW r) { ((cpx<T> *)dst)[j] = cmul(mk<T>(r.x0 * a.scale, r.x1 * a.scale), r.w); });
                            else stage_loop<STEP>(j0, a.n_in,
                                [&](int j) { const int64_t jj = (int64_t)j * a.inner + m0; XW r; r.x0 = lane_o[2 * jj]; r.x1 = lane_o[nn - 1 - 2 * jj]; r.w = a.aux1[jj]; return r; },
                                [&](int j, XW r) { ((cpx<T> *)dst)[j] = cmul(mk<T>(r.x0 * a.scale, r.x1 * a.scale), r.w); });
                            folded = true;
                        }
                    }
                    if (folded) {}
                    else if (a.stream_in) stage_loop<STEP>(j0, a.n_in, [&](int j) { return gload<T, true>(in + (int64_t)j * a.elem_in); }, [&](int j, cpx<T> v) { ((cpx<T> *)dst)[j] = v; });
                    else stage_loop<STEP>(j0, a.n_in, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, cpx<T> v) { ((cpx<T> *)dst)[j] = v; });
                } else {
                    const T *in = (const T *)a.in + base;
                    bool gathered = false;
                    if constexpr (ROWOUT) {
                        if (a.makhoul) {   // element j of lane (o, n2) is v[m], m = j inner + n2, of the n = this->n * inner long lane o
                            const T *lane_o = (const T *)a.in + Lo * a.outer_in;
                            const int64_t m0 = Li, nn = (int64_t)a.n * a.inner;
                            // makhoul = 1: Makhoul's permutation (DCT-II); makhoul = 3 (round 5): the EVEN EXTENSION e[m] = x[m] (m <= nn/2), x[nn - m] otherwise, of a DCT-I lane of
                            // nn/2 + 1 points (exec.hip: real_fourstep with dct1 = true)
                            const bool ext = a.makhoul == 3;
                            auto src = [&](int j) -> int64_t { const int64_t m = (int64_t)j * a.inner + m0; return ext ? (2 * m <= nn ? m : nn - m) : (2 * j < a.n ? 2 * m : 2 * (nn - 1 - m) + 1); };
                            if (a.stream_in) stage_loop<STEP, UR>(j0, a.n_in, [&](int j) { return __builtin_nontemporal_load(lane_o + src(j)); }, [&](int j, T v) { ((T *)dst)[j] = v; });
                            else stage_loop<STEP, UR>(j0, a.n_in, [&](int j) { return lane_o[src(j)]; }, [&](int j, T v) { ((T *)dst)[j] = v; });
                            gathered = true;
                        }
                    }
                    if (gathered) {}
                    else if (a.stream_in) stage_loop<STEP, UR>(j0, a.n_in, [&](int j) { return __builtin_nontemporal_load(in + (int64_t)j * a.elem_in); }, [&](int j, T v) { ((T *)dst)[j] = v; });
                    else stage_loop<STEP, UR>(j0, a.n_in, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, T v) { ((T *)dst)[j] = v; });
                }
            }
        } else if constexpr (!DIRECT_IN) {
            const int64_t lsafe = live ? lane : 0;
            if constexpr (IN_CPLX) {
                const cpx<T> *in = (const cpx<T> *)a.in + lsafe * a.pitch_in;
                cpx<T> *raw = (cpx<T> *)lds;
                if (sizeof(T) == 4 && a.vec_in) {   // two c64 per 16-byte load
                    const int nv = a.n_in >> 1;
                    if (a.stream_in) stage_loop<TPL>(t, nv, [&](int j) { return __builtin_nontemporal_load((const vec4f *)in + j); }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    else stage_loop<TPL>(t, nv, [&](int j) { return ((const vec4f *)in)[j]; }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    for (int j = 2 * nv + t; j < a.n_in; j += TPL) raw[j] = in[j];
                } else {
                    stage_loop<TPL>(t, a.n_in, [&](int j) { return in[j]; }, [&](int j, cpx<T> v) { raw[j] = v; });
                }
            } else {
                const T *in = (const T *)a.in + lsafe * a.pitch_in;
                T *raw = (T *)lds;
                if (a.vec_in) {                      // 2 doubles / 4 floats per 16-byte load
                    constexpr int W = 16 / sizeof(T);
                    const int nv = a.n_in / W;
                    // (stream_in: the input comes from HBM, not from the Infinity Cache -- exec.hip: MallModel)
                    if (a.stream_in) stage_loop<TPL>(t, nv, [&](int j) { return __builtin_nontemporal_load((const vec4f *)in + j); }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    else stage_loop<TPL>(t, nv, [&](int j) { return ((const vec4f *)in)[j]; }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    for (int j = W * nv + t; j < a.n_in; j += TPL) raw[j] = in[j];
                } else {
                    stage_loop<TPL>(t, a.n_in, [&](int j) { return in[j]; }, [&](int j, T v) { raw[j] = v; });
                }
            }
        }
        if constexpr (CS >= 4) {
#pragma unroll
            for (int i = 0; i < CS4_NS; ++i) { const int e = t + i * TPL; if (e < CS4_E0) cs_twl[e * LPB + ll] = cmul(cs4_shi[i], cs4_slo[i]); }
        }
        if constexpr (TBL_BYTES != 0) {
#pragma unroll
            for (int i = 0; i < TBL_PER_THREAD; ++i) { const int idx = (int)threadIdx.x + i * THREADS; if (idx < TBL_ENTRIES) cs_twl[idx] = tbl_v[i]; }
        }
        if constexpr (!DIRECT_IN) __syncthreads();
        if constexpr (OP == G_DCT3_EVEN) {
            // DCT-III: V[k] = 0.5 s (x[k] - i x[n-k]) e^{+i pi k/(2n)}, k = 0..F (x[n] := 0), computed ONCE per k into the lane
            // region (over the raw lane); the PRE fold below then needs V[i] and V[F-i] only.  pre_elem would build both
            // from four raw reads, two table loads and two complex multiplies per element -- each V[k] twice.
            // (round 6: the fold reading x[k] and x[n-k] from GLOBAL memory itself -- no staged raw lane, one LDS round trip and two barriers fewer, but 18 eight-byte loads per
            //  thread instead of 8 sixteen-byte ones -- measured cfg4 nddct3 97.5 -> 102.2 us: not kept)
            constexpr int NK = (F + 1 + TPL - 1) / TPL;
            cpx<T> vk[NK];
            const T *xr = (const T *)lds;
            const int n = 2 * F;
            const T hs = (T)0.5 * a.scale;
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = t + i * TPL;
                if (k <= F) vk[i] = cmul(mk<T>(xr[k] * hs, k ? -xr[n - k] * hs : (T)0), cconj(a.aux2[k]));
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int k = t + i * TPL;
                if (k <= F) ((cpx<T> *)lds)[k] = vk[i];
            }
            __syncthreads();
        }
        // ---- PRE into the first pass's register pattern ----
        cpx<T> v[E];
        {
            constexpr int R0 = RL::at(0), NB0 = F / R0, NBF0 = FFT::slots(0);
            const void *raw = (const void *)lds;
            if constexpr (DIRECT_IN) {
                const int64_t lsafe = live ? lane : 0;
                if constexpr (IN_CPLX) raw = (const void *)((const cpx<T> *)a.in + lsafe * a.pitch_in);
                else raw = (const void *)((const T *)a.in + lsafe * a.pitch_in);
            }
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
                if (FFT::full(0) || t + q * TPL < NB0) {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const int i = t + q * TPL + r * NB0;
                        if constexpr (OP == G_R2C_EVEN || OP == G_C2C_FWD) v[q * R0 + r] = ((const cpx<T> *)raw)[i];   // z[i] = (x[2i], x[2i+1])
                        else if constexpr (OP == G_C2C_INV) v[q * R0 + r] = cconj(((const cpx<T> *)raw)[i]);
                        else if constexpr (OP == G_DCT3_EVEN) v[q * R0 + r] = herm_fold<T>(((const cpx<T> *)raw)[i], cconj(((const cpx<T> *)raw)[F - i]), a.aux1[i]);
                        else v[q * R0 + r] = pre_elem<T, OP, ZiNone>(a, raw, i);
                    }
                }
        }
        if constexpr (CS >= 4) {   // (INV: the lane was conjugated above -- the same table serves both directions)
            const cpx<T> base = cmul(cs4_bhi, cs4_blo);
#pragma unroll
            for (int e = 0; e < CS4_E0; ++e)      // (a partial last round: the thread's element does not exist -- the PRE loop above left it alone, so do we)
                if (FFT::full(0) || t + (e / RL::at(0)) * TPL < F / RL::at(0)) v[e] = cmul(v[e], cmul(base, cs_twl[e * LPB + ll]));
        }
        if constexpr (CS == 1 || CS == 2) {
            if (a.cs_grid3) {   // the stage twiddle W_N^(j k1) (INV: the lane was conjugated above, conj(x conj(w)) = conj(x) w), from the tile's LDS table
                constexpr int R0 = RL::at(0), NB0 = F / R0, NBF0 = FFT::slots(0);
#pragma unroll
                for (int q = 0; q < NBF0; ++q)
                    if (FFT::full(0) || t + q * TPL < NB0) {
#pragma unroll
                        for (int r = 0; r < R0; ++r) v[q * R0 + r] = cmul(v[q * R0 + r], cs_twl[t + q * TPL + r * NB0]);
                    }
            }
        }
        // (the first exchange inside passes() starts with a barrier, so the raw lane is dead by then)
        FFT::template passes<0>(v, a.twp, lds, t);
        // ---- Z in natural order ----
        __syncthreads();
        {
            constexpr int RL_ = RL::at(RL::NP - 1), NBL = F / RL_, NBFL = FFT::slots(RL::NP - 1);
            cpx<T> *z = (cpx<T> *)lds;
#pragma unroll
            for (int q = 0; q < NBFL; ++q)
                if (FFT::full(RL::NP - 1) || t + q * TPL < NBL) {
#pragma unroll
                    for (int r = 0; r < RL_; ++r) z[ZiPhi::map(t + q * TPL + r * NBL)] = v[q * RL_ + r];
                }
        }
        __syncthreads();
        // ---- POST gather + store ----
        if constexpr (COL && !ROWOUT) {
            const int cl = threadIdx.x % LPB, j0 = threadIdx.x / LPB;
            const int64_t L = lane0 + cl;
            CsPos csp; csp.live = true;
            int64_t base = 0;
            int64_t Lo = 0, Li = 0;
            if constexpr (CS >= 1 && CS <= 3) { csp = cs_pos(a, cl); if (!csp.live) return; }
            else { if (L >= a.nlanes) return; lane_split(a, lane0, cl, Lo, Li); base = Lo * a.outer_out + Li; }
            const cpx<T> *res = (const cpx<T> *)(smem + (size_t)cl * LANE_LDS * 2 * sizeof(T));
            if constexpr (PAIR) {
                // the split twiddles of this thread's pairs, loaded up front: inside the loops below every iteration would wait for its own table load behind
                // the previous iteration's stores (F / TPL / 2 + 1 = 5 dependent round trips with E = 8)
                constexpr int NIT = (F / 2 + TPL) / TPL;
                cpx<T> w1[NIT];
                const cpx<T> *aux1p = TBL_BYTES != 0 ? cs_twl : a.aux1, *aux2p = TBL_BYTES != 0 ? cs_twl + (F / 2 + 1) : a.aux2;
#pragma unroll
                for (int i = 0; i < NIT; ++i) { const int k = j0 + i * TPL; if (k <= F / 2) w1[i] = aux1p[k]; }
                if constexpr (OUT_CPLX) {
                    cpx<T> *out = (cpx<T> *)a.out + base;
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        const int k = j0 + i * TPL;
                        if (k > F / 2) break;
                        const PairOut<cpx<T>> r = post_pair<cpx<T>>(a, res, k, w1[i], aux2p);
#pragma unroll
                        for (int z = 0; z < 4; ++z)
                            if (r.q[z] >= 0) {
                                if (XCD || a.keep_out) out[(int64_t)r.q[z] * a.elem_out] = r.v[z]; else gstore<T, true>(out + (int64_t)r.q[z] * a.elem_out, r.v[z]);
                            }
                    }
                } else {
                    T *out = (T *)a.out + base;
#pragma unroll
                    for (int i = 0; i < NIT; ++i) {
                        const int k = j0 + i * TPL;
                        if (k > F / 2) break;
                        const PairOut<T> r = post_pair<T>(a, res, k, w1[i], aux2p);
#pragma unroll
                        for (int z = 0; z < 4; ++z)
                            if (r.q[z] >= 0) {
                                if constexpr (XCD) out[(int64_t)r.q[z] * a.elem_out] = r.v[z];
                                else __builtin_nontemporal_store(r.v[z], out + (int64_t)r.q[z] * a.elem_out);
                            }
                    }
                }
            } else if constexpr (CS == 1 || CS == 2) {
                const int k1 = csp.k1;
                cpx<T> *out = (cpx<T> *)a.out + csp.o * a.cs_outer_out + csp.ii;   // row 0 of (o, i)
#pragma unroll
                for (int qi = 0; qi < CSNQ; ++qi) {
                    const int q = j0 + qi * (THREADS / LPB);
                    cpx<T> val = post_cplx<T, OP, ZiPhi>(a, res, q);
                    int kk = k1, r2 = q;
                    bool skip = false;
                    if constexpr (CS == 2) {
                        if (k1 == 0) skip = q > F / 2;
                        else if (2 * k1 == a.cs_f1) skip = q >= F / 2;
                        else if (q >= F / 2) { kk = a.cs_f1 - k1; r2 = F - 1 - q; val.y = -val.y; }
                    }
                    if (!skip) gstore<T, true>(out + (int64_t)kk * a.cs_pitch + (int64_t)r2 * a.elem_out, val);
                }
            } else if constexpr (CS == 5 || CS == 6) {
                const int k1 = (int)Li;
                if (2 * k1 > a.cs_f1) return;
                const int64_t ob = Lo * a.outer_out;   // output lane o
                for (int q = j0; q < F; q += THREADS / LPB) {
                    cpx<T> val = post_cplx<T, OP, ZiPhi>(a, res, q);
                    int kk = k1, r2 = q;
                    bool mir = false;
                    if (k1 == 0) { if (q > F / 2) continue; }
                    else if (2 * k1 == a.cs_f1) { if (q >= F / 2) continue; }
                    else if (q >= F / 2) { kk = a.cs_f1 - k1; r2 = F - 1 - q; val.y = -val.y; mir = true; }
                    const int64_t k = kk + (int64_t)a.cs_f1 * r2;   // 0..n/2, every value once
                    // the tile's 128 bytes of adjacent k1 land on one line at k, but on TWO lines at the mirrored index N1 - k1 (shifted by one
                    // element): keep_out = 1 writes those with plain stores so that the L2 merges the pieces of neighbouring tiles
                    if constexpr (CS == 5) {
                        if (a.makhoul == 3) { ((T *)a.out)[ob + k] = val.x * a.scale; continue; }     // DCT-I: y[k] = Re X[k] / 2 (times the pre-scale), plain 4 / 8-byte stores
                        if (mir && a.keep_out) ((cpx<T> *)a.out)[ob + k] = val; else gstore<T, true>((cpx<T> *)a.out + ob + k, val);
                    } else {
                        const cpx<T> tk = cmul(val, a.fc1 ? cmul(a.fc1[kk], a.fc2[r2]) : a.aux2[k]);
                        T *out = (T *)a.out + ob;
                        const T y0 = tk.x * a.scale, y1 = -tk.y * a.scale;
                        if (mir && a.keep_out) out[k] = y0; else __builtin_nontemporal_store(y0, out + k);
                        if (k > 0 && 2 * k < a.cs_n) { if (!mir && a.keep_out) out[a.cs_n - k] = y1; else __builtin_nontemporal_store(y1, out + (a.cs_n - k)); }
                    }
                }
            } else if constexpr (CS == 3) {
                cpx<T> *out = (cpx<T> *)a.out + (csp.o * a.cs_k1n + csp.k1) * a.outer_out + csp.ii;
                if (a.cs_grid3) {
#pragma unroll
                    for (int qi = 0; qi < CSNQ; ++qi) {
                        const int q = j0 + qi * (THREADS / LPB);
                        out[(int64_t)q * a.elem_out] = cmul(post_cplx<T, OP, ZiPhi>(a, res, q), cconj(cs_twl[q]));
                    }
                } else {
                    cpx<T> csw_hi[CSNQ], csw_lo[CSNQ];
                    cs3_twiddles(a, csp.k1, j0, csw_hi, csw_lo);
#pragma unroll
                    for (int qi = 0; qi < CSNQ; ++qi) {
                        const int q = j0 + qi * (THREADS / LPB);
                        out[(int64_t)q * a.elem_out] = cmul(post_cplx<T, OP, ZiPhi>(a, res, q), cconj(cmul(csw_hi[qi], csw_lo[qi])));
                    }
                }
            } else if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + base;
                for (int q = j0; q < a.n_out; q += THREADS / LPB) {
                    const cpx<T> val = post_cplx<T, OP, ZiPhi>(a, res, q);
                    if (XCD || a.keep_out) out[(int64_t)q * a.elem_out] = val; else gstore<T, true>(out + (int64_t)q * a.elem_out, val);
                }
            } else {
                T *out = (T *)a.out + base;
                if constexpr (OP == G_C2R_EVEN && !XCD) {
                    if (a.makhoul) {   // last pass of the inverse real four-step, DCT-III: element q of lane (o, n2) is v[m], m = q inner + n2 -> y[2m] / y[2(n-1-m)+1]
                        T *lane_o = (T *)a.out + Lo * a.outer_out;
                        const int64_t m0 = Li, nn = (int64_t)a.n_out * a.inner;
                        for (int q = j0; q < a.n_out; q += THREADS / LPB) {
                            const int64_t m = (int64_t)q * a.inner + m0;
                            lane_o[2 * q < a.n_out ? 2 * m : 2 * (nn - 1 - m) + 1] = post_real<T, OP, ZiPhi>(a, res, q);
                        }
                        return;
                    }
                }
                for (int q = j0; q < a.n_out; q += THREADS / LPB) {
                    const T val = post_real<T, OP, ZiPhi>(a, res, q);
                    // narrow tiles write 8-32 byte pieces of lines shared with neighbouring tiles: keep them
                    // cacheable so that the XCD's L2 merges whole lines before they go to HBM
                    if constexpr (XCD) out[(int64_t)q * a.elem_out] = val; else __builtin_nontemporal_store(val, out + (int64_t)q * a.elem_out);
                }
            }
        } else {
            const cpx<T> *res = (const cpx<T> *)lds;
            using OT = typename cond_type<OUT_CPLX, cpx<T>, T>::type;
            // DCT-I / DCT-II scatter four real outputs per spectrum pair (k, n-k, F-k, F+k): worth staging.
            // (DCT-III was tried the same way as DCT-IV above: 88 -> 94 us on cfg4 -- its cost is the PRE fold, not the POST)
            // (measured: 117 -> 104 us on cfg4; the ops with one contiguous output per thread lose 5-20 % to the
            // two extra barriers, so they keep their direct stores)
            if constexpr (OP == G_DCT4_EVEN) {
                // DCT-IV: y[2k] = Re(Z[k] c_k), y[n-1-2k] = -Im(Z[k] c_k) -- one LDS read and one twiddle per PAIR of
                // outputs (realops.h post_real reads and multiplies once per output), staged through LDS for 16-byte stores
                if (a.vec_out) {
                    constexpr int NK = (F + TPL - 1) / TPL;
                    T o0[NK], o1[NK];
#pragma unroll
                    for (int i = 0; i < NK; ++i) {
                        const int k = t + i * TPL;
                        if (k < F) { const cpx<T> u = cmul(res[ZiPhi::map(k)], a.aux2[k]); o0[i] = u.x; o1[i] = -u.y; }
                    }
                    __syncthreads();
                    T *stage = (T *)lds;
                    const int n = 2 * F;
#pragma unroll
                    for (int i = 0; i < NK; ++i) {
                        const int k = t + i * TPL;
                        if (k < F) { stage[2 * k] = o0[i]; stage[n - 1 - 2 * k] = o1[i]; }
                    }
                    __syncthreads();
                    if (!live) return;
                    T *out = (T *)a.out + lane * a.pitch_out;
                    constexpr int W = 16 / sizeof(T);
                    const int nv = n / W;
                    for (int j = t; j < nv; j += TPL) __builtin_nontemporal_store(((const vec4f *)stage)[j], (vec4f *)out + j);
                    for (int j = W * nv + t; j < n; j += TPL) out[j] = stage[j];
                    return;
                }
            }
            if constexpr (OP == G_R2C_EVEN) {
                // R2C rows: F + 1 complex per lane -- lanes start and end mid-line and the split produces two scattered streams
                // (k ascending, F - k descending).  When the output lanes are dense the LPB lanes of this workgroup are one
                // contiguous chunk: the outputs go registers -> LDS (own lane region, natural order) -> coalesced 16-byte
                // stores over the whole chunk.  (8192 x 8192 f32 rows: 102 us -> see DESIGN.md section 3.2)
                if (a.chunk_out) {
                    constexpr int NSL = 4 * ((F / 2) / TPL + 1);
                    cpx<T> o[NSL];
                    int oq[NSL];
#pragma unroll
                    for (int i = 0; i < NSL / 4; ++i) {
                        const int k = t + i * TPL;
                        const PairOut<cpx<T>> r = post_pair<cpx<T>>(a, res, k <= F / 2 ? k : 0);
#pragma unroll
                        for (int z = 0; z < 4; ++z) { o[4 * i + z] = r.v[z]; oq[4 * i + z] = k <= F / 2 ? r.q[z] : -1; }
                    }
                    __syncthreads();                       // every read of Z is done: the lane region becomes the output lane
                    cpx<T> *stage = (cpx<T> *)lds;
#pragma unroll
                    for (int i = 0; i < NSL; ++i) if (oq[i] >= 0) stage[oq[i]] = o[i];
                    __syncthreads();
                    constexpr int NOUT = F + 1;
                    const int64_t e0 = (int64_t)blockIdx.x * LPB * NOUT;            // (identity tile map: chunk_out rows never use the XCD map)
                    const int64_t etot = a.nlanes * NOUT;
                    cpx<T> *outc = (cpx<T> *)a.out + e0;
                    // complex elements per store: a 16-byte vector (2 x c64) when every workgroup's chunk starts 16-byte aligned
                    // (LPB (F + 1) even), else one element (odd lane counts of the specialised partial-round configurations)
                    constexpr int EPV = (sizeof(cpx<T>) == 8 && (LPB * NOUT) % 2 == 0) ? 2 : 1;
                    constexpr int NV = (LPB * NOUT + EPV - 1) / EPV;
                    for (int v = threadIdx.x; v < NV; v += THREADS) {
                        const int g = v * EPV;
                        if (e0 + g >= etot) break;
                        const cpx<T> *l0 = (const cpx<T> *)(smem + (size_t)(g / NOUT) * LANE_LDS * 2 * sizeof(T));
                        if constexpr (EPV == 1) {
                            gstore<T, true>(outc + g, l0[g % NOUT]);
                        } else {
                            const cpx<T> c0 = l0[g % NOUT];
                            if (e0 + g + 1 < etot && g + 1 < LPB * NOUT) {
                                const cpx<T> *l1 = (const cpx<T> *)(smem + (size_t)((g + 1) / NOUT) * LANE_LDS * 2 * sizeof(T));
                                const cpx<T> c1 = l1[(g + 1) % NOUT];
                                vec4f w; w.x = c0.x; w.y = c0.y; w.z = c1.x; w.w = c1.y;
                                __builtin_nontemporal_store(w, (vec4f *)(outc + g));
                            } else {
                                gstore<T, true>(outc + g, c0);
                            }
                        }
                    }
                    return;
                }
            }
            if (PAIR && !OUT_CPLX && a.vec_out) {
                // Row layout, 16-byte aligned lanes: outputs go registers -> LDS (raw order) -> 16-byte
                // non-temporal stores, instead of 4/8-byte stores straight from the POST gather.
                constexpr int NSLOT = PAIR ? 4 * ((F / 2) / TPL + 1) : 2 * E;
                OT o[NSLOT];
                int oq[NSLOT];
                if constexpr (PAIR) {
#pragma unroll
                    for (int i = 0; i < NSLOT / 4; ++i) {
                        const int k = t + i * TPL;
                        const PairOut<OT> r = post_pair<OT>(a, res, k <= F / 2 ? k : 0);
#pragma unroll
                        for (int z = 0; z < 4; ++z) { o[4 * i + z] = r.v[z]; oq[4 * i + z] = k <= F / 2 ? r.q[z] : -1; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NSLOT; ++i) {
                        const int q = t + i * TPL;
                        oq[i] = q < a.n_out ? q : -1;
                        if (q < a.n_out) {
                            if constexpr (OUT_CPLX) o[i] = post_cplx<T, OP, ZiPhi>(a, res, q); else o[i] = post_real<T, OP, ZiPhi>(a, res, q);
                        }
                    }
                }
                __syncthreads();                       // every read of Z is done: the lane region becomes the raw output
                OT *stage = (OT *)lds;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) if (oq[i] >= 0) stage[oq[i]] = o[i];
                __syncthreads();
                if (!live) return;
                OT *out = (OT *)a.out + lane * a.pitch_out;
                constexpr int W = 16 / sizeof(OT);
                const int nv = a.n_out / W;
                for (int j = t; j < nv; j += TPL) __builtin_nontemporal_store(((const vec4f *)stage)[j], (vec4f *)out + j);
                for (int j = W * nv + t; j < a.n_out; j += TPL) out[j] = stage[j];
                return;
            }
            if (!live) return;
            if constexpr (PAIR) {
                OT *out = (OT *)a.out + lane * a.pitch_out;
                for (int k = t; k <= F / 2; k += TPL) {
                    const PairOut<OT> r = post_pair<OT>(a, res, k);
#pragma unroll
                    for (int z = 0; z < 4; ++z)
                        if (r.q[z] >= 0) {
                            if constexpr (OUT_CPLX) gstore<T, ROW_NT>((cpx<T> *)out + r.q[z], r.v[z]);
                            else __builtin_nontemporal_store(r.v[z], out + r.q[z]);
                        }
                }
            } else if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + lane * a.pitch_out;
                if constexpr (ROWOUT) {   // (an intermediate the next pass re-reads: keep_out = cache-allocating stores)
                    if (a.keep_out) { for (int q = t; q < a.n_out; q += TPL) out[q] = post_cplx<T, OP, ZiPhi>(a, res, q); return; }
                }
                for (int q = t; q < a.n_out; q += TPL) gstore<T, true>(out + q, post_cplx<T, OP, ZiPhi>(a, res, q));
            } else {
                T *out = (T *)a.out + lane * a.pitch_out;
                if constexpr (OP == G_C2R_EVEN) {
                    // x[2k] = Re, x[2k+1] = -Im of the conj-trick result: one LDS read and one 8/16-byte store per pair
                    // (realops.h post_real would read each element twice and store 4/8 bytes at a time)
                    if (a.vec_out) {
                        for (int k = t; k < F; k += TPL) { const cpx<T> c = res[ZiPhi::map(k)]; gstore<T, true>((cpx<T> *)out + k, mk<T>(c.x, -c.y)); }
                        return;
                    }
                }
                for (int q = t; q < a.n_out; q += TPL) __builtin_nontemporal_store(post_real<T, OP, ZiPhi>(a, res, q), out + q);
            }
        }
    }
};

template <typename K, typename T> __global__ __launch_bounds__(K::THREADS) void k_pow2_real(const RealArgs<T> a) { K::run(a); }

}  // namespace ndfft
