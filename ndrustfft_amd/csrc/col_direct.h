// col_direct.h -- LANE-FASTEST register kernel for the passes of the four-step routes whose tile is a set of ADJACENT lanes with a strided
// transform axis (complex lanes, power-of-two F = 64..1024).
//
// The column kernels of pow2_real.h stage the whole F x LPB complex tile in LDS twice (raw tile in, natural-order result out) because their
// thread map is lane-major (thread = lane * TPL + t): a wavefront then owns ONE lane, and only an LDS transpose makes its global accesses
// contiguous.  Here thread = t * LPB + lane: a wavefront owns one butterfly slot of LPB adjacent lanes, so
//   * the loads go global -> registers directly, already in the first pass's pattern (element t + q TPL + r F/R0 of LPB adjacent lanes =
//     one 128-byte row segment per (q, r)), with all E of them in flight;
//   * the passes are pow2_kernel.h's with the HALF exchange (real parts, then imaginary parts, through F reals per lane);
//   * the stores go registers -> global directly in the last pass's pattern -- as rows k of adjacent lanes (column store) or, for the
//     transposing first pass of the row four-step, as runs of consecutive k of one lane (a wavefront writes 64 / LPB consecutive elements of
//     LPB lanes: whole 128-byte lines for c128).
// LDS per tile: LPB x (F + F/16 + 1) x sizeof(T) -- a quarter of the staged tile plus exchange of the lane-major kernel for the same lanes
// (F = 512 c128, 8 lanes: 17 KiB instead of 70 KiB; F = 1024: 34 KiB instead of 139 KiB), no staging barriers, two LDS round trips fewer.
#pragma once
#include "pow2_real.h"

namespace ndfft {

// MODE (the CS numbering of pow2_real.h):
//   0  first pass of the row four-step: column load, ROW store (lane L -> out + L pitch_out)
//   4  second pass of the row four-step: twiddle W_N^(j k1) on load (k1 = the inner index), column store at row k2
//   5  second pass of the REAL four-step, R2C: half spectrum, the mirrored half conjugated into place (pow2_real.h CS = 5)
//   6  the same with the DCT-II post-twiddle: two real outputs per element (CS = 6)
//   7  FIRST pass of the inverse real four-step, C2R: lanes (o, k1), k1 = 0..N1/2; element k2 is Xh[k1 + N1 k2] of the Hermitian extension (read at
//      the mirrored index and conjugated beyond n/2; scaled, imaginary parts of DC and Nyquist dropped: src/lib.rs:511-521), unnormalised inverse FFT
//      over k2, times W_n^(-n2 k1), ROW store s[(o, k1)][n2] -- the column C2R kernels of length N1 finish the lane (exec.hip: real_fourstep_inv)
//   8  the same for DCT-III: the element is V[k] = 0.5 s (x[k] - i x[n-k]) e^(+i pi k / 2n) built from two real loads (realops.h: G_DCT3_EVEN)
//   9  second pass of the fused DCT-IV four-step: as mode 4, then y[2k] = Re(Z[k] c_k), y[n-1-2k] = -Im(Z[k] c_k), k = k1 + F1 k2 (c_k = aux2[k]; src/lib.rs:726-741
//      through realops.h: G_DCT4_EVEN); a tile writes the even (or odd) reals of its lines, its mirror tile the others: mirror_pair_tile
template <typename T, int F, int TPL, int LPB, typename RL, int OP, int MODE> struct ColDirectKernel {
    static_assert(OP == G_C2C_FWD || OP == G_C2C_INV, "complex lanes only");
    static_assert(MODE == 0 || MODE == 4 || ((MODE >= 5 && MODE <= 9) && OP == G_C2C_FWD), "bad mode");
    using FFT = Pow2Kernel<T, F, TPL, LPB, true, RL, 0, 1, 0>;
    static constexpr int E = FFT::E, THREADS = TPL * LPB;
    static_assert(E * TPL == F, "whole butterfly rounds only");
    // The four-step twiddle W_N^(i k1) of element i = t + d (d = q TPL + r F/R: E values) of lane k1 is W^(t k1) x W^(d k1).  W^(t k1) is one per thread (two table
    // loads); the E x LPB values W^(d k1) are a tile's: threads t < E load one each, multiply and leave it in LDS behind the exchange area.  Before round 6 every
    // thread gathered 2 x E table entries of its own (16 scattered 16-byte loads per thread beside 8 loads of data: 38-54 vector loads per wave, SQ counters in
    // profiles/r09/r09f_sq_long_round5_kernels.json); the number of complex multiplies per element is unchanged (two).
    static constexpr bool TWIDDLED = MODE >= 4;
    static constexpr int NSTEP = (E + TPL - 1) / TPL;      // step twiddles per thread row: 1 for every ahead-of-time recipe (TPL >= E), more for hiprtc recipes with E > TPL
    static constexpr size_t STEP_BYTES = TWIDDLED ? (size_t)E * LPB * 2 * sizeof(T) : 0;
    static constexpr size_t LDS_BYTES = FFT::LDS_BYTES + STEP_BYTES;
    static __device__ __forceinline__ cpx<T> tw_at(const RealArgs<T> &a, int m) { return cmul(a.cs_twhi[m >> a.cs_logB], a.cs_twlo[m & ((1 << a.cs_logB) - 1)]); }

    // MODE 7 / 8 (see above)
    static __device__ __forceinline__ void run_inv(const RealArgs<T> &a, int64_t L, int64_t o, int k1, bool live, int t, int cl, char *lds, cpx<T> *stepl) {
        cpx<T> v[E];
        constexpr int R0 = RL::at(0), NB0 = F / R0, NBF0 = FFT::slots(0);
        constexpr int RL_ = RL::at(RL::NP - 1), NBL = F / RL_, NBFL = FFT::slots(RL::NP - 1);
        const int64_t nn = a.cs_n;
        const T hs = (T)0.5 * a.scale;
        // the store twiddles' table entries: issued in front of the data loads (older loads return first), combined behind them
        const int k1s = live ? k1 : 0;
        cpx<T> s_hi[NSTEP], s_lo[NSTEP];
        const int mask = (1 << a.cs_logB) - 1;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) { const int e = t + i * TPL; if (e < E) { const int m = ((e / RL_) * TPL + (e % RL_) * NBL) * k1s; s_hi[i] = a.cs_twhi[m >> a.cs_logB]; s_lo[i] = a.cs_twlo[m & mask]; } }
        const int mb = t * k1s;
        const cpx<T> b_hi = a.cs_twhi[mb >> a.cs_logB], b_lo = a.cs_twlo[mb & mask];
#pragma unroll
        for (int q = 0; q < NBF0; ++q)
#pragma unroll
            for (int r = 0; r < R0; ++r) {
                cpx<T> x = mk<T>((T)0, (T)0);
                if (live) {
                    const int64_t k = k1 + (int64_t)a.cs_f1 * (t + q * TPL + r * NB0);
                    const bool mir = 2 * k > nn;
                    const int64_t src = mir ? nn - k : k;
                    if constexpr (MODE == 7) {
                        x = a.stream_in ? gload<T, true>((const cpx<T> *)a.in + o * a.outer_in + src) : ((const cpx<T> *)a.in)[o * a.outer_in + src];
                        x.x *= a.scale; x.y *= a.scale;
                        if (src == 0 || 2 * src == nn) x.y = (T)0;
                    } else {
                        const T *xr = (const T *)a.in + o * a.outer_in;
                        // e^{+i pi src/(2n)}: src = k1 + N1 i, or mirrored (N1 - k1) + N1 (N2 - 1 - i) -- two small tables (engine.h: rfs_c1 / rfs_c2) instead of a stream of n/2 + 1 entries
                        const int i_ = t + q * TPL + r * NB0;
                        const cpx<T> cw = a.fc1 ? cmul(a.fc1[mir ? a.cs_f1 - k1 : k1], a.fc2[mir ? F - 1 - i_ : i_]) : a.aux2[src];
                        x = cmul(mk<T>(xr[src] * hs, src ? -xr[nn - src] * hs : (T)0), cconj(cw));
                    }
                    if (!mir) x.y = -x.y;   // conj of the Hermitian extension (mirrored elements are conjugates already): inverse FFT by forward butterflies
                }
                v[q * R0 + r] = x;
            }
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) { const int e = t + i * TPL; if (e < E) stepl[e * LPB + cl] = cmul(s_hi[i], s_lo[i]); }
        const cpx<T> base = cmul(b_hi, b_lo);
        // (the passes' exchange barriers publish stepl; it lies behind the exchange area and is read only in the store loop)
        FFT::template passes<0>(v, a.twp, lds, t);
        if (!live) return;
        cpx<T> *out = (cpx<T> *)a.out + (o * a.cs_k1n + k1) * a.pitch_out;
        // (round 5: loading the table entries of all E outputs before the first store measured WORSE here -- E more complex registers across the store phase; round 6: they
        //  come from the tile's LDS table, nothing is loaded from global memory between the stores)
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
#pragma unroll
            for (int r = 0; r < RL_; ++r) {
                const int kq = t + q * TPL + r * NBL;
                const cpx<T> u = cmul(v[q * RL_ + r], cmul(base, stepl[(q * RL_ + r) * LPB + cl]));
                out[kq] = mk<T>(u.x, -u.y);   // conj(r W^(n2 k1)) = IFFT value times W^(-n2 k1); plain store: the next launch re-reads it
            }
    }

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int cl = threadIdx.x % LPB, t = threadIdx.x / LPB;
        // (xcd_chunk: runs of consecutive tiles on one XCD, so that lines shared by neighbouring tiles -- the mirrored index N1 - k1 is shifted
        //  by one element, real outputs are half lines -- merge in that XCD's L2 when they are written with plain stores: keep_out)
        int64_t tile = xcd_block(blockIdx.x, gridDim.x, a.xcd_chunk);
        if constexpr (MODE == 9) { if (a.inner % LPB == 0) tile = mirror_pair_tile(blockIdx.x, a.inner / LPB); }
        const int64_t L = tile * LPB + cl;
        const int64_t o = L / a.inner;
        const int k1 = (int)(L - o * a.inner);
        // (MODE 5 / 6: the inner index runs over a pitch padded to whole 128-byte lines; k1 > N1/2 is padding)
        const bool live = L < a.nlanes && (MODE < 5 || MODE == 9 || 2 * k1 <= a.cs_f1);
        cpx<T> *stepl = (cpx<T> *)(smem + FFT::LDS_BYTES);
        if constexpr (MODE == 7 || MODE == 8) { run_inv(a, L, o, k1, live, t, cl, smem + (size_t)cl * FFT::LANE_LDS * sizeof(T), stepl); return; }
        char *lds = smem + (size_t)cl * FFT::LANE_LDS * sizeof(T);
        cpx<T> v[E];
        {
            constexpr int R0 = RL::at(0), NB0 = F / R0, NBF0 = FFT::slots(0);
            // the load twiddles' table entries, in front of the data loads (see TWIDDLED)
            cpx<T> s_hi[NSTEP], s_lo[NSTEP], b_hi = mk<T>((T)1, (T)0), b_lo = b_hi;
            if constexpr (TWIDDLED) {
                const int k1s = live ? k1 : 0, mask = (1 << a.cs_logB) - 1;
#pragma unroll
                for (int i = 0; i < NSTEP; ++i) { const int e = t + i * TPL; if (e < E) { const int m = ((e / R0) * TPL + (e % R0) * NB0) * k1s; s_hi[i] = a.cs_twhi[m >> a.cs_logB]; s_lo[i] = a.cs_twlo[m & mask]; } }
                const int mb = t * k1s;
                b_hi = a.cs_twhi[mb >> a.cs_logB]; b_lo = a.cs_twlo[mb & mask];
            }
            const cpx<T> *in = (const cpx<T> *)a.in + o * a.outer_in + k1;
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const int i = t + q * TPL + r * NB0;
                    if (live) v[q * R0 + r] = a.stream_in ? gload<T, true>(in + (int64_t)i * a.elem_in) : in[(int64_t)i * a.elem_in];
                    else v[q * R0 + r] = mk<T>((T)0, (T)0);
                }
            if constexpr (OP == G_C2C_INV) {
#pragma unroll
                for (int i = 0; i < E; ++i) v[i].y = -v[i].y;
            }
            if constexpr (TWIDDLED) {   // W_N^(i k1), after the conjugation: the same table serves both directions
#pragma unroll
                for (int i = 0; i < NSTEP; ++i) { const int e = t + i * TPL; if (e < E) stepl[e * LPB + cl] = cmul(s_hi[i], s_lo[i]); }
                __syncthreads();
                const cpx<T> base = cmul(b_hi, b_lo);
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = cmul(v[e], cmul(base, stepl[e * LPB + cl]));
            }
        }
        FFT::template passes<0>(v, a.twp, lds, t);
        if (!live) return;
        constexpr int RL_ = RL::at(RL::NP - 1), NBL = F / RL_, NBFL = FFT::slots(RL::NP - 1);
        if constexpr (OP == G_C2C_INV) {
#pragma unroll
            for (int i = 0; i < E; ++i) { v[i].x *= a.scale; v[i].y *= -a.scale; }   // conj + normalisation (src/lib.rs:326-330)
        }
        // MODE 9: the post-twiddles c_k of all E outputs are loaded before the first store -- inside the store loop every table load waits behind the previous element's
        // stores (may-alias), E dependent L2 round trips per tile: nddct4 64 x 262144 f64 147 -> 133 us (round 5; MODE 6 measured the same either way and keeps its loop)
        cpx<T> ck[MODE == 9 ? E : 1];
        if constexpr (MODE == 9) {
#pragma unroll
            for (int q = 0; q < NBFL; ++q)
#pragma unroll
                for (int r = 0; r < RL_; ++r) {
                    const int kq = t + q * TPL + r * NBL;
                    ck[q * RL_ + r] = a.aux2[k1 + (int64_t)a.cs_f1 * kq];
                }
        }
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
#pragma unroll
            for (int r = 0; r < RL_; ++r) {
                const int kq = t + q * TPL + r * NBL;
                cpx<T> val = v[q * RL_ + r];
                if constexpr (MODE == 0) {
                    cpx<T> *out = (cpx<T> *)a.out + L * a.pitch_out + kq;
                    if (a.keep_out) *out = val; else gstore<T, true>(out, val);
                } else if constexpr (MODE == 4) {
                    gstore<T, true>((cpx<T> *)a.out + o * a.outer_out + k1 + (int64_t)kq * a.elem_out, val);
                } else if constexpr (MODE == 9) {
                    const int64_t k = k1 + (int64_t)a.cs_f1 * kq;
                    const cpx<T> tk = cmul(val, ck[q * RL_ + r]);
                    T *out = (T *)a.out + o * a.outer_out;
                    out[2 * k] = tk.x;                       // plain stores: the mirror tile completes every line in the same L2
                    out[(int64_t)a.cs_n - 1 - 2 * k] = -tk.y;
                } else {
                    int kk = k1, r2 = kq;
                    bool mir = false;
                    if (k1 == 0) { if (kq > F / 2) continue; }
                    else if (2 * k1 == a.cs_f1) { if (kq >= F / 2) continue; }
                    else if (kq >= F / 2) { kk = a.cs_f1 - k1; r2 = F - 1 - kq; val.y = -val.y; mir = true; }
                    const int64_t k = kk + (int64_t)a.cs_f1 * r2;   // 0..n/2, every value once
                    const int64_t ob = o * a.outer_out;
                    if constexpr (MODE == 5) {
                        if (a.makhoul == 3) { ((T *)a.out)[ob + k] = val.x * a.scale; continue; }     // DCT-I (real_fourstep with dct1): y[k] = Re X[k] / 2 times the pre-scale
                        if ((mir && a.keep_out) || (a.keep_out & 2)) ((cpx<T> *)a.out)[ob + k] = val; else gstore<T, true>((cpx<T> *)a.out + ob + k, val);
                    } else {
                        const cpx<T> tk = cmul(val, a.fc1 ? cmul(a.fc1[kk], a.fc2[r2]) : a.aux2[k]);
                        T *out = (T *)a.out + ob;
                        const T y0 = tk.x * a.scale, y1 = -tk.y * a.scale;
                        if ((mir && a.keep_out) || (a.keep_out & 2)) out[k] = y0; else __builtin_nontemporal_store(y0, out + k);
                        if (k > 0 && 2 * k < a.cs_n) { if ((!mir && a.keep_out) || (a.keep_out & 2)) out[a.cs_n - k] = y1; else __builtin_nontemporal_store(y1, out + (a.cs_n - k)); }
                    }
                }
            }
    }
};

// Entry point with a floor on waves per SIMD (= a cap on VGPRs): without it these kernels take 126-159 VGPRs -- ONE 512-thread workgroup per CU.
#ifndef NDFFT_COLDIRECT_MIN_WAVES
#define NDFFT_COLDIRECT_MIN_WAVES 4
#endif
template <typename K, typename T> __global__ __launch_bounds__(K::THREADS, NDFFT_COLDIRECT_MIN_WAVES) void k_col_direct(const RealArgs<T> a) { K::run(a); }

}  // namespace ndfft
