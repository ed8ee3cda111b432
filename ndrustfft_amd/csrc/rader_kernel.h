// rader_kernel.h -- lengths with ONE large prime factor on the register-resident Stockham engine, without Bluestein's
// zero padding: the inner complex FFT length is F = MC * P with P prime, P - 1 smooth (13-smooth, or with one factor 17 / 19 -- f32 also
// 23 / 29 / 31 -- that gets its own pass), and a small cofactor MC: one butterfly (2..16) or a product of two (up to 32 in f64, 48 in f32);
// MC is coprime to P because P is F's largest prime factor and divides it once.
//
//   Good-Thomas (no twiddles): x[(n1 P + n2 MC) mod F] -> X[(k1 C1 + k2 C2) mod F],  C1 = P (P^-1 mod MC), C2 = MC (MC^-1 mod P):
//       X[k1, k2] = sum_n1 W_MC^(n1 k1) sum_n2 W_P^(n2 k2) x[n1, n2]
//   Rader for each of the MC length-P transforms (g a primitive root of P, M = P - 1):
//       a[q] = x[n1, g^q],  b[q] = W_P^(g^-q):   X[n1, g^-k] = x[n1, 0] + (a (*) b)[k]   (cyclic convolution of length M)
//       X[n1, 0] = x[n1, 0] + A[0],  A = FFT_M(a);  (a (*) b) = conj FFT_M(conj(A * bhat)),  bhat = FFT_M(b) / M, and adding
//       x[n1, 0] to bin 0 of the product adds it to every output of the convolution.
//
// Per lane: stage the raw lane -> LDS; PRE (realops.h) gathered through the index table straight into the first pass's
// register pattern of MC independent FFT_M (MC * TPL threads per lane, the passes of pow2_kernel.h with any radix list);
// * bhat, conj in registers; the same passes in REVERSE order (they start from the pattern the first FFT ends in, so the
// product never goes through LDS); scatter to the Rader output order;
// for MC > 1 the length-MC transforms across the MC sub-transforms (one butterfly, or two factors in registers); Z in LDS in natural
// order; POST gather (realops.h).
// FFT work per lane: 2 MC (P - 1) ~ 2 F points against Bluestein's 2 M' with M' = 2^k >= 2F - 1 (2F .. 4F), in a workgroup
// of 1/2 .. 1/4 of the LDS.  Specialised with hiprtc per (P, MC, op, dtype, layout) at first use (jit.hip: launch_jit_rader).
// The lane semantics are the reference's (src/lib.rs:313-338, 497-531, 688-741) through realops.h, exactly as in blue_kernel.h.
#pragma once
#include "pow2_real.h"
#include "reg_kernel.h"

namespace ndfft {

constexpr int rader_inv_mod(int a, int m) {   // a^-1 mod m for coprime a, m (0 when m == 1)
    a %= m;
    for (int x = 1; x < m; ++x) if ((a * x) % m == 1) return x;
    return 0;
}

// MC = MC1 * MC2: the cofactor transform is one butterfly (MC2 = 1: MC = 2..16) or a two-factor Cooley-Tukey in registers (reg_kernel.h: RegFft2::fft,
// e.g. 18 = 6 x 3, 27 = 9 x 3, 28 = 7 x 4; its twiddles W_MC^k come through a.chirp)
//
// SYM (round 5; DCT-I with an ODD cofactor MC > 1, e.g. nddct1 n = 512: F = 511 = 7 x 73): the even extension e of length 2F is split by Good-Thomas
// over 2 x F (F odd) instead of being packed two reals per complex: u_a[b] = e[m], m = b (mod F), m = a (mod 2), are two REAL EVEN sequences of
// length F, and with z = u_0 + i u_1, i.e.
//       z[b] = (x[b], x[F - b]) s  for even b,   (x[F - b], x[b]) s  for odd b          (s = the pre-scale, src/lib.rs:692-696)
// Z = FFT_F(z) = U_0 + i U_1 with U_0, U_1 real, and DFT_2F(e)[q] = U_0[q h] + (-1)^q U_1[q h], h = (F + 1) / 2 = 2^-1 mod F:
//       y[q] = (Re Z[j] + Im Z[j]) / 2,  j = q / 2              for even q,
//       y[q] = (Re Z[j] - Im Z[j]) / 2,  j = ((q + F) / 2) mod F for odd q             (no split twiddle at all)
// z is EVEN, z[-b] = z[b]: in the Good-Thomas grid x[n1, n2] = x[MC - n1, P - n2], so the length-P spectra of rows n1 and MC - n1 are mirror images,
// Y[MC - n1][k2] = Y[n1][-k2].  Only rows 0 .. (MC - 1) / 2 run Rader's convolution -- 4 of 7 for F = 511 -- on (MC + 1) / 2 x TPL threads per lane;
// each result goes to its own place of the [k2][n1] grid or to the mirrored one, and because Z is even too only the columns k2 <= (P - 1) / 2 get
// their cofactor transform (POST reads Z[F - j] for the others).
template <typename T, int P, int MC1, int MC2, int TPL, int LPB, typename RL, int OP, bool COL = false, bool SYM = false> struct RaderKernel {
    static constexpr int MC = MC1 * MC2;
    static_assert(!SYM || (OP == G_DCT1 && (MC & 1)), "SYM: DCT-I with an odd cofactor");
    // HALF (SYM with cofactor 1, F = P prime: nddct1 n = 128, 8192): the Rader sequence a[q] = z[g^q] is itself periodic, a[q + (P-1)/2] = z[-g^q] = a[q], so FFT_(P-1)(a) has only even
    // bins and the cyclic convolution collapses to one of HALF the length on a[0 .. (P-1)/2): (a (*) b)[t] = IFFT_M(FFT_M(a) . bh)[t] with M = (P-1)/2 and
    // bh = FFT_M(b[q] + b[q + M]) / M (plan.hip: build_rader_tables); sum_q a[q] = 2 A[0]; each output is Z[g^-t] = Z[P - g^-t].
    static constexpr bool HALF = SYM && MC == 1;
    static constexpr int MT = P - 1;                                // length of the index tables g^i, g^-i
    static constexpr int ROWS = SYM ? (MC + 1) / 2 : MC;            // sub-transforms that run the convolution
    using COF = RegFft2<T, MC1, MC2, 1, false>;
    static constexpr int M = (SYM && MC1 * MC2 == 1) ? (P - 1) / 2 : P - 1, F = P * MC;       // convolution length
    using RLR = RadixReversed<RL>;
    using FFT = Pow2Kernel<T, M, TPL, LPB * ROWS, false, RL, 0, 1, 0>;
    using FFT2 = Pow2Kernel<T, M, TPL, LPB * ROWS, false, RLR, 0, 1, 0>;
    static_assert(FFT2::E == FFT::E, "same radices, same registers");
    static constexpr int E = FFT::E;
    static constexpr int LTHREADS = TPL * ROWS;                     // threads of one lane
    static constexpr int THREADS = LTHREADS * LPB;
    static constexpr int SUB_LDS = M + (M >> 4) + 2;                // complex elements of one sub-transform's exchange region
    // ... and the raw lane / Z.  SYM keeps only the bins it computes, Z[k1, k2] with k2 <= (P - 1) / 2, at k2 * MC + k1: about half the LDS per lane,
    // i.e. twice the lanes per CU (nddct1 n = 512: 545 -> 313 complex elements)
    static constexpr int ZLEN = SYM ? ((P + 1) / 2) * MC : F;
    static constexpr int ZRAW = (SYM && (F + 2) / 2 > ZLEN + (ZLEN >> 4) + 3) ? (F + 2) / 2 : ZLEN + (ZLEN >> 4) + 3;
    static constexpr int LANE_MIN = (ROWS * SUB_LDS > ZRAW) ? ROWS * SUB_LDS : ZRAW;
    // odd pitch: with few threads per lane the threads of a wave sit in DIFFERENT lanes at the same offset -- an even pitch (in 16-byte elements) puts them on the same
    // LDS banks (F = 34, 2 threads per lane, pitch 40: 16-way conflicts, nddct2 n = 68 179 us; odd pitch: see profiles/r04/r04zd_rader_pitch.txt)
    static constexpr int LANE_LDS = LANE_MIN | 1;
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * 2 * sizeof(T);
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN || OP == G_C2R_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN || OP == G_R2C_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    static constexpr int PINV = rader_inv_mod(P, MC), MINV = rader_inv_mod(MC, P);   // P^-1 mod MC, MC^-1 mod P
    static_assert(FFT::LANE_LDS <= SUB_LDS, "exchange region of one sub-transform");

    // (the remainder in batches of U / 2, U / 4, ...: see pow2_real.h stage_loop -- one load at a time is one round trip to memory each)
    template <int STEP, int U = 8, typename LD, typename ST> static __device__ __forceinline__ void stage_loop(int j0, int n, LD ld, ST st) {
        int j = j0;
        for (; j + (U - 1) * STEP < n; j += U * STEP) {
            decltype(ld(0)) tmp[U];
#pragma unroll
            for (int u = 0; u < U; ++u) tmp[u] = ld(j + u * STEP);
#pragma unroll
            for (int u = 0; u < U; ++u) st(j + u * STEP, tmp[u]);
        }
        if constexpr (U >= 4) stage_loop<STEP, U / 2>(j, n, ld, st);
        else for (; j < n; j += STEP) st(j, ld(j));
    }

    // inner-FFT input element i (natural order) from the raw lane
    static __device__ __forceinline__ cpx<T> pre(const RealArgs<T> &a, const void *raw, int i) {
        if constexpr (SYM) {
            const T u = ((const T *)raw)[i] * a.scale, w = ((const T *)raw)[F - i] * a.scale;
            return (i & 1) ? mk<T>(w, u) : mk<T>(u, w);
        } else
        if constexpr (OP == G_C2C_FWD || OP == G_R2C_EVEN) return ((const cpx<T> *)raw)[i];   // R2C even: z[i] = (x[2i], x[2i+1])
        else if constexpr (OP == G_C2C_INV) return cconj(((const cpx<T> *)raw)[i]);
        else if constexpr (OP == G_R2C_ODD) return mk<T>(((const T *)raw)[i], (T)0);
        else return pre_elem<T, OP, ZiNone>(a, raw, i);
    }

    // real output element q from the natural-order Z
    static __device__ __forceinline__ T post_r(const RealArgs<T> &a, const cpx<T> *res, int q) {
        if constexpr (SYM) {
            int j = (q & 1) ? (q + F) >> 1 : q >> 1;
            if (j >= F) j -= F;
            // bin j = (k1, k2) = (j mod MC, j mod P) of the Good-Thomas output map; Z[-j] = Z[j] and only k2 <= (P - 1) / 2 was computed
            int k2 = j % P, k1 = j % MC;
            if (k2 > (P - 1) / 2) { k2 = P - k2; k1 = k1 ? MC - k1 : 0; }
            const cpx<T> c = res[ZiPhi::map(k2 * MC + k1)];
            return (T)0.5 * ((q & 1) ? c.x - c.y : c.x + c.y);
        } else {
            return post_real<T, OP, ZiPhi>(a, res, q);
        }
    }

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int tl = threadIdx.x % LTHREADS, ll = threadIdx.x / LTHREADS;
        const int t = tl % TPL, n1 = tl / TPL;
        const int64_t lane0 = (int64_t)blockIdx.x * LPB;
        const int64_t lane = lane0 + ll;
        const bool live = lane < a.nlanes;
        char *lds = smem + (size_t)ll * LANE_LDS * 2 * sizeof(T);       // the lane: raw data, then Z
        char *sub = lds + (size_t)n1 * SUB_LDS * 2 * sizeof(T);        // this thread's sub-transform
        // ---- stage the raw lane(s) ----
        if constexpr (COL) {
            const int cl = threadIdx.x % LPB, j0 = threadIdx.x / LPB;
            const int64_t L = lane0 + cl;
            if (L < a.nlanes) {
                const int64_t base = (L / a.inner) * a.outer_in + (L % a.inner);
                char *dst = smem + (size_t)cl * LANE_LDS * 2 * sizeof(T);
                constexpr int STEP = THREADS / LPB;
                if constexpr (IN_CPLX) {
                    const cpx<T> *in = (const cpx<T> *)a.in + base;
                    stage_loop<STEP>(j0, a.n_in, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, cpx<T> v) { ((cpx<T> *)dst)[j] = v; });
                } else {
                    const T *in = (const T *)a.in + base;
                    stage_loop<STEP>(j0, a.n_in, [&](int j) { return in[(int64_t)j * a.elem_in]; }, [&](int j, T v) { ((T *)dst)[j] = v; });
                }
            }
        } else {
            const int64_t lsafe = live ? lane : 0;
            if constexpr (IN_CPLX) {
                const cpx<T> *in = (const cpx<T> *)a.in + lsafe * a.pitch_in;
                cpx<T> *raw = (cpx<T> *)lds;
                stage_loop<LTHREADS>(tl, a.n_in, [&](int j) { return in[j]; }, [&](int j, cpx<T> v) { raw[j] = v; });
            } else {
                const T *in = (const T *)a.in + lsafe * a.pitch_in;
                T *raw = (T *)lds;
                // 16-byte staging loads where the lanes are 16-byte aligned in global memory (vec_in) and in LDS, streaming (nt) when the residency model says the
                // input comes from HBM (stream_in) -- round 5: cfg4 nddct1 from HBM 152 us.  F64 ONLY: LANE_LDS is odd (bank spreading), so a lane base in LDS is
                // 16-byte aligned for 16-byte elements and never for f32's 8-byte ones -- f32 rows keep the 4-byte loads LTHREADS apart (and the 8-byte stores below)
                if ((LANE_LDS * 2 * sizeof(T)) % 16 == 0 && a.vec_in) {
                    constexpr int W = 16 / sizeof(T);
                    const int nv = a.n_in / W;
                    if (a.stream_in) stage_loop<LTHREADS>(tl, nv, [&](int j) { return __builtin_nontemporal_load((const vec4f *)in + j); }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    else stage_loop<LTHREADS>(tl, nv, [&](int j) { return ((const vec4f *)in)[j]; }, [&](int j, vec4f v) { ((vec4f *)raw)[j] = v; });
                    for (int j = W * nv + tl; j < a.n_in; j += LTHREADS) raw[j] = in[j];
                } else {
                    stage_loop<LTHREADS>(tl, a.n_in, [&](int j) { return in[j]; }, [&](int j, T v) { raw[j] = v; });
                }
            }
        }
        __syncthreads();
        // ---- PRE, gathered in Rader order, in the first pass's register pattern ----
        constexpr int R0 = RL::at(0), NB0 = FFT::nbfly(0), NBF0 = FFT::slots(0);
        constexpr int RLAST = RL::at(RL::NP - 1), NBL = FFT::nbfly(RL::NP - 1), NBFL = FFT::slots(RL::NP - 1);
        constexpr bool FULL0 = FFT::full(0), FULLL = FFT::full(RL::NP - 1);
        const int32_t *gpow = a.rader_tab, *ginv = a.rader_tab + MT;     // g^i mod P, g^-i mod P
        const int row0 = (n1 * P) % F;                                    // (n1, n2 = 0)
        const cpx<T> x0 = pre(a, (const void *)lds, row0);
        cpx<T> v[E];
#pragma unroll
        for (int q = 0; q < NBF0; ++q)
            if (FULL0 || t + q * TPL < NB0) {
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    int i = row0 + MC * gpow[t + q * TPL + r * NB0];
                    if (MC > 1 && i >= F) i -= F;
                    v[q * R0 + r] = pre(a, (const void *)lds, i);
                }
            }
        // (the first exchange inside passes() starts with a barrier, so the raw lane is dead by then)
        FFT::template passes<0>(v, a.twp, sub, t);
        // ---- * bhat (+ x0 in bin 0), conj: in registers, already in the first-pass pattern of the reversed radix list ----
        cpx<T> X0 = x0;
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
            if (FULLL || t + q * TPL < NBL) {
#pragma unroll
                for (int r = 0; r < RLAST; ++r) {
                    cpx<T> c = cmul(v[q * RLAST + r], a.bhat[t + q * TPL + r * NBL]);
                    if (q == 0 && r == 0) {
                        if (t == 0) { X0 = HALF ? cadd(x0, cadd(v[0], v[0])) : cadd(x0, v[0]); c = cadd(c, x0); }
                    }
                    v[q * RLAST + r] = cconj(c);
                }
            }
        FFT2::template passes<0>(v, a.twp_rev, sub, t);     // (ends in the pattern of RL's FIRST pass)
        __syncthreads();
        cpx<T> *zz = (cpx<T> *)lds;
        if constexpr (MC == 1) {
            // ---- Z[g^-k] = conj(.), Z[0] = x0 + A[0] ----
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
                if (FULL0 || t + q * TPL < NB0) {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        int k2 = ginv[t + q * TPL + r * NB0];
                        if constexpr (HALF) { if (k2 > (P - 1) / 2) k2 = P - k2; }      // Z is even: the compact half (post_r), one place per convolution output
                        zz[ZiPhi::map(k2)] = cconj(v[q * R0 + r]);
                    }
                }
            if (t == 0) zz[0] = X0;
        } else {
            // ---- Y[n1][k2] -> LDS as [k2][n1]; radix-MC butterflies across n1; Z[(k1 C1 + k2 C2) mod F] ----
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
                if (FULL0 || t + q * TPL < NB0) {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const int k2 = ginv[t + q * TPL + r * NB0];
                        const cpx<T> y = cconj(v[q * R0 + r]);
                        if constexpr (SYM) {
                            // only the columns k2 <= (P - 1) / 2 of the grid are transformed (Z is even as well): a value lands in its own place, or -- from the
                            // upper half -- in the mirrored place of row MC - n1, Y[MC - n1][-k2] = Y[n1][k2]; row 0 is its own mirror
                            if (k2 <= (P - 1) / 2) zz[k2 * MC + n1] = y;
                            else if (n1 > 0) zz[(P - k2) * MC + (MC - n1)] = y;
                        } else {
                            zz[k2 * MC + n1] = y;
                        }
                    }
                }
            if (t == 0) {
                zz[n1] = X0;
                if constexpr (SYM) { if (n1 > 0) zz[MC - n1] = X0; }
            }
            __syncthreads();
            constexpr int PK = SYM ? (P + 1) / 2 : P;                       // columns k2 that get a cofactor transform
            constexpr int NS = (PK + LTHREADS - 1) / LTHREADS;
            cpx<T> w[NS][MC];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k2 = tl + s * LTHREADS;
                if (k2 < PK) {
#pragma unroll
                    for (int j = 0; j < MC; ++j) w[s][j] = zz[k2 * MC + j];
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k2 = tl + s * LTHREADS;
                if (k2 < PK) {
                    COF::fft(w[s], a.chirp);                              // register slot j holds output k1 = COF::out_index(j)
                    [[maybe_unused]] const int kb = MC * ((k2 * MINV) % P);              // k2 C2 mod F
#pragma unroll
                    for (int j = 0; j < MC; ++j) {
                        if constexpr (SYM) {
                            zz[ZiPhi::map(k2 * MC + COF::out_index(j))] = w[s][j];       // compact: bin (k1, k2) at k2 MC + k1 (post_r)
                        } else {
                            int k = kb + P * ((COF::out_index(j) * PINV) % MC);   // + k1 C1 mod F
                            if (k >= F) k -= F;
                            zz[ZiPhi::map(k)] = w[s][j];
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- POST gather + store ----
        if constexpr (COL) {
            const int cl = threadIdx.x % LPB, j0 = threadIdx.x / LPB;
            const int64_t L = lane0 + cl;
            if (L >= a.nlanes) return;
            const int64_t base = (L / a.inner) * a.outer_out + (L % a.inner);
            const cpx<T> *res = (const cpx<T> *)(smem + (size_t)cl * LANE_LDS * 2 * sizeof(T));
            if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + base;
                for (int q = j0; q < a.n_out; q += THREADS / LPB) gstore<T, true>(out + (int64_t)q * a.elem_out, post_cplx<T, OP, ZiPhi>(a, res, q));
            } else {
                T *out = (T *)a.out + base;
                for (int q = j0; q < a.n_out; q += THREADS / LPB) __builtin_nontemporal_store(post_r(a, res, q), out + (int64_t)q * a.elem_out);
            }
        } else {
            if constexpr (SYM && (LANE_LDS * 2 * sizeof(T)) % 16 == 0) {
                // DCT-I rows, 16-byte aligned lanes (vec_out), f64 only (see the staging loads above): outputs go registers -> LDS (the lane region, natural order) -> 16-byte non-temporal
                // stores instead of 8-byte stores LTHREADS elements apart (the staged POST of pow2_real.h)
                if (a.vec_out) {
                    constexpr int NQ = (F + 1 + LTHREADS - 1) / LTHREADS;
                    const cpx<T> *res0 = (const cpx<T> *)lds;
                    T o[NQ];
#pragma unroll
                    for (int i = 0; i < NQ; ++i) { const int q = tl + i * LTHREADS; if (q <= F) o[i] = post_r(a, res0, q); }
                    __syncthreads();                       // every read of Z is done: the lane region becomes the output lane
                    T *stage = (T *)lds;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) { const int q = tl + i * LTHREADS; if (q <= F) stage[q] = o[i]; }
                    __syncthreads();
                    if (!live) return;
                    T *out = (T *)a.out + lane * a.pitch_out;
                    constexpr int W = 16 / sizeof(T);
                    constexpr int NV = (F + 1) / W;
                    for (int j = tl; j < NV; j += LTHREADS) __builtin_nontemporal_store(((const vec4f *)stage)[j], (vec4f *)out + j);
                    for (int j = W * NV + tl; j <= F; j += LTHREADS) out[j] = stage[j];
                    return;
                }
            }
            if (!live) return;
            const cpx<T> *res = (const cpx<T> *)lds;
            if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + lane * a.pitch_out;
                for (int q = tl; q < a.n_out; q += LTHREADS) gstore<T, true>(out + q, post_cplx<T, OP, ZiPhi>(a, res, q));
            } else {
                T *out = (T *)a.out + lane * a.pitch_out;
                for (int q = tl; q < a.n_out; q += LTHREADS) __builtin_nontemporal_store(post_r(a, res, q), out + q);
            }
        }
    }
};

}  // namespace ndfft
