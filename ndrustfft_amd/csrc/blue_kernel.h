// blue_kernel.h -- Bluestein on the register-resident Stockham engine: any op whose inner complex FFT
// length F has a prime factor > 13 (prime lane lengths, DCT-I with n-1 not smooth, ...), one pass over HBM.
//
//   Z[k] = chirp[k] * IFFT_M( FFT_M(z * chirp, zero padded to M) * bhat )[k],   chirp[j] = e^{-i pi j^2/F},
//   M >= 2F - 1 (a power of two, or a cheaper 13-smooth length: jit.hip blue_pick_len), bhat = FFT_M(conj chirp, wrapped) / M
//   (plan.hip builds both tables in long double)
//
// Per lane: stage raw lane -> LDS; PRE (realops.h) * chirp -> registers in the first pass's pattern, zero
// padded; the power-of-two passes of pow2_kernel.h; * bhat and conj in registers; the same passes in REVERSE
// order (they start from the pattern the first FFT ends in); conj * chirp -> Z in LDS (natural order, k < F);
// POST gather (realops.h) -> global.  Both FFTs are forward butterflies (IFFT = conj . FFT . conj).
// Specialised with hiprtc per (M, op, dtype, layout) at first use (jit.hip) -- the LDS kernel
// (generic_kernel.h) runs the same algorithm with run-time radices at 4-6 % of the HBM roofline.
// The lane semantics are the reference's (src/lib.rs:313-338, 497-531, 688-741) through realops.h.
#pragma once
#include "pow2_real.h"

namespace ndfft {

template <typename T, int M, int TPL, int LPB, typename RL, int OP, bool COL = false> struct BlueKernel {
    static constexpr int E = Pow2Kernel<T, M, TPL, LPB, false, RL, 0, 1, 0>::E;     // = M / TPL when every pass has whole rounds (partial rounds: pow2_kernel.h)
    static constexpr int THREADS = TPL * LPB;
    static constexpr int LANE_LDS = COL ? ((M + (M >> 4) + 2) | 1) : ((M + (M >> 4) + 3) & ~1);   // complex elements per lane
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * 2 * sizeof(T);
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN || OP == G_C2R_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN || OP == G_R2C_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    using FFT = Pow2Kernel<T, M, TPL, LPB, false, RL, 0, 1, 0>;
    using FFT2 = Pow2Kernel<T, M, TPL, LPB, false, RadixReversed<RL>, 0, 1, 0>;   // the passes back to front (pow2_real.h: RadixReversed)

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

    // FFT input element i (before the chirp) from the raw lane
    static __device__ __forceinline__ cpx<T> pre(const RealArgs<T> &a, const void *raw, int i) {
        if constexpr (OP == G_C2C_FWD || OP == G_R2C_EVEN) return ((const cpx<T> *)raw)[i];   // R2C even: z[i] = (x[2i], x[2i+1])
        else if constexpr (OP == G_C2C_INV) return cconj(((const cpx<T> *)raw)[i]);
        else if constexpr (OP == G_R2C_ODD) return mk<T>(((const T *)raw)[i], (T)0);
        else return pre_elem<T, OP, ZiNone>(a, raw, i);
    }

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        const int64_t lane0 = (int64_t)blockIdx.x * LPB;
        const int64_t lane = lane0 + ll;
        const bool live = lane < a.nlanes;
        char *lds = smem + (size_t)ll * LANE_LDS * 2 * sizeof(T);
        const int F = a.F;
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
                stage_loop<TPL>(t, a.n_in, [&](int j) { return in[j]; }, [&](int j, cpx<T> v) { raw[j] = v; });
            } else {
                const T *in = (const T *)a.in + lsafe * a.pitch_in;
                T *raw = (T *)lds;
                stage_loop<TPL>(t, a.n_in, [&](int j) { return in[j]; }, [&](int j, T v) { raw[j] = v; });
            }
        }
        __syncthreads();
        // ---- PRE * chirp, zero padded, in the first pass's register pattern ----
        constexpr int R0 = RL::at(0), NB0 = FFT::nbfly(0), NBF0 = FFT::slots(0);
        constexpr int RLAST = RL::at(RL::NP - 1), NBL = FFT::nbfly(RL::NP - 1), NBFL = FFT::slots(RL::NP - 1);
        constexpr bool FULL0 = FFT::full(0), FULLL = FFT::full(RL::NP - 1);
        cpx<T> v[E];
#pragma unroll
        for (int q = 0; q < NBF0; ++q)
            if (FULL0 || t + q * TPL < NB0) {
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const int i = t + q * TPL + r * NB0;
                    v[q * R0 + r] = i < F ? cmul(pre(a, (const void *)lds, i), a.chirp[i]) : mk<T>((T)0, (T)0);
                }
            }
        // (the first exchange inside passes() starts with a barrier, so the raw lane is dead by then)
        FFT::template passes<0>(v, a.twp, lds, t);
        // ---- * bhat, conj: in registers -- the reversed pass order starts from exactly this pattern ----
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
            if (FULLL || t + q * TPL < NBL) {
#pragma unroll
                for (int r = 0; r < RLAST; ++r) v[q * RLAST + r] = cconj(cmul(v[q * RLAST + r], a.bhat[t + q * TPL + r * NBL]));
            }
        FFT2::template passes<0>(v, a.twp_rev, lds, t);
        // ---- Z[k] = conj(.) * chirp[k], k < F, natural order (the reversed list ends in the pattern of RL's first pass) ----
        __syncthreads();
        {
            cpx<T> *z = (cpx<T> *)lds;
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
                if (FULL0 || t + q * TPL < NB0) {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const int o = t + q * TPL + r * NB0;
                        if (o < F) z[ZiPhi::map(o)] = cmul(cconj(v[q * R0 + r]), a.chirp[o]);
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
                for (int q = j0; q < a.n_out; q += THREADS / LPB) __builtin_nontemporal_store(post_real<T, OP, ZiPhi>(a, res, q), out + (int64_t)q * a.elem_out);
            }
        } else {
            if (!live) return;
            const cpx<T> *res = (const cpx<T> *)lds;
            if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + lane * a.pitch_out;
                for (int q = t; q < a.n_out; q += TPL) gstore<T, true>(out + q, post_cplx<T, OP, ZiPhi>(a, res, q));
            } else {
                T *out = (T *)a.out + lane * a.pitch_out;
                for (int q = t; q < a.n_out; q += TPL) __builtin_nontemporal_store(post_real<T, OP, ZiPhi>(a, res, q), out + q);
            }
        }
    }
};

}  // namespace ndfft
