// plain_kernel.h -- the odd-n variants of the real-data ops (R2C / C2R / DCT-II..IV with n odd: the inner complex FFT has length
// F = n, or 2n for DCT-IV, never a power of two) on the register-resident Stockham engine, for smooth F.  RealPow2Kernel (pow2_real.h)
// pairs bins k and F - k in registers and only covers the even-n forms; before this kernel the odd forms ran on the LDS kernel
// (generic_kernel.h) at 12-30 % of the HBM roofline.  Same skeleton as blue_kernel.h without the convolution:
//   stage raw lane -> LDS; PRE (realops.h) -> registers in the first pass's pattern; the passes of pow2_kernel.h (any radix list, partial
//   rounds allowed); Z -> LDS in natural order; POST gather (realops.h) -> global.  Rows and column tiles.
// Specialised with hiprtc per (F, op, dtype, layout) at first use (jit.hip: launch_jit_plain).
// The lane semantics are the reference's (src/lib.rs:497-531, 688-741) through realops.h.
#pragma once
#include "pow2_real.h"

namespace ndfft {

template <typename T, int F, int TPL, int LPB, typename RL, int OP, bool COL = false> struct PlainRealKernel {
    using FFT = Pow2Kernel<T, F, TPL, LPB, false, RL, 0, 1, 0>;
    static constexpr int E = FFT::E;
    static constexpr int THREADS = TPL * LPB;
    static constexpr int LANE_LDS = COL ? ((F + (F >> 4) + 3) | 1) : ((F + (F >> 4) + 4) & ~1);   // complex elements per lane (raw lane, exchange, Z)
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * 2 * sizeof(T);
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN || OP == G_C2R_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN || OP == G_R2C_ODD || OP == G_C2C_FWD || OP == G_C2C_INV;
    static_assert(FFT::LANE_LDS <= LANE_LDS, "exchange region");

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

    static __device__ __forceinline__ cpx<T> pre(const RealArgs<T> &a, const void *raw, int i) {
        if constexpr (OP == G_R2C_ODD) return mk<T>(((const T *)raw)[i], (T)0);
        else return pre_elem<T, OP, ZiNone>(a, raw, i);
    }

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        const int64_t lane0 = (int64_t)blockIdx.x * LPB;
        const int64_t lane = lane0 + ll;
        const bool live = lane < a.nlanes;
        char *lds = smem + (size_t)ll * LANE_LDS * 2 * sizeof(T);
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
        // ---- PRE in the first pass's register pattern ----
        constexpr int R0 = RL::at(0), NB0 = FFT::nbfly(0), NBF0 = FFT::slots(0);
        constexpr int RLAST = RL::at(RL::NP - 1), NBL = FFT::nbfly(RL::NP - 1), NBFL = FFT::slots(RL::NP - 1);
        cpx<T> v[E];
#pragma unroll
        for (int q = 0; q < NBF0; ++q)
            if (FFT::full(0) || t + q * TPL < NB0) {
#pragma unroll
                for (int r = 0; r < R0; ++r) v[q * R0 + r] = pre(a, (const void *)lds, t + q * TPL + r * NB0);
            }
        // (the first exchange inside passes() starts with a barrier, so the raw lane is dead by then; a single-pass FFT writes nothing)
        FFT::template passes<0>(v, a.twp, lds, t);
        // ---- Z in natural order ----
        __syncthreads();
        {
            cpx<T> *z = (cpx<T> *)lds;
#pragma unroll
            for (int q = 0; q < NBFL; ++q)
                if (FFT::full(RL::NP - 1) || t + q * TPL < NBL) {
#pragma unroll
                    for (int r = 0; r < RLAST; ++r) z[ZiPhi::map(t + q * TPL + r * NBL)] = v[q * RLAST + r];
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
