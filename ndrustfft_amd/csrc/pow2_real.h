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

namespace ndfft {

template <typename T> struct RealArgs {
    const void *in; void *out;
    int64_t nlanes, pitch_in, pitch_out;   // pitches in elements of the in / out element type
    int32_t n, F, n_in, n_out;
    T scale;
    const cpx<T> *aux1, *aux2, *twp;
};

struct ZiNone { static __device__ __forceinline__ int map(int p) { return p; } };
struct ZiPhi { static __device__ __forceinline__ int map(int p) { return p + (p >> 4); } };

template <typename T, int F, int TPL, int LPB, typename RL, int OP> struct RealPow2Kernel {
    static constexpr int E = F / TPL;
    static constexpr int THREADS = TPL * LPB;
    static constexpr int LANE_LDS = F + (F >> 4) + 2;   // complex elements: padded Z, or F+1 raw complex (C2R)
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * 2 * sizeof(T);
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN;
    using FFT = Pow2Kernel<T, F, TPL, LPB, false, RL, 0, 1, 0>;

    static __device__ __forceinline__ void run(const RealArgs<T> &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        const int64_t lane = (int64_t)blockIdx.x * LPB + ll;
        const bool live = lane < a.nlanes;
        const int64_t lsafe = live ? lane : 0;
        char *lds = smem + (size_t)ll * LANE_LDS * 2 * sizeof(T);
        // ---- stage the raw lane ----
        if constexpr (IN_CPLX) {
            const cpx<T> *in = (const cpx<T> *)a.in + lsafe * a.pitch_in;
            cpx<T> *raw = (cpx<T> *)lds;
            for (int j = t; j < a.n_in; j += TPL) raw[j] = in[j];
        } else {
            const T *in = (const T *)a.in + lsafe * a.pitch_in;
            T *raw = (T *)lds;
            for (int j = t; j < a.n_in; j += TPL) raw[j] = in[j];
        }
        __syncthreads();
        // ---- PRE into the first pass's register pattern ----
        cpx<T> v[E];
        {
            constexpr int R0 = RL::at(0), NB0 = F / R0, NBF0 = E / R0;
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const int i = t + q * TPL + r * NB0;
                    if constexpr (OP == G_R2C_EVEN) v[q * R0 + r] = ((const cpx<T> *)lds)[i];   // z[i] = (x[2i], x[2i+1])
                    else v[q * R0 + r] = pre_elem<T, OP, ZiNone>(a, (const void *)lds, i);
                }
        }
        // (the first exchange inside passes() starts with a barrier, so the raw lane is dead by then)
        FFT::template passes<0>(v, a.twp, lds, t);
        // ---- Z in natural order ----
        __syncthreads();
        {
            constexpr int RL_ = RL::at(RL::NP - 1), NBL = F / RL_, NBFL = E / RL_;
            cpx<T> *z = (cpx<T> *)lds;
#pragma unroll
            for (int q = 0; q < NBFL; ++q)
#pragma unroll
                for (int r = 0; r < RL_; ++r) z[ZiPhi::map(t + q * TPL + r * NBL)] = v[q * RL_ + r];
        }
        __syncthreads();
        if (!live) return;
        // ---- POST gather + store ----
        const cpx<T> *res = (const cpx<T> *)lds;
        if constexpr (OUT_CPLX) {
            cpx<T> *out = (cpx<T> *)a.out + lane * a.pitch_out;
            for (int q = t; q < a.n_out; q += TPL) gstore<T, true>(out + q, post_cplx<T, OP, ZiPhi>(a, res, q));
        } else {
            T *out = (T *)a.out + lane * a.pitch_out;
            for (int q = t; q < a.n_out; q += TPL) __builtin_nontemporal_store(post_real<T, OP, ZiPhi>(a, res, q), out + q);
        }
    }
};

template <typename K, typename T> __global__ __launch_bounds__(K::THREADS) void k_pow2_real(const RealArgs<T> a) { K::run(a); }

}  // namespace ndfft
