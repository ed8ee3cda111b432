// tinymat_kernel.h -- the REAL-DATA transforms (R2C, C2R, DCT-I..IV) on lanes of 2..16 points, one thread per lane:
// a lane is at most 18 reals, so the transform is applied as its definition -- a dense real matrix of at most 18 x 16
// (built in long double by the plan, read through scalar loads: every thread uses the same entries) times the lane
// vector held in registers.  n^2 multiply-adds per lane cost less than the memory traffic of the lane at these sizes,
// and every op, parity of n and normalisation point of the reference is ONE formula (plan.hip: build_tiny_mats):
//   R2C  fft_r2c_lane  lib.rs:497-503   C2R  ifft_r2c_lane lib.rs:506-523 (DC / Nyquist imaginary parts have zero columns)
//   DCT  dct1..4_lane  lib.rs:688-734   (the x2 / x1 / custom pre-scale is the scalar `scale`)
// Layouts as tiny_kernel.h: strided axis with adjacent lanes contiguous -> coalesced as it is; dense rows -> staged
// through LDS; anything else -> direct strided accesses.
#pragma once
#include "pow2_kernel.h"

namespace ndfft {

// NI / NO: reals per input / output lane (a complex lane of m elements counts 2 m)
template <typename T, int NI, int NO, bool IN_CPLX, bool OUT_CPLX, bool STAGE> struct TinyMat {
    static constexpr int THREADS = 256;
    static constexpr int PI_ = NI | 1, PO_ = NO | 1, PMAX = PI_ > PO_ ? PI_ : PO_;
    static constexpr size_t LDS_BYTES = STAGE ? (size_t)THREADS * PMAX * sizeof(T) : 0;

    static __device__ __forceinline__ void run(const TinyArgs &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int64_t L = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const bool live = L < a.nlanes;
        const T *__restrict__ M = (const T *)a.mat;
        T x[NI], y[NO];
        if constexpr (STAGE) {
            T *lds = (T *)smem;
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * NI, total = a.nlanes * NI;
            const T *in = (const T *)a.in + c0;
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int g = threadIdx.x + k * THREADS;
                if (c0 + g < total) lds[(g / NI) * PI_ + g % NI] = in[g];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NI; ++j) x[j] = lds[threadIdx.x * PI_ + j];
            __syncthreads();                                   // the output image reuses the buffer with another pitch
        } else {
            const int64_t Ls = live ? L : 0;
            const int64_t base = (Ls / a.inner) * a.outer_in + (Ls % a.inner) * a.lane_in;
            if constexpr (IN_CPLX) {
                const cpx<T> *in = (const cpx<T> *)a.in + base;
#pragma unroll
                for (int j = 0; j < NI / 2; ++j) { const cpx<T> c = in[(int64_t)j * a.elem_in]; x[2 * j] = c.x; x[2 * j + 1] = c.y; }
            } else {
                const T *in = (const T *)a.in + base;
#pragma unroll
                for (int j = 0; j < NI; ++j) x[j] = in[(int64_t)j * a.elem_in];
            }
        }
        const T sc = (T)a.scale;
#pragma unroll
        for (int j = 0; j < NI; ++j) x[j] *= sc;               // the reference scales the INPUT lane (lib.rs:511-515, 692-696)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            T acc = 0;
#pragma unroll
            for (int j = 0; j < NI; ++j) acc += M[o * NI + j] * x[j];
            y[o] = acc;
        }
        if constexpr (STAGE) {
            T *lds = (T *)smem;
#pragma unroll
            for (int o = 0; o < NO; ++o) lds[threadIdx.x * PO_ + o] = y[o];
            __syncthreads();
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * NO, total = a.nlanes * NO;
            T *out = (T *)a.out + c0;
#pragma unroll
            for (int k = 0; k < NO; ++k) {
                const int g = threadIdx.x + k * THREADS;
                if (c0 + g < total) __builtin_nontemporal_store(lds[(g / NO) * PO_ + g % NO], out + g);
            }
        } else {
            if (!live) return;
            const int64_t base = (L / a.inner) * a.outer_out + (L % a.inner) * a.lane_out;
            if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + base;
#pragma unroll
                for (int o = 0; o < NO / 2; ++o) gstore<T, true>(out + (int64_t)o * a.elem_out, mk<T>(y[2 * o], y[2 * o + 1]));
            } else {
                T *out = (T *)a.out + base;
#pragma unroll
                for (int o = 0; o < NO; ++o) __builtin_nontemporal_store(y[o], out + (int64_t)o * a.elem_out);
            }
        }
    }
};

template <typename K> __global__ __launch_bounds__(K::THREADS) void k_tinymat(const TinyArgs a) { K::run(a); }

}  // namespace ndfft
