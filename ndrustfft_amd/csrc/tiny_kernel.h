// tiny_kernel.h -- C2C lanes of 2..13 and 16 points, ONE THREAD PER LANE: the whole lane lives in the thread's
// registers and is transformed by one radix-n butterfly (butterflies.h), no LDS exchange, no shuffles.
// Replaces FftHandler::fft_lane / ifft_lane (src/lib.rs:313-331) and the lane loop around them for lane lengths the
// reference's own tests use (n = 6, src/lib.rs:903-1406) and for the short axes of 3-D arrays.
//   * strided axis, adjacent lanes contiguous (strategy ii, src/lib.rs:125-137): adjacent threads own adjacent lanes, so
//     every load / store instruction of a wave is one contiguous run -- the gather / scatter of the reference disappears;
//   * dense contiguous lanes (strategy i): the workgroup's 256 lanes are one contiguous chunk; it is staged through LDS
//     (coalesced global accesses, padded lane pitch) so that global memory never sees the stride-n pattern;
//   * any other layout: direct strided accesses (correct, not tuned).
#pragma once
#include "pow2_kernel.h"   // gstore, butterflies

namespace ndfft {

template <typename T, int N, bool STAGE> struct TinyFft {
    static constexpr int THREADS = 256;
    static constexpr int P = N | 1;                                    // padded lane pitch in LDS
    static constexpr size_t LDS_BYTES = STAGE ? (size_t)THREADS * P * sizeof(cpx<T>) : 0;

    static __device__ __forceinline__ void run(const TinyArgs &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int64_t L = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const bool live = L < a.nlanes;
        cpx<T> v[N];
        if constexpr (STAGE) {
            // dense lanes: the workgroup's chunk [blockIdx * 256 * N, +256 * N) is contiguous
            cpx<T> *lds = (cpx<T> *)smem;
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * N, total = a.nlanes * N;
            const cpx<T> *in = (const cpx<T> *)a.in + c0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                if (c0 + g < total) lds[(g / N) * P + g % N] = in[g];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = lds[threadIdx.x * P + j];
        } else {
            const int64_t Ls = live ? L : 0;
            const cpx<T> *in = (const cpx<T> *)a.in + (Ls / a.inner) * a.outer_in + (Ls % a.inner) * a.lane_in;
#pragma unroll
            for (int j = 0; j < N; ++j) v[j] = in[(int64_t)j * a.elem_in];
        }
        if (a.inverse) {
#pragma unroll
            for (int j = 0; j < N; ++j) v[j].y = -v[j].y;
        }
        Bfly<T, N>::run(v);
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int j = 0; j < N; ++j) { v[j].x *= sc; v[j].y *= -sc; }   // conj + norm_default (lib.rs:333-338)
        }
        if constexpr (STAGE) {
            cpx<T> *lds = (cpx<T> *)smem;
#pragma unroll
            for (int j = 0; j < N; ++j) lds[threadIdx.x * P + j] = v[j];       // own region only: no barrier needed before
            __syncthreads();
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * N, total = a.nlanes * N;
            cpx<T> *out = (cpx<T> *)a.out + c0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                if (c0 + g < total) gstore<T, true>(out + g, lds[(g / N) * P + g % N]);
            }
        } else {
            if (!live) return;
            cpx<T> *out = (cpx<T> *)a.out + (L / a.inner) * a.outer_out + (L % a.inner) * a.lane_out;
#pragma unroll
            for (int j = 0; j < N; ++j) gstore<T, true>(out + (int64_t)j * a.elem_out, v[j]);
        }
    }
};

template <typename K> __global__ __launch_bounds__(K::THREADS) void k_tiny(const TinyArgs a) { K::run(a); }

}  // namespace ndfft
