// reg_kernel.h -- lanes of 14 .. ~100 points, ONE THREAD PER LANE, the whole transform in that thread's registers:
// n = N1 * N2 with both factors butterflies of butterflies.h (2..13, 16, and the primes 17..31; N2 = 1 for a prime n),
// Cooley-Tukey inside the thread --
//   A[k1][n2] = sum_{n1} x[n1 N2 + n2] W_N1^(n1 k1)    (N2 radix-N1 butterflies)
//   A[k1][n2] *= W_n^(n2 k1)                            (constants: every index is known after unrolling, the table is
//                                                        read through scalar loads)
//   X[k1 + N1 k2] = sum_{n2} A[k1][n2] W_N2^(n2 k2)    (N1 radix-N2 butterflies)
// -- no LDS exchange, no barrier between passes, no shuffles.  Specialised per (T, N1, N2, layout) with hiprtc at first use
// (jit.hip), like the other plan-time kernels; lanes this short ran at 9-35 % of the roofline on the general
// register kernel (4-8 threads per lane: 64-byte global accesses and an LDS exchange per pass).
// Replaces FftHandler::fft_lane / ifft_lane (src/lib.rs:313-331) and the lane loop around them.
// Layouts as tiny_kernel.h: dense rows are staged through LDS as one contiguous chunk per workgroup (coalesced global
// accesses), a strided axis with adjacent lanes contiguous is coalesced as it is, anything else uses direct accesses.
#pragma once
#include "pow2_kernel.h"

namespace ndfft {

__device__ __forceinline__ int64_t std_min64(int64_t a, int64_t b) { return a < b ? a : b; }

template <typename T, int N1, int N2, int LANES, bool STAGE> struct RegFft2 {
    static constexpr int N = N1 * N2;
    static constexpr int THREADS = LANES;                            // one thread per lane
    static constexpr int P = N | 1;                                  // padded lane pitch in LDS
    static constexpr size_t LDS_BYTES = STAGE ? (size_t)LANES * P * sizeof(cpx<T>) : 0;

    // v[n1 * N2 + n2] in, X[k1 + N1 k2] left in slot k1 * N2 + k2
    static __device__ __forceinline__ void fft(cpx<T> (&v)[N], const cpx<T> *__restrict__ tw) {
        if constexpr (N2 == 1) {
            Bfly<T, N1>::run(v);                                         // a prime length: the lane is one butterfly
        } else {
#pragma unroll
            for (int n2 = 0; n2 < N2; ++n2) {
                cpx<T> a[N1];
#pragma unroll
                for (int n1 = 0; n1 < N1; ++n1) a[n1] = v[n1 * N2 + n2];
                Bfly<T, N1>::run(a);
#pragma unroll
                for (int k1 = 0; k1 < N1; ++k1) {
                    if ((n2 * k1) % N != 0) a[k1] = cmul(a[k1], tw[(n2 * k1) % N]);
                    v[k1 * N2 + n2] = a[k1];
                }
            }
#pragma unroll
            for (int k1 = 0; k1 < N1; ++k1) {
                cpx<T> b[N2];
#pragma unroll
                for (int n2 = 0; n2 < N2; ++n2) b[n2] = v[k1 * N2 + n2];
                Bfly<T, N2>::run(b);
#pragma unroll
                for (int k2 = 0; k2 < N2; ++k2) v[k1 * N2 + k2] = b[k2];
            }
        }
    }
    // output index held by register slot s = k1 * N2 + k2
    static constexpr int out_index(int s) { return s / N2 + N1 * (s % N2); }

    static __device__ __forceinline__ void run(const TinyArgs &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int64_t L = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const bool live = L < a.nlanes;
        const cpx<T> *__restrict__ tw = (const cpx<T> *)a.mat;       // W_n^k, k < n
        cpx<T> v[N];
        if constexpr (STAGE) {
            cpx<T> *lds = (cpx<T> *)smem;
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * N, total = a.nlanes * N;
            const cpx<T> *in = (const cpx<T> *)a.in + c0;
            // all N coalesced loads of this thread in flight before the first LDS store (v is the landing zone): a loop that
            // stores each element as it arrives keeps 4 x 16 bytes per thread in flight -- 12 KiB per CU, a third of what HBM needs
            // (the tail workgroup clamps its addresses instead of predicating: every thread executes every load and every LDS
            //  store -- positions of lanes that do not exist receive copies of the last element and are never stored)
            const int last = (int)std_min64(total - c0, (int64_t)THREADS * N) - 1;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                v[k] = in[g < last ? g : last];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                lds[(g / N) * P + g % N] = v[k];
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
        fft(v, tw);
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int j = 0; j < N; ++j) { v[j].x *= sc; v[j].y *= -sc; }   // conj + norm_default (lib.rs:333-338)
        }
        if constexpr (STAGE) {
            cpx<T> *lds = (cpx<T> *)smem;
#pragma unroll
            for (int s = 0; s < N; ++s) lds[threadIdx.x * P + out_index(s)] = v[s];   // own region only
            __syncthreads();
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * N, total = a.nlanes * N;
            cpx<T> *out = (cpx<T> *)a.out + c0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                v[k] = lds[(g / N) * P + g % N];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int g = threadIdx.x + k * THREADS;
                if (c0 + g < total) gstore<T, true>(out + g, v[k]);
            }
        } else {
            if (!live) return;
            cpx<T> *out = (cpx<T> *)a.out + (L / a.inner) * a.outer_out + (L % a.inner) * a.lane_out;
#pragma unroll
            for (int s = 0; s < N; ++s) gstore<T, true>(out + (int64_t)out_index(s) * a.elem_out, v[s]);
        }
    }
};

}  // namespace ndfft
