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
#include "realops.h"

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


// ---------------------------------------------------------------------------------------------------------------------
// The REAL-DATA transforms (R2C, C2R, DCT-I..IV; even and odd n) on lanes of 17 .. ~100 points, one thread per lane:
// raw lane -> PRE fold (realops.h, the same formulas the LDS kernels use: Hermitian fold with the reference's pre-scale and
// DC / Nyquist zeroing lib.rs:511-521, Makhoul permutation, DCT-IV pre-twiddle, ...) -> complex FFT of length F = F1 * F2 in
// registers (RegFft2::fft) -> POST gather (real-FFT split, post-twiddles, un-permutation) -> output lane.  Every index is a
// compile-time constant after unrolling, so the raw lane, Z and the outputs live in registers and the op tables (aux1,
// aux2) are read through scalar loads.  Replaces fft_r2c_lane / ifft_r2c_lane (lib.rs:497-523) and dct1..4_lane (lib.rs:688-734).
// OP is the kernel op (device_common.h: GenOp); N the handler length; F1 * F2 = F the inner complex FFT length.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct RegOpArgs { int n, F; T scale; const cpx<T> *aux1, *aux2; };
struct RegZi { static __device__ __forceinline__ int map(int p) { return p; } };

template <typename T, int OP, int N, int F1, int F2, int LANES, bool STAGE> struct RegReal {
    static constexpr int F = F1 * F2;
    static constexpr bool IN_CPLX = OP == G_C2R_EVEN || OP == G_C2R_ODD;
    static constexpr bool OUT_CPLX = OP == G_R2C_EVEN || OP == G_R2C_ODD;
    static constexpr int M = N / 2 + 1;
    static constexpr int NI = IN_CPLX ? 2 * M : N, NO = OUT_CPLX ? 2 * M : N;   // reals per input / output lane
    static constexpr int THREADS = LANES;
    static constexpr int PI_ = NI | 1, PO_ = NO | 1, PMAX = PI_ > PO_ ? PI_ : PO_;
    static constexpr size_t LDS_BYTES = STAGE ? (size_t)LANES * PMAX * sizeof(T) : 0;
    using FFT = RegFft2<T, F1, F2, LANES, false>;

    static __device__ __forceinline__ void run(const RegRealArgs &ra) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const TinyArgs &a = ra.t;
        const int64_t L = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const bool live = L < a.nlanes;
        T raw[NI + 2];                                      // (+2: the odd-n C2R fold may index one complex past the half spectrum)
        raw[NI] = (T)0; raw[NI + 1] = (T)0;
        if constexpr (STAGE) {
            T *lds = (T *)smem;
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * NI, total = a.nlanes * NI;
            const T *in = (const T *)a.in + c0;
            const int last = (int)std_min64(total - c0, (int64_t)THREADS * NI) - 1;
#ifdef NDFFT_REPRO_MASKED_TAIL
            // REPRODUCER ONLY (tools/repro_masked_tail.py, NDFFT_REPRO_MASKED_TAIL=1): the tail workgroup PREDICATES its loads, the form this
            // kernel was first written in.  On the MI355X (ROCm 7.2, hiprtc) every lane of the last workgroup came out wrong for n = 40 and 48
            // in f64 -- the raw lane lives partly in AGPRs there (256 VGPRs + AGPR spill) and the exec-masked loads into those registers lost
            // their values.  The product clamps the addresses instead (below): every thread executes every load.
#pragma unroll
            for (int k = 0; k < NI; ++k) { const int g = threadIdx.x + k * THREADS; raw[k] = (T)0; if (g <= last) raw[k] = in[g]; }
#else
#pragma unroll
            for (int k = 0; k < NI; ++k) { const int g = threadIdx.x + k * THREADS; raw[k] = in[g < last ? g : last]; }
#endif
#pragma unroll
            for (int k = 0; k < NI; ++k) { const int g = threadIdx.x + k * THREADS; lds[(g / NI) * PI_ + g % NI] = raw[k]; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NI; ++j) raw[j] = lds[threadIdx.x * PI_ + j];
            __syncthreads();
        } else {
            const int64_t Ls = live ? L : 0;
            const int64_t base = (Ls / a.inner) * a.outer_in + (Ls % a.inner) * a.lane_in;
            if constexpr (IN_CPLX) {
                const cpx<T> *in = (const cpx<T> *)a.in + base;
#pragma unroll
                for (int j = 0; j < M; ++j) { const cpx<T> c = in[(int64_t)j * a.elem_in]; raw[2 * j] = c.x; raw[2 * j + 1] = c.y; }
            } else {
                const T *in = (const T *)a.in + base;
#pragma unroll
                for (int j = 0; j < N; ++j) raw[j] = in[(int64_t)j * a.elem_in];
            }
        }
        RegOpArgs<T> oa;
        oa.n = N; oa.F = F; oa.scale = (T)a.scale; oa.aux1 = (const cpx<T> *)ra.aux1; oa.aux2 = (const cpx<T> *)ra.aux2;
        // ---- PRE ----
        cpx<T> z[F];
#pragma unroll
        for (int i = 0; i < F; ++i) {
            if constexpr (OP == G_R2C_EVEN) z[i] = mk<T>(raw[2 * i], raw[2 * i + 1]);
            else if constexpr (OP == G_R2C_ODD) z[i] = mk<T>(raw[i], (T)0);
            else z[i] = pre_elem<T, OP, RegZi>(oa, (const void *)raw, i);
        }
        FFT::fft(z, (const cpx<T> *)a.mat);
        cpx<T> nat[F + 1];                                 // natural order (a renaming); one spare slot for folds that peek at F
#pragma unroll
        for (int s = 0; s < F; ++s) nat[FFT::out_index(s)] = z[s];
        nat[F] = nat[0];
        // ---- POST ----
        T y[NO];
        if constexpr (OUT_CPLX) {
#pragma unroll
            for (int q = 0; q < M; ++q) { const cpx<T> c = post_cplx<T, OP, RegZi>(oa, nat, q); y[2 * q] = c.x; y[2 * q + 1] = c.y; }
        } else {
#pragma unroll
            for (int q = 0; q < N; ++q) y[q] = post_real<T, OP, RegZi>(oa, nat, q);
        }
        if constexpr (STAGE) {
            T *lds = (T *)smem;
#pragma unroll
            for (int o = 0; o < NO; ++o) lds[threadIdx.x * PO_ + o] = y[o];
            __syncthreads();
            const int64_t c0 = (int64_t)blockIdx.x * THREADS * NO, total = a.nlanes * NO;
            T *out = (T *)a.out + c0;
#pragma unroll
            for (int k = 0; k < NO; ++k) { const int g = threadIdx.x + k * THREADS; y[k] = lds[(g / NO) * PO_ + g % NO]; }
#pragma unroll
            for (int k = 0; k < NO; ++k) { const int g = threadIdx.x + k * THREADS; if (c0 + g < total) __builtin_nontemporal_store(y[k], out + g); }
        } else {
            if (!live) return;
            const int64_t base = (L / a.inner) * a.outer_out + (L % a.inner) * a.lane_out;
            if constexpr (OUT_CPLX) {
                cpx<T> *out = (cpx<T> *)a.out + base;
#pragma unroll
                for (int o = 0; o < M; ++o) gstore<T, true>(out + (int64_t)o * a.elem_out, mk<T>(y[2 * o], y[2 * o + 1]));
            } else {
                T *out = (T *)a.out + base;
#pragma unroll
                for (int o = 0; o < N; ++o) __builtin_nontemporal_store(y[o], out + (int64_t)o * a.elem_out);
            }
        }
    }
};

}  // namespace ndfft
