// kernels_pow2.hip -- the north-star kernel: batched C2C FFT along the contiguous axis,
// power-of-two n, ONE read and ONE write of HBM per point (BASELINE cfg2 / cfg5 / cfg3-B).
//
// Replaces, per lane, FftHandler::fft_lane / ifft_lane (src/lib.rs:313-331) together with the
// strategy-(i) row loop that calls it (src/lib.rs:117-124, par 187-194).
//
// Design (CDNA4):
//  * A lane lives in REGISTERS: TPL threads x E = n/TPL complex elements each.  The first radix
//    pass reads global memory directly in the Stockham input pattern x[j + r n/R] -- consecutive
//    threads, consecutive 16-byte elements, 1 KiB per wave instruction -- and the last pass writes
//    y[j + r n/R] the same way, so there is no staging copy on either side.
//  * Between passes the lane is exchanged through LDS at its natural Stockham index, padded by one
//    element every 16 (phi(p) = p + p/16) so the stride-R writes of early passes and the unit
//    stride reads of the next pass are both bank-conflict free.
//  * HALF exchange: real parts, then imaginary parts, through ONE real-sized buffer.  n=4096 f64
//    needs 34 KiB instead of 68 KiB, i.e. 4 workgroups per CU instead of 2 -- the occupancy that
//    keeps enough HBM requests in flight while other workgroups are in their butterfly phase.
//  * Twiddles come from per-pass tables laid out [r-1][k] so that a wave reads them coalesced;
//    they are built on the host in long double (plan.hip) and stay L2-resident (<= 64 KiB).
//  * No MFMA: ~1.9 flop/byte, the kernel is HBM-bound by design.
#include "butterflies.h"
#include "engine.h"

namespace ndfft {

template <int... Rs> struct RadixList {
    static constexpr int NP = sizeof...(Rs);
    static constexpr int at(int i) { constexpr int r[NP] = {Rs...}; return r[i]; }
    // Ns before pass i
    static constexpr int ns(int i) { int s = 1; for (int p = 0; p < i; ++p) s *= at(p); return s; }
    // offset (in complex elements) of pass i's twiddle block inside twp; pass 0 has none
    static constexpr int twoff(int i) { int o = 0; for (int p = 1; p < i; ++p) o += (at(p) - 1) * ns(p); return o; }
};

__device__ __forceinline__ int phi(int p) { return p + (p >> 4); }

template <typename T, int N, int TPL, int LPB, bool HALF, typename RL> struct Pow2Kernel {
    static constexpr int E = N / TPL;
    static constexpr int THREADS = TPL * LPB;
    static constexpr int LANE_LDS = N + (N >> 4) + 1;                      // padded elements per lane
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * (HALF ? sizeof(T) : 2 * sizeof(T));

    template <int P>
    static __device__ __forceinline__ void passes(cpx<T> (&v)[E], const cpx<T> *__restrict__ twp, char *lds, int t) {
        constexpr int R = RL::at(P), Ns = RL::ns(P), NBF = E / R;
        if constexpr (P > 0) {
            const cpx<T> *tw = twp + RL::twoff(P);
#pragma unroll
            for (int q = 0; q < NBF; ++q) {
                const int k = (t + q * TPL) & (Ns - 1);
#pragma unroll
                for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], tw[(r - 1) * Ns + k]);
            }
        }
#pragma unroll
        for (int q = 0; q < NBF; ++q) Bfly<T, R>::run(&v[q * R]);
        if constexpr (P + 1 < RL::NP) {
            constexpr int R2 = RL::at(P + 1), NB2 = N / R2, NBF2 = E / R2;
            if constexpr (HALF) {
                T *s = (T *)lds;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < NBF; ++q) {
                        const int j = t + q * TPL, k = j & (Ns - 1), o = (j - k) * R + k;
#pragma unroll
                        for (int r = 0; r < R; ++r) s[phi(o + r * Ns)] = half ? v[q * R + r].y : v[q * R + r].x;
                    }
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < NBF2; ++q) {
                        const int j = t + q * TPL;
#pragma unroll
                        for (int r = 0; r < R2; ++r) {
                            const T x = s[phi(j + r * NB2)];
                            if (half) v[q * R2 + r].y = x; else v[q * R2 + r].x = x;
                        }
                    }
                }
            } else {
                cpx<T> *s = (cpx<T> *)lds;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBF; ++q) {
                    const int j = t + q * TPL, k = j & (Ns - 1), o = (j - k) * R + k;
#pragma unroll
                    for (int r = 0; r < R; ++r) s[phi(o + r * Ns)] = v[q * R + r];
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBF2; ++q) {
                    const int j = t + q * TPL;
#pragma unroll
                    for (int r = 0; r < R2; ++r) v[q * R2 + r] = s[phi(j + r * NB2)];
                }
            }
            passes<P + 1>(v, twp, lds, t);
        }
    }

    static __device__ __forceinline__ void run(const Pow2Args &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        const int64_t lane = (int64_t)blockIdx.x * LPB + ll;
        const bool live = lane < a.nlanes;
        const cpx<T> *in = (const cpx<T> *)a.in + (live ? lane : 0) * a.pitch_in;
        cpx<T> *out = (cpx<T> *)a.out + (live ? lane : 0) * a.pitch_out;
        char *lds = smem + (size_t)ll * LANE_LDS * (HALF ? sizeof(T) : 2 * sizeof(T));
        cpx<T> v[E];
        {
            constexpr int R0 = RL::at(0), NB0 = N / R0, NBF0 = E / R0;
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
#pragma unroll
                for (int r = 0; r < R0; ++r) v[q * R0 + r] = in[t + q * TPL + r * NB0];
        }
        if (a.inverse) {
#pragma unroll
            for (int i = 0; i < E; ++i) v[i].y = -v[i].y;
        }
        passes<0>(v, (const cpx<T> *)a.twp, lds, t);
        if (!live) return;
        constexpr int RL_ = RL::at(RL::NP - 1), NBL = N / RL_, NBFL = E / RL_;
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int i = 0; i < E; ++i) { v[i].x *= sc; v[i].y *= -sc; }   // conj + norm_default (lib.rs:333-338)
        }
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
#pragma unroll
            for (int r = 0; r < RL_; ++r) out[t + q * TPL + r * NBL] = v[q * RL_ + r];
    }
};

template <typename K> __global__ __launch_bounds__(K::THREADS) void k_pow2(const Pow2Args a) { K::run(a); }

// N, threads-per-lane, radices.  E = N / TPL must be a multiple of every radix.
#define NDFFT_POW2_CONFIGS(X) \
    X(64, 8, 8, 8)            \
    X(128, 8, 16, 8)          \
    X(256, 16, 16, 16)        \
    X(512, 64, 8, 8, 8)       \
    X(1024, 64, 16, 8, 8)     \
    X(2048, 128, 16, 16, 8)   \
    X(4096, 256, 16, 16, 16)  \
    X(8192, 512, 16, 16, 8, 4) \
    X(16384, 1024, 16, 16, 16, 4)

static constexpr int lpb_for(int tpl) { return tpl >= 256 ? 1 : 256 / tpl; }

template <int N> struct Pow2Cfg;
#define NDFFT_DEF_CFG(N_, TPL_, ...)                \
    template <> struct Pow2Cfg<N_> {                \
        static constexpr int TPL = TPL_;            \
        using RL = RadixList<__VA_ARGS__>;          \
    };
NDFFT_POW2_CONFIGS(NDFFT_DEF_CFG)

bool pow2_supported(int dtype, int n) {
    (void)dtype;
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...) case N_: return true;
        NDFFT_POW2_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return false;
    }
}

template <typename RL> static void build_tw(int n, HostTable &out) {
    const long double kPiL = 3.14159265358979323846264338327950288L;
    for (int p = 1; p < RL::NP; ++p) {
        const int R = RL::at(p), Ns = RL::ns(p);
        // tw[(r-1)*Ns + k] = e^{-2 pi i r k/(Ns R)}
        for (int r = 1; r < R; ++r)
            for (int k = 0; k < Ns; ++k) {
                const unsigned long long num = ((unsigned long long)r * k) % ((unsigned long long)Ns * R);
                const long double ang = 2.0L * kPiL * (long double)num / (long double)((unsigned long long)Ns * R);
                out.re.push_back(cosl(ang)); out.im.push_back(-sinl(ang));
            }
    }
    (void)n;
}

void pow2_build_twiddles(int dtype, int n, HostTable &out) {
    (void)dtype;
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...) case N_: build_tw<Pow2Cfg<N_>::RL>(n, out); break;
        NDFFT_POW2_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: break;
    }
}

template <typename T, int N> static int launch_one(const Pow2Args &a, hipStream_t s) {
    constexpr int TPL = Pow2Cfg<N>::TPL, LPB = lpb_for(TPL);
    using K = Pow2Kernel<T, N, TPL, LPB, true, typename Pow2Cfg<N>::RL>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_pow2<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int64_t nblk = (a.nlanes + LPB - 1) / LPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL(k_pow2<K>, dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

int launch_pow2(int dtype, int n, const Pow2Args &a, hipStream_t s) {
    switch (n) {
#define NDFFT_CASE(N_, TPL_, ...) \
    case N_: return dtype == NDFFT_F32 ? launch_one<float, N_>(a, s) : launch_one<double, N_>(a, s);
        NDFFT_POW2_CONFIGS(NDFFT_CASE)
#undef NDFFT_CASE
        default: return fail(NDFFT_ERR_UNSUPPORTED, "pow2 kernel: unsupported n");
    }
}

}  // namespace ndfft
