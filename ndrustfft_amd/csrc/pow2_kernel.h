// pow2_kernel.h -- register-resident Stockham kernel template (see kernels_pow2.hip for the design notes).
#pragma once
#include "butterflies.h"
#ifndef __HIPCC_RTC__
#include "engine.h"
#include <cmath>
#endif

namespace ndfft {

template <int... Rs> struct RadixList {
    static constexpr int NP = sizeof...(Rs);
    static constexpr int at(int i) { constexpr int r[NP] = {Rs...}; return r[i]; }
    // Ns before pass i
    static constexpr int ns(int i) { int s = 1; for (int p = 0; p < i; ++p) s *= at(p); return s; }
    // offset (in complex elements) of pass i's twiddle block inside twp; pass 0 has none
    static constexpr int twoff(int i) { int o = 0; for (int p = 1; p < i; ++p) o += (at(p) - 1) * ns(p); return o; }
};

// LDS padding of the exchange: one element per 2^NDFFT_PHI_SHIFT (16).  One per 8 makes the stride-8 writes of the
// first passes conflict-free but measured no faster, so 16 stays the default.
#ifndef NDFFT_PHI_SHIFT
#define NDFFT_PHI_SHIFT 4
#endif
__device__ __forceinline__ int phi(int p) { return p + (p >> NDFFT_PHI_SHIFT); }
// phi(base + r * STRIDE): when STRIDE is a multiple of 16 the padding term splits, phi(base) + r * (STRIDE + STRIDE/16),
// i.e. a compile-time offset per r that folds into the LDS instruction's immediate
template <int STRIDE> __device__ __forceinline__ int phi_at(int base, int pbase, int r) {
    if constexpr (STRIDE % (1 << NDFFT_PHI_SHIFT) == 0) return pbase + r * (STRIDE + (STRIDE >> NDFFT_PHI_SHIFT));
    else return phi(base + r * STRIDE);
}
// j mod Ns for a compile-time Ns (a mask when Ns is a power of two, a multiply-shift otherwise)
template <int Ns> __device__ __forceinline__ int kmod(int j) {
    if constexpr ((Ns & (Ns - 1)) == 0) return j & (Ns - 1);
    else return j % Ns;
}

// Streaming (touch-once) global accesses.  A lane is read once and written once per call, so the
// output is stored non-temporally: it then neither evicts the still-to-be-read input from the
// 256 MiB Infinity Cache nor the twiddle tables from L2 (measured on MI355X: a 2 x 256 MiB copy
// with this access pattern runs 6.9 TB/s with nt stores vs 5.2 TB/s without, tools/membench.hip).
template <typename T> struct vec_of;
template <> struct vec_of<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct vec_of<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T, bool NT> __device__ __forceinline__ cpx<T> gload(const cpx<T> *p) {
    if constexpr (NT) {
        typename vec_of<T>::type v = __builtin_nontemporal_load((const typename vec_of<T>::type *)p);
        return mk<T>(v.x, v.y);
    } else {
        return *p;
    }
}
typedef float vec4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ vec4f gload4(const float *p) {
    if constexpr (NT) return __builtin_nontemporal_load((const vec4f *)p);
    else return *(const vec4f *)p;
}
template <bool NT> __device__ __forceinline__ void gstore4(float *p, vec4f v) {
    if constexpr (NT) __builtin_nontemporal_store(v, (vec4f *)p);
    else *(vec4f *)p = v;
}
template <typename T, bool NT> __device__ __forceinline__ void gstore(cpx<T> *p, cpx<T> v) {
    if constexpr (NT) {
        typename vec_of<T>::type w; w.x = v.x; w.y = v.y;
        __builtin_nontemporal_store(w, (typename vec_of<T>::type *)p);
    } else {
        *p = v;
    }
}

// FLAGS is for ablation builds in tools/kbench.hip only (product kernels use 0):
//   1 = skip twiddle multiplies, 2 = skip the LDS exchange, 4 = skip butterflies
// product flags: 8 = fused four-step twiddle on load (Pow2Args::twlo), 16 = derive twiddle powers for big late-pass tables,
//   32 = LATENCY form (round 5): the twiddles of pass p + 1 are loaded BEFORE the exchange that follows pass p and the exchange's barriers wait for the LDS only
//        (raw s_barrier behind s_waitcnt lgkmcnt(0); __syncthreads() would also drain the table loads), so the table's L2 round trip overlaps the exchange instead of
//        following it.  Costs up to E - 1 complex registers across the exchange: for calls of few tiles, where one tile's latency is the call's (pow2_real.h: small grids)
// NT: bit 0 = non-temporal stores, bit 1 = non-temporal loads
// VEC = 2 (f32 only): global loads/stores move TWO adjacent complex elements (16 B) per lane; the
//   first and last pass then own adjacent butterfly pairs j = 2t, 2t+1 instead of j = t, t+TPL.
//   Needs E/R >= 2 in those passes, even lane pitches and 16-byte aligned bases (checked on the host).
// PSPLIT = 2: every exchange runs in TWO position rounds through a buffer of N / 2 elements -- round A moves the elements whose Stockham
//   position is below N / 2 (written by the butterflies j < N / (2 R), read as r < R' / 2 by every thread), round B the rest.  Twice the
//   barriers, half the LDS: n = 16384 then takes 70 KiB per lane instead of 139 KiB, i.e. TWO workgroups per CU, so that one lane's load /
//   store phases overlap the other's exchanges (round 3; needs whole butterfly rounds and power-of-two sizes).
// barrier of the exchange: LDS traffic only (see FLAGS bit 5)
template <bool RAW> __device__ __forceinline__ void xbar() {
#ifdef NDFFT_LDS_BARRIER_OVERRIDE          // (a build for another target supplies its own LDS-only barrier)
    if constexpr (RAW) NDFFT_LDS_BARRIER_OVERRIDE();
#else
    if constexpr (RAW) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    else __syncthreads();
}
template <typename T, int N, int TPL, int LPB, bool HALF, typename RL, int FLAGS = 0, int MINW = 1, int NT = 1, int VEC = 1, int PSPLIT = 1> struct Pow2Kernel {
    static constexpr bool PREF = (FLAGS & 32) != 0;
    static constexpr int MIN_WAVES = MINW;
    // Pass p has N / R_p butterflies, dealt to the TPL threads of the lane in SLOTS(p) rounds: j = t + q TPL.
    // When TPL does not divide N / R_p the last round is PARTIAL (threads with j >= N / R_p idle): that is what
    // lets radix lists like 840 = 8.7.5.3 run here although no E is a multiple of every radix.
    static constexpr int nbfly(int p) { return N / RL::at(p); }
    static constexpr int slots(int p) { return (nbfly(p) + TPL - 1) / TPL; }
    static constexpr bool full(int p) { return nbfly(p) % TPL == 0; }
    static constexpr int calc_e() { int e = 0; for (int p = 0; p < RL::NP; ++p) { const int x = slots(p) * RL::at(p); if (x > e) e = x; } return e; }
    static constexpr int E = calc_e();                                     // = N / TPL when every pass is full
    static constexpr int THREADS = TPL * LPB;
    static_assert(PSPLIT == 1 || PSPLIT == 2, "PSPLIT is 1 or 2");
    static constexpr int LANE_LDS = N / PSPLIT + ((N / PSPLIT) >> NDFFT_PHI_SHIFT) + 1;   // padded elements per lane
    static constexpr size_t LDS_BYTES = (size_t)LPB * LANE_LDS * (HALF ? sizeof(T) : 2 * sizeof(T));

    // butterfly index owned by thread t, slot q, in pass P
    template <int P> static __device__ __forceinline__ int jof(int t, int q) {
        if constexpr (VEC == 2 && (P == 0 || P == RL::NP - 1)) return 2 * t + (q & 1) + (q >> 1) * (2 * TPL);
        else return t + q * TPL;
    }

    // twiddles of pass P for this thread, in the order the multiply loop below consumes them (LATENCY form)
    template <int P> static __device__ __forceinline__ void load_tw(cpx<T> (&w)[E], const cpx<T> *__restrict__ twp, int t) {
        constexpr int R = RL::at(P), Ns = RL::ns(P), NBF = slots(P), NB = nbfly(P);
        const cpx<T> *tw = twp + RL::twoff(P);
#pragma unroll
        for (int q = 0; q < NBF; ++q) {
            const int j = jof<P>(t, q), k = kmod<Ns>(j);
            if (full(P) || j < NB) {
#pragma unroll
                for (int r = 1; r < R; ++r) w[q * R + r] = tw[(r - 1) * Ns + k];
            }
        }
    }
    template <int P>
    static __device__ __forceinline__ void passes(cpx<T> (&v)[E], const cpx<T> *__restrict__ twp, char *lds, int t) {
        if constexpr (PREF) { cpx<T> w[E]; passes_w<P>(v, twp, lds, t, w); }
        else { cpx<T> w[1]; passes_w<P>(v, twp, lds, t, w); }
    }
    template <int P, int WN>
    static __device__ __forceinline__ void passes_w(cpx<T> (&v)[E], const cpx<T> *__restrict__ twp, char *lds, int t, cpx<T> (&wpre)[WN]) {
        constexpr int R = RL::at(P), Ns = RL::ns(P), NBF = slots(P), NB = nbfly(P);
        constexpr bool FULL = full(P);
        constexpr bool TW_POWERS = (FLAGS & 16) != 0 && (size_t)(R - 1) * Ns * sizeof(cpx<T>) > 32 * 1024;
        static_assert(!PREF || !TW_POWERS, "the latency form loads plain tables");
        if constexpr (P > 0 && !(FLAGS & 1) && PREF) {
#pragma unroll
            for (int q = 0; q < NBF; ++q) {
                if (FULL || jof<P>(t, q) < NB) {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], wpre[q * R + r]);
                }
            }
        } else
        if constexpr (P > 0 && !(FLAGS & 1)) {
            const cpx<T> *tw = twp + RL::twoff(P);
#pragma unroll
            for (int q = 0; q < NBF; ++q) {
                const int j = jof<P>(t, q), k = kmod<Ns>(j);
                if (FULL || j < NB) {
                    if constexpr (TW_POWERS && (R == 8 || R == 16)) {
                        // big late-pass tables (> 32 KiB) miss L1: load W^k, W^2k, W^4k (, W^8k) only and build the other
                        // powers with at most two (three) complex multiplications each -- 3-4 L2 loads instead of 7-15
                        cpx<T> w[R];
                        w[1] = tw[k]; w[2] = tw[Ns + k]; w[4] = tw[3 * Ns + k];
                        w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
                        if constexpr (R == 16) {
                            w[8] = tw[7 * Ns + k];
#pragma unroll
                            for (int r = 9; r < 16; ++r) w[r] = cmul(w[r - 8], w[8]);
                        }
#pragma unroll
                        for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], w[r]);
                    } else {
#pragma unroll
                        for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], tw[(r - 1) * Ns + k]);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NBF; ++q) if constexpr (!(FLAGS & 4)) { if (FULL || jof<P>(t, q) < NB) Bfly<T, R>::run(&v[q * R]); }
        if constexpr (P + 1 < RL::NP) {
            constexpr int R2 = RL::at(P + 1), NB2 = N / R2, NBF2 = slots(P + 1);
            constexpr bool FULL2 = full(P + 1);
            if constexpr (PREF && !(FLAGS & 1)) load_tw<P + 1>(wpre, twp, t);     // in flight during the exchange below
            if constexpr (FLAGS & 2) {
            } else if constexpr (HALF) {
                T *s = (T *)lds;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    // (PSPLIT = 2: round A's values wait in `keep` -- every thread still owns unwritten round-B values in v)
                    T keep[PSPLIT == 2 ? NBF2 * (R2 / 2) : 1];
#pragma unroll
                    for (int ps = 0; ps < PSPLIT; ++ps) {
                        xbar<PREF>();
#pragma unroll
                        for (int q = 0; q < NBF; ++q) {
                            const int j = jof<P>(t, q), k = kmod<Ns>(j), o = (j - k) * R + k - ps * (N / 2), po = phi(o);
                            if ((FULL || j < NB) && (PSPLIT == 1 || (j < NB / 2) == (ps == 0))) {
#pragma unroll
                                for (int r = 0; r < R; ++r) s[phi_at<Ns>(o, po, r)] = half ? v[q * R + r].y : v[q * R + r].x;
                            }
                        }
                        xbar<PREF>();
#pragma unroll
                        for (int q = 0; q < NBF2; ++q) {
                            const int j = jof<P + 1>(t, q), pj = phi(j);
                            if (FULL2 || j < NB2) {
#pragma unroll
                                for (int r = 0; r < R2 / PSPLIT; ++r) {
                                    const T x = s[phi_at<NB2>(j, pj, r)];     // round B: position j + (r + R2 / 2) NB2 - N / 2 = j + r NB2
                                    if constexpr (PSPLIT == 2) {
                                        if (ps == 0) { keep[q * (R2 / 2) + r] = x; continue; }
                                    }
                                    const int rr = r + ps * (R2 / 2);
                                    if (half) v[q * R2 + rr].y = x; else v[q * R2 + rr].x = x;
                                }
                            }
                        }
                    }
                    if constexpr (PSPLIT == 2) {
#pragma unroll
                        for (int q = 0; q < NBF2; ++q)
#pragma unroll
                            for (int r = 0; r < R2 / 2; ++r) { if (half) v[q * R2 + r].y = keep[q * (R2 / 2) + r]; else v[q * R2 + r].x = keep[q * (R2 / 2) + r]; }
                    }
                }
            } else {
                cpx<T> *s = (cpx<T> *)lds;
                cpx<T> keep[PSPLIT == 2 ? NBF2 * (R2 / 2) : 1];
#pragma unroll
                for (int ps = 0; ps < PSPLIT; ++ps) {
                    xbar<PREF>();
#pragma unroll
                    for (int q = 0; q < NBF; ++q) {
                        const int j = jof<P>(t, q), k = kmod<Ns>(j), o = (j - k) * R + k - ps * (N / 2), po = phi(o);
                        if ((FULL || j < NB) && (PSPLIT == 1 || (j < NB / 2) == (ps == 0))) {
#pragma unroll
                            for (int r = 0; r < R; ++r) s[phi_at<Ns>(o, po, r)] = v[q * R + r];
                        }
                    }
                    xbar<PREF>();
#pragma unroll
                    for (int q = 0; q < NBF2; ++q) {
                        const int j = jof<P + 1>(t, q), pj = phi(j);
                        if (FULL2 || j < NB2) {
#pragma unroll
                            for (int r = 0; r < R2 / PSPLIT; ++r) {
                                const cpx<T> x = s[phi_at<NB2>(j, pj, r)];
                                if constexpr (PSPLIT == 2) {
                                    if (ps == 0) { keep[q * (R2 / 2) + r] = x; continue; }
                                }
                                v[q * R2 + r + ps * (R2 / 2)] = x;
                            }
                        }
                    }
                }
                if constexpr (PSPLIT == 2) {
#pragma unroll
                    for (int q = 0; q < NBF2; ++q)
#pragma unroll
                        for (int r = 0; r < R2 / 2; ++r) v[q * R2 + r] = keep[q * (R2 / 2) + r];
                }
            }
            passes_w<P + 1>(v, twp, lds, t, wpre);
        }
    }

    static __device__ __forceinline__ void run(const Pow2Args &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x % TPL, ll = threadIdx.x / TPL;
        const int64_t lane = (int64_t)xcd_block(blockIdx.x, gridDim.x, a.xcd_chunk) * LPB + ll;
        const bool live = lane < a.nlanes;
        const cpx<T> *in = (const cpx<T> *)a.in + (live ? lane : 0) * a.pitch_in;
        cpx<T> *out = (cpx<T> *)a.out + (live ? lane : 0) * a.pitch_out;
        char *lds = smem + (size_t)ll * LANE_LDS * (HALF ? sizeof(T) : 2 * sizeof(T));
        cpx<T> v[E];
        {
            constexpr int R0 = RL::at(0), NB0 = N / R0, NBF0 = slots(0);
            if constexpr (VEC == 2) {
                static_assert(NBF0 % 2 == 0 && full(0) && full(RL::NP - 1), "VEC=2 needs an even number of whole butterfly rounds in the first and last pass");
#pragma unroll
                for (int q = 0; q < NBF0; q += 2)
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const vec4f w = gload4<(NT & 2) != 0>((const float *)(in + jof<0>(t, q) + r * NB0));
                        v[q * R0 + r] = mk<T>(w.x, w.y); v[(q + 1) * R0 + r] = mk<T>(w.z, w.w);
                    }
            } else {
#pragma unroll
                for (int q = 0; q < NBF0; ++q)
                    if (full(0) || t + q * TPL < NB0) {
#pragma unroll
                        for (int r = 0; r < R0; ++r) v[q * R0 + r] = gload<T, (NT & 2) != 0>(in + t + q * TPL + r * NB0);
                    }
            }
        }
        if (a.inverse) {
#pragma unroll
            for (int i = 0; i < E; ++i) v[i].y = -v[i].y;
        }
        if constexpr ((FLAGS & 8) != 0) {   // fused four-step twiddle (after the conjugation, so the same table serves both directions)
            constexpr int R0 = RL::at(0), NB0 = N / R0, NBF0 = slots(0);
            const int k1 = (int)(lane % a.f1), mask = (1 << a.logB) - 1;
            const cpx<T> *lo = (const cpx<T> *)a.twlo, *hi = (const cpx<T> *)a.twhi;
#pragma unroll
            for (int q = 0; q < NBF0; ++q)
                if (full(0) || jof<0>(t, q) < NB0) {
#pragma unroll
                    for (int r = 0; r < R0; ++r) {
                        const int m = (jof<0>(t, q) + r * NB0) * k1;
                        v[q * R0 + r] = cmul(v[q * R0 + r], cmul(hi[m >> a.logB], lo[m & mask]));
                    }
                }
        }
        passes<0>(v, (const cpx<T> *)a.twp, lds, t);
        if (!live) return;
        constexpr int RL_ = RL::at(RL::NP - 1), NBL = N / RL_, NBFL = slots(RL::NP - 1);
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int i = 0; i < E; ++i) { v[i].x *= sc; v[i].y *= -sc; }   // conj + norm_default (lib.rs:333-338)
        }
        if constexpr (VEC == 2) {
            static_assert(NBFL % 2 == 0, "VEC=2 needs an even number of butterflies per thread in the last pass");
#pragma unroll
            for (int q = 0; q < NBFL; q += 2)
#pragma unroll
                for (int r = 0; r < RL_; ++r) {
                    vec4f w; w.x = v[q * RL_ + r].x; w.y = v[q * RL_ + r].y; w.z = v[(q + 1) * RL_ + r].x; w.w = v[(q + 1) * RL_ + r].y;
                    gstore4<(NT & 1) != 0>((float *)(out + jof<RL::NP - 1>(t, q) + r * NBL), w);
                }
        } else {
#pragma unroll
            for (int q = 0; q < NBFL; ++q)
                if (full(RL::NP - 1) || t + q * TPL < NBL) {
#pragma unroll
                    for (int r = 0; r < RL_; ++r) gstore<T, (NT & 1) != 0>(out + t + q * TPL + r * NBL, v[q * RL_ + r]);
                }
        }
    }
};

template <typename K> __global__ __launch_bounds__(K::THREADS, K::MIN_WAVES) void k_pow2(const Pow2Args a) { K::run(a); }

#ifndef __HIPCC_RTC__
// host: per-pass transposed twiddles, tw_p[(r-1)*Ns + k] = e^{-2 pi i r k/(Ns R)}, long double
template <typename RL> inline void build_tw(HostTable &out) {
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
}


#endif

}  // namespace ndfft
