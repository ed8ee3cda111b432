// tools/pow2_persist.h -- PROBE (round 4, measured slower, not part of the library: tools/kbench.hip f64_persist, profiles/r07/r07a_*, r07b_*):
// the register-resident Stockham kernel of pow2_kernel.h as a PERSISTENT, software-pipelined grid.
//
// k_pow2 runs one workgroup per lane block: load the lane, butterfly passes with LDS exchanges, store.  A launch of a few
// rounds of workgroups (4096 x 4096 c128: 4096 workgroups on 1024 slots) pays ~10 us of its ~93 us once per launch
// (4096 lanes 93 us, every further 4096 lanes 81 us).  Here a workgroup owns the lane blocks
// b, b + G, b + 2 G, ... (G = gridDim.x, a multiple of 8, so xcd_block() keeps a workgroup's blocks on its own XCD) and issues
// the global loads of lane block i + 1 BEFORE the passes of lane block i: every workgroup has a lane of loads in flight
// while it computes.
//
// Two things follow from gfx950's in-order vmcnt counter:
//  * a twiddle load issued after the prefetch could only be waited for together with it, so the twiddles must not come
//    through the vector memory path at all: the workgroup copies the rows it needs into LDS once (n = 4096 f64: 2040
//    entries, 32 KiB) and every pass reads them from there (ds_read_b128, lgkmcnt);
//  * the prefetched lane costs E complex registers per thread (n = 4096 f64: 32 VGPRs): two workgroups per CU at <= 128 VGPRs
//    instead of four at 64 -- the lanes in flight per CU stay four.
#pragma once
#include "pow2_kernel.h"

namespace ndfft {

template <typename K> struct Pow2Persist;

template <typename T, int N, int TPL, int LPB, bool HALF, typename RL, int FLAGS, int MINW, int NT, int PSPLIT>
struct Pow2Persist<Pow2Kernel<T, N, TPL, LPB, HALF, RL, FLAGS, MINW, NT, 1, PSPLIT>> {
    using K = Pow2Kernel<T, N, TPL, LPB, HALF, RL, FLAGS, MINW, NT, 1, PSPLIT>;
    static constexpr int E = K::E, THREADS = K::THREADS, MIN_WAVES = MINW, NP = RL::NP;
    static_assert(HALF && PSPLIT == 1 && LPB == 1, "persistent form: half exchange, one lane per workgroup");
    static_assert((FLAGS & ~16) == 0, "persistent form: only the twiddle-powers flag");
    static constexpr bool all_full() { for (int p = 0; p < NP; ++p) if (!K::full(p)) return false; return true; }
    static_assert(all_full(), "persistent form: whole butterfly rounds in every pass");

    // twiddle rows kept in LDS per pass: W^k, W^2k, W^4k for the passes whose global table exceeds 32 KiB (pow2_kernel.h: TW_POWERS), all R - 1 otherwise
    static constexpr bool tw_powers(int p) { return (FLAGS & 16) != 0 && (RL::at(p) == 8) && (size_t)(RL::at(p) - 1) * RL::ns(p) * sizeof(cpx<T>) > 32 * 1024; }
    static constexpr int tw_rows(int p) { return tw_powers(p) ? 3 : RL::at(p) - 1; }
    static constexpr int tw_lds_off(int p) { int o = 0; for (int q = 1; q < p; ++q) o += tw_rows(q) * RL::ns(q); return o; }
    static constexpr int TW_ENTRIES = tw_lds_off(NP);
    static constexpr size_t EXCH_BYTES = ((size_t)K::LANE_LDS * sizeof(T) + 15) / 16 * 16;
    static constexpr size_t LDS_BYTES = EXCH_BYTES + (size_t)TW_ENTRIES * sizeof(cpx<T>);

    static constexpr int R0 = RL::at(0), NB0 = N / R0, NBF0 = K::slots(0);
    static constexpr int RL_ = RL::at(NP - 1), NBL = N / RL_, NBFL = K::slots(NP - 1);

    static __device__ __forceinline__ void load_lane(cpx<T> (&v)[E], const cpx<T> *__restrict__ in, int t) {
#pragma unroll
        for (int q = 0; q < NBF0; ++q)
#pragma unroll
            for (int r = 0; r < R0; ++r) v[q * R0 + r] = gload<T, (NT & 2) != 0>(in + t + q * TPL + r * NB0);
    }
    static __device__ __forceinline__ void store_lane(const cpx<T> (&v)[E], cpx<T> *__restrict__ out, int t) {
#pragma unroll
        for (int q = 0; q < NBFL; ++q)
#pragma unroll
            for (int r = 0; r < RL_; ++r) gstore<T, (NT & 1) != 0>(out + t + q * TPL + r * NBL, v[q * RL_ + r]);
    }

    // global per-pass tables (pow2_kernel.h: build_tw) -> the LDS rows; once per workgroup
    template <int P> static __device__ __forceinline__ void stage_tw(cpx<T> *tws, const cpx<T> *__restrict__ twp) {
        if constexpr (P < NP) {
            constexpr int Ns = RL::ns(P), rows = tw_rows(P);
            const cpx<T> *src = twp + RL::twoff(P);
            cpx<T> *dst = tws + tw_lds_off(P);
            for (int i = threadIdx.x; i < rows * Ns; i += THREADS) {
                const int row = i / Ns, k = i - row * Ns;
                const int srow = tw_powers(P) ? (row == 2 ? 3 : row) : row;       // rows r - 1 = 0, 1, 3 hold W^k, W^2k, W^4k
                dst[i] = src[srow * Ns + k];
            }
            stage_tw<P + 1>(tws, twp);
        }
    }

    template <int P> static __device__ __forceinline__ void passes(cpx<T> (&v)[E], const cpx<T> *tws, T *s, int t) {
        constexpr int R = RL::at(P), Ns = RL::ns(P), NBF = K::slots(P);
        if constexpr (P > 0) {
            const cpx<T> *tw = tws + tw_lds_off(P);
#pragma unroll
            for (int q = 0; q < NBF; ++q) {
                const int k = kmod<Ns>(t + q * TPL);
                if constexpr (tw_powers(P)) {
                    cpx<T> w[8];
                    w[1] = tw[k]; w[2] = tw[Ns + k]; w[4] = tw[2 * Ns + k];
                    w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
#pragma unroll
                    for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], w[r]);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], tw[(r - 1) * Ns + k]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NBF; ++q) Bfly<T, R>::run(&v[q * R]);
        if constexpr (P + 1 < NP) {
            constexpr int R2 = RL::at(P + 1), NB2 = N / R2, NBF2 = K::slots(P + 1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBF; ++q) {
                    const int j = t + q * TPL, k = kmod<Ns>(j), o = (j - k) * R + k, po = phi(o);
#pragma unroll
                    for (int r = 0; r < R; ++r) s[phi_at<Ns>(o, po, r)] = half ? v[q * R + r].y : v[q * R + r].x;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < NBF2; ++q) {
                    const int j = t + q * TPL, pj = phi(j);
#pragma unroll
                    for (int r = 0; r < R2; ++r) {
                        const T x = s[phi_at<NB2>(j, pj, r)];
                        if (half) v[q * R2 + r].y = x; else v[q * R2 + r].x = x;
                    }
                }
            }
            passes<P + 1>(v, tws, s, t);
        }
    }

    // transform the lane in `cur` and store it as lane `lane`
    static __device__ __forceinline__ void finish(cpx<T> (&cur)[E], const Pow2Args &a, int64_t lane, const cpx<T> *tws, T *s, int t) {
        if (a.inverse) {
#pragma unroll
            for (int i = 0; i < E; ++i) cur[i].y = -cur[i].y;
        }
        passes<0>(cur, tws, s, t);
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int i = 0; i < E; ++i) { cur[i].x *= sc; cur[i].y *= -sc; }       // conj + norm_default (lib.rs:333-338)
        }
        store_lane(cur, (cpx<T> *)a.out + lane * a.pitch_out, t);
    }

    // Two register sets alternate (A computes while B's loads are in flight, then B computes while A's are), so nothing is copied in
    // the steady state, and the loop body is straight-line code: the only vmcnt wait is the one in front of the first use of the
    // current set, and it leaves the previous lane's stores and the prefetch in flight (a prefetch under an `if` would make the
    // compiler's wait at the join as strict as the path WITHOUT the prefetch needs: vmcnt(0)).  The last lane runs after the loop.
    static __device__ __forceinline__ void run(const Pow2Args &a) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        const int t = threadIdx.x;
        const unsigned nblk = (unsigned)a.nlanes, G = gridDim.x;
        unsigned vb = blockIdx.x;
        if (vb >= nblk) return;
        T *s = (T *)smem;
        cpx<T> *tws = (cpx<T> *)(smem + EXCH_BYTES);
        stage_tw<1>(tws, (const cpx<T> *)a.twp);
        const cpx<T> *in = (const cpx<T> *)a.in;
        int64_t lane = xcd_block(vb, nblk, a.xcd_chunk);
        cpx<T> va[E], vb2[E];
        load_lane(va, in + lane * a.pitch_in, t);
        __syncthreads();                                                         // the twiddle rows are in LDS
        for (;;) {
            if (vb + G >= nblk) break;                                           // uniform over the workgroup
            int64_t nlane = xcd_block(vb + G, nblk, a.xcd_chunk);
            load_lane(vb2, in + nlane * a.pitch_in, t);
            finish(va, a, lane, tws, s, t);
            vb += G; lane = nlane;
            if (vb + G >= nblk) {
#pragma unroll
                for (int i = 0; i < E; ++i) va[i] = vb2[i];
                break;
            }
            nlane = xcd_block(vb + G, nblk, a.xcd_chunk);
            load_lane(va, in + nlane * a.pitch_in, t);
            finish(vb2, a, lane, tws, s, t);
            vb += G; lane = nlane;
        }
        finish(va, a, lane, tws, s, t);
    }
};

template <typename P> __global__ __launch_bounds__(P::THREADS, P::MIN_WAVES) void k_pow2_persist(const Pow2Args a) { P::run(a); }


// ---- dynamic persistent grid (no prefetch, the product kernel's occupancy) ------------------------------------------
// A static b, b + G, ... split measured ~10 % SLOWER per lane than one workgroup per lane (tools/drainprobe.hip, profiles/r07): workgroups
// drift apart and nothing evens them out.  Here a workgroup takes its next lane block from a counter of its XCD (blockIdx % 8): the
// blocks of an XCD are handed out in the order a one-workgroup-per-block launch would start them, to whichever workgroup is free.
// ctr: 8 counters, 128 bytes apart, + the exit counter at [8 * 32]; all zero before the first launch -- the last workgroup to
// leave resets them, and launches on one stream are serialised, so they are zero again when the next launch starts.
template <typename K> struct Pow2Dyn;
template <typename T, int N, int TPL, int LPB, bool HALF, typename RL, int FLAGS, int MINW, int NT, int PSPLIT>
struct Pow2Dyn<Pow2Kernel<T, N, TPL, LPB, HALF, RL, FLAGS, MINW, NT, 1, PSPLIT>> {
    using K = Pow2Kernel<T, N, TPL, LPB, HALF, RL, FLAGS, MINW, NT, 1, PSPLIT>;
    using P = Pow2Persist<K>;
    static constexpr int E = K::E, THREADS = K::THREADS, MIN_WAVES = MINW;
    static constexpr size_t LDS_BYTES = K::LDS_BYTES;
    static __device__ __forceinline__ void run(const Pow2Args &a, unsigned *ctr) {
        extern __shared__ __attribute__((aligned(16))) char smem[];
        __shared__ unsigned s_j;
        const int t = threadIdx.x;
        const unsigned nblk = (unsigned)a.nlanes, x = blockIdx.x & 7u;
        if (t == 0) s_j = atomicAdd(&ctr[x * 32], 1u);
        __syncthreads();
        for (;;) {
            const unsigned vb = s_j * 8u + x;
            if (vb >= nblk) break;
            const int64_t lane = xcd_block(vb, nblk, a.xcd_chunk);
            cpx<T> v[E];
            P::load_lane(v, (const cpx<T> *)a.in + lane * a.pitch_in, t);
            __syncthreads();                                                     // every thread has read s_j
            if (t == 0) s_j = atomicAdd(&ctr[x * 32], 1u);                       // the NEXT block's index travels with this lane's loads
            if (a.inverse) {
#pragma unroll
                for (int i = 0; i < E; ++i) v[i].y = -v[i].y;
            }
            K::template passes<0>(v, (const cpx<T> *)a.twp, smem, t);            // (its barriers publish s_j)
            if (a.inverse) {
                const T sc = (T)a.scale;
#pragma unroll
                for (int i = 0; i < E; ++i) { v[i].x *= sc; v[i].y *= -sc; }
            }
            P::store_lane(v, (cpx<T> *)a.out + lane * a.pitch_out, t);
        }
        if (t == 0) {
            __threadfence();
            if (atomicAdd(&ctr[8 * 32], 1u) == gridDim.x - 1) {                  // last one out
                for (int i = 0; i <= 8; ++i) ctr[i * 32] = 0;
                __threadfence();
            }
        }
    }
};
template <typename D> __global__ __launch_bounds__(D::THREADS, D::MIN_WAVES) void k_pow2_dyn(const Pow2Args a, unsigned *ctr) { D::run(a, ctr); }

}  // namespace ndfft
