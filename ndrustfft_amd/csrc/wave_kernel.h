// wave_kernel.h -- SHORT contiguous C2C lanes (n = 2 .. 64, powers of two) without LDS: every transposition between
// "which index bits are spread over the lanes of a wavefront" and "which index bits are spread over a thread's
// registers" is done with cross-lane operations (DPP row shifts / quad permutes, v_permlane16/32_swap) -- the
// "wavefront shuffles for the small-radix transposes" of the north star.
//
// Replaces, for one wavefront's worth of lanes, FftHandler::fft_lane / ifft_lane (src/lib.rs:313-331) and the
// strategy-(i) row loop around them (src/lib.rs:117-124, par 187-194).
//
// One wavefront (64 threads x 8 complex registers) owns a CONTIGUOUS chunk of 512 complex elements = 512 / n whole
// lanes.  The chunk is addressed by a 9-bit flat index g = L n + x (x: position in the lane, L: lane in the chunk).
//   load   fully coalesced, 16 bytes per thread per instruction (1 KiB per wave instruction) -- short lanes read this
//          way reach the copy ceiling; the old kernel read 64-byte pieces per lane (f32 n = 64: 72 % of the roofline)
//   swaps  a "bit swap" exchanges one index bit held in the thread number with one held in the register number: every
//          thread keeps half of its registers and trades the other half with lane ^ 2^b.  lane ^ 1, ^ 2: quad_perm DPP +
//          select; ^ 4, ^ 8: row_shr / row_shl DPP with a bank mask (one instruction per dword); ^ 16, ^ 32:
//          v_permlane16_swap / v_permlane32_swap (one instruction per dword PAIR, new on gfx950)
//   FFT    decimation in frequency, at most two radix-8 passes entirely in registers (the pass's index bits are
//          register bits at that moment), twiddles W_n^(n_lo k) from the plan's table
//   store  swaps back to the coalesced pattern; the bit order of the thread number is left wherever it ended up: a
//          wave instruction still writes one contiguous KiB, only WHICH thread writes which 16 bytes changes
// The swap schedule is computed at compile time from (element size, log2 n): see WaveSched.
// No LDS, no barrier; 128 (f32) .. 224 (f64) cross-lane instructions per 512 elements.
#pragma once
#include "pow2_kernel.h"   // vec4f, gstore, butterflies

namespace ndfft {

// ---- compile-time schedule --------------------------------------------------------------------------------------
// flat index bit ids 0..8; a location is a thread bit (0..5) or a register bit (0..2)
struct WaveSched {
    int nsteps = 0;
    int kind[32] = {};        // 0: swap(thread bit p0, register bit p1); 1: butterfly pass p0 (0 or 1)
    int p0[32] = {}, p1[32] = {};
    int pass_bits[2] = {0, 0};            // index bits transformed by pass 0 / pass 1
    int pass_regpos[2][3] = {{0, 0, 0}, {0, 0, 0}};   // register bit holding the pass's i-th (least significant first) index bit
    int nlo_thrpos[3] = {0, 0, 0};        // pass 0 twiddle: thread bit holding x_j, j < logn - 3
    int out_thr[6] = {};                  // OUTPUT flat bit carried by each thread bit at store time
    int out_reg[3] = {};                  // ... and by each register bit
};

constexpr WaveSched wave_make_sched(int dw /* dwords per complex element: 2 or 4 */, int logn) {
    WaveSched s;
    int thr[6] = {}, reg[3] = {};          // flat INPUT bit id at each location
    if (dw == 2) { reg[0] = 0; for (int k = 0; k < 6; ++k) thr[k] = k + 1; reg[1] = 7; reg[2] = 8; }
    else { for (int k = 0; k < 6; ++k) thr[k] = k; reg[0] = 6; reg[1] = 7; reg[2] = 8; }
    auto need = [&](unsigned mask) {       // bring every id of `mask` into registers (mask has <= 3 bits set)
        for (int tp = 0; tp < 6; ++tp) {
            if (!((mask >> thr[tp]) & 1u)) continue;
            int rb = -1;
            for (int r = 0; r < 3; ++r) if (!((mask >> reg[r]) & 1u)) { rb = r; break; }
            s.kind[s.nsteps] = 0; s.p0[s.nsteps] = tp; s.p1[s.nsteps] = rb; ++s.nsteps;
            const int t = thr[tp]; thr[tp] = reg[rb]; reg[rb] = t;
        }
    };
    auto regpos_of = [&](int id) { for (int r = 0; r < 3; ++r) if (reg[r] == id) return r; return -1; };
    auto thrpos_of = [&](int id) { for (int k = 0; k < 6; ++k) if (thr[k] == id) return k; return -1; };
    const int b0 = logn < 3 ? logn : 3, b1 = logn - b0;
    s.pass_bits[0] = b0; s.pass_bits[1] = b1;
    // pass 0: the TOP b0 bits of x (decimation in frequency); produces the LOW b0 bits of the output index
    unsigned m0 = 0; for (int i = 0; i < b0; ++i) m0 |= 1u << (logn - b0 + i);
    need(m0);
    for (int i = 0; i < b0; ++i) s.pass_regpos[0][i] = regpos_of(logn - b0 + i);
    for (int j = 0; j < b1; ++j) s.nlo_thrpos[j] = thrpos_of(j);
    s.kind[s.nsteps] = 1; s.p0[s.nsteps] = 0; ++s.nsteps;
    if (b1 > 0) {
        unsigned m1 = 0; for (int j = 0; j < b1; ++j) m1 |= 1u << j;
        need(m1);
        for (int j = 0; j < b1; ++j) s.pass_regpos[1][j] = regpos_of(j);
        s.kind[s.nsteps] = 1; s.p0[s.nsteps] = 1; ++s.nsteps;
    }
    // output bit carried by input bit id: pass-0 bits become out bits 0..b0-1, pass-1 bits out bits b0.., lane bits stay
    auto outbit = [&](int id) { return id >= logn ? id : (id >= logn - b0 ? id - (logn - b0) : id + b0); };
    // store pattern: 16 bytes per thread -> the register bits must carry out bits {0, 7, 8} (8-byte elements) or {6, 7, 8}
    unsigned ms = 0;
    for (int id = 0; id < 9; ++id) {
        const int ob = outbit(id);
        const bool isreg = dw == 2 ? (ob == 0 || ob == 7 || ob == 8) : (ob >= 6);
        if (isreg) ms |= 1u << id;
    }
    need(ms);
    for (int k = 0; k < 6; ++k) s.out_thr[k] = outbit(thr[k]);
    for (int r = 0; r < 3; ++r) s.out_reg[r] = outbit(reg[r]);
    return s;
}

// ---- the bit swap: thread bit TB <-> the register bit that separates `lo` from `hi` ---------------------------------
// before: thread t holds lo (register bit 0) and hi (register bit 1); after: threads with TB = 0 hold (own lo, partner's lo),
// threads with TB = 1 hold (partner's hi, own hi), partner = t ^ 2^TB
#ifdef NDFFT_WAVE_SWAP_OVERRIDE   // host-side builds of these sources (no cross-lane hardware) supply the exchange
template <int TB> __device__ __forceinline__ void wave_swap_u32(unsigned &lo, unsigned &hi, bool) { NDFFT_WAVE_SWAP_OVERRIDE(lo, hi, TB); }
#else
template <int TB> __device__ __forceinline__ void wave_swap_u32(unsigned &lo, unsigned &hi, bool bit) {
    if constexpr (TB == 0 || TB == 1) {
        constexpr int ctrl = TB == 0 ? 0xB1 /* quad_perm [1,0,3,2] */ : 0x4E /* quad_perm [2,3,0,1] */;
        const unsigned plo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, 0xf, 0xf, true);
        const unsigned phi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, 0xf, 0xf, true);
        const unsigned nlo = bit ? phi : lo, nhi = bit ? hi : plo;
        lo = nlo; hi = nhi;
    } else if constexpr (TB == 2 || TB == 3) {
        // lanes with the bit set (banks 1,3 / 2,3 of each row of 16) take lo from the partner's hi below them,
        // lanes with the bit clear take hi from the partner's lo above them: one DPP move each, masked by bank
        constexpr int shr = TB == 2 ? 0x114 : 0x118, shl = TB == 2 ? 0x104 : 0x108;
        constexpr int bm_set = TB == 2 ? 0xA : 0xC, bm_clr = TB == 2 ? 0x5 : 0x3;
        const unsigned nlo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)hi, shr, 0xf, bm_set, false);
        const unsigned nhi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)lo, shl, 0xf, bm_clr, false);
        lo = nlo; hi = nhi;
    } else if constexpr (TB == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false);   // lo[odd rows] <-> hi[even rows]
        lo = r[0]; hi = r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);   // lo[32..63] <-> hi[0..31]
        lo = r[0]; hi = r[1];
    }
}
#endif
template <int TB> __device__ __forceinline__ void wave_swap(float &lo, float &hi, bool bit) {
    unsigned a = __float_as_uint(lo), b = __float_as_uint(hi);
    wave_swap_u32<TB>(a, b, bit);
    lo = __uint_as_float(a); hi = __uint_as_float(b);
}
template <int TB> __device__ __forceinline__ void wave_swap(double &lo, double &hi, bool bit) {
    unsigned al = (unsigned)__double2loint(lo), ah = (unsigned)__double2hiint(lo);
    unsigned bl = (unsigned)__double2loint(hi), bh = (unsigned)__double2hiint(hi);
    wave_swap_u32<TB>(al, bl, bit);
    wave_swap_u32<TB>(ah, bh, bit);
    lo = __hiloint2double((int)ah, (int)al); hi = __hiloint2double((int)bh, (int)bl);
}

template <typename T, int LOGN> struct WaveFft {
    static constexpr int N = 1 << LOGN;
    static constexpr int DW = (int)(sizeof(cpx<T>) / 4);
    static constexpr int THREADS = 256;              // 4 independent wavefronts per workgroup
    static constexpr int CHUNK = 512;                // complex elements per wavefront
    static constexpr WaveSched S = wave_make_sched(DW, LOGN);

    template <int TB, int RB> static __device__ __forceinline__ void swap_bits(cpx<T> (&v)[8], int lane) {
        const bool bit = (lane >> TB) & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (!((e >> RB) & 1)) {
                T lx = v[e].x, ly = v[e].y, hx = v[e | (1 << RB)].x, hy = v[e | (1 << RB)].y;
                wave_swap<TB>(lx, hx, bit);
                wave_swap<TB>(ly, hy, bit);
                v[e] = mk<T>(lx, ly); v[e | (1 << RB)] = mk<T>(hx, hy);
            }
    }

    // radix-2^B butterflies over the register bits S.pass_regpos[P][0..B), for every value of the other register bits
    template <int P> static __device__ __forceinline__ void pass(cpx<T> (&v)[8], int lane, const cpx<T> *tw) {
        constexpr int B = S.pass_bits[P], R = 1 << B;
        if constexpr (B > 0) {
            constexpr int p0 = S.pass_regpos[P][0], p1 = B > 1 ? S.pass_regpos[P][1] : -1, p2 = B > 2 ? S.pass_regpos[P][2] : -1;
            // n_lo = x mod (N / 8): the index bits pass 1 will transform, read from the thread number
            int nlo = 0;
            if constexpr (P == 0 && S.pass_bits[1] > 0) {
#pragma unroll
                for (int j = 0; j < S.pass_bits[1]; ++j) nlo |= ((lane >> S.nlo_thrpos[j]) & 1) << j;
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // c enumerates the register bits NOT in the pass: skip values of c that have a pass bit set
                if (((c >> p0) & 1) || (p1 >= 0 && ((c >> p1) & 1)) || (p2 >= 0 && ((c >> p2) & 1))) continue;
                cpx<T> a[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = c | ((r & 1) << p0) | (p1 >= 0 ? ((r >> 1) & 1) << p1 : 0) | (p2 >= 0 ? ((r >> 2) & 1) << p2 : 0);
                    a[r] = v[e];
                }
                Bfly<T, R>::run(a);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = c | ((r & 1) << p0) | (p1 >= 0 ? ((r >> 1) & 1) << p1 : 0) | (p2 >= 0 ? ((r >> 2) & 1) << p2 : 0);
                    if constexpr (P == 0 && S.pass_bits[1] > 0) {
                        if (r > 0) a[r] = cmul(a[r], tw[nlo * r]);          // W_N^(n_lo k), k = r
                    }
                    v[e] = a[r];
                }
            }
        }
    }

    template <int I> static __device__ __forceinline__ void steps(cpx<T> (&v)[8], int lane, const cpx<T> *tw) {
        if constexpr (I < S.nsteps) {
            if constexpr (S.kind[I] == 0) swap_bits<S.p0[I], S.p1[I]>(v, lane);
            else pass<S.p0[I]>(v, lane, tw);
            steps<I + 1>(v, lane, tw);
        }
    }

    static __device__ __forceinline__ void run(const WaveArgs &a) {
        const int lane = threadIdx.x & 63;
        const int64_t wave = (int64_t)xcd_block(blockIdx.x, gridDim.x, a.xcd_chunk) * (THREADS / 64) + (threadIdx.x >> 6);
        const int64_t base = wave * CHUNK;
        if (base >= a.total) return;                       // whole wavefront out of range (no barriers in this kernel)
        const int64_t left = a.total - base;               // elements of this chunk that exist (even; whole lanes)
        const cpx<T> *in = (const cpx<T> *)a.in + base;
        cpx<T> *out = (cpx<T> *)a.out + base;
        cpx<T> v[8];
        // ---- coalesced load: 16 bytes per thread per instruction ----
        if constexpr (DW == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = i * 128 + 2 * lane;
                vec4f w = {0.f, 0.f, 0.f, 0.f};
                if (g < left) w = *(const vec4f *)(in + g);
                v[2 * i] = mk<T>(w.x, w.y); v[2 * i + 1] = mk<T>(w.z, w.w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = i * 64 + lane;
                v[i] = g < left ? in[g] : mk<T>((T)0, (T)0);
            }
        }
        if (a.inverse) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e].y = -v[e].y;
        }
        steps<0>(v, lane, (const cpx<T> *)a.tw);
        if (a.inverse) {
            const T sc = (T)a.scale;
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e].x *= sc; v[e].y *= -sc; }   // conj + norm_default (lib.rs:333-338)
        }
        // ---- coalesced store: this thread's part of the output flat index from its lane bits ----
        int gt = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) gt |= ((lane >> k) & 1) << S.out_thr[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int g = gt;
#pragma unroll
            for (int r = 0; r < 3; ++r) g |= ((e >> r) & 1) << S.out_reg[r];
            if constexpr (DW == 2) {
                // registers whose out bit 0 is clear pair up with the one that has it set: one 16-byte store
                constexpr int r0 = S.out_reg[0] == 0 ? 0 : (S.out_reg[1] == 0 ? 1 : 2);
                if (!((e >> r0) & 1)) {
                    const cpx<T> lo = v[e], hi = v[e | (1 << r0)];
                    vec4f w; w.x = lo.x; w.y = lo.y; w.z = hi.x; w.w = hi.y;
                    if (g < left) __builtin_nontemporal_store(w, (vec4f *)(out + g));
                }
            } else {
                if (g < left) gstore<T, true>(out + g, v[e]);
            }
        }
    }
};

template <typename K> __global__ __launch_bounds__(K::THREADS) void k_wave(const WaveArgs a) { K::run(a); }

}  // namespace ndfft
