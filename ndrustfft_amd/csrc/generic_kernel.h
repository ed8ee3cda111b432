// generic_kernel.h -- the any-n, any-op lane kernel (gfx950); instantiated by kernels_generic_*.hip.
//
// One workgroup owns `lpb` whole lanes in LDS and makes ONE pass over HBM:
//   LOAD   global -> LDS, coalesced either along the lane (IO_ROW) or across adjacent lanes
//          (IO_COL: the LDS-padded transpose that replaces the reference's per-lane
//          x.to_vec() gather, src/lib.rs:133, 155);
//   PRE    op-specific fold of the raw lane into the complex FFT input Z[0..F)
//          (R2C packing, C2R/DCT-III Hermitian fold, Makhoul permutation, DCT-IV pre-twiddle,
//          the "before" normalisations of C2R and DCT: src/lib.rs:511-521, 692-696);
//   FFT    Stockham autosort radix passes ping-ponging two LDS buffers (Bluestein when F has a
//          prime factor > 13), twiddles from a precomputed HBM table that stays in L1/L2;
//   STORE  op-specific gather out of the FFT result (R2C split, DCT post-twiddles, the "after"
//          normalisation of the C2C inverse, src/lib.rs:326-330), LDS -> global coalesced
//          (replaces y.assign(&outvec), src/lib.rs:134).
#pragma once
#include "butterflies.h"
#include "engine.h"
#include "realops.h"

namespace ndfft {

// LDS header: per-lane global offsets, radix lists and per-pass fast-division constants
struct GenHeader {
    int64_t off_in[kMaxLpb], off_out[kMaxLpb];
    int32_t radix[kMaxPasses], radixM[kMaxPasses];
};
constexpr size_t kGenHeaderBytes = (sizeof(GenHeader) + 15) & ~size_t(15);

template <typename T> struct GenCtx {
    GenHeader *h;
    cpx<T> *buf[2];
    int lanes;                   // lanes this block really owns (<= lpb)
    int fl, fj0, fstep, flstep;  // FFT-phase thread map: lane = fl (+= flstep), index = fj0 (+= fstep)
};

// Z buffers are padded by one element every 8 so that the stride-R writes of the first radix
// passes (R = 4, 8: addresses 8j+q -> 9j+q) spread over all LDS banks for 8- and 16-byte elements
__device__ __forceinline__ int zi(int p) { return p + (p >> 3); }
struct ZiPad8 { static __device__ __forceinline__ int map(int p) { return p + (p >> 3); } };

// j mod d for j < 2^17, d < 2^15, with m = ceil(2^32 / d)
__device__ __forceinline__ int fast_mod(int j, int d, uint32_t m) {
    const uint32_t q = (uint32_t)(((uint64_t)(uint32_t)j * m) >> 32);
    return j - (int)q * d;
}

// ---------------------------------------------------------------------------------------------
// one Stockham pass over all lanes of the block.
// SRC / DST say where the pass reads / writes: LDS (padded Z buffer) or, for the FIRST / LAST pass of the
// ops whose PRE / POST is elementwise (C2C, R2C), straight from / to global memory -- x[j + r n/R] and
// y[j + q n/R] are unit-stride in j, so those accesses are coalesced and the LDS staging round trip
// (and its barrier) disappears.
// ---------------------------------------------------------------------------------------------
enum PassIO : int { IO_LDS = 0, IO_GLOBAL = 1 };

// what a fused pass needs from the kernel arguments -- passed BY VALUE: handing `const GenArgs&` to a
// non-inlined function would force the whole kernarg struct into scratch memory
template <typename T> struct PassGlobals { const void *in; void *out; int n_out; T scale; };

// (fused passes are only used on unit-stride lanes, so the element index is the memory offset)
template <typename T, int OP> __device__ __forceinline__ cpx<T> first_pass_load(const PassGlobals<T> &a, int64_t base, int i) {
    if constexpr (OP == G_R2C_EVEN) {          // z[i] = (x[2i], x[2i+1])
        const T *p = (const T *)a.in + base;
        return mk<T>(p[2 * i], p[2 * i + 1]);
    } else if constexpr (OP == G_R2C_ODD) {
        return mk<T>(((const T *)a.in)[base + i], (T)0);
    } else {
        cpx<T> v = ((const cpx<T> *)a.in)[base + i];
        if (OP == G_C2C_INV) v.y = -v.y;
        return v;
    }
}

template <typename T, int OP, int R, int SRC, int DST>
__device__ __forceinline__ void stockham_pass(const GenCtx<T> &c, const PassGlobals<T> &a, const cpx<T> *__restrict__ src,
                                              cpx<T> *__restrict__ dst, const cpx<T> *__restrict__ tw, int len, int Ns, int pitch) {
    const int nb = len / R;
    const uint32_t magic = Ns > 1 ? (uint32_t)((0x100000000ull + (uint32_t)Ns - 1) / (uint32_t)Ns) : 0u;
    for (int l = c.fl; l < c.lanes; l += c.flstep) {
        const cpx<T> *s = src + l * pitch;
        cpx<T> *d = dst + l * pitch;
        for (int j = c.fj0; j < nb; j += c.fstep) {
            cpx<T> v[R];
            if constexpr (SRC == IO_GLOBAL) {
                const int64_t base = c.h->off_in[l];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = first_pass_load<T, OP>(a, base, j + r * nb);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = s[zi(j + r * nb)];
            }
            int k = 0;
            if (Ns > 1) {
                k = fast_mod(j, Ns, magic);
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[(r - 1) * Ns + k]);   // this pass's transposed block
            }
            Bfly<T, R>::run(v);
            const int o = (j - k) * R + k;
            if constexpr (DST == IO_GLOBAL) {   // last pass: Ns = len / R, so o + q Ns = j + q nb
                cpx<T> *outp = (cpx<T> *)a.out + c.h->off_out[l];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    const int idx = o + q * Ns;
                    if (idx < a.n_out) {
                        cpx<T> w = v[q];
                        if (OP == G_C2C_INV) { w.x *= a.scale; w.y *= -a.scale; }
                        outp[idx] = w;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < R; ++q) d[zi(o + q * Ns)] = v[q];
            }
        }
    }
}

// in-place variant: ONE LDS buffer per lane.  The host guarantees that every thread owns at most one
// butterfly of the pass, so its R values sit in registers across the barrier that separates "all reads
// of this pass" from "first write of this pass" -- half the LDS footprint, twice the lanes per CU.
template <typename T, int OP, int R, int SRC, int DST>
__device__ __forceinline__ void stockham_pass_inplace(const GenCtx<T> &c, const PassGlobals<T> &a, cpx<T> *__restrict__ buf,
                                                      const cpx<T> *__restrict__ tw, int len, int Ns, int pitch) {
    const int nb = len / R, l = c.fl, j = c.fj0;
    const bool active = l < c.lanes && j < nb;
    cpx<T> *s = buf + l * pitch;
    cpx<T> v[R];
    int k = 0;
    if (active) {
        if constexpr (SRC == IO_GLOBAL) {
            const int64_t base = c.h->off_in[l];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = first_pass_load<T, OP>(a, base, j + r * nb);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = s[zi(j + r * nb)];
        }
        if (Ns > 1) {
            const uint32_t magic = (uint32_t)((0x100000000ull + (uint32_t)Ns - 1) / (uint32_t)Ns);
            k = fast_mod(j, Ns, magic);
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[(r - 1) * Ns + k]);
        }
        Bfly<T, R>::run(v);
    }
    const int o = (j - k) * R + k;
    if constexpr (DST == IO_GLOBAL) {
        if (active) {
            cpx<T> *outp = (cpx<T> *)a.out + c.h->off_out[l];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                const int idx = o + q * Ns;
                if (idx < a.n_out) {
                    cpx<T> w = v[q];
                    if (OP == G_C2C_INV) { w.x *= a.scale; w.y *= -a.scale; }
                    outp[idx] = w;
                }
            }
        }
    } else {
        if constexpr (SRC == IO_LDS) __syncthreads();
        if (active) {
#pragma unroll
            for (int q = 0; q < R; ++q) s[zi(o + q * Ns)] = v[q];
        }
    }
}

template <typename T, int OP, int SRC, int DST, bool BIG>
__device__ __forceinline__ void pass_switch_inplace(int R, const GenCtx<T> &c, const PassGlobals<T> &a, cpx<T> *buf,
                                                    const cpx<T> *tw, int len, int Ns, int pitch) {
    switch (R) {
#define NDFFT_R(RR) case RR: stockham_pass_inplace<T, OP, RR, SRC, DST>(c, a, buf, tw, len, Ns, pitch); break;
        NDFFT_R(2) NDFFT_R(3) NDFFT_R(4) NDFFT_R(5) NDFFT_R(6) NDFFT_R(7) NDFFT_R(8) NDFFT_R(9) NDFFT_R(10)
        default:
            if constexpr (BIG) {
                switch (R) {
                    NDFFT_R(11)
                    default: stockham_pass_inplace<T, OP, 13, SRC, DST>(c, a, buf, tw, len, Ns, pitch); break;
                }
            }
            break;
#undef NDFFT_R
    }
}

template <typename T, int OP, int SRC, int DST, bool BIG>
__device__ __forceinline__ void pass_switch(int R, const GenCtx<T> &c, const PassGlobals<T> &a, const cpx<T> *s, cpx<T> *d,
                                            const cpx<T> *tw, int len, int Ns, int pitch) {
    switch (R) {
#define NDFFT_R(RR) case RR: stockham_pass<T, OP, RR, SRC, DST>(c, a, s, d, tw, len, Ns, pitch); break;
        NDFFT_R(2) NDFFT_R(3) NDFFT_R(4) NDFFT_R(5) NDFFT_R(6) NDFFT_R(7) NDFFT_R(8) NDFFT_R(9) NDFFT_R(10)
        default:
            // the register-hungry prime radices live only in the BIG instantiation (compiled for <= 512 threads,
            // so the allocator may use up to 256 VGPRs instead of spilling: 264 = 8*3*11 runs 566 vs 975 us)
            if constexpr (BIG) {
                switch (R) {
                    NDFFT_R(11)
                    default: stockham_pass<T, OP, 13, SRC, DST>(c, a, s, d, tw, len, Ns, pitch); break;
                }
            }
            break;
#undef NDFFT_R
    }
}

// runs the radix passes on LDS buffers; returns the index of the buffer holding the result.
// fuse_in / fuse_out: first pass reads global / last pass writes global (see stockham_pass).
template <typename T, int OP, bool BIG, bool INPLACE>
__device__ __forceinline__ int run_passes(const GenCtx<T> &c, const PassGlobals<T> &a, int cur, int len, int npass, const int32_t *radix,
                          const cpx<T> *tw, int pitch, bool fuse_in, bool fuse_out) {
    constexpr bool can_in = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_EVEN || OP == G_R2C_ODD;
    constexpr bool can_out = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_ODD;
    int Ns = 1;
    for (int p = 0; p < npass; ++p) {
        const int R = radix[p];
        const bool gin = can_in && fuse_in && p == 0, gout = can_out && fuse_out && p == npass - 1;
        if (!gin) __syncthreads();
        const cpx<T> *s = c.buf[cur];
        cpx<T> *d = c.buf[cur ^ 1];
        if constexpr (can_in && INPLACE) {
            {   // only the elementwise ops are ever instantiated in place
                cpx<T> *b = c.buf[0];
                if constexpr (can_out) {
                    if (gin && gout) pass_switch_inplace<T, OP, IO_GLOBAL, IO_GLOBAL, BIG>(R, c, a, b, tw, len, Ns, pitch);
                    else if (gin) pass_switch_inplace<T, OP, IO_GLOBAL, IO_LDS, BIG>(R, c, a, b, tw, len, Ns, pitch);
                    else if (gout) pass_switch_inplace<T, OP, IO_LDS, IO_GLOBAL, BIG>(R, c, a, b, tw, len, Ns, pitch);
                    else pass_switch_inplace<T, OP, IO_LDS, IO_LDS, BIG>(R, c, a, b, tw, len, Ns, pitch);
                } else {
                    if (gin) pass_switch_inplace<T, OP, IO_GLOBAL, IO_LDS, BIG>(R, c, a, b, tw, len, Ns, pitch);
                    else pass_switch_inplace<T, OP, IO_LDS, IO_LDS, BIG>(R, c, a, b, tw, len, Ns, pitch);
                }
                if (p > 0) tw += (R - 1) * Ns;
                Ns *= R;
                continue;
            }
        } else if constexpr (can_in && can_out) {
            if (gin && gout) pass_switch<T, OP, IO_GLOBAL, IO_GLOBAL, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
            else if (gin) pass_switch<T, OP, IO_GLOBAL, IO_LDS, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
            else if (gout) pass_switch<T, OP, IO_LDS, IO_GLOBAL, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
            else pass_switch<T, OP, IO_LDS, IO_LDS, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
        } else if constexpr (can_in) {
            if (gin) pass_switch<T, OP, IO_GLOBAL, IO_LDS, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
            else pass_switch<T, OP, IO_LDS, IO_LDS, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
        } else {
            pass_switch<T, OP, IO_LDS, IO_LDS, BIG>(R, c, a, s, d, tw, len, Ns, pitch);
        }
        cur ^= 1;
        if (p > 0) tw += (R - 1) * Ns;   // next pass's twiddle block (pass 0 has none)
        Ns *= R;
    }
    __syncthreads();
    return cur;
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t lane_offset(const LaneGeom &g, int64_t lane) {
    int64_t off = 0;
    for (int d = g.nb - 1; d >= 0; --d) {
        const int64_t e = g.bshape[d], i = lane % e;
        lane /= e;
        off += i * g.bstride[d];
    }
    return off;
}

template <typename T, int OP, bool BIG, bool INPLACE> __global__ __launch_bounds__((BIG || INPLACE) ? 512 : 1024) void k_generic(const GenArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GenCtx<T> c;
    c.h = (GenHeader *)smem;
    c.buf[0] = (cpx<T> *)(smem + kGenHeaderBytes);
    c.buf[1] = INPLACE ? c.buf[0] : c.buf[0] + (size_t)a.lpb * a.pitch;
    const int64_t lane0 = (int64_t)blockIdx.x * a.lpb;
    c.lanes = (int)min((int64_t)a.lpb, a.nlanes - lane0);
    const int tid = threadIdx.x, nthr = blockDim.x, pitch = a.pitch;
    // division-free thread maps (all counts are powers of two chosen on the host)
    c.fl = tid >> a.fft_tpl_log; c.fj0 = tid & ((1 << a.fft_tpl_log) - 1);
    c.fstep = 1 << a.fft_tpl_log; c.flstep = nthr >> a.fft_tpl_log;

    if (tid < kMaxPasses) { c.h->radix[tid] = a.radix[tid]; c.h->radixM[tid] = a.radixM[tid]; }
    if (tid < c.lanes) {
        c.h->off_in[tid] = lane_offset(a.gin, lane0 + tid);
        c.h->off_out[tid] = lane_offset(a.gout, lane0 + tid);
    }
    __syncthreads();

    // ---- LOAD ------------------------------------------------------------------------------
    // ops whose PRE is elementwise load straight into Z (buffer 0); the others stage the raw lane
    // in buffer 1 and fold it into buffer 0.
    constexpr bool direct = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_EVEN || OP == G_R2C_ODD;
    constexpr bool store_direct = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_ODD;
    // row-layout elementwise ops: the first pass loads global memory itself and the last pass stores it
    const bool fuse_in = direct && a.load_mode == IO_ROW && a.gin.axis_stride == 1 && !a.blue && a.npass > 0;
    const bool fuse_out = store_direct && a.store_mode == IO_ROW && a.gout.axis_stride == 1 && !a.blue && a.npass > 0;
    constexpr bool in_cplx = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_C2R_EVEN || OP == G_C2R_ODD;
    constexpr bool out_cplx = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_EVEN || OP == G_R2C_ODD;
    if (!fuse_in) {
        int l0, lstep, j0, jstep;
        if (a.load_mode == IO_ROW) { l0 = tid >> a.io_tpl_log; lstep = nthr >> a.io_tpl_log; j0 = tid & ((1 << a.io_tpl_log) - 1); jstep = 1 << a.io_tpl_log; }
        else { l0 = tid & ((1 << a.lpb_log) - 1); lstep = 1 << a.lpb_log; j0 = tid >> a.lpb_log; jstep = nthr >> a.lpb_log; }
        const int n_in = a.n_in;
        const int64_t as = a.gin.axis_stride;
        for (int l = l0; l < c.lanes; l += lstep) {
            const int64_t base = c.h->off_in[l];
            for (int j = j0; j < n_in; j += jstep) {
                const int64_t g = base + (int64_t)j * as;
                if (in_cplx) {
                    cpx<T> v = ((const cpx<T> *)a.in)[g];
                    if (OP == G_C2C_INV) v.y = -v.y;
                    if (direct) c.buf[0][l * pitch + zi(j)] = v; else c.buf[1][l * pitch + j] = v;
                } else {
                    const T v = ((const T *)a.in)[g];
                    if (OP == G_R2C_ODD) c.buf[0][l * pitch + zi(j)] = mk<T>(v, (T)0);
                    else if (OP == G_R2C_EVEN) ((T *)c.buf[0])[(l * pitch + zi(j >> 1)) * 2 + (j & 1)] = v;   // packs pairs in place
                    else ((T *)c.buf[1])[l * 2 * pitch + j] = v;
                }
            }
        }
    }
    // ---- PRE -------------------------------------------------------------------------------
    if (!direct) {
        __syncthreads();
        const int F = a.F;
        for (int l = c.fl; l < c.lanes; l += c.flstep)
            for (int i = c.fj0; i < F; i += c.fstep)
                c.buf[0][l * pitch + zi(i)] = pre_elem<T, OP, ZiPad8>(a, (const void *)(c.buf[1] + l * pitch), i);
    }
    // ---- FFT -------------------------------------------------------------------------------
    int cur = 0;
    const PassGlobals<T> pg{a.in, a.out, a.n_out, a.scale};
    {
        // one call site for the (force-inlined) pass runner: plain FFT = one round; Bluestein = two rounds
        // of FFT_M with the chirp / chirp-spectrum multiplies in front:
        //   X[k] = chirp[k] * IFFT_M( FFT_M(z * chirp, zero padded) * bhat )[k]      (bhat carries 1/M)
        const int rounds = a.blue ? 2 : 1;
        const int len = a.blue ? a.M : a.F, np = a.blue ? a.npassM : a.npass;
        const int32_t *radix = a.blue ? c.h->radixM : c.h->radix;
        const cpx<T> *tw = a.blue ? a.twM : a.tw;
        for (int round = 0; round < rounds; ++round) {
            if (a.blue) {
                __syncthreads();
                const int F = a.F, M = a.M;
                for (int l = c.fl; l < c.lanes; l += c.flstep) {
                    cpx<T> *z = c.buf[cur] + l * pitch;
                    if (round == 0) for (int i = c.fj0; i < M; i += c.fstep) z[zi(i)] = i < F ? cmul(z[zi(i)], a.chirp[i]) : mk<T>((T)0, (T)0);
                    else for (int i = c.fj0; i < M; i += c.fstep) z[zi(i)] = cconj(cmul(z[zi(i)], a.bhat[i]));
                }
            }
            cur = run_passes<T, OP, BIG, INPLACE>(c, pg, cur, len, np, radix, tw, pitch, fuse_in, fuse_out);
        }
        if (a.blue) {
            const int F = a.F;
            for (int l = c.fl; l < c.lanes; l += c.flstep) {
                cpx<T> *z = c.buf[cur] + l * pitch;
                for (int i = c.fj0; i < F; i += c.fstep) z[zi(i)] = cmul(cconj(z[zi(i)]), a.chirp[i]);
            }
            __syncthreads();
        }
    }
    // ---- STORE (with POST gather) ----------------------------------------------------------
    if (!fuse_out) {
        int l0, lstep, j0, jstep;
        if (a.store_mode == IO_ROW) { l0 = tid >> a.io_tpl_log; lstep = nthr >> a.io_tpl_log; j0 = tid & ((1 << a.io_tpl_log) - 1); jstep = 1 << a.io_tpl_log; }
        else { l0 = tid & ((1 << a.lpb_log) - 1); lstep = 1 << a.lpb_log; j0 = tid >> a.lpb_log; jstep = nthr >> a.lpb_log; }
        const int n_out = a.n_out;
        const int64_t as = a.gout.axis_stride;
        for (int l = l0; l < c.lanes; l += lstep) {
            const int64_t base = c.h->off_out[l];
            const cpx<T> *res = c.buf[cur] + l * pitch;
            for (int q = j0; q < n_out; q += jstep) {
                const int64_t g = base + (int64_t)q * as;
                if (out_cplx) ((cpx<T> *)a.out)[g] = post_cplx<T, OP, ZiPad8>(a, res, q);
                else ((T *)a.out)[g] = post_real<T, OP, ZiPad8>(a, res, q);
            }
        }
    }
}


template <typename T, int OP, bool BIG, bool INPLACE = false> static int launch_op(const GenArgs<T> &a, int threads, size_t lds_bytes, hipStream_t s) {
    constexpr bool elementwise = OP == G_C2C_FWD || OP == G_C2C_INV || OP == G_R2C_EVEN || OP == G_R2C_ODD;
    if constexpr (elementwise && !INPLACE) {
        if (a.inplace) return launch_op<T, OP, BIG, true>(a, threads, lds_bytes, s);
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_generic<T, OP, BIG, INPLACE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int64_t nblk = (a.nlanes + a.lpb - 1) / a.lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    hipLaunchKernelGGL((k_generic<T, OP, BIG, INPLACE>), dim3((unsigned)nblk), dim3(threads), lds_bytes, s, a);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

template <typename T, bool BIG> int launch_generic_class(const GenArgs<T> &a, int threads, size_t lds_bytes, hipStream_t s) {
    switch (a.op) {
#define NDFFT_OPCASE(OP) case OP: return launch_op<T, OP, BIG>(a, threads, lds_bytes, s);
        NDFFT_OPCASE(G_C2C_FWD) NDFFT_OPCASE(G_C2C_INV) NDFFT_OPCASE(G_R2C_EVEN) NDFFT_OPCASE(G_R2C_ODD)
        NDFFT_OPCASE(G_C2R_EVEN) NDFFT_OPCASE(G_C2R_ODD) NDFFT_OPCASE(G_DCT1) NDFFT_OPCASE(G_DCT2_EVEN)
        NDFFT_OPCASE(G_DCT2_ODD) NDFFT_OPCASE(G_DCT3_EVEN) NDFFT_OPCASE(G_DCT3_ODD) NDFFT_OPCASE(G_DCT4_EVEN)
        NDFFT_OPCASE(G_DCT4_ODD)
#undef NDFFT_OPCASE
        default: return fail(NDFFT_ERR_INVALID_ARG, "bad generic op");
    }
}

}  // namespace ndfft
