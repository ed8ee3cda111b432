// realops.h -- op-specific folds shared by the generic LDS kernel and the register-resident real-op
// kernels: PRE builds the complex FFT input Z[i] from the raw lane, POST gathers output element q
// from the FFT result.  `Args` supplies n, F, scale, aux1, aux2; `ZI::map` is the LDS padding of the
// Z buffer the caller uses.  Reference semantics restated here:
//   R2C  (realfft forward)            lib.rs:497-503      C2R (scale, zero DC/Nyquist imag, inverse) lib.rs:506-531
//   DCT-I..IV (x2 pre-scale, rustdct) lib.rs:688-741      C2C inverse scale after                    lib.rs:321-338
#pragma once
#include "device_common.h"

namespace ndfft {

// ---------------------------------------------------------------------------------------------
// PRE: Z[i] from the raw lane (raw real lanes are addressed as T*, raw complex as cpx<T>*; raw
// lanes are NOT padded)
// ---------------------------------------------------------------------------------------------
template <typename T, typename Args>
__device__ __forceinline__ cpx<T> c2r_input(const Args &a, const cpx<T> *X, int k, int F_nyq) {
    // lib.rs:511-521: scale first, then force DC (and even-n Nyquist) imaginary parts to zero
    cpx<T> v = X[k];
    v.x *= a.scale; v.y *= a.scale;
    if (k == 0 || k == F_nyq) v.y = (T)0;
    return v;
}

template <typename T>
__device__ __forceinline__ cpx<T> herm_fold(cpx<T> a_, cpx<T> b_, cpx<T> w) {
    // Zt = (a + b) + i * conj(w) * (a - b); returns conj(Zt) (inverse FFT via forward butterflies)
    cpx<T> s = cadd(a_, b_), d = csub(a_, b_);
    cpx<T> t = cmul(d, cconj(w));
    return mk<T>(s.x - t.y, -(s.y + t.x));
}

template <typename T, int OP, typename ZI, typename Args>
__device__ __forceinline__ cpx<T> pre_elem(const Args &a, const void *raw_, int i) {
    const T *xr = (const T *)raw_;
    const cpx<T> *xc = (const cpx<T> *)raw_;
    const int n = a.n, F = a.F;
    switch (OP) {
        case G_C2R_EVEN: {
            cpx<T> A = c2r_input<T>(a, xc, i, F), B = cconj(c2r_input<T>(a, xc, F - i, F));
            return herm_fold<T>(A, B, a.aux1[i]);
        }
        case G_C2R_ODD: {
            const int m = n / 2 + 1;
            cpx<T> v = c2r_input<T>(a, xc, i < m ? i : n - i, -1);
            // full spectrum value is v (i<m) or conj(v); we store its conjugate
            return i < m ? cconj(v) : v;
        }
        case G_DCT1: {   // even extension of length L = 2(n-1), packed two reals per complex
            const int L = 2 * (n - 1), j0 = 2 * i, j1 = 2 * i + 1;
            T e0 = xr[j0 < n ? j0 : L - j0], e1 = xr[j1 < n ? j1 : L - j1];
            return mk<T>(e0 * a.scale, e1 * a.scale);
        }
        case G_DCT2_EVEN: {   // Makhoul: v[p] = x[2p] (p < n/2), v[p] = x[2(n-1-p)+1] otherwise
            const int h = n / 2, p0 = 2 * i, p1 = 2 * i + 1;
            T v0 = xr[p0 < h ? 2 * p0 : 2 * (n - 1 - p0) + 1], v1 = xr[p1 < h ? 2 * p1 : 2 * (n - 1 - p1) + 1];
            return mk<T>(v0 * a.scale, v1 * a.scale);
        }
        case G_DCT2_ODD: {
            const int h = (n + 1) / 2;
            return mk<T>(xr[i < h ? 2 * i : 2 * (n - 1 - i) + 1] * a.scale, (T)0);
        }
        case G_DCT3_EVEN: {
            // V[k] = 0.5 (x[k] - i x[n-k]) e^{+i pi k/(2n)}, k in [0,F], x[n] := 0 ; then Hermitian fold
            const int k0 = i, k1 = F - i;
            const T hs = (T)0.5 * a.scale;
            cpx<T> v0 = cmul(mk<T>(xr[k0] * hs, k0 ? -xr[n - k0] * hs : (T)0), cconj(a.aux2[k0]));
            cpx<T> v1 = cmul(mk<T>(xr[k1] * hs, -xr[n - k1] * hs), cconj(a.aux2[k1]));   // k1 >= 1 always
            return herm_fold<T>(v0, cconj(v1), a.aux1[i]);
        }
        case G_DCT3_ODD: {
            const T hs = (T)0.5 * a.scale;
            cpx<T> v = cmul(mk<T>(xr[i] * hs, i ? -xr[n - i] * hs : (T)0), cconj(a.aux2[i]));
            return cconj(v);
        }
        case G_DCT4_EVEN:
            return cmul(mk<T>(xr[2 * i] * a.scale, xr[n - 1 - 2 * i] * a.scale), a.aux1[i]);
        case G_DCT4_ODD: {
            if (i >= n) return mk<T>((T)0, (T)0);
            const T x = xr[i] * a.scale;
            return mk<T>(x * a.aux1[i].x, x * a.aux1[i].y);
        }
        default: return mk<T>((T)0, (T)0);
    }
}

// ---------------------------------------------------------------------------------------------
// POST: output element q from the FFT result `res` (length F, padded by zi)
// ---------------------------------------------------------------------------------------------
template <typename T, typename ZI>
__device__ __forceinline__ cpx<T> r2c_split(const cpx<T> *res, int k, int F, cpx<T> w) {
    // X[k] = (Z[k] + conj Z[F-k])/2 + w (Z[k] - conj Z[F-k])/(2i)
    cpx<T> A = res[ZI::map(k == F ? 0 : k)], B = cconj(res[ZI::map(k == 0 ? 0 : F - k)]);
    cpx<T> e = mk<T>((A.x + B.x) * (T)0.5, (A.y + B.y) * (T)0.5);
    cpx<T> o = mk<T>((A.y - B.y) * (T)0.5, -(A.x - B.x) * (T)0.5);
    return cadd(e, cmul(o, w));
}

// both halves of the real-FFT split from ONE pair of reads:  X[k] = E + T,  X[F-k] = conj(E - T),
// E = (Z[k] + conj Z[F-k])/2,  T = W_{2F}^k (Z[k] - conj Z[F-k])/(2i)     (W^{F-k} = -conj W^k)
template <typename T, typename ZI>
__device__ __forceinline__ void r2c_split_pair(const cpx<T> *res, int k, int F, cpx<T> w, cpx<T> &xk, cpx<T> &xfk) {
    cpx<T> A = res[ZI::map(k == F ? 0 : k)], B = cconj(res[ZI::map(k == 0 ? 0 : F - k)]);
    cpx<T> e = mk<T>((A.x + B.x) * (T)0.5, (A.y + B.y) * (T)0.5);
    cpx<T> o = mk<T>((A.y - B.y) * (T)0.5, -(A.x - B.x) * (T)0.5);
    cpx<T> t = cmul(o, w);
    xk = cadd(e, t);
    xfk = cconj(csub(e, t));
}

template <typename T, int OP, typename ZI, typename Args> __device__ __forceinline__ T post_real(const Args &a, const cpx<T> *res, int q) {
    const int n = a.n, F = a.F;
    switch (OP) {
        case G_C2R_EVEN: { cpx<T> c = res[ZI::map(q >> 1)]; return (q & 1) ? -c.y : c.x; }
        case G_C2R_ODD: return res[ZI::map(q)].x;
        case G_DCT1: return (T)0.5 * r2c_split<T, ZI>(res, q, F, a.aux1[q]).x;
        case G_DCT2_EVEN: {
            const int k = q <= F ? q : n - q;
            cpx<T> t = cmul(r2c_split<T, ZI>(res, k, F, a.aux1[k]), a.aux2[k]);
            return q <= F ? t.x : -t.y;
        }
        case G_DCT2_ODD: { cpx<T> t = cmul(res[ZI::map(q)], a.aux2[q]); return t.x; }
        case G_DCT3_EVEN: {
            const int p = (q & 1) ? n - 1 - (q >> 1) : (q >> 1);
            cpx<T> c = res[ZI::map(p >> 1)];
            return (p & 1) ? -c.y : c.x;
        }
        case G_DCT3_ODD: { const int p = (q & 1) ? n - 1 - (q >> 1) : (q >> 1); return res[ZI::map(p)].x; }
        case G_DCT4_EVEN: {
            const int k = (q & 1) ? (n - 1 - q) >> 1 : q >> 1;
            cpx<T> u = cmul(res[ZI::map(k)], a.aux2[k]);
            return (q & 1) ? -u.y : u.x;
        }
        case G_DCT4_ODD: { cpx<T> u = cmul(res[ZI::map(q)], a.aux2[q]); return u.x; }
        default: return (T)0;
    }
}

template <typename T, int OP, typename ZI, typename Args> __device__ __forceinline__ cpx<T> post_cplx(const Args &a, const cpx<T> *res, int q) {
    switch (OP) {
        case G_C2C_INV: { cpx<T> c = res[ZI::map(q)]; return mk<T>(c.x * a.scale, -c.y * a.scale); }   // lib.rs:326-330
        case G_R2C_EVEN: return r2c_split<T, ZI>(res, q, a.F, a.aux1[q]);
        default: return res[ZI::map(q)];   // G_C2C_FWD, G_R2C_ODD
    }
}

}  // namespace ndfft
