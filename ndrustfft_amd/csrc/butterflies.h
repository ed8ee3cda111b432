// butterflies.h -- forward radix-R DFT butterflies on register arrays (device only).
// Inverse transforms never need their own butterflies: ifft(x) = conj(fft(conj(x))) and the
// conjugations are folded into the global load / store of the kernels.
#pragma once
#include "device_common.h"

namespace ndfft {

template <int R> struct OddTab;
// cos / sin of 2 pi k / R in constant memory: folded to literals where the index is known after unrolling,
// a scalar load otherwise (a function-local table would live in scratch)
#define NDFFT_ODDTAB(R, ...)                                   \
    static __constant__ const double kUnitCos##R[R] = __VA_ARGS__;
#define NDFFT_ODDTAB_S(R, ...)                                 \
    static __constant__ const double kUnitSin##R[R] = __VA_ARGS__; \
    template <> struct OddTab<R> {                             \
        static __device__ __forceinline__ double c(int k) { return kUnitCos##R[k]; } \
        static __device__ __forceinline__ double s(int k) { return kUnitSin##R[k]; } \
    };

NDFFT_ODDTAB(3, {1.0, -0.5, -0.5})
NDFFT_ODDTAB_S(3, {0.0, 0.8660254037844386467637, -0.8660254037844386467637})
NDFFT_ODDTAB(5, {1.0, 0.3090169943749474241023, -0.8090169943749474241023, -0.8090169943749474241023, 0.3090169943749474241023})
NDFFT_ODDTAB_S(5, {0.0, 0.9510565162951535721164, 0.5877852522924731291687, -0.5877852522924731291687, -0.9510565162951535721164})
NDFFT_ODDTAB(7, {1.0, 0.623489801858733530525, -0.2225209339563144042889, -0.9009688679024191262361, -0.9009688679024191262361, -0.2225209339563144042889, 0.623489801858733530525})
NDFFT_ODDTAB_S(7, {0.0, 0.7818314824680298087084, 0.9749279121818236070181, 0.4338837391175581204758, -0.4338837391175581204758, -0.9749279121818236070181, -0.7818314824680298087084})
NDFFT_ODDTAB(11, {1.0, 0.8412535328311811688618, 0.4154150130018864255293, -0.1423148382732851404438, -0.6548607339452850640569, -0.9594929736144973898904, -0.9594929736144973898904, -0.6548607339452850640569, -0.1423148382732851404438, 0.4154150130018864255293, 0.8412535328311811688618})
NDFFT_ODDTAB_S(11, {0.0, 0.5406408174555975821076, 0.9096319953545183714117, 0.9898214418809327323761, 0.755749574354258283774, 0.2817325568414296977114, -0.2817325568414296977114, -0.755749574354258283774, -0.9898214418809327323761, -0.9096319953545183714117, -0.5406408174555975821076})
NDFFT_ODDTAB(13, {1.0, 0.8854560256532098959004, 0.5680647467311558025118, 0.1205366802553230533491, -0.3546048870425356259696, -0.7485107481711010986346, -0.970941817426052027157, -0.970941817426052027157, -0.7485107481711010986346, -0.3546048870425356259696, 0.1205366802553230533491, 0.5680647467311558025118, 0.8854560256532098959004})
NDFFT_ODDTAB_S(13, {0.0, 0.464723172043768545656, 0.8229838658936563945796, 0.9927088740980539928008, 0.9350162426854148234398, 0.6631226582407952023768, 0.2393156642875577671488, -0.2393156642875577671488, -0.6631226582407952023768, -0.9350162426854148234398, -0.9927088740980539928008, -0.8229838658936563945796, -0.464723172043768545656})

// unit roots for the composite radices (same c/s tables, used as inner twiddles by BflyComp)
NDFFT_ODDTAB(6, {1.0, 0.5, -0.5, -1.0, -0.5, 0.5})
NDFFT_ODDTAB_S(6, {0.0, 0.8660254037844386467637, 0.8660254037844386467637, 0.0, -0.8660254037844386467637, -0.8660254037844386467637})
NDFFT_ODDTAB(9, {1.0, 0.7660444431189780352024, 0.1736481776669303488517, -0.5, -0.9396926207859083840541, -0.9396926207859083840541, -0.5, 0.1736481776669303488517, 0.7660444431189780352024})
NDFFT_ODDTAB_S(9, {0.0, 0.6427876096865393263226, 0.9848077530122080593667, 0.8660254037844386467637, 0.3420201433256687330441, -0.3420201433256687330441, -0.8660254037844386467637, -0.9848077530122080593667, -0.6427876096865393263226})
NDFFT_ODDTAB(10, {1.0, 0.8090169943749474241023, 0.3090169943749474241023, -0.3090169943749474241023, -0.8090169943749474241023, -1.0, -0.8090169943749474241023, -0.3090169943749474241023, 0.3090169943749474241023, 0.8090169943749474241023})
NDFFT_ODDTAB_S(10, {0.0, 0.5877852522924731291687, 0.9510565162951535721164, 0.9510565162951535721164, 0.5877852522924731291687, 0.0, -0.5877852522924731291687, -0.9510565162951535721164, -0.9510565162951535721164, -0.5877852522924731291687})
NDFFT_ODDTAB(12, {1.0, 0.8660254037844386467637, 0.5, 0.0, -0.5, -0.8660254037844386467637, -1.0, -0.8660254037844386467637, -0.5, 0.0, 0.5, 0.8660254037844386467637})
NDFFT_ODDTAB_S(12, {0.0, 0.5, 0.8660254037844386467637, 1.0, 0.8660254037844386467637, 0.5, 0.0, -0.5, -0.8660254037844386467637, -1.0, -0.8660254037844386467637, -0.5})

template <typename T> __device__ __forceinline__ void bfly2(cpx<T> &a, cpx<T> &b) {
    cpx<T> t = a; a = cadd(t, b); b = csub(t, b);
}

// forward DFT-4, in place, natural order out
template <typename T> __device__ __forceinline__ void bfly4(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d) {
    cpx<T> s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = cmul_mi(csub(b, d));
    a = cadd(s0, s2); b = cadd(s1, s3); c = csub(s0, s2); d = csub(s1, s3);
}

template <typename T, int R> struct Bfly;

template <typename T> struct Bfly<T, 2> {
    static __device__ __forceinline__ void run(cpx<T> *v) { bfly2<T>(v[0], v[1]); }
};
template <typename T> struct Bfly<T, 4> {
    static __device__ __forceinline__ void run(cpx<T> *v) { bfly4<T>(v[0], v[1], v[2], v[3]); }
};
template <typename T> struct Bfly<T, 8> {
    static __device__ __forceinline__ void run(cpx<T> *v) {
        const T h = (T)0.70710678118654752440084436210485;
        bfly4<T>(v[0], v[2], v[4], v[6]);   // E0..E3 in v0,v2,v4,v6
        bfly4<T>(v[1], v[3], v[5], v[7]);   // O0..O3 in v1,v3,v5,v7
        cpx<T> o1 = mk<T>((v[3].x + v[3].y) * h, (v[3].y - v[3].x) * h);      // O1 * (1-i)/sqrt2
        cpx<T> o2 = cmul_mi(v[5]);                                             // O2 * -i
        cpx<T> o3 = mk<T>((v[7].y - v[7].x) * h, -(v[7].x + v[7].y) * h);     // O3 * (-1-i)/sqrt2
        cpx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
        v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
        v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
        v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
    }
};

// forward DFT-16 = 4 x DFT-4 (stride 4), internal twiddles W16^{jk}, 4 x DFT-4; natural order out
template <typename T> struct Bfly<T, 16> {
    static __device__ __forceinline__ void run(cpx<T> *v) {
        const T h = (T)0.70710678118654752440084436210485;
        const T c1 = (T)0.92387953251128675612818318939679, s1 = (T)0.38268343236508977172845998403040;
#pragma unroll
        for (int j = 0; j < 4; ++j) bfly4<T>(v[j], v[j + 4], v[j + 8], v[j + 12]);
        // element (j, k) = v[j + 4k] gets W16^{jk}
        // j = 1: W^1, W^2, W^3 ; j = 2: W^2, W^4, W^6 ; j = 3: W^3, W^6, W^9
        v[5] = cmul(v[5], mk<T>(c1, -s1));
        v[9] = mk<T>((v[9].x + v[9].y) * h, (v[9].y - v[9].x) * h);
        v[13] = cmul(v[13], mk<T>(s1, -c1));
        v[6] = mk<T>((v[6].x + v[6].y) * h, (v[6].y - v[6].x) * h);
        v[10] = cmul_mi(v[10]);
        v[14] = mk<T>((v[14].y - v[14].x) * h, -(v[14].x + v[14].y) * h);
        v[7] = cmul(v[7], mk<T>(s1, -c1));
        v[11] = mk<T>((v[11].y - v[11].x) * h, -(v[11].x + v[11].y) * h);
        v[15] = cmul(v[15], mk<T>(-c1, s1));
#pragma unroll
        for (int k = 0; k < 4; ++k) bfly4<T>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
        // now v[4k + q] holds X[k + 4q]; permute to natural order (4x4 transpose)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = k + 1; q < 4; ++q) { cpx<T> t = v[4 * k + q]; v[4 * k + q] = v[4 * q + k]; v[4 * q + k] = t; }
    }
};

// odd prime radices: pair (q, R-q) shares a_r = x_r + x_{R-r}, b_r = x_r - x_{R-r}
template <typename T, int R> struct BflyOdd {
    static __device__ __forceinline__ void run(cpx<T> *v) {
        constexpr int H = (R - 1) / 2;
        cpx<T> a[H + 1], b[H + 1];
        const cpx<T> x0 = v[0];
        cpx<T> y0 = x0;
#pragma unroll
        for (int r = 1; r <= H; ++r) { a[r] = cadd(v[r], v[R - r]); b[r] = csub(v[r], v[R - r]); y0 = cadd(y0, a[r]); }
        v[0] = y0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            T cr = x0.x, ci = x0.y, sr = 0, si = 0;
#pragma unroll
            for (int r = 1; r <= H; ++r) {
                const T c = (T)OddTab<R>::c((r * q) % R), s = (T)OddTab<R>::s((r * q) % R);
                cr += a[r].x * c; ci += a[r].y * c; sr += b[r].x * s; si += b[r].y * s;
            }
            v[q] = mk<T>(cr + si, ci - sr);
            v[R - q] = mk<T>(cr - si, ci + sr);
        }
    }
};
template <typename T> struct Bfly<T, 3> : BflyOdd<T, 3> {};
template <typename T> struct Bfly<T, 5> : BflyOdd<T, 5> {};
template <typename T> struct Bfly<T, 7> : BflyOdd<T, 7> {};
template <typename T> struct Bfly<T, 11> : BflyOdd<T, 11> {};
template <typename T> struct Bfly<T, 13> : BflyOdd<T, 13> {};

// composite radix R = R1*R2 in registers (Cooley-Tukey inside one butterfly), natural order in and out:
//   X[k1 + R1 k2] = sum_{n2} W_R2^{n2 k2} ( W_R^{n2 k1} sum_{n1} x[n1 R2 + n2] W_R1^{n1 k1} )
// fewer Stockham passes (1000 = 10*10*10 instead of 8*5*5*5) = fewer LDS round trips and barriers
template <typename T, int R1, int R2> struct BflyComp {
    static __device__ __forceinline__ void run(cpx<T> *v) {
        constexpr int R = R1 * R2;
        cpx<T> a[R];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) {
            cpx<T> t[R1];
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) t[n1] = v[n1 * R2 + n2];
            Bfly<T, R1>::run(t);
#pragma unroll
            for (int k1 = 0; k1 < R1; ++k1) {
                const int m = (n2 * k1) % R;
                if (m == 0) a[k1 * R2 + n2] = t[k1];
                else a[k1 * R2 + n2] = cmul(t[k1], mk<T>((T)OddTab<R>::c(m), -(T)OddTab<R>::s(m)));
            }
        }
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            cpx<T> t[R2];
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[k1 * R2 + n2];
            Bfly<T, R2>::run(t);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = t[k2];
        }
    }
};
template <typename T> struct Bfly<T, 6> : BflyComp<T, 2, 3> {};
template <typename T> struct Bfly<T, 9> : BflyComp<T, 3, 3> {};
template <typename T> struct Bfly<T, 10> : BflyComp<T, 2, 5> {};
template <typename T> struct Bfly<T, 12> : BflyComp<T, 4, 3> {};

}  // namespace ndfft
