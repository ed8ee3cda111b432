// big.hip -- pieces of the four-step path for lanes longer than one workgroup's LDS
// (the reference accepts any n: FftHandler::new(n) is infallible, src/lib.rs:294).
//   F = F1 * F2 :  X[k1 + F1 k2] = sum_{n2} W_F2^{n2 k2} ( W_F^{n2 k1} sum_{n1} x[n1 F2 + n2] W_F1^{n1 k1} )
// = transpose, batched length-F1 row FFTs, twiddle, transpose, batched length-F2 row FFTs, transpose,
// all on the existing row kernels (exec.hip: big_fft).  The real-data ops add an elementwise PRE and
// POST over global memory that reuse realops.h.
#include <algorithm>

#include "pow2_real.h"

namespace ndfft {

template <typename T>
__global__ __launch_bounds__(256) void k_big_twiddle(cpx<T> *data, int64_t total, int F1, int F2, const cpx<T> *twlo,
                                                     const cpx<T> *twhi, int logB, int conj, T scale) {
    // data[lane][n2][k1] *= W_F^{n2 k1} * scale
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i % ((int64_t)F1 * F2);
        const int64_t n2 = r / F1, k1 = r - n2 * F1, m = n2 * k1;
        cpx<T> w = cmul(twhi[m >> logB], twlo[m & (((int64_t)1 << logB) - 1)]);
        if (conj) w.y = -w.y;
        cpx<T> v = cmul(data[i], w);
        v.x *= scale; v.y *= scale;
        data[i] = v;
    }
}

template <typename T>
int launch_big_twiddle(cpx<T> *data, int64_t lanes, int F1, int F2, const cpx<T> *twlo, const cpx<T> *twhi, int logB, int conj,
                       T scale, hipStream_t s) {
    const int64_t total = lanes * F1 * F2;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(k_big_twiddle<T>, dim3(grid), dim3(256), 0, s, data, total, F1, F2, twlo, twhi, logB, conj, scale);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template int launch_big_twiddle<float>(cpx<float> *, int64_t, int, int, const cpx<float> *, const cpx<float> *, int, int, float, hipStream_t);
template int launch_big_twiddle<double>(double2 *, int64_t, int, int, const double2 *, const double2 *, int, int, double, hipStream_t);

// Bluestein over global memory (FftConfig::bigblue), the three elementwise stages around two FFT_M:
//   0: a[l][j] = z[l][j] * chirp[j] (j < F; conj(z) for the inverse), 0 for F <= j < M
//   1: b[l][k] = conj(a[l][k] * bhat[k])                                  (in place)
//   2: out[l][k] = conj(b[l][k]) * chirp[k], k < F  (inverse: its conjugate, * scale)
template <typename T>
__global__ __launch_bounds__(256) void k_blue_stage(int stage, cpx<T> *dst, int64_t pitch_dst, const cpx<T> *src, int64_t pitch_src, int64_t lanes,
                                                    int F, int M, const cpx<T> *chirp, const cpx<T> *bhat, int inverse, T scale) {
    const int len = stage == 2 ? F : M;
    const int64_t total = lanes * len;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t l = i / len; const int j = (int)(i - l * len);
        if (stage == 0) {
            cpx<T> v = mk<T>((T)0, (T)0);
            if (j < F) { v = src[l * pitch_src + j]; if (inverse) v.y = -v.y; v = cmul(v, chirp[j]); }
            dst[l * pitch_dst + j] = v;
        } else if (stage == 1) {
            dst[l * pitch_dst + j] = cconj(cmul(src[l * pitch_src + j], bhat[j]));
        } else {
            cpx<T> v = cmul(cconj(src[l * pitch_src + j]), chirp[j]);
            if (inverse) v.y = -v.y;
            v.x *= scale; v.y *= scale;
            dst[l * pitch_dst + j] = v;
        }
    }
}
template <typename T>
int launch_blue_stage(int stage, cpx<T> *dst, int64_t pitch_dst, const cpx<T> *src, int64_t pitch_src, int64_t lanes, int F, int M,
                      const cpx<T> *chirp, const cpx<T> *bhat, int inverse, T scale, hipStream_t s) {
    const int64_t total = lanes * (stage == 2 ? F : M);
    if (total <= 0) return NDFFT_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(k_blue_stage<T>, dim3(grid), dim3(256), 0, s, stage, dst, pitch_dst, src, pitch_src, lanes, F, M, chirp, bhat, inverse, scale);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template int launch_blue_stage<float>(int, cpx<float> *, int64_t, const cpx<float> *, int64_t, int64_t, int, int, const cpx<float> *, const cpx<float> *, int, float, hipStream_t);
template int launch_blue_stage<double>(int, double2 *, int64_t, const double2 *, int64_t, int64_t, int, int, const double2 *, const double2 *, int, double, hipStream_t);

template <typename T, int OP> __global__ __launch_bounds__(256) void k_big_pre(const RealArgs<T> a, cpx<T> *z, int conj_z) {
    constexpr bool in_cplx = OP == G_C2R_EVEN || OP == G_C2R_ODD;
    const int64_t total = a.nlanes * a.F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t lane = i / a.F; const int k = (int)(i - lane * a.F);
        const void *raw = in_cplx ? (const void *)((const cpx<T> *)a.in + lane * a.pitch_in) : (const void *)((const T *)a.in + lane * a.pitch_in);
        cpx<T> v;
        if constexpr (OP == G_R2C_EVEN) v = ((const cpx<T> *)raw)[k];
        else if constexpr (OP == G_R2C_ODD) v = mk<T>(((const T *)raw)[k], (T)0);
        else v = pre_elem<T, OP, ZiNone>(a, raw, k);
        if (conj_z) v.y = -v.y;
        z[i] = v;
    }
}
template <typename T, int OP> __global__ __launch_bounds__(256) void k_big_post(const RealArgs<T> a, const cpx<T> *z) {
    constexpr bool out_cplx = OP == G_R2C_EVEN || OP == G_R2C_ODD;
    const int64_t total = a.nlanes * a.n_out;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t lane = i / a.n_out; const int q = (int)(i - lane * a.n_out);
        const cpx<T> *res = z + lane * a.F;
        if constexpr (out_cplx) ((cpx<T> *)a.out)[lane * a.pitch_out + q] = post_cplx<T, OP, ZiNone>(a, res, q);
        else ((T *)a.out)[lane * a.pitch_out + q] = post_real<T, OP, ZiNone>(a, res, q);
    }
}

// Long lanes in an arbitrary strided layout (neither unit-stride lanes nor a C-layout block): pack the lanes into
// a dense [lane][len] scratch array / unpack them from one.  Element size esz = 4, 8 or 16 bytes.
template <typename E>
__global__ __launch_bounds__(256) void k_pack_lanes(const E *strided, E *dense, LaneGeom g, int64_t lanes, int64_t len, int64_t pitch, int unpack) {
    const int64_t total = lanes * len;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t lane = i / len; const int64_t j = i - lane * len;
        int64_t off = j * g.axis_stride;
        const int64_t d0 = lane * pitch + j;
        for (int b = g.nb - 1; b >= 0; --b) { const int64_t c = lane % g.bshape[b]; lane /= g.bshape[b]; off += c * g.bstride[b]; }
        if (unpack) ((E *)strided)[off] = dense[d0]; else dense[d0] = strided[off];
    }
}
int launch_pack_lanes(const void *strided, void *dense, const LaneGeom &g, int64_t lanes, int64_t len, int64_t pitch, int esz, int unpack, hipStream_t s) {
    const int64_t total = lanes * len;
    if (total <= 0) return NDFFT_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    if (esz == 4) hipLaunchKernelGGL(k_pack_lanes<float>, dim3(grid), dim3(256), 0, s, (const float *)strided, (float *)dense, g, lanes, len, pitch, unpack);
    else if (esz == 8) hipLaunchKernelGGL(k_pack_lanes<double>, dim3(grid), dim3(256), 0, s, (const double *)strided, (double *)dense, g, lanes, len, pitch, unpack);
    else hipLaunchKernelGGL(k_pack_lanes<double2>, dim3(grid), dim3(256), 0, s, (const double2 *)strided, (double2 *)dense, g, lanes, len, pitch, unpack);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

// POST for the ops whose outputs come in spectrum pairs (k, F - k): R2C, DCT-I, DCT-II.  One thread reads Z[k] and Z[F - k] ONCE and writes every output
// derived from them (k_big_post above reads each Z element twice and, for DCT-II, multiplies each twiddle twice: 99 us for 64 x 262144 f64, this form: see DESIGN 3.5).
template <typename T, int OP> __global__ __launch_bounds__(256) void k_big_post_pair(const RealArgs<T> a, const cpx<T> *z) {
    const int F = a.F, half = F / 2 + 1;
    const int64_t total = a.nlanes * half;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t lane = i / half; const int k = (int)(i - lane * half), kf = F - k;
        const cpx<T> *res = z + lane * F;
        cpx<T> xk, xf;
        r2c_split_pair<T, ZiNone>(res, k, F, a.aux1[k], xk, xf);
        if constexpr (OP == G_R2C_EVEN) {
            cpx<T> *out = (cpx<T> *)a.out + lane * a.pitch_out;
            out[k] = xk;
            if (kf != k) out[kf] = xf;
        } else if constexpr (OP == G_DCT1) {
            T *out = (T *)a.out + lane * a.pitch_out;
            out[k] = (T)0.5 * xk.x;
            if (kf != k) out[kf] = (T)0.5 * xf.x;
        } else {   // G_DCT2_EVEN: y[k] = Re(X[k] c_k), y[n - k] = -Im(X[k] c_k)
            T *out = (T *)a.out + lane * a.pitch_out;
            const int n = 2 * F;
            const cpx<T> tk = cmul(xk, a.aux2[k]);
            out[k] = tk.x;
            if (k > 0) out[n - k] = -tk.y;
            if (kf != k) {
                const cpx<T> tf = cmul(xf, a.aux2[kf]);
                out[kf] = tf.x;
                if (kf < F) out[n - kf] = -tf.y;
            }
        }
    }
}

#define NDFFT_BIG_OPS(X) X(G_R2C_EVEN) X(G_R2C_ODD) X(G_C2R_EVEN) X(G_C2R_ODD) X(G_DCT1) X(G_DCT2_EVEN) X(G_DCT2_ODD) \
    X(G_DCT3_EVEN) X(G_DCT3_ODD) X(G_DCT4_EVEN) X(G_DCT4_ODD)

template <typename T> int launch_big_pre(int op, const RealArgs<T> &a, cpx<T> *z, hipStream_t s, int conj_z) {
    const unsigned grid = (unsigned)std::min<int64_t>((a.nlanes * a.F + 255) / 256, 8192);
    switch (op) {
#define NDFFT_C(OP) case OP: hipLaunchKernelGGL((k_big_pre<T, OP>), dim3(grid), dim3(256), 0, s, a, z, conj_z); break;
        NDFFT_BIG_OPS(NDFFT_C)
#undef NDFFT_C
        default: return fail(NDFFT_ERR_INVALID_ARG, "big pre: bad op");
    }
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template <typename T> int launch_big_post(int op, const RealArgs<T> &a, const cpx<T> *z, hipStream_t s) {
    if (op == G_R2C_EVEN || op == G_DCT1 || op == G_DCT2_EVEN) {
        const unsigned gp = (unsigned)std::min<int64_t>((a.nlanes * (a.F / 2 + 1) + 255) / 256, 16384);
        if (op == G_R2C_EVEN) hipLaunchKernelGGL((k_big_post_pair<T, G_R2C_EVEN>), dim3(gp), dim3(256), 0, s, a, z);
        else if (op == G_DCT1) hipLaunchKernelGGL((k_big_post_pair<T, G_DCT1>), dim3(gp), dim3(256), 0, s, a, z);
        else hipLaunchKernelGGL((k_big_post_pair<T, G_DCT2_EVEN>), dim3(gp), dim3(256), 0, s, a, z);
        NDFFT_HIP(hipGetLastError());
        return NDFFT_OK;
    }
    const unsigned grid = (unsigned)std::min<int64_t>((a.nlanes * a.n_out + 255) / 256, 8192);
    switch (op) {
#define NDFFT_C(OP) case OP: hipLaunchKernelGGL((k_big_post<T, OP>), dim3(grid), dim3(256), 0, s, a, z); break;
        NDFFT_BIG_OPS(NDFFT_C)
#undef NDFFT_C
        default: return fail(NDFFT_ERR_INVALID_ARG, "big post: bad op");
    }
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
template int launch_big_pre<float>(int, const RealArgs<float> &, cpx<float> *, hipStream_t, int);
template int launch_big_pre<double>(int, const RealArgs<double> &, double2 *, hipStream_t, int);
template int launch_big_post<float>(int, const RealArgs<float> &, const cpx<float> *, hipStream_t);
template int launch_big_post<double>(int, const RealArgs<double> &, const double2 *, hipStream_t);

}  // namespace ndfft
