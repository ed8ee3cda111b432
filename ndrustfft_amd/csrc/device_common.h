// device_common.h -- the device-side vocabulary shared by every kernel header.  It is also the first of
// the headers handed to hiprtc when a register-resident kernel is specialised at plan time (jit.hip), so it
// must compile WITHOUT host headers: under __HIPCC_RTC__ nothing is included.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef __hip_internal::int64_t int64_t;   // hiprtc has no <stdint.h>; same widths as the host's
typedef __hip_internal::int32_t int32_t;
typedef __hip_internal::uint32_t uint32_t;
typedef __hip_internal::uint64_t uint64_t;
#endif

namespace ndfft {

// ------------------------------------------------------------------------------------------
// complex helpers (Complex<T> = {re, im} interleaved = float2 / double2)
// ------------------------------------------------------------------------------------------
template <typename T> struct vec2;
// f32 complex is a NATIVE 2-vector (same layout as float2): component-wise arithmetic then compiles to packed
// math (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, two floats per lane per issue).  A wave64 VALU instruction
// takes 4 cycles on the 16-lane SIMDs and the f32 kernels are issue-bound before they are HBM-bound.
typedef float ndfft_c32 __attribute__((ext_vector_type(2)));
template <> struct vec2<float> { using type = ndfft_c32; };
template <> struct vec2<double> { using type = double2; };
template <typename T> using cpx = typename vec2<T>::type;

template <typename T> __host__ __device__ inline cpx<T> mk(T a, T b) { cpx<T> r; r.x = a; r.y = b; return r; }
template <typename C> __host__ __device__ inline C cadd(C a, C b) { a.x += b.x; a.y += b.y; return a; }
template <typename C> __host__ __device__ inline C csub(C a, C b) { a.x -= b.x; a.y -= b.y; return a; }
template <typename C> __host__ __device__ inline C cmul(C a, C b) {
    C r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r;
}
__host__ __device__ inline ndfft_c32 cadd(ndfft_c32 a, ndfft_c32 b) { return a + b; }
__host__ __device__ inline ndfft_c32 csub(ndfft_c32 a, ndfft_c32 b) { return a - b; }
__host__ __device__ inline ndfft_c32 cmul(ndfft_c32 a, ndfft_c32 b) { const ndfft_c32 bs = {-b.y, b.x}; return a.xx * b + a.yy * bs; }
template <typename C> __host__ __device__ inline C cconj(C a) { a.y = -a.y; return a; }
// multiply by -i (forward quarter turn): (x, y) -> (y, -x)
template <typename C> __host__ __device__ inline C cmul_mi(C a) { C r; r.x = a.y; r.y = -a.x; return r; }

// internal op codes of the generic kernel = public ndfft_op, with parity variants resolved on host
enum GenOp : int {
    G_C2C_FWD = 0, G_C2C_INV,
    G_R2C_EVEN, G_R2C_ODD, G_C2R_EVEN, G_C2R_ODD,
    G_DCT1,               // n >= 2, F = n-1
    G_DCT2_EVEN, G_DCT2_ODD, G_DCT3_EVEN, G_DCT3_ODD,
    G_DCT4_EVEN, G_DCT4_ODD
};

// arguments of the register-resident C2C row kernels (pow2_kernel.h)
struct Pow2Args {
    const void *in; void *out;
    int64_t nlanes;
    int64_t pitch_in, pitch_out;   // elements between consecutive lanes
    int32_t inverse;
    double scale;
    const void *twp;
    // four-step second stage: element i of lane L is first multiplied by W_F^{i * (L % f1)} = twhi[m >> logB] * twlo[m & (2^logB - 1)]
    const void *twlo = nullptr, *twhi = nullptr;
    int32_t logB = 0, f1 = 1;
    // XCD-aware workgroup -> lane-block map (xcd_block): 0 = identity
    int32_t xcd_chunk = 0;
    // load policy of the input: -1 = the launcher decides by size, 0 = default policy (the input is expected in the Infinity Cache),
    // 1 = streaming (nt) loads (the input comes from HBM) -- exec.hip: MallModel
    int32_t stream_in = -1;
};

// arguments of the LDS-free wavefront kernel for short dense C2C lanes (wave_kernel.h)
struct WaveArgs {
    const void *in; void *out;
    int64_t total;            // complex elements = lanes * n (lanes are dense: pitch == n)
    int32_t inverse;
    double scale;
    const void *tw;           // tw[k] = e^{-2 pi i k / n}, k < n
    int32_t xcd_chunk;        // XCD-aware workgroup -> chunk map (xcd_block below), 0 = identity
};

// arguments of the thread-per-lane kernel for very short C2C lanes (tiny_kernel.h)
struct TinyArgs {
    const void *in; void *out;
    int64_t nlanes;
    int64_t inner;                     // lane L = (o, i) = (L / inner, L % inner)
    int64_t outer_in, outer_out;       // stride of o (elements)
    int64_t lane_in, lane_out;         // stride of i
    int64_t elem_in, elem_out;         // stride of the element index j
    int32_t inverse;
    double scale;
    const void *mat = nullptr;         // tinymat_kernel.h: the transform as a dense real matrix, NO x NI, row-major
};

// arguments of the thread-per-lane real-op register kernel (reg_kernel.h: RegReal)
struct RegRealArgs {
    TinyArgs t;                      // layout, scale (t.mat = W_F^k table of the inner FFT)
    const void *aux1, *aux2;         // op tables of the plan slot (plan.hip)
};

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8; MI355X_MICROARCH.md).  With the identity map every
// XCD touches every eighth lane of the array: 8 interleaved streams per 512 KiB of addresses, and every XCD's L2 /
// TLB sees every page.  xcd_block() hands XCD x, out of each group of 8 C consecutive lane blocks, the C CONTIGUOUS
// blocks [x C, (x + 1) C): each XCD then streams whole multi-megabyte runs (measured cache-cold on 2 GiB arrays,
// tools/coldcopy.hip: 5.2 -> 5.8 TB/s for the copy with the kernel's access shape, 6.0 with streaming loads).  Blocks
// past the last whole group keep the identity map.  A speed choice only: any bijection gives the same results.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblk, int chunk) {
    if (chunk <= 0) return b;
    const unsigned grp = 8u * (unsigned)chunk;
    const unsigned g = b / grp;
    if ((g + 1) * grp > nblk) return b;
    const unsigned r = b - g * grp;
    return g * grp + (r & 7u) * (unsigned)chunk + (r >> 3);
}

}  // namespace ndfft
