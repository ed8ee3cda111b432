// transpose.hip -- batched, LDS-padded 2-D transpose (gfx950).
// out[b][c][r] = in[b][r][c].  Used when the transform axis is not the contiguous one AND a tile
// of adjacent lanes would not fit LDS (long lanes, e.g. BASELINE cfg3-A: 8192-long f32 lanes at
// stride 8192): lanes are made contiguous by one coalesced transpose, transformed by the row
// kernels, and transposed back -- replacing the reference's per-lane x.to_vec() / y.assign()
// strided copies (src/lib.rs:133-134), which touch one cache line per element.
#include "engine.h"

namespace ndfft {

// Tile order: 8x8 groups of tiles, so that the ~1000 concurrently resident workgroups read AND write
// 8-tile-wide (2 KiB) contiguous runs instead of 256-byte pieces scattered over the whole array.
__device__ __forceinline__ bool tile_coords(int64_t tiles_x, int64_t tiles_y, int64_t &tx, int64_t &ty) {
    if (tiles_x < 8 || tiles_y < 8) {   // small tile grids (many small batches): plain row-major order
        tx = blockIdx.x % tiles_x; ty = blockIdx.x / tiles_x;
        return ty < tiles_y;
    }
    const int64_t b = blockIdx.x, gpr = (tiles_x + 7) / 8;
    const int64_t g = b >> 6, w = b & 63;
    tx = (g % gpr) * 8 + (w & 7);
    ty = (g / gpr) * 8 + (w >> 3);
    return tx < tiles_x && ty < tiles_y;
}
static inline unsigned swizzled_blocks(int64_t tiles_x, int64_t tiles_y) {
    if (tiles_x < 8 || tiles_y < 8) return (unsigned)(tiles_x * tiles_y);
    return (unsigned)(((tiles_x + 7) / 8) * ((tiles_y + 7) / 8) * 64);
}

template <typename E, int TILE>
__global__ __launch_bounds__(256) void k_transpose(const E *__restrict__ in, E *__restrict__ out, int64_t rows, int64_t cols,
                                                   int64_t ld_in, int64_t ld_out, int64_t bs_in, int64_t bs_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    E(*tile)[TILE + 1] = (E(*)[TILE + 1])smem;   // +1 element of padding: column reads hit distinct banks
    const E *src = in + (int64_t)blockIdx.z * bs_in;
    E *dst = out + (int64_t)blockIdx.z * bs_out;
    int64_t tcx, tcy;
    if (!tile_coords((cols + TILE - 1) / TILE, (rows + TILE - 1) / TILE, tcx, tcy)) return;
    const int64_t c0 = tcx * TILE, r0 = tcy * TILE;
    const int tx = threadIdx.x % TILE, ty = threadIdx.x / TILE;
    constexpr int RSTEP = 256 / TILE;
#pragma unroll
    for (int ii = 0; ii < TILE / RSTEP; ++ii) {          // (trip count known at compile time: all loads in flight)
        const int i = ty + ii * RSTEP;
        const int64_t r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = src[r * ld_in + c];
    }
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < TILE / RSTEP; ++ii) {
        const int i = ty + ii * RSTEP;
        const int64_t c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) __builtin_nontemporal_store(tile[tx][i], &dst[c * ld_out + r]);
    }
}

// 16-byte-vectorised variant for 4- and 8-byte elements: V = 16/sizeof(E) elements per global access on
// BOTH sides (rows of the input, rows of the output); the element shuffle happens in LDS.
template <typename E, int V>
__global__ __launch_bounds__(256) void k_transpose_vec(const E *__restrict__ in, E *__restrict__ out, int64_t rows, int64_t cols,
                                                       int64_t ld_in, int64_t ld_out, int64_t bs_in, int64_t bs_out) {
    constexpr int TILE = 64, VPR = TILE / V;               // vectors per tile row
    typedef E vecE __attribute__((ext_vector_type(V)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    E(*tile)[TILE + 1] = (E(*)[TILE + 1])smem;
    const E *src = in + (int64_t)blockIdx.z * bs_in;
    E *dst = out + (int64_t)blockIdx.z * bs_out;
    int64_t tcx, tcy;
    if (!tile_coords((cols + TILE - 1) / TILE, (rows + TILE - 1) / TILE, tcx, tcy)) return;
    const int64_t c0 = tcx * TILE, r0 = tcy * TILE;
#pragma unroll
    for (int idx = threadIdx.x; idx < TILE * VPR; idx += 256) {
        const int r = idx / VPR, cv = (idx % VPR) * V;
        const int64_t gr = r0 + r, gc = c0 + cv;
        if (gr < rows) {
            if (gc + V <= cols) {
                const vecE v = *(const vecE *)(src + gr * ld_in + gc);
#pragma unroll
                for (int k = 0; k < V; ++k) tile[r][cv + k] = v[k];
            } else {
                for (int k = 0; k < V; ++k) if (gc + k < cols) tile[r][cv + k] = src[gr * ld_in + gc + k];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int idx = threadIdx.x; idx < TILE * VPR; idx += 256) {
        const int c = idx / VPR, rv = (idx % VPR) * V;
        const int64_t gc = c0 + c, gr = r0 + rv;
        if (gc < cols) {
            if (gr + V <= rows) {
                vecE v;
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = tile[rv + k][c];
                __builtin_nontemporal_store(v, (vecE *)(dst + gc * ld_out + gr));
            } else {
                for (int k = 0; k < V; ++k) if (gr + k < rows) dst[gc * ld_out + gr + k] = tile[rv + k][c];
            }
        }
    }
}

struct alignas(16) E16 { double a, b; };
typedef double E16v __attribute__((ext_vector_type(2)));

int launch_transpose(const void *in, void *out, int64_t batch, int64_t rows, int64_t cols, int64_t ld_in,
                     int64_t ld_out, int64_t bstride_in, int64_t bstride_out, int elem_bytes, hipStream_t s) {
    if (batch <= 0 || rows <= 0 || cols <= 0) return NDFFT_OK;
    if (batch > 65535) return fail(NDFFT_ERR_UNSUPPORTED, "transpose: batch too large");
    const int V = 16 / elem_bytes;
    const bool vec = elem_bytes < 16 && ld_in % V == 0 && ld_out % V == 0 && bstride_in % V == 0 && bstride_out % V == 0 &&
                     (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0;
    if (elem_bytes == 4) {
        dim3 g(swizzled_blocks((cols + 63) / 64, (rows + 63) / 64), 1, (unsigned)batch);
        if (vec) hipLaunchKernelGGL((k_transpose_vec<float, 4>), g, dim3(256), 64 * 65 * 4, s, (const float *)in, (float *)out, rows, cols, ld_in, ld_out, bstride_in, bstride_out);
        else hipLaunchKernelGGL((k_transpose<float, 64>), g, dim3(256), 64 * 65 * 4, s, (const float *)in, (float *)out, rows, cols, ld_in, ld_out, bstride_in, bstride_out);
    } else if (elem_bytes == 8) {
        dim3 g(swizzled_blocks((cols + 63) / 64, (rows + 63) / 64), 1, (unsigned)batch);
        if (vec) hipLaunchKernelGGL((k_transpose_vec<double, 2>), g, dim3(256), 64 * 65 * 8, s, (const double *)in, (double *)out, rows, cols, ld_in, ld_out, bstride_in, bstride_out);
        else hipLaunchKernelGGL((k_transpose<double, 64>), g, dim3(256), 64 * 65 * 8, s, (const double *)in, (double *)out, rows, cols, ld_in, ld_out, bstride_in, bstride_out);
    } else if (elem_bytes == 16) {
        dim3 g(swizzled_blocks((cols + 31) / 32, (rows + 31) / 32), 1, (unsigned)batch);
        hipLaunchKernelGGL((k_transpose<E16v, 32>), g, dim3(256), 32 * 33 * 16, s, (const E16v *)in, (E16v *)out, rows, cols, ld_in, ld_out, bstride_in, bstride_out);
    } else {
        return fail(NDFFT_ERR_INVALID_ARG, "transpose: element size");
    }
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

}  // namespace ndfft
