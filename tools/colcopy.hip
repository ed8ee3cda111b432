// tools/colcopy.hip -- developer tool: what does the memory system give a COLUMN-TILE access shape?
// A 1024-thread workgroup copies a tile of W adjacent 16-byte columns x ROWS rows of a row-major [ROWS][COLS] array of
// 16-byte elements (row pitch COLS * 16 B): W * 16 bytes per row segment.  This is the access pattern of a
// register-resident single-pass column FFT (one tile = W whole lanes in the registers of one workgroup), without the FFT.
// MAP 1: the 128 / (16 W) tiles that share each 128-byte line are dispatched back to back on ONE XCD (blocks b, b+8, ...).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/colcopy.hip -o tools/colcopy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// ROWS x W elements per tile, 1024 threads: thread (l = tid % W, t = tid / W) moves rows t + (1024 / W) r
template <int W, int ROWS, int MAP, int NTL, int NTS> __global__ __launch_bounds__(1024) void k_col(const v4f *in, v4f *out, int cols, unsigned ntiles) {
    extern __shared__ char pad[];
    constexpr int TPR = 1024 / W, E = ROWS / TPR;   // threads per column, rows per thread
    unsigned tile = blockIdx.x;
    if constexpr (MAP == 1) {
        constexpr unsigned S = 8 / W;               // tiles per 128-byte line
        if constexpr (S > 1) {
            const unsigned grp = 8 * S, g = blockIdx.x / grp, r = blockIdx.x % grp;
            if ((g + 1) * grp <= ntiles) tile = g * grp + (r & 7) * S + (r >> 3);
        }
    }
    const int l = threadIdx.x % W, t = threadIdx.x / W;
    const size_t base = (size_t)tile * W + l;
    v4f v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const v4f *p = in + base + (size_t)(t + r * TPR) * cols;
        if constexpr (NTL) v[r] = __builtin_nontemporal_load(p); else v[r] = *p;
    }
#pragma unroll
    for (int r = 0; r < E; ++r) {
        v4f *p = out + base + (size_t)(t + r * TPR) * cols;
        if constexpr (NTS) __builtin_nontemporal_store(v[r], p); else *p = v[r];
    }
}

struct Var { std::string name; std::function<void()> go; };
int main(int argc, char **argv) {
    const int rows = 4096, cols = argc > 1 ? atoi(argv[1]) : 4096;     // 4096 x 4096 x 16 B = 256 MiB
    const int npairs = argc > 2 ? atoi(argv[2]) : 6, rounds = 7;
    const size_t n = (size_t)rows * cols;
    std::vector<v4f *> a(npairs), b(npairs);
    for (int i = 0; i < npairs; ++i) { CK(hipMalloc(&a[i], n * 16)); CK(hipMalloc(&b[i], n * 16)); CK(hipMemset(a[i], 1 + i, n * 16)); }
    std::vector<Var> vs;
    static size_t cnt = 0;
#define COL(W, MAP, NTL, NTS, LDSKB) vs.push_back({"col W" #W " (" + std::to_string(W * 16) + " B rows) map" #MAP " ntl" #NTL " nts" #NTS " lds" #LDSKB "K", [=]() { \
        const size_t k = cnt++ % npairs; hipLaunchKernelGGL((k_col<W, 4096, MAP, NTL, NTS>), dim3(cols / W), dim3(1024), LDSKB * 1024, 0, a[k], b[k], cols, (unsigned)(cols / W)); }});
    COL(4, 0, 0, 1, 0) COL(4, 1, 0, 1, 0) COL(4, 1, 1, 1, 0) COL(4, 1, 0, 0, 0) COL(4, 1, 1, 0, 0) COL(4, 0, 0, 0, 0)
    COL(4, 1, 0, 1, 100) COL(4, 1, 0, 0, 100)
    COL(2, 0, 0, 1, 0) COL(2, 1, 0, 1, 0) COL(2, 1, 0, 0, 0) COL(2, 1, 1, 0, 0)
    COL(8, 0, 0, 1, 0) COL(8, 0, 1, 1, 0) COL(8, 0, 0, 0, 0)
    COL(1, 0, 0, 0, 0) COL(1, 1, 0, 0, 0)
    CK(hipFuncSetAttribute((const void *)k_col<4, 4096, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_col<4, 4096, 1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int inner = 6;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            vs[i].go();
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; ++k) vs[i].go();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[i].push_back(ms * 1000.f / inner); CK(hipGetLastError());
        }
    printf("column tiles of a %d x %d array of 16-byte elements, %d rotating pairs\n%-48s %10s %10s %8s\n", rows, cols, npairs, "variant", "median_us", "GB/s", "of 8T");
    const double bytes = 2.0 * n * 16;
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf("%-48s %10.2f %10.1f %8.3f\n", vs[i].name.c_str(), med, bytes / med / 1e3, bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
