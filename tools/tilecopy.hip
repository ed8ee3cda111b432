// tools/tilecopy.hip -- developer tool: copy bandwidth of 2-D tiles (R rows x WB bytes) whose rows are PITCH bytes apart,
// the access shape of the column four-step's stages (exec.hip: col_split): which tile width / row count does the memory
// system like when the pitch is a large power of two?  Input array [NROWS][ROWB bytes]; tile (rb, cb) covers rows
// rb, rb + RS, rb + 2 RS, ... (R of them, RS = row stride in rows) and bytes [cb WB, (cb + 1) WB).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/tilecopy.hip -o tools/tilecopy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// T threads; tile R rows x WB bytes (WB / 16 vectors per row); thread -> (vector within row fastest, row)
template <int T, int R, int WB, int NTL, int NTS> __global__ __launch_bounds__(T) void k_tile(const v4f *in, v4f *out, int rowv /* vectors per array row */, int rs /* row stride of a tile, in rows */,
                                                                                           int tiles_per_rowblock /* rowv / (WB/16) */) {
    constexpr int VPR = WB / 16, RPI = T / VPR, E = R / RPI;   // rows per instruction, instructions
    const unsigned cb = blockIdx.x % tiles_per_rowblock, rb = blockIdx.x / tiles_per_rowblock;
    // row block rb: rows (rb / rs) * (R * rs) + rb % rs + k * rs
    const size_t row0 = (size_t)(rb / rs) * ((size_t)R * rs) + rb % rs;
    const int vx = threadIdx.x % VPR, ry = threadIdx.x / VPR;
    v4f v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const v4f *p = in + (row0 + (size_t)(ry + e * RPI) * rs) * rowv + (size_t)cb * VPR + vx;
        if constexpr (NTL) v[e] = __builtin_nontemporal_load(p); else v[e] = *p;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        v4f *p = out + (row0 + (size_t)(ry + e * RPI) * rs) * rowv + (size_t)cb * VPR + vx;
        if constexpr (NTS) __builtin_nontemporal_store(v[e], p); else *p = v[e];
    }
}

struct Var { std::string name; std::function<void()> go; };
int main(int argc, char **argv) {
    const int nrows = 8192, rowb = 32768;                 // 8192 x 8192 f32 = 256 MiB, pitch 32 KiB (cfg3-A's input)
    const int npairs = argc > 1 ? atoi(argv[1]) : 6, rounds = 7;
    const size_t nv = (size_t)nrows * rowb / 16;
    std::vector<v4f *> a(npairs), b(npairs);
    for (int i = 0; i < npairs; ++i) { CK(hipMalloc(&a[i], nv * 16)); CK(hipMalloc(&b[i], nv * 16)); CK(hipMemset(a[i], 1 + i, nv * 16)); }
    std::vector<Var> vs;
    static size_t cnt = 0;
    const int rowv = rowb / 16;
#define TILE(T, R, WB, RS, NTL, NTS) vs.push_back({"T" #T " tile " #R " rows x " #WB " B, row stride " #RS " rows, ntl" #NTL " nts" #NTS, [=]() { \
        const size_t k = cnt++ % npairs; const int tpr = rowv / (WB / 16); const unsigned nb = (unsigned)((size_t)(nrows / R) * tpr); \
        hipLaunchKernelGGL((k_tile<T, R, WB, NTL, NTS>), dim3(nb), dim3(T), 0, 0, a[k], b[k], rowv, RS, tpr); }});
    // stage-A-like: 128 rows, 64 rows apart (2 MiB pitch), different widths
    TILE(256, 128, 128, 64, 0, 1) TILE(256, 128, 256, 64, 0, 1) TILE(256, 128, 512, 64, 0, 1) TILE(256, 128, 1024, 64, 0, 1) TILE(512, 128, 2048, 64, 0, 1)
    TILE(256, 128, 128, 64, 1, 1) TILE(256, 128, 256, 64, 1, 1) TILE(256, 128, 512, 64, 1, 1) TILE(256, 128, 1024, 64, 1, 1)
    TILE(256, 128, 128, 64, 1, 0) TILE(256, 128, 512, 64, 1, 0)
    // fewer / more rows per tile at the same stride
    TILE(256, 64, 256, 128, 1, 1) TILE(256, 64, 512, 128, 1, 1) TILE(256, 32, 1024, 256, 1, 1) TILE(256, 256, 256, 32, 1, 1)
    // adjacent rows (row stride 1): what contiguity buys
    TILE(256, 128, 128, 1, 1, 1) TILE(256, 128, 512, 1, 1, 1) TILE(256, 64, 1024, 1, 1, 1)
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
    printf("2-D tile copies of an %d x %d-byte array, %d rotating pairs\n%-66s %10s %10s %8s\n", nrows, rowb, npairs, "variant", "median_us", "GB/s", "of 8T");
    const double bytes = 2.0 * nv * 16;
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf("%-66s %10.2f %10.1f %8.3f\n", vs[i].name.c_str(), med, bytes / med / 1e3, bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
