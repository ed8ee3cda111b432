// tools/coldcopy.hip -- developer tool: the CACHE-COLD copy ceiling of the FFT kernel's access shape.
// Every workgroup moves one contiguous 64 KiB "lane" (what k_pow2<double,4096> reads and writes), the arrays are
// far larger than the 256 MiB Infinity Cache, and the cache-policy bits of the loads and stores (sc0 / sc1 / nt),
// the workgroup shape, the occupancy and the blockIdx -> lane map are swept.  Interleaved rounds, HIP events.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/coldcopy.hip -o tools/coldcopy
//   run  : tools/coldcopy [MiB per array = 2048] [rounds = 7]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0, 6 sc1 nt, 7 sc0 nt
template <int POL> __device__ __forceinline__ v4f ldg(const v4f *p) {
    v4f v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL> __device__ __forceinline__ void stg(v4f *p, v4f v) {
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// One workgroup of T threads per 64 KiB lane: thread t loads elements t + r T, r = 0..E-1 (E = 4096 / T), all loads
// first, then all stores -- the register kernel's first and last pass.  MAP 1: XCD x (= blockIdx % 8) owns a
// contiguous eighth of the lanes instead of every eighth lane.
// MAP: 0 identity; C > 0: workgroups are dealt round-robin over the 8 XCDs, so XCD x = blockIdx % 8 gets, out of every
// group of 8 C consecutive lanes, the C CONTIGUOUS lanes [x C, (x+1) C) instead of every eighth one; 1 = one eighth of all
template <int T, int LD, int ST, int MAP> __global__ __launch_bounds__(T) void k_lane(const v4f *in, v4f *out, unsigned nlanes) {
    extern __shared__ char occupancy_pad[];
    constexpr int E = 4096 / T;
    unsigned lane = blockIdx.x;
    if constexpr (MAP == 1) lane = (blockIdx.x & 7) * (nlanes >> 3) + (blockIdx.x >> 3);
    if constexpr (MAP > 1) {
        const unsigned g = blockIdx.x / (8 * MAP), r = blockIdx.x % (8 * MAP);
        lane = g * (8 * MAP) + (r & 7) * MAP + (r >> 3);
    }
    const v4f *s = in + (size_t)lane * 4096 + threadIdx.x;
    v4f *d = out + (size_t)lane * 4096 + threadIdx.x;
    v4f v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = ldg<LD>(s + r * T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < E; ++r) stg<ST>(d + r * T, v[r]);
}

// read-only / write-only with the same shape (what each direction can do alone)
template <int T, int LD> __global__ __launch_bounds__(T) void k_lane_read(const v4f *in, v4f *out, unsigned nlanes) {
    extern __shared__ char occupancy_pad[];
    constexpr int E = 4096 / T;
    const v4f *s = in + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v4f v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = ldg<LD>(s + r * T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float acc = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    if (acc == 1.2345e30f) out[0].x = acc;
}
template <int T, int ST> __global__ __launch_bounds__(T) void k_lane_write(const v4f *in, v4f *out, unsigned nlanes) {
    extern __shared__ char occupancy_pad[];
    constexpr int E = 4096 / T;
    v4f *d = out + (size_t)blockIdx.x * 4096 + threadIdx.x;
    v4f v; v.x = 1.f; v.y = 2.f; v.z = 3.f; v.w = (float)threadIdx.x;
#pragma unroll
    for (int r = 0; r < E; ++r) stg<ST>(d + r * T, v);
}

struct Var { std::string name; std::function<void()> go; double bytes; bool check; };

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoll(argv[1]) : 2048;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const unsigned nlanes = (unsigned)(mib * 16);          // 64 KiB lanes
    const size_t n = (size_t)nlanes * 4096;               // 16-byte elements
    v4f *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    {
        std::vector<float> h(1 << 20);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i * 2654435761u % 1000003);
        for (size_t off = 0; off < n * 4; off += h.size()) CK(hipMemcpy((float *)a + off, h.data(), std::min(h.size(), n * 4 - off) * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemset(b, 0, n * 16));
    std::vector<Var> vs;
    const double cb = 2.0 * n * 16;
#define LANE(T, LD, ST, MAP, LDSKB) \
    vs.push_back({"lane T" #T " ld" #LD " st" #ST " map" #MAP " lds" #LDSKB "K", [=]() { hipLaunchKernelGGL((k_lane<T, LD, ST, MAP>), dim3(nlanes), dim3(T), LDSKB * 1024, 0, a, b, nlanes); }, cb, true});
    // the product kernel's shape: 512 threads, 34 KiB of LDS (4 workgroups per CU), plain loads, nt stores
    LANE(512, 0, 1, 0, 34)
    // store policy sweep
    LANE(512, 0, 0, 0, 34) LANE(512, 0, 2, 0, 34) LANE(512, 0, 3, 0, 34) LANE(512, 0, 4, 0, 34) LANE(512, 0, 5, 0, 34) LANE(512, 0, 6, 0, 34) LANE(512, 0, 7, 0, 34)
    // load policy sweep (nt stores)
    LANE(512, 1, 1, 0, 34) LANE(512, 2, 1, 0, 34) LANE(512, 3, 1, 0, 34) LANE(512, 4, 1, 0, 34) LANE(512, 5, 1, 0, 34) LANE(512, 6, 1, 0, 34) LANE(512, 7, 1, 0, 34)
    // both streaming
    LANE(512, 1, 0, 0, 34) LANE(512, 4, 4, 0, 34) LANE(512, 2, 2, 0, 34) LANE(512, 6, 6, 0, 34) LANE(512, 1, 6, 0, 34) LANE(512, 1, 4, 0, 34)
    // occupancy: 8 / 4 / 2 / 1 workgroups per CU
    LANE(512, 0, 1, 0, 0) LANE(512, 0, 1, 0, 17) LANE(512, 0, 1, 0, 68) LANE(512, 0, 1, 0, 136)
    // workgroup shape
    LANE(256, 0, 1, 0, 17) LANE(256, 0, 1, 0, 34) LANE(1024, 0, 1, 0, 68) LANE(1024, 0, 1, 0, 34) LANE(1024, 1, 1, 0, 68)
    // XCD-contiguous lane maps: one eighth per XCD (1), or chunks of C lanes per XCD
    LANE(512, 0, 1, 1, 34) LANE(512, 1, 1, 1, 34) LANE(512, 0, 0, 1, 34) LANE(512, 1, 0, 1, 34) LANE(1024, 0, 1, 1, 68) LANE(1024, 1, 1, 1, 68)
    LANE(512, 1, 1, 2, 34) LANE(512, 1, 1, 4, 34) LANE(512, 1, 1, 8, 34) LANE(512, 1, 1, 16, 34) LANE(512, 1, 1, 32, 34) LANE(512, 1, 1, 64, 34) LANE(512, 1, 1, 128, 34) LANE(512, 1, 1, 512, 34)
    LANE(512, 0, 1, 4, 34) LANE(512, 0, 1, 16, 34) LANE(512, 0, 1, 32, 34) LANE(512, 0, 1, 64, 34) LANE(512, 0, 1, 128, 34) LANE(512, 0, 1, 512, 34)
    LANE(512, 0, 0, 32, 34) LANE(512, 1, 0, 32, 34)
#define RD(T, LD, LDSKB) vs.push_back({"read  T" #T " ld" #LD " lds" #LDSKB "K", [=]() { hipLaunchKernelGGL((k_lane_read<T, LD>), dim3(nlanes), dim3(T), LDSKB * 1024, 0, a, b, nlanes); }, cb / 2, false});
#define WR(T, ST, LDSKB) vs.push_back({"write T" #T " st" #ST " lds" #LDSKB "K", [=]() { hipLaunchKernelGGL((k_lane_write<T, ST>), dim3(nlanes), dim3(T), LDSKB * 1024, 0, a, b, nlanes); }, cb / 2, false});
    RD(512, 0, 34) RD(512, 1, 34) RD(512, 2, 34) RD(512, 4, 34)
    WR(512, 0, 34) WR(512, 1, 34) WR(512, 2, 34) WR(512, 3, 34) WR(512, 4, 34) WR(512, 6, 34)
    vs.push_back({"hipMemcpyDtoD", [=]() { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); }, cb, true});
    for (auto &v : vs) {
        const void *fn = nullptr; (void)fn;
    }
    // the kernels with > 64 KiB of dynamic LDS need the opt-in
    CK(hipFuncSetAttribute((const void *)k_lane<512, 0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_lane<1024, 0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_lane<1024, 1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_lane<1024, 0, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_lane<1024, 1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ha(4096), hb(4096);
    const int inner = 2;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            if (r == 0 && vs[i].check) CK(hipMemset(b, 0, n * 16));
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; ++k) vs[i].go();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t[i].push_back(ms * 1000.f / inner);
            CK(hipGetLastError());
            if (r == 0 && vs[i].check) {   // spot-check three 16 KiB windows (first, middle, last)
                for (size_t w : {(size_t)0, n * 2, n * 4 - 4096}) {
                    CK(hipMemcpy(ha.data(), (float *)a + w, 16384, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hb.data(), (float *)b + w, 16384, hipMemcpyDeviceToHost));
                    if (ha != hb) { printf("!! %s: copy differs at float %zu\n", vs[i].name.c_str(), w); break; }
                }
            }
        }
    printf("arrays %zu MiB each (cache-cold)\n%-40s %10s %10s %10s %8s\n", mib, "variant", "median_us", "min_us", "GB/s(med)", "of 8T");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf("%-40s %10.2f %10.2f %10.1f %8.3f\n", vs[i].name.c_str(), med, t[i][0], vs[i].bytes / med / 1e3, vs[i].bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
