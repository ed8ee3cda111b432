// tools/drainprobe.hip -- developer tool: where does the CONSTANT cost of a short cache-cold launch go?
// k_pow2<double,4096> moves 4096 lanes in ~93 us and every further 4096 lanes in ~81 us: ~10 us per launch
// are not bandwidth.  This probe times copy kernels of the FFT kernel's access shape (one 64 KiB lane per workgroup pass, 512 threads
// x 8 x 16 B) on L = 1024 .. 16384 lanes, walking a 4 GiB + 4 GiB arena so that every launch is cache-cold, and fits
// t(L) = a + b L per variant: a = what a launch pays once (fill, drain, end-of-kernel L2 write-back), 4096 b = what the bytes cost.
// Variants: store / load policy bits, one workgroup per lane vs a persistent grid with the next lane's loads issued before the current
// lane's "compute" (a timed s_sleep stand-in for the butterfly passes), workgroups per CU.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ndrustfft_amd/csrc tools/drainprobe.hip -o tools/drainprobe
//   run  : tools/drainprobe [rounds = 5] [compute_cycles = 6000]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// POL: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt
template <int POL> __device__ __forceinline__ v4f ldg(const v4f *p) {
    v4f v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL> __device__ __forceinline__ void stg(v4f *p, v4f v) {
    if constexpr (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
    if constexpr (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblk, int chunk) {
    if (chunk <= 0) return b;
    const unsigned grp = 8u * (unsigned)chunk, g = b / grp;
    if ((g + 1) * grp > nblk) return b;
    const unsigned r = b - g * grp;
    return g * grp + (r & 7u) * (unsigned)chunk + (r >> 3);
}
// stand-in for the butterfly passes: the wave does nothing for ~cycles (s_sleep 8 = 512 clocks)
__device__ __forceinline__ void fake_compute(int cycles) {
    for (int c = 0; c < cycles; c += 512) __builtin_amdgcn_s_sleep(8);
}

constexpr int T = 512, E = 8;

// one workgroup per lane (the product kernel's structure)
template <int LD, int ST> __global__ __launch_bounds__(T) void k_one(const v4f *in, v4f *out, unsigned nlanes, int chunk, int cycles) {
    extern __shared__ char occupancy_pad[];
    const unsigned lane = xcd_block(blockIdx.x, nlanes, chunk);
    const v4f *s = in + (size_t)lane * 4096 + threadIdx.x;
    v4f *d = out + (size_t)lane * 4096 + threadIdx.x;
    v4f v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = ldg<LD>(s + r * T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fake_compute(cycles);
#pragma unroll
    for (int r = 0; r < E; ++r) stg<ST>(d + r * T, v[r]);
}

// persistent grid: workgroup b owns lanes b, b + G, ...; PF = 1: the next lane's loads are issued before this lane's compute
template <int LD, int ST, int PF> __global__ __launch_bounds__(T) void k_persist(const v4f *in, v4f *out, unsigned nlanes, int chunk, int cycles) {
    extern __shared__ char occupancy_pad[];
    const unsigned G = gridDim.x;
    unsigned vb = blockIdx.x;
    if (vb >= nlanes) return;
    v4f nxt[E], v[E];
    if constexpr (PF) {
        const v4f *s = in + (size_t)xcd_block(vb, nlanes, chunk) * 4096 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < E; ++r) nxt[r] = ldg<LD>(s + r * T);
    }
    for (; vb < nlanes; vb += G) {
        const unsigned lane = xcd_block(vb, nlanes, chunk);
        if constexpr (PF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the prefetched lane has landed (and the previous lane's stores were acknowledged)
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = nxt[r];
            if (vb + G < nlanes) {
                const v4f *s = in + (size_t)xcd_block(vb + G, nlanes, chunk) * 4096 + threadIdx.x;
#pragma unroll
                for (int r = 0; r < E; ++r) nxt[r] = ldg<LD>(s + r * T);
            }
        } else {
            const v4f *s = in + (size_t)lane * 4096 + threadIdx.x;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = ldg<LD>(s + r * T);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        fake_compute(cycles);
        v4f *d = out + (size_t)lane * 4096 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < E; ++r) stg<ST>(d + r * T, v[r]);
    }
}


// dynamic persistent grid: the next lane comes from a counter of the workgroup's XCD (blockIdx % 8); the last workgroup out resets the counters
template <int LD, int ST, int PF> __global__ __launch_bounds__(T) void k_dyn(const v4f *in, v4f *out, unsigned nlanes, int chunk, int cycles, unsigned *ctr) {
    extern __shared__ char occupancy_pad[];
    __shared__ unsigned s_j;
    const unsigned x = blockIdx.x & 7u;
    v4f nxt[E], v[E];
    if (threadIdx.x == 0) s_j = atomicAdd(&ctr[x * 32], 1u);
    __syncthreads();
    unsigned vb = s_j * 8u + x;
    if constexpr (PF) {
        if (vb < nlanes) {
            const v4f *s = in + (size_t)xcd_block(vb, nlanes, chunk) * 4096 + threadIdx.x;
#pragma unroll
            for (int r = 0; r < E; ++r) nxt[r] = ldg<LD>(s + r * T);
        }
    }
    while (vb < nlanes) {
        const unsigned lane = xcd_block(vb, nlanes, chunk);
        __syncthreads();
        if (threadIdx.x == 0) s_j = atomicAdd(&ctr[x * 32], 1u);               // the next index, fetched while this lane is on its way
        if constexpr (PF) {
            __syncthreads();
            const unsigned nvb = s_j * 8u + x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = nxt[r];
            if (nvb < nlanes) {
                const v4f *s = in + (size_t)xcd_block(nvb, nlanes, chunk) * 4096 + threadIdx.x;
#pragma unroll
                for (int r = 0; r < E; ++r) nxt[r] = ldg<LD>(s + r * T);
            }
        } else {
            const v4f *s = in + (size_t)lane * 4096 + threadIdx.x;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = ldg<LD>(s + r * T);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        fake_compute(cycles);
        v4f *d = out + (size_t)lane * 4096 + threadIdx.x;
#pragma unroll
        for (int r = 0; r < E; ++r) stg<ST>(d + r * T, v[r]);
        __syncthreads();
        vb = s_j * 8u + x;
    }
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&ctr[8 * 32], 1u) == gridDim.x - 1) { for (int i = 0; i <= 8; ++i) ctr[i * 32] = 0; __threadfence(); }
    }
}

struct Var { std::string name; std::function<void(const v4f *, v4f *, unsigned)> go; };

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const int cycles = argc > 2 ? atoi(argv[2]) : 6000;
    const unsigned arena_lanes = 65536;                                  // 4 GiB per side
    const size_t n = (size_t)arena_lanes * 4096;
    v4f *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const unsigned cus = (unsigned)pr.multiProcessorCount;
    std::vector<Var> vs;
#define ONE(LD, ST, LDSKB, CHUNK, CYC) \
    vs.push_back({"one/lane ld" #LD " st" #ST " lds" #LDSKB "K chunk" #CHUNK " cyc" #CYC, [=](const v4f *i, v4f *o, unsigned L) { \
        hipLaunchKernelGGL((k_one<LD, ST>), dim3(L), dim3(T), LDSKB * 1024, 0, i, o, L, CHUNK, CYC); }});
#define PER(LD, ST, PF, WGCU, LDSKB, CHUNK, CYC) \
    vs.push_back({"persist pf" #PF " ld" #LD " st" #ST " " #WGCU "wg/cu chunk" #CHUNK " cyc" #CYC, [=](const v4f *i, v4f *o, unsigned L) { \
        hipLaunchKernelGGL((k_persist<LD, ST, PF>), dim3(std::min(L, WGCU * cus)), dim3(T), LDSKB * 1024, 0, i, o, L, CHUNK, CYC); }});
    unsigned *ctr; CK(hipMalloc(&ctr, 4096)); CK(hipMemset(ctr, 0, 4096));
#define DYN(LD, ST, PF, WGCU, LDSKB, CHUNK, CYC) \
    vs.push_back({"dynamic pf" #PF " ld" #LD " st" #ST " " #WGCU "wg/cu chunk" #CHUNK " cyc" #CYC, [=](const v4f *i, v4f *o, unsigned L) { \
        hipLaunchKernelGGL((k_dyn<LD, ST, PF>), dim3(std::min(L, WGCU * cus) & ~7u), dim3(T), LDSKB * 1024, 0, i, o, L, CHUNK, CYC, ctr); }});
    const int C = cycles;
    // the product's structure (4 workgroups per CU through 34 KiB of LDS), store policies: nt / plain / write-through (sc0 sc1) / both
    ONE(1, 1, 34, 8, C) ONE(1, 0, 34, 8, C) ONE(1, 3, 34, 8, C) ONE(1, 4, 34, 8, C) ONE(1, 2, 34, 8, C)
    ONE(0, 1, 34, 8, C)
    ONE(1, 1, 34, 8, 0)                                             // no compute phase at all
    // persistent, no prefetch: 4 per CU
    PER(1, 1, 0, 4, 34, 8, C)
    // persistent with prefetch: 2, 3, 4 per CU
    PER(1, 1, 1, 2, 68, 8, C) PER(1, 1, 1, 3, 50, 8, C) PER(1, 1, 1, 4, 34, 8, C)
    PER(1, 4, 1, 2, 68, 8, C) PER(1, 3, 1, 2, 68, 8, C)
    PER(1, 1, 1, 2, 68, 8, 0)
    DYN(1, 1, 0, 4, 34, 8, C) DYN(1, 1, 0, 3, 50, 8, C) DYN(1, 1, 1, 2, 68, 8, C) DYN(1, 1, 1, 4, 34, 8, C) DYN(1, 1, 0, 4, 34, 8, 0) DYN(1, 1, 0, 8, 17, 8, C)
    ONE(1, 1, 17, 8, C) ONE(1, 1, 34, 8, C)
    for (auto f : {(const void *)k_one<1, 1>, (const void *)k_persist<1, 1, 1>, (const void *)k_persist<1, 4, 1>, (const void *)k_persist<1, 3, 1>, (const void *)k_persist<1, 1, 0>, (const void *)k_dyn<1, 1, 0>, (const void *)k_dyn<1, 1, 1>})
        CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));

    const std::vector<unsigned> Ls = {1024, 2048, 4096, 8192, 16384};
    std::vector<std::vector<std::vector<float>>> t(vs.size(), std::vector<std::vector<float>>(Ls.size()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t cursor = 0;
    // clocks up
    for (int k = 0; k < 200; ++k) vs[0].go(a, b, 16384);
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i)
            for (size_t li = 0; li < Ls.size(); ++li) {
                const unsigned L = Ls[li];
                const int inner = (int)std::max<unsigned>(4, 65536 / L);        // one walk over the arena
                auto at = [&]() { const size_t off = cursor; cursor = (cursor + L) % arena_lanes; if (cursor + L > arena_lanes) cursor = 0; return off * 4096; };
                { const size_t o = at(); vs[i].go(a + o, b + o, L); }
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < inner; ++k) { const size_t o = at(); vs[i].go(a + o, b + o, L); }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[i][li].push_back(ms * 1000.f / inner);
                CK(hipGetLastError());
            }
    printf("compute stand-in %d cycles; us per launch (median of %d) at L lanes of 64 KiB; fit over L >= 2048: t = a + b L\n", cycles, rounds);
    printf("%-52s", "variant");
    for (unsigned L : Ls) printf(" %8u", L);
    printf(" %8s %10s %8s\n", "a_us", "us/4096", "frac4096");
    for (size_t i = 0; i < vs.size(); ++i) {
        printf("%-52s", vs[i].name.c_str());
        std::vector<double> med;
        for (size_t li = 0; li < Ls.size(); ++li) { auto &x = t[i][li]; std::sort(x.begin(), x.end()); med.push_back(x[x.size() / 2]); printf(" %8.2f", med.back()); }
        // least squares over L >= 2048
        double sx = 0, sy = 0, sxx = 0, sxy = 0; int m = 0;
        for (size_t li = 1; li < Ls.size(); ++li) { const double x = Ls[li], y = med[li]; sx += x; sy += y; sxx += x * x; sxy += x * y; ++m; }
        const double bb = (m * sxy - sx * sy) / (m * sxx - sx * sx), aa = (sy - bb * sx) / m;
        printf(" %8.2f %10.2f %8.3f\n", aa, bb * 4096, 2.0 * 4096 * 65536 / med[2] / 1e3 / 8000.0);
    }
    return 0;
}
