// tools/membench.hip -- developer tool: which streaming pattern does the MI355X memory system like?
// Establishes the copy ceiling the FFT kernels are measured against (DESIGN.md "HBM ceiling").
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double2 V;   // 16 bytes

template <int NT> __device__ __forceinline__ V ld(const V *p) {
    if constexpr (NT & 2) { V r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
    else return *p;
}
template <int NT> __device__ __forceinline__ void st(V *p, V v) {
    if constexpr (NT & 1) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

// grid-stride: iteration `it` touches a window of gridDim*THREADS*U consecutive elements
template <int THREADS, int U, int NT> __global__ __launch_bounds__(THREADS) void k_gs(const V *in, V *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    for (size_t base = (size_t)blockIdx.x * THREADS + threadIdx.x; base < n; base += stride * U) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * stride < n) v[u] = ld<NT>(in + base + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * stride < n) st<NT>(out + base + u * stride, v[u]);
    }
}

// chunked: each block owns contiguous chunks of THREADS*U elements (U loads per thread at stride THREADS),
// chunks dealt round-robin to a persistent grid
template <int THREADS, int U, int NT> __global__ __launch_bounds__(THREADS) void k_chunk(const V *in, V *out, size_t n) {
    const size_t chunk = (size_t)THREADS * U, nchunks = n / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const V *s = in + c * chunk; V *d = out + c * chunk;
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(s + threadIdx.x + u * THREADS);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(d + threadIdx.x + u * THREADS, v[u]);
    }
}

// chunked with software prefetch: loads of chunk c+grid are issued before the stores of chunk c
template <int THREADS, int U, int NT> __global__ __launch_bounds__(THREADS) void k_chunk_pf(const V *in, V *out, size_t n) {
    const size_t chunk = (size_t)THREADS * U, nchunks = n / chunk;
    size_t c = blockIdx.x;
    if (c >= nchunks) return;
    V v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + c * chunk + threadIdx.x + u * THREADS);
    for (;;) {
        const size_t cn = c + gridDim.x;
        const bool more = cn < nchunks;
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = ld<NT>(in + cn * chunk + threadIdx.x + u * THREADS);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(out + c * chunk + threadIdx.x + u * THREADS, v[u]);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u];
        c = cn;
    }
}

template <int THREADS, int U> __global__ __launch_bounds__(THREADS) void k_read(const V *in, V *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    double acc = 0;
    for (size_t base = (size_t)blockIdx.x * THREADS + threadIdx.x; base < n; base += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * stride < n) { V v = in[base + u * stride]; acc += v.x + v.y; }
    }
    if (acc == 1.2345e300) out[0].x = acc;
}
template <int THREADS, int U> __global__ __launch_bounds__(THREADS) void k_write(const V *in, V *out, size_t n) {
    const size_t stride = (size_t)gridDim.x * THREADS;
    V v; v.x = 1.0; v.y = 2.0;
    for (size_t base = (size_t)blockIdx.x * THREADS + threadIdx.x; base < n; base += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * stride < n) out[base + u * stride] = v;
    }
}

struct Var { std::string name; std::function<void()> go; double bytes; };

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoll(argv[1]) : 256;      // per array
    const int rounds = argc > 2 ? atoi(argv[2]) : 9;
    const size_t n = mib * 1024 * 1024 / 16;
    V *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
    std::vector<Var> vs;
    const double cb = 2.0 * n * 16;
#define GS(T, U, NT, G) vs.push_back({"gs   thr" #T " U" #U " nt" #NT " grid" #G, [=]() { hipLaunchKernelGGL((k_gs<T, U, NT>), dim3(G), dim3(T), 0, 0, a, b, n); }, cb});
#define CH(T, U, NT, G) vs.push_back({"chunk thr" #T " U" #U " nt" #NT " grid" #G, [=]() { hipLaunchKernelGGL((k_chunk<T, U, NT>), dim3(G), dim3(T), 0, 0, a, b, n); }, cb});
#define PF(T, U, NT, G) vs.push_back({"chkpf thr" #T " U" #U " nt" #NT " grid" #G, [=]() { hipLaunchKernelGGL((k_chunk_pf<T, U, NT>), dim3(G), dim3(T), 0, 0, a, b, n); }, cb});
    GS(256, 1, 0, 2048) GS(256, 1, 0, 8192) GS(256, 1, 0, 65536)
    GS(256, 4, 0, 1024) GS(256, 4, 0, 2048) GS(256, 4, 0, 4096) GS(256, 4, 0, 16384)
    GS(256, 8, 0, 1024) GS(256, 8, 0, 2048) GS(256, 16, 0, 1024) GS(256, 16, 0, 512)
    GS(512, 4, 0, 1024) GS(1024, 4, 0, 512) GS(1024, 2, 0, 1024)
    GS(256, 4, 1, 2048) GS(256, 4, 2, 2048) GS(256, 4, 3, 2048) GS(256, 8, 3, 1024) GS(256, 1, 3, 8192)
    CH(256, 16, 0, 1024) CH(256, 16, 0, 2048) CH(256, 16, 0, 512) CH(256, 16, 3, 1024) CH(256, 16, 1, 1024)
    CH(256, 16, 0, 65536) CH(256, 4, 0, 2048) CH(256, 4, 0, 4096) CH(512, 8, 0, 1024) CH(512, 8, 0, 512) CH(1024, 4, 0, 512)
    PF(256, 16, 0, 512) PF(256, 16, 0, 1024) PF(256, 16, 3, 1024) PF(256, 8, 0, 1024) PF(256, 8, 0, 2048) PF(512, 8, 0, 512) PF(512, 8, 0, 1024)
    vs.push_back({"read  thr256 U4 grid2048", [=]() { hipLaunchKernelGGL((k_read<256, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); }, 1.0 * n * 16});
    vs.push_back({"read  thr256 U8 grid2048", [=]() { hipLaunchKernelGGL((k_read<256, 8>), dim3(2048), dim3(256), 0, 0, a, b, n); }, 1.0 * n * 16});
    vs.push_back({"write thr256 U4 grid2048", [=]() { hipLaunchKernelGGL((k_write<256, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); }, 1.0 * n * 16});
    vs.push_back({"hipMemcpyDtoD", [=]() { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); }, cb});

    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int inner = 5;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            vs[i].go();
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; ++k) vs[i].go();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t[i].push_back(ms * 1000.f / inner);
            CK(hipGetLastError());
        }
    printf("array %zu MiB each\n%-40s %10s %10s %10s\n", mib, "variant", "median_us", "min_us", "GB/s(med)");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf("%-40s %10.2f %10.2f %10.1f\n", vs[i].name.c_str(), med, t[i][0], vs[i].bytes / med / 1e3);
    }
    return 0;
}
