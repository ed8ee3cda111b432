// tools/kbench.hip -- developer microbenchmark (not part of the product library).
// Times variants of the register-resident pow2 kernel and copy kernels with the same access
// pattern on one MI355X, interleaved rounds in ONE process (guide rule 24), HIP events.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ndrustfft_amd/csrc tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "pow2_kernel.h"

using namespace ndfft;

namespace ndfft {   // the two hooks engine.h declares, so this tool links without the library
int fail(int code, const std::string &msg) { fprintf(stderr, "fail: %s\n", msg.c_str()); return code; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- copy kernels: what the memory system gives this access pattern -----------------------------
// each block moves one contiguous 64 KiB "lane": thread t loads x[t + 256 r], r = 0..15 (16 B each)
template <int THREADS, int E> __global__ __launch_bounds__(THREADS) void k_copy_lane(const double2 *in, double2 *out) {
    extern __shared__ char dummy[];
    const double2 *src = in + (size_t)blockIdx.x * THREADS * E;
    double2 *dst = out + (size_t)blockIdx.x * THREADS * E;
    double2 v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = src[threadIdx.x + r * THREADS];
#pragma unroll
    for (int r = 0; r < E; ++r) dst[threadIdx.x + r * THREADS] = v[r];
}
// grid-stride streaming copy, 16 B per thread per iteration
__global__ __launch_bounds__(256) void k_copy_stream(const double2 *in, double2 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

struct Variant { std::string name; std::function<void()> launch; double bytes; bool check; };

template <typename T> static void *upload_tw(const HostTable &t) {
    std::vector<T> h(2 * t.re.size());
    for (size_t i = 0; i < t.re.size(); ++i) { h[2 * i] = (T)t.re[i]; h[2 * i + 1] = (T)t.im[i]; }
    void *d; CK(hipMalloc(&d, std::max<size_t>(h.size() * sizeof(T), 16)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

template <typename K, typename T, typename RL> static Variant fft_variant(const char *name, const void *in, void *out, int64_t lanes, int n, size_t extra_lds = 0) {
    HostTable t; build_tw<RL>(t);
    void *tw = upload_tw<T>(t);
    CK(hipFuncSetAttribute((const void *)k_pow2<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Pow2Args a; a.in = in; a.out = out; a.nlanes = lanes; a.pitch_in = n; a.pitch_out = n; a.inverse = 0; a.scale = 1.0; a.twp = tw;
    const size_t lds = K::LDS_BYTES + extra_lds;
    const int per_blk = K::THREADS / (n / K::E);
    const unsigned nblk = (unsigned)((lanes + per_blk - 1) / per_blk);
    Variant v;
    v.name = name;
    v.launch = [=]() { hipLaunchKernelGGL(k_pow2<K>, dim3(nblk), dim3(K::THREADS), lds, 0, a); };
    v.bytes = 2.0 * lanes * n * 2 * sizeof(T);
    v.check = true;
    return v;
}

int main(int argc, char **argv) {
    const int n = 4096;
    const int64_t lanes = argc > 1 ? atoll(argv[1]) : 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 15;
    const size_t elems = (size_t)lanes * n;
    double2 *din, *dout, *dref;
    CK(hipMalloc(&din, elems * 16)); CK(hipMalloc(&dout, elems * 16)); CK(hipMalloc(&dref, elems * 16));
    {
        std::vector<double2> h(elems);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < elems; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].x = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2 - 1;
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].y = (double)(s >> 11) * (1.0 / 9007199254740992.0) * 2 - 1;
        }
        CK(hipMemcpy(din, h.data(), elems * 16, hipMemcpyHostToDevice));
    }
    using R16 = RadixList<16, 16, 16>;
    using R8 = RadixList<8, 8, 8, 8>;
    using R1684 = RadixList<16, 16, 4, 4>;
    std::vector<Variant> vs;
#define FV(name, TPL, HALF, RL, FLAGS, MINW, NT) vs.push_back(fft_variant<Pow2Kernel<double, 4096, TPL, 1, HALF, RL, FLAGS, MINW, NT>, double, RL>(name, din, dout, lanes, n));
    FV("half_256x16_r16^3 nt0", 256, true, R16, 0, 1, 0)
    FV("half_256x16_r16^3 nt1(st)", 256, true, R16, 0, 1, 1)
    FV("half_256x16_r16^3 nt2(ld)", 256, true, R16, 0, 1, 2)
    FV("half_256x16_r16^3 nt3", 256, true, R16, 0, 1, 3)
    FV("full_256x16_r16^3 nt1", 256, false, R16, 0, 1, 1)
    FV("half_512x8_r8^4 nt0", 512, true, R8, 0, 1, 0)
    FV("half_512x8_r8^4 nt1", 512, true, R8, 0, 1, 1)
    FV("half_512x8_r8^4 nt3", 512, true, R8, 0, 1, 3)
    FV("full_512x8_r8^4 nt1", 512, false, R8, 0, 1, 1)
    FV("ablate: no twiddles nt1", 256, true, R16, 1, 1, 1)
    FV("ablate: no LDS exchange nt1", 256, true, R16, 2, 1, 1)
    FV("ablate: load+store only nt1", 256, true, R16, 7, 1, 1)
    FV("ablate: load+store only nt3", 256, true, R16, 7, 1, 3)
    FV("ablate: 512x8 load+store only nt1", 512, true, R8, 7, 1, 1)
    for (auto &v : vs) if (v.name.rfind("ablate", 0) == 0) v.check = false;
    {   // copy ceilings, occupancy limited through dummy LDS like the FFT kernel (34.8 KiB -> 4 blocks/CU)
        const double bytes = 2.0 * elems * 16;
        for (size_t lds : {(size_t)0, (size_t)34880, (size_t)69700}) {
            CK(hipFuncSetAttribute((const void *)k_copy_lane<256, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            char nm[64]; snprintf(nm, sizeof nm, "copy lane-pattern 256x16, lds=%zu", lds);
            vs.push_back({nm, [=]() { hipLaunchKernelGGL((k_copy_lane<256, 16>), dim3((unsigned)lanes), dim3(256), lds, 0, din, dout); }, bytes, false});
        }
        vs.push_back({"copy stream grid=2048", [=]() { hipLaunchKernelGGL(k_copy_stream, dim3(2048), dim3(256), 0, 0, din, dout, elems); }, bytes, false});
        vs.push_back({"copy stream grid=8192", [=]() { hipLaunchKernelGGL(k_copy_stream, dim3(8192), dim3(256), 0, 0, din, dout, elems); }, bytes, false});
    }
    // reference output from the product variant
    vs[0].launch(); CK(hipDeviceSynchronize());
    CK(hipMemcpy(dref, dout, elems * 16, hipMemcpyDeviceToDevice));
    std::vector<double2> href(1 << 16), hgot(1 << 16);
    CK(hipMemcpy(href.data(), dref, href.size() * 16, hipMemcpyDeviceToHost));

    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int inner = 10;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            vs[i].launch();   // warm
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; ++k) vs[i].launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            t[i].push_back(ms * 1000.f / inner);
            if (r == 0 && vs[i].check) {
                CK(hipMemcpy(hgot.data(), dout, hgot.size() * 16, hipMemcpyDeviceToHost));
                double md = 0, mr = 0;
                for (size_t k = 0; k < hgot.size(); ++k) {
                    md = std::max(md, std::max(fabs(hgot[k].x - href[k].x), fabs(hgot[k].y - href[k].y)));
                    mr = std::max(mr, std::max(fabs(href[k].x), fabs(href[k].y)));
                }
                if (md / mr > 1e-12) printf("!! %s differs from product output: rel %.3e\n", vs[i].name.c_str(), md / mr);
            }
            CK(hipGetLastError());
        }
    printf("%-42s %10s %10s %10s %8s\n", "variant", "median_us", "min_us", "GB/s(med)", "frac8T");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2], mn = t[i][0];
        printf("%-42s %10.2f %10.2f %10.1f %8.3f\n", vs[i].name.c_str(), med, mn, vs[i].bytes / med / 1e3, vs[i].bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
