// tools/kbench.hip -- developer microbenchmark (not part of the product library).
// Times variants of the register-resident pow2 kernel and copy kernels with the same access
// pattern on one MI355X, interleaved rounds in ONE process (guide rule 24), HIP events.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -I ndrustfft_amd/csrc -I tools tools/kbench.hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "pow2_kernel.h"
#include "pow2_persist.h"

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

template <typename K, typename T, typename RL> static Variant fft_variant(const char *name, const void *in, void *out, int64_t lanes, int n, size_t extra_lds = 0, int xcd_chunk = 0, int rot = 1) {
    HostTable t; build_tw<RL>(t);
    void *tw = upload_tw<T>(t);
    CK(hipFuncSetAttribute((const void *)k_pow2<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Pow2Args a; a.in = in; a.out = out; a.nlanes = lanes; a.pitch_in = n; a.pitch_out = n; a.inverse = 0; a.scale = 1.0; a.twp = tw; a.xcd_chunk = xcd_chunk;
    const size_t lds = K::LDS_BYTES + extra_lds;
    const int per_blk = K::THREADS / (n / K::E);
    const unsigned nblk = (unsigned)((lanes + per_blk - 1) / per_blk);
    Variant v;
    v.name = name;
    static size_t counter = 0;   // rot > 1: successive launches walk over `rot` distinct (in, out) pairs (cache-cold at any size)
    const size_t pair_bytes = (size_t)lanes * n * 2 * sizeof(T);
    v.launch = [=]() {
        Pow2Args b = a;
        const size_t k = rot > 1 ? (counter++ % (size_t)rot) : 0;
        b.in = (const char *)a.in + k * pair_bytes; b.out = (char *)a.out + k * pair_bytes;
        hipLaunchKernelGGL(k_pow2<K>, dim3(nblk), dim3(K::THREADS), lds, 0, b);
    };
    v.bytes = 2.0 * lanes * n * 2 * sizeof(T);
    v.check = true;
    return v;
}

// persistent, software-pipelined form (pow2_persist.h): grid = wg_per_cu x CUs workgroups (0 = what the occupancy query allows)
template <typename K, typename T, typename RL> static Variant persist_variant(const char *name, const void *in, void *out, int64_t lanes, int n, int wg_per_cu, int xcd_chunk, int rot) {
    using P = Pow2Persist<K>;
    HostTable t; build_tw<RL>(t);
    void *tw = upload_tw<T>(t);
    CK(hipFuncSetAttribute((const void *)k_pow2_persist<P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_pow2_persist<P>, P::THREADS, P::LDS_BYTES));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int per = wg_per_cu > 0 ? wg_per_cu : occ;
    Pow2Args a; a.in = in; a.out = out; a.nlanes = lanes; a.pitch_in = n; a.pitch_out = n; a.inverse = 0; a.scale = 1.0; a.twp = tw; a.xcd_chunk = xcd_chunk;
    const int per_blk = K::THREADS / (n / K::E);
    const unsigned nblk = (unsigned)((lanes + per_blk - 1) / per_blk);
    const unsigned grid = std::min<unsigned>(nblk, (unsigned)(per * pr.multiProcessorCount));
    static char nm[64][96]; static int ni = 0;
    snprintf(nm[ni], 96, "%s [occ %d, grid %u]", name, occ, grid);
    Variant v; v.name = nm[ni++];
    static size_t counter = 0;
    const size_t pair_bytes = (size_t)lanes * n * 2 * sizeof(T);
    v.launch = [=]() {
        Pow2Args b = a;
        const size_t k = rot > 1 ? (counter++ % (size_t)rot) : 0;
        b.in = (const char *)a.in + k * pair_bytes; b.out = (char *)a.out + k * pair_bytes;
        hipLaunchKernelGGL(k_pow2_persist<P>, dim3(grid), dim3(P::THREADS), P::LDS_BYTES, 0, b);
    };
    v.bytes = 2.0 * lanes * n * 2 * sizeof(T);
    v.check = true;
    return v;
}

// dynamic persistent grid (pow2_persist.h: Pow2Dyn): wg_per_cu x CUs workgroups take lane blocks from per-XCD counters
template <typename K, typename T, typename RL> static Variant dyn_variant(const char *name, const void *in, void *out, int64_t lanes, int n, int wg_per_cu, int xcd_chunk, int rot) {
    using D = Pow2Dyn<K>;
    HostTable t; build_tw<RL>(t);
    void *tw = upload_tw<T>(t);
    CK(hipFuncSetAttribute((const void *)k_pow2_dyn<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_pow2_dyn<D>, D::THREADS, D::LDS_BYTES));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int per = wg_per_cu > 0 ? wg_per_cu : occ;
    static unsigned *ctr = nullptr;
    if (!ctr) { CK(hipMalloc(&ctr, 4096)); CK(hipMemset(ctr, 0, 4096)); }
    Pow2Args a; a.in = in; a.out = out; a.nlanes = lanes; a.pitch_in = n; a.pitch_out = n; a.inverse = 0; a.scale = 1.0; a.twp = tw; a.xcd_chunk = xcd_chunk;
    const unsigned grid = std::min<unsigned>((unsigned)lanes, (unsigned)(per * pr.multiProcessorCount)) & ~7u;
    static char nm[64][96]; static int ni = 0;
    snprintf(nm[ni], 96, "%s [occ %d, grid %u]", name, occ, grid);
    Variant v; v.name = nm[ni++];
    static size_t counter = 0;
    const size_t pair_bytes = (size_t)lanes * n * 2 * sizeof(T);
    unsigned *c = ctr;
    v.launch = [=]() {
        Pow2Args b = a;
        const size_t k = rot > 1 ? (counter++ % (size_t)rot) : 0;
        b.in = (const char *)a.in + k * pair_bytes; b.out = (char *)a.out + k * pair_bytes;
        hipLaunchKernelGGL(k_pow2_dyn<D>, dim3(grid), dim3(D::THREADS), D::LDS_BYTES, 0, b, c);
    };
    v.bytes = 2.0 * lanes * n * 2 * sizeof(T);
    v.check = true;
    return v;
}

template <typename T> struct Bench {
    int n; int64_t lanes; int rounds; int rot = 1;
    cpx<T> *din, *dout, *dref;
    std::vector<Variant> vs;
    void init() {
        const size_t elems = (size_t)lanes * n * rot;
        CK(hipMalloc(&din, elems * sizeof(cpx<T>))); CK(hipMalloc(&dout, elems * sizeof(cpx<T>))); CK(hipMalloc(&dref, (size_t)lanes * n * sizeof(cpx<T>)));
        std::vector<cpx<T>> h(elems);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < elems; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].x = (T)((double)(s >> 11) * (1.0 / 9007199254740992.0) * 2 - 1);
            s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i].y = (T)((double)(s >> 11) * (1.0 / 9007199254740992.0) * 2 - 1);
        }
        CK(hipMemcpy(din, h.data(), elems * sizeof(cpx<T>), hipMemcpyHostToDevice));
    }
    template <typename K, typename RL> void add(const char *name, int xcd_chunk = 0) { vs.push_back(fft_variant<K, T, RL>(name, din, dout, lanes, n, 0, xcd_chunk, rot)); }
    template <typename K, typename RL> void addp(const char *name, int wg_per_cu, int xcd_chunk) { vs.push_back(persist_variant<K, T, RL>(name, din, dout, lanes, n, wg_per_cu, xcd_chunk, rot)); }
    template <typename K, typename RL> void addd(const char *name, int wg_per_cu, int xcd_chunk) { vs.push_back(dyn_variant<K, T, RL>(name, din, dout, lanes, n, wg_per_cu, xcd_chunk, rot)); }
    void run(double tol) {
        const size_t elems = (size_t)lanes * n;
        vs[0].launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(dref, dout, elems * sizeof(cpx<T>), hipMemcpyDeviceToDevice));
        std::vector<cpx<T>> href(1 << 16), hgot(1 << 16);
        CK(hipMemcpy(href.data(), dref, href.size() * sizeof(cpx<T>), hipMemcpyDeviceToHost));
        std::vector<std::vector<float>> t(vs.size());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int inner = 10;
        for (int r = 0; r < rounds; ++r)
            for (size_t i = 0; i < vs.size(); ++i) {
                vs[i].launch();
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < inner; ++k) vs[i].launch();
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[i].push_back(ms * 1000.f / inner);
                if (r == 0 && vs[i].check) {
                    CK(hipMemcpy(hgot.data(), dout, hgot.size() * sizeof(cpx<T>), hipMemcpyDeviceToHost));
                    double md = 0, mr = 0;
                    for (size_t k = 0; k < hgot.size(); ++k) {
                        md = std::max(md, std::max(fabs((double)hgot[k].x - href[k].x), fabs((double)hgot[k].y - href[k].y)));
                        mr = std::max(mr, std::max(fabs((double)href[k].x), fabs((double)href[k].y)));
                    }
                    if (md / mr > tol) printf("!! %s differs from variant 0: rel %.3e\n", vs[i].name.c_str(), md / mr);
                }
                CK(hipGetLastError());
            }
        printf("%-46s %10s %10s %10s %8s\n", "variant", "median_us", "min_us", "GB/s(med)", "frac8T");
        for (size_t i = 0; i < vs.size(); ++i) {
            std::sort(t[i].begin(), t[i].end());
            const double med = t[i][t[i].size() / 2], mn = t[i][0];
            printf("%-46s %10.2f %10.2f %10.1f %8.3f\n", vs[i].name.c_str(), med, mn, vs[i].bytes / med / 1e3, vs[i].bytes / med / 1e3 / 8000.0);
        }
    }
};

#define V(B, T, N, name, TPL, HALF, NT, VEC, ...) B.template add<Pow2Kernel<T, N, TPL, (TPL >= 256 ? 1 : 256 / TPL), HALF, RadixList<__VA_ARGS__>, 0, 1, NT, VEC>, RadixList<__VA_ARGS__>>(name)

int main(int argc, char **argv) {
    const std::string what = argc > 1 ? argv[1] : "f32_8192";
    const int rounds = argc > 2 ? atoi(argv[2]) : 11;
    if (what == "f32_8192") {
        Bench<float> b{8192, 4096, rounds}; b.init();
        V(b, float, 8192, "half 512x16 8.16.8.8 vec2 (product)", 512, true, 1, 2, 8, 16, 8, 8);
        V(b, float, 8192, "full 512x16 8.16.8.8 vec2", 512, false, 1, 2, 8, 16, 8, 8);
        V(b, float, 8192, "half 1024x8 4.8.8.8.4 vec2", 1024, true, 1, 2, 4, 8, 8, 8, 4);
        V(b, float, 8192, "full 1024x8 4.8.8.8.4 vec2", 1024, false, 1, 2, 4, 8, 8, 8, 4);
        V(b, float, 8192, "half 256x32 16.16.8.4 vec2", 256, true, 1, 2, 16, 16, 8, 4);
        V(b, float, 8192, "full 256x32 16.16.8.4 vec2", 256, false, 1, 2, 16, 16, 8, 4);
        V(b, float, 8192, "half 256x32 16.32?no 8.8.8.16 vec2", 256, true, 1, 2, 8, 8, 8, 16);
        V(b, float, 8192, "half 512x16 16.16.8.4 vec1", 512, true, 1, 1, 16, 16, 8, 4);
        V(b, float, 8192, "half 512x16 8.16.8.8 vec2 nt3", 512, true, 3, 2, 8, 16, 8, 8);
        b.run(1e-5);
    } else if (what == "f32_4096") {
        Bench<float> b{4096, 8192, rounds}; b.init();
        V(b, float, 4096, "half 256x16 8.8.8.8 vec2 (product)", 256, true, 1, 2, 8, 8, 8, 8);
        V(b, float, 4096, "full 256x16 8.8.8.8 vec2", 256, false, 1, 2, 8, 8, 8, 8);
        V(b, float, 4096, "half 512x8 4.8.8.4.4 vec2", 512, true, 1, 2, 4, 8, 8, 4, 4);
        V(b, float, 4096, "full 512x8 4.8.8.4.4 vec2", 512, false, 1, 2, 4, 8, 8, 4, 4);
        V(b, float, 4096, "full 128x32 16.16.16 vec2", 128, false, 1, 2, 16, 16, 16);
        V(b, float, 4096, "half 128x32 16.16.16 vec2", 128, true, 1, 2, 16, 16, 16);
        b.run(1e-5);
    } else if (what == "f64_8192") {
        Bench<double> b{8192, 2048, rounds}; b.init();
        V(b, double, 8192, "half 512x16 16.16.8.4 (product)", 512, true, 1, 1, 16, 16, 8, 4);
        V(b, double, 8192, "half 1024x8 8.8.8.8.2", 1024, true, 1, 1, 8, 8, 8, 8, 2);
        V(b, double, 8192, "half 1024x8 8.8.8.4.4", 1024, true, 1, 1, 8, 8, 8, 4, 4);
        V(b, double, 8192, "half 512x16 8.8.8.16", 512, true, 1, 1, 8, 8, 8, 16);
        V(b, double, 8192, "half 512x16 16.8.8.8", 512, true, 1, 1, 16, 8, 8, 8);
        V(b, double, 8192, "half 256x32 16.16.32?no 16.16.8.4", 256, true, 1, 1, 16, 16, 8, 4);
        b.run(1e-12);
    } else if (what == "creep") {
        // does the per-launch time drift under sustained back-to-back launches? (clock / power management)
        Bench<double> b{4096, 4096, 1}; b.init();
        V(b, double, 4096, "fft 512x8 nt1", 512, true, 1, 1, 8, 8, 8, 8);
        {
            Variant v; v.name = "copy_lane 512x8"; const double2 *i2 = (const double2 *)b.din; double2 *o2 = (double2 *)b.dout;
            v.launch = [=]() { hipLaunchKernelGGL((k_copy_lane<512, 8>), dim3(4096), dim3(512), 0, 0, i2, o2); };
            v.bytes = 2.0 * 4096 * 4096 * 16; v.check = false; b.vs.push_back(v);
        }
        const int groups = argc > 2 ? atoi(argv[2]) : 40, per = 25;
        std::vector<hipEvent_t> ev(groups + 1);
        for (auto &e : ev) CK(hipEventCreate(&e));
        for (auto &v : b.vs) {
            CK(hipDeviceSynchronize());
            usleep(300000);
            CK(hipEventRecord(ev[0], 0));
            for (int g = 0; g < groups; ++g) { for (int k = 0; k < per; ++k) v.launch(); CK(hipEventRecord(ev[g + 1], 0)); }
            CK(hipDeviceSynchronize());
            printf("%s: us/launch per group of %d:", v.name.c_str(), per);
            for (int g = 0; g < groups; ++g) { float ms; CK(hipEventElapsedTime(&ms, ev[g], ev[g + 1])); printf(" %.1f", ms * 1000.f / per); }
            printf("\n");
        }
    } else if (what == "f64_4096") {
        const int64_t lanes = argc > 3 ? atoll(argv[3]) : 8192;
        Bench<double> b{4096, lanes, rounds}; b.init();
        V(b, double, 4096, "half 512x8 8.8.8.8 nt1 (st)", 512, true, 1, 1, 8, 8, 8, 8);
        V(b, double, 4096, "half 512x8 8.8.8.8 nt3 (ld+st)", 512, true, 3, 1, 8, 8, 8, 8);
        V(b, double, 4096, "half 512x8 8.8.8.8 nt0", 512, true, 0, 1, 8, 8, 8, 8);
        V(b, double, 4096, "half 512x8 8.8.8.8 nt2 (ld)", 512, true, 2, 1, 8, 8, 8, 8);
        V(b, double, 4096, "half 256x16 16.16.16 nt1", 256, true, 1, 1, 16, 16, 16);
        V(b, double, 4096, "half 256x16 16.16.16 nt3", 256, true, 3, 1, 16, 16, 16);
        b.run(1e-12);
    } else if (what == "f64_cold") {
        // cache-cold: 32768 lanes = 2 GiB in + 2 GiB out per launch, far beyond the 256 MiB Infinity Cache
        const int64_t lanes = argc > 3 ? atoll(argv[3]) : 32768;
        Bench<double> b{4096, lanes, rounds}; b.init();
#define VC(name, TPL, LPB, FL, MINW, NT, ...) b.template add<Pow2Kernel<double, 4096, TPL, LPB, true, RadixList<__VA_ARGS__>, FL, MINW, NT, 1>, RadixList<__VA_ARGS__>>(name)
        VC("512x8 8.8.8.8 TWP nt1 (product)", 512, 1, 16, 1, 1, 8, 8, 8, 8);
        VC("512x8 8.8.8.8 TWP nt3", 512, 1, 16, 1, 3, 8, 8, 8, 8);
        VC("512x8 8.8.8.8 TWP nt0", 512, 1, 16, 1, 0, 8, 8, 8, 8);
        VC("512x8 8.8.8.8 TWP nt2", 512, 1, 16, 1, 2, 8, 8, 8, 8);
        VC("512x8 8.8.8.8 plain-tw nt1", 512, 1, 0, 1, 1, 8, 8, 8, 8);
        VC("2 lanes/WG 1024thr TWP nt1", 512, 2, 16, 1, 1, 8, 8, 8, 8);
        VC("2 lanes/WG 1024thr TWP nt3", 512, 2, 16, 1, 3, 8, 8, 8, 8);
        VC("256x16 16.16.16 nt1", 256, 1, 0, 1, 1, 16, 16, 16);
        VC("256x16 16.16.16 nt3", 256, 1, 0, 1, 3, 16, 16, 16);
        VC("256x16 4 lanes/WG nt1", 256, 4, 0, 1, 1, 16, 16, 16);
        VC("1024x4 4.4.4.4.4.4 nt1", 1024, 1, 0, 1, 1, 4, 4, 4, 4, 4, 4);
        VC("ABLATE load+store only", 512, 1, 7, 1, 1, 8, 8, 8, 8);
        VC("ABLATE no LDS exchange", 512, 1, 2, 1, 1, 8, 8, 8, 8);
        b.run(1e-12);
    } else if (what == "f64_persist") {
        // persistent software-pipelined grid (pow2_persist.h) against the product kernel; lanes = 4096 with rot = 6 is bench.py's cache-cold region
        const int64_t lanes = argc > 3 ? atoll(argv[3]) : 4096;
        Bench<double> b{4096, lanes, rounds}; b.rot = argc > 4 ? atoi(argv[4]) : 6; b.init();
        using RL8 = RadixList<8, 8, 8, 8>;
        using K1 = Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 1, 1, 1>;
        using K3 = Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 1, 3, 1>;
        using K1w4 = Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 4, 1, 1>;
        using K3w4 = Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 4, 3, 1>;
        b.template add<K1, RL8>("product nt1 chunk 8", 8);
        b.template add<K3, RL8>("product nt3 chunk 8", 8);
        b.template addd<K3, RL8>("dyn nt3 c8 4/CU", 0, 8);
        b.template addd<K1, RL8>("dyn nt1 c8 4/CU", 0, 8);
        b.template addd<K3, RL8>("dyn nt3 c8 3/CU", 3, 8);
        b.template addd<K3, RL8>("dyn nt3 c32 4/CU", 0, 32);
        b.template addp<K3w4, RL8>("persist nt3 c8", 0, 8);
        b.template addp<K3w4, RL8>("persist nt3 c32", 0, 32);
        b.template add<K3, RL8>("product nt3 chunk 8 (again)", 8);
        b.run(1e-12);
    } else if (what == "f64_map") {
        // XCD-aware lane-block maps (device_common.h: xcd_block) x lanes per workgroup x load policy; lanes = 4096 is the
        // Infinity-Cache-warm bench shape, 32768 the cache-cold one
        const int64_t lanes = argc > 3 ? atoll(argv[3]) : 32768;
        Bench<double> b{4096, lanes, rounds}; b.rot = argc > 4 ? atoi(argv[4]) : 1; b.init();
        using RL8 = RadixList<8, 8, 8, 8>;
        const int eighth1 = (int)(lanes / 8), eighth2 = (int)(lanes / 16);
        static char names[64][64]; int ni = 0;
        for (int chunk : {0, 2, 8, 32, 128, 512, -1}) {
            const int c1 = chunk < 0 ? eighth1 : chunk, c2 = chunk < 0 ? eighth2 : chunk / 2;
            snprintf(names[ni], 64, "1 lane/WG nt1 chunk %d", c1);  b.template add<Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 1, 1, 1>, RL8>(names[ni++], c1);
            snprintf(names[ni], 64, "1 lane/WG nt3 chunk %d", c1);  b.template add<Pow2Kernel<double, 4096, 512, 1, true, RL8, 16, 1, 3, 1>, RL8>(names[ni++], c1);
            snprintf(names[ni], 64, "2 lanes/WG nt1 chunk %d", c2); b.template add<Pow2Kernel<double, 4096, 512, 2, true, RL8, 16, 1, 1, 1>, RL8>(names[ni++], c2);
            snprintf(names[ni], 64, "2 lanes/WG nt3 chunk %d", c2); b.template add<Pow2Kernel<double, 4096, 512, 2, true, RL8, 16, 1, 3, 1>, RL8>(names[ni++], c2);
        }
        b.run(1e-12);
    } else if (what == "f64_full") {
        { Bench<double> b{4096, 4096, rounds}; b.init();
          V(b, double, 4096, "4096 half 512x8 8.8.8.8 (product)", 512, true, 1, 1, 8, 8, 8, 8);
          V(b, double, 4096, "4096 full 512x8 8.8.8.8", 512, false, 1, 1, 8, 8, 8, 8);
          b.template add<Pow2Kernel<double, 4096, 512, 1, true, RadixList<8, 8, 8, 8>, 16, 1, 1, 1>, RadixList<8, 8, 8, 8>>("4096 half 512x8 8.8.8.8 TW_POWERS");
          b.run(1e-12); }
        { Bench<double> b{8192, 2048, rounds}; b.init();
          V(b, double, 8192, "8192 half 512x16 8.8.8.16 (product)", 512, true, 1, 1, 8, 8, 8, 16);
          V(b, double, 8192, "8192 full 512x16 8.8.8.16", 512, false, 1, 1, 8, 8, 8, 16); b.run(1e-12); }
        { Bench<double> b{2048, 8192, rounds}; b.init();
          V(b, double, 2048, "2048 half 128x16 16.16.8 (product)", 128, true, 1, 1, 16, 16, 8);
          V(b, double, 2048, "2048 full 128x16 16.16.8", 128, false, 1, 1, 16, 16, 8); b.run(1e-12); }
    } else if (what == "f32_small") {
        { Bench<float> b{64, 524288, rounds}; b.init();
          V(b, float, 64, "64 half 4x16 8.8 vec2 (product)", 4, true, 1, 2, 8, 8);
          V(b, float, 64, "64 full 4x16 8.8 vec2", 4, false, 1, 2, 8, 8);
          V(b, float, 64, "64 half 8x8 8.8 vec1", 8, true, 1, 1, 8, 8);
          V(b, float, 64, "64 full 8x8 8.8 vec1", 8, false, 1, 1, 8, 8);
          V(b, float, 64, "64 full 2x32 16.4?no 8.8 vec2", 2, false, 1, 2, 8, 8); b.run(1e-5); }
        { Bench<float> b{128, 262144, rounds}; b.init();
          V(b, float, 128, "128 half 8x16 8.2.8 vec2 (product)", 8, true, 1, 2, 8, 2, 8);
          V(b, float, 128, "128 full 8x16 8.2.8 vec2", 8, false, 1, 2, 8, 2, 8);
          V(b, float, 128, "128 full 8x16 8.16 vec1", 8, false, 1, 1, 8, 16); b.run(1e-5); }
        { Bench<float> b{256, 131072, rounds}; b.init();
          V(b, float, 256, "256 half 16x16 8.4.8 vec2 (product)", 16, true, 1, 2, 8, 4, 8);
          V(b, float, 256, "256 full 16x16 8.4.8 vec2", 16, false, 1, 2, 8, 4, 8);
          V(b, float, 256, "256 full 16x16 16.16 vec1", 16, false, 1, 1, 16, 16); b.run(1e-5); }
        { Bench<float> b{512, 65536, rounds}; b.init();
          V(b, float, 512, "512 half 32x16 8.8.8 vec2 (product)", 32, true, 1, 2, 8, 8, 8);
          V(b, float, 512, "512 full 32x16 8.8.8 vec2", 32, false, 1, 2, 8, 8, 8); b.run(1e-5); }
    } else if (what == "f32_halffull") {
        { Bench<float> b{1024, 32768, rounds}; b.init();
          V(b, float, 1024, "1024 half 64x16 8.16.8 (product)", 64, true, 1, 2, 8, 16, 8);
          V(b, float, 1024, "1024 full 64x16 8.16.8", 64, false, 1, 2, 8, 16, 8); b.run(1e-5); }
        { Bench<float> b{2048, 16384, rounds}; b.init();
          V(b, float, 2048, "2048 half 128x16 8.4.8.8 (product)", 128, true, 1, 2, 8, 4, 8, 8);
          V(b, float, 2048, "2048 full 128x16 8.4.8.8", 128, false, 1, 2, 8, 4, 8, 8);
          V(b, float, 2048, "2048 full 128x16 16.16.8 vec1", 128, false, 1, 1, 16, 16, 8); b.run(1e-5); }
        { Bench<float> b{4096, 8192, rounds}; b.init();
          V(b, float, 4096, "4096 half 256x16 8.8.8.8 (product)", 256, true, 1, 2, 8, 8, 8, 8);
          V(b, float, 4096, "4096 full 256x16 8.8.8.8", 256, false, 1, 2, 8, 8, 8, 8);
          V(b, float, 4096, "4096 full 256x16 8.16.8?  8.8.8.8->16.16.16 no vec", 256, false, 1, 1, 16, 16, 16); b.run(1e-5); }
    } else if (what == "f32_16384") {
        Bench<float> b{16384, 2048, rounds}; b.init();
        V(b, float, 16384, "half 1024x16 8.16.16.8 vec2 (product)", 1024, true, 1, 2, 8, 16, 16, 8);
        V(b, float, 16384, "half 512x32 16.16.8.8 vec2", 512, true, 1, 2, 16, 16, 8, 8);
        V(b, float, 16384, "half 512x32 8.16.16.8 vec2", 512, true, 1, 2, 8, 16, 16, 8);
        V(b, float, 16384, "half 512x32 16.8.8.16 vec2", 512, true, 1, 2, 16, 8, 8, 16);
        V(b, float, 16384, "full 512x32 16.16.8.8 vec2", 512, false, 1, 2, 16, 16, 8, 8);
        V(b, float, 16384, "half 1024x16 16.16.16.4 vec1", 1024, true, 1, 1, 16, 16, 16, 4);
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 0, 8, 1, 2>, RadixList<8, 16, 16, 8>>("half 1024x16 8.16.16.8 vec2 MINW=8");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 0, 8, 1, 1>, RadixList<8, 16, 16, 8>>("half 1024x16 8.16.16.8 vec1 MINW=8");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 16, 1, 1, 2>, RadixList<8, 16, 16, 8>>("TW_POWERS 8.16.16.8");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 2, 1, 1, 2>, RadixList<8, 16, 16, 8>>("ABLATE no LDS exchange");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 4, 1, 1, 2>, RadixList<8, 16, 16, 8>>("ABLATE no butterflies");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 7, 1, 1, 2>, RadixList<8, 16, 16, 8>>("ABLATE load+store only");
        b.template add<Pow2Kernel<float, 16384, 1024, 1, true, RadixList<8, 16, 16, 8>, 1, 1, 1, 2>, RadixList<8, 16, 16, 8>>("ABLATE no twiddles");
        b.run(1e-5);
    } else if (what == "f64_16384") {
        Bench<double> b{16384, 1024, rounds}; b.init();
        V(b, double, 16384, "half 1024x16 16.16.16.4 (product)", 1024, true, 1, 1, 16, 16, 16, 4);
        V(b, double, 16384, "half 1024x16 8.8.16.16", 1024, true, 1, 1, 8, 8, 16, 16);
        V(b, double, 16384, "half 512x32 16.16.16.4", 512, true, 1, 1, 16, 16, 16, 4);
        V(b, double, 16384, "half 1024x16 8.8.8.8.4", 1024, true, 1, 1, 8, 8, 8, 8, 4);
        b.template add<Pow2Kernel<double, 16384, 1024, 1, true, RadixList<16, 16, 16, 4>, 16, 1, 1, 1>, RadixList<16, 16, 16, 4>>("TW_POWERS 16.16.16.4");
        b.template add<Pow2Kernel<double, 16384, 1024, 1, true, RadixList<8, 8, 16, 16>, 16, 1, 1, 1>, RadixList<8, 8, 16, 16>>("TW_POWERS 8.8.16.16");
        b.run(1e-12);
    }
    return 0;
}
