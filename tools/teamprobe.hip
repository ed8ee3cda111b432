// tools/teamprobe.hip -- developer tool (round 3): can the intermediate of the column four-step stay in ONE XCD's L2?
//
// BASELINE configs[2] part A (ndfft_r2c along axis 0 of 8192 x 8192 f32) runs as two launches: stage A (column FFTs of length 128 over
// a = row / 64) writes an intermediate S[k1][b][i] of the size of the output, stage B (length-64 FFTs over b) reads it back: 1.07 GB of
// traffic for 0.54 GB of algorithmic bytes.  This probe moves the SAME bytes in the same tile shapes without any FFT, two ways:
//   two   : two launches, the intermediate goes through HBM / the Infinity Cache (today's col_split)
//   team  : ONE persistent launch.  The array is cut into strips of 64 columns (intermediate of a strip: 2 MiB); the workgroups of one
//           XCD (read from the hardware register XCC_ID -- not assumed from blockIdx) form a team that claims strips from a global
//           counter and works through "all A tiles of the strip, then all B tiles" from a per-XCD ticket counter; a B tile waits for
//           the strip's A-tile count.  Producer and consumer share an L2, so the intermediate is read back from L2.
// Every wait is bounded (a poll budget, then an error flag): the probe cannot hang the GPU.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/teamprobe.hip -o tools/teamprobe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include <unistd.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int ROWS = 8192, COLS = 8192, F1 = 128, F2 = 64, WS = 64;     // WS: strip width in columns
constexpr int NSTRIP = COLS / WS, K1 = 64;                               // (the real transform has 65 k1 rows; 64 keeps the probe square)
constexpr size_t STRIP_ELEMS = (size_t)K1 * F2 * WS;                     // complex elements of one strip's intermediate

// A tile (strip s, b): rows 64 a + b, a < 128, columns [64 s, 64 s + 64) of the real input  ->  S[s][k1][b][i], k1 < 64 (complex)
__device__ __forceinline__ void tile_a(const float *in, float2 *S, int s, int b, int nt_in) {
    const int t = threadIdx.x;                       // 512 threads: 16 threads per 256-byte row segment, 32 rows per instruction
    const int piece = t & 15, r0 = t >> 4;
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int a = r0 + 32 * u;
        const v4f *p = (const v4f *)(in + (size_t)(F2 * a + b) * COLS + (size_t)s * WS) + piece;
        v[u] = nt_in ? __builtin_nontemporal_load(p) : *p;
    }
    // 64 k1 rows x 512 bytes: 32 threads per row, 16 rows per instruction; values are a cheap function of the loads
    const int q = t & 31, k0 = t >> 5;
    float2 *dst = S + (size_t)s * STRIP_ELEMS;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k1 = k0 + 16 * u;
        v4f w = v[u] + v[(u + 1) & 3];
        *(v4f *)(dst + ((size_t)k1 * F2 + b) * WS + 2 * q) = w;      // plain store: stays in this XCD's L2
    }
}
// B tile (strip s, k1): S[s][k1][b][i], b < 64  ->  output rows k1 + 128 k2, k2 < 64, columns [64 s, 64 s + 64) (complex)
__device__ __forceinline__ void tile_b(const float2 *S, float2 *out, int s, int k1) {
    const int t = threadIdx.x;                       // 32 threads per 512-byte row, 16 rows per instruction
    const int q = t & 31, r0 = t >> 5;
    const float2 *src = S + (size_t)s * STRIP_ELEMS + (size_t)k1 * F2 * WS;
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const v4f *)(src + (size_t)(r0 + 16 * u) * WS + 2 * q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k2 = r0 + 16 * u;
        v4f w = v[u] * 1.0001f;
        __builtin_nontemporal_store(w, (v4f *)(out + (size_t)(k1 + K1 * k2) * COLS + (size_t)s * WS + 2 * q));
    }
}

// ---- two launches (tile order of today's kernels: strip fastest) -----------------------------------------------------------
__global__ __launch_bounds__(512) void k_a(const float *in, float2 *S, int nt_in) { tile_a(in, S, blockIdx.x % NSTRIP, blockIdx.x / NSTRIP, nt_in); }
__global__ __launch_bounds__(512) void k_b(const float2 *S, float2 *out) { tile_b(S, out, blockIdx.x % NSTRIP, blockIdx.x / NSTRIP); }

// ---- one persistent launch, XCD teams, STATIC pipelined schedule ---------------------------------------------------------------
// (first attempt, kept in git history: a dynamic ticket per tile -- 3-4 dependent memory round trips per 64 KiB tile made it 6x SLOWER
//  than two launches.)  Now: team x = blockIdx.x % 8 (checked against XCC_ID), rank r = blockIdx.x / 8 of T = gridDim.x / 8; the team's
// strips are x, x + 8, ...; per strip a workgroup does its share of A tiles, ONE atomic add, and -- one strip later -- one wait:
//     A(0); for k: { A(k+1); wait(k); B(k) }
// Counters only grow (target = 64 * launch number): no reset, no "last workgroup".
template <int WSX> struct Geo {
    static constexpr int NSTRIPX = COLS / WSX;
    static constexpr size_t STRIPX = (size_t)K1 * F2 * WSX;
};
template <int WSX> __device__ __forceinline__ void tile_ax(const float *in, float2 *S, int s, int b, int nt_in) {
    constexpr int TPS = WSX * 4 / 16, RPI = 512 / TPS, U = F1 / RPI;
    const int t = threadIdx.x, piece = t % TPS, r0 = t / TPS;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const v4f *p = (const v4f *)(in + (size_t)(F2 * (r0 + RPI * u) + b) * COLS + (size_t)s * WSX) + piece;
        v[u] = nt_in ? __builtin_nontemporal_load(p) : *p;
    }
    constexpr int TPS2 = WSX * 8 / 16, RPI2 = 512 / TPS2, U2 = K1 / RPI2;
    const int q = t % TPS2, k0 = t / TPS2;
    float2 *dst = S + (size_t)s * Geo<WSX>::STRIPX;
#pragma unroll
    for (int u = 0; u < U2; ++u) {
        const v4f w = v[u % U] + v[(u + 1) % U];
        *(v4f *)(dst + ((size_t)(k0 + RPI2 * u) * F2 + b) * WSX + 2 * q) = w;      // plain store: stays in this XCD's L2
    }
}
template <int WSX> __device__ __forceinline__ void tile_bx(const float2 *S, float2 *out, int s, int k1, int nt_s = 0) {
    constexpr int TPS2 = WSX * 8 / 16, RPI2 = 512 / TPS2, U2 = F2 / RPI2;
    const int t = threadIdx.x, q = t % TPS2, r0 = t / TPS2;
    const float2 *src = S + (size_t)s * Geo<WSX>::STRIPX + (size_t)k1 * F2 * WSX;
    v4f v[U2];
#pragma unroll
    for (int u = 0; u < U2; ++u) { const v4f *p = (const v4f *)(src + (size_t)(r0 + RPI2 * u) * WSX + 2 * q); v[u] = nt_s ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
    for (int u = 0; u < U2; ++u)
        __builtin_nontemporal_store(v[u] * 1.0001f, (v4f *)(out + (size_t)(k1 + K1 * (r0 + RPI2 * u)) * COLS + (size_t)s * WSX + 2 * q));
}
template <int WSX> __global__ __launch_bounds__(512) void k_ax(const float *in, float2 *S, int nt_in) { tile_ax<WSX>(in, S, blockIdx.x % Geo<WSX>::NSTRIPX, blockIdx.x / Geo<WSX>::NSTRIPX, nt_in); }
template <int WSX> __global__ __launch_bounds__(512) void k_bx(const float2 *S, float2 *out) { tile_bx<WSX>(S, out, blockIdx.x % Geo<WSX>::NSTRIPX, blockIdx.x / Geo<WSX>::NSTRIPX); }

struct TeamCtl {
    unsigned error;                  // 1: a bounded wait ran out, 2: blockIdx % 8 is not the XCD
    unsigned pad_[15];
    unsigned done_a[COLS / 32];      // A tiles finished, per strip; grows by F2 per launch
    unsigned long long prof[1024][6];   // per workgroup: 10 ns ticks in A tiles, drain + barrier, poll, acquire, B tiles; polls made
};
#ifndef POLL_BUDGET
#define POLL_BUDGET (1u << 22)
#endif
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
__device__ __forceinline__ unsigned uload(const unsigned *p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }

template <int WSX, int DEPTH> __global__ __launch_bounds__(512) void k_team(const float *in, float2 *S, float2 *out, TeamCtl *ctl, int nt_in, unsigned launch_no) {
    const unsigned x = blockIdx.x & 7u, r = blockIdx.x >> 3, T = gridDim.x >> 3;
    if (xcc_id() != x) { if (threadIdx.x == 0) atomicExch(&ctl->error, 2u); return; }   // (uniform: the whole workgroup is on one XCD)
    constexpr int NK = Geo<WSX>::NSTRIPX / 8;
    const unsigned target = (unsigned)F2 * launch_no;
    unsigned long long tA = 0, tD = 0, tP = 0, tI = 0, tB = 0, nP = 0;
    auto now = [] { return __builtin_amdgcn_s_memrealtime(); };
    auto strip_at = [&](int k) { return (nt_in & 64) ? (int)x * NK + k : (int)x + 8 * k; };
    auto stage_a = [&](int k) {
        const int s = strip_at(k);
        unsigned mine = 0;
        const unsigned long long t0 = now();
        for (unsigned b = r; b < (unsigned)F2; b += T) { tile_ax<WSX>(in, S, s, (int)b, nt_in & 1); ++mine; }
        const unsigned long long t1 = now();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stores have been acknowledged by the L2
        __syncthreads();
        if (threadIdx.x == 0 && mine) atomicAdd(&ctl->done_a[s], mine);
        tA += t1 - t0; tD += now() - t1;
    };
    auto stage_b = [&](int k) -> bool {
        const int s = strip_at(k);
        const unsigned long long t0 = now();
        if (nt_in & 8) {                                        // poll with ONE wave, the others wait at the barrier
            if (threadIdx.x < 64) {
                for (unsigned polls = 0; uload(&ctl->done_a[s]) < target; ++polls) {
                    ++nP;
                    if (polls > POLL_BUDGET) { if (threadIdx.x == 0) atomicExch(&ctl->error, 1u); break; }
                    __builtin_amdgcn_s_sleep(32);
                }
            }
            __syncthreads();
        } else {
            for (unsigned polls = 0; uload(&ctl->done_a[s]) < target; ++polls) {
                ++nP;
                if (polls > POLL_BUDGET) { if (threadIdx.x == 0) atomicExch(&ctl->error, 1u); return false; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        const unsigned long long t1 = now();
        if (!(nt_in & 16)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's L1 (the L2 is the producers' L2)
        const unsigned long long t2 = now();
        for (unsigned k1 = r; k1 < (unsigned)K1; k1 += T) tile_bx<WSX>(S, out, s, (int)k1, (nt_in >> 5) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tP += t1 - t0; tI += t2 - t1; tB += now() - t2;
        return true;
    };
    for (int k = 0; k < DEPTH && k < NK; ++k) stage_a(k);
    for (int k = 0; k < NK; ++k) {
        if (k + DEPTH < NK) stage_a(k + DEPTH);
        if (!stage_b(k)) return;
    }
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned long long *p = ctl->prof[blockIdx.x];
        p[0] = tA; p[1] = tD; p[2] = tP; p[3] = tI; p[4] = tB; p[5] = nP;
    }
}

// where do workgroups land?  hist[blockIdx.x % 8][xcc]
__global__ void k_where(unsigned *hist) { if (threadIdx.x == 0) atomicAdd(&hist[(blockIdx.x & 7) * 8 + xcc_id()], 1u); }

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const char *only = argc > 2 ? argv[2] : "all";
    const int npairs = 3;
    std::vector<float *> in(npairs); std::vector<float2 *> S(npairs), out(npairs);
    for (int i = 0; i < npairs; ++i) {
        CK(hipMalloc(&in[i], (size_t)ROWS * COLS * 4)); CK(hipMalloc(&S[i], (size_t)K1 * F2 * COLS * 8)); CK(hipMalloc(&out[i], (size_t)(ROWS / 2 + 1) * COLS * 8));
        CK(hipMemset(in[i], 0x11 * (i + 1), (size_t)ROWS * COLS * 4));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)ROWS * COLS * 4 + (double)(ROWS / 2) * COLS * 8;
    auto time_it = [&](const char *name, auto go) {
        for (int k = 0; k < 3; ++k) go(k);
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < reps; ++k) go(k);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms * 1000.f / reps); CK(hipGetLastError());
        }
        std::sort(ts.begin(), ts.end());
        printf("%-56s %9.2f us   %7.1f GB/s algorithmic   %.3f of 8 TB/s\n", name, ts[2], bytes / ts[2] / 1e3, bytes / ts[2] / 8e6);
    };
    auto two = [&](auto ws_tag, const char *nm) {
        constexpr int WSX = decltype(ws_tag)::value;
        time_it(nm, [&](int k) {
            hipLaunchKernelGGL((k_ax<WSX>), dim3(Geo<WSX>::NSTRIPX * F2), dim3(512), 0, 0, in[k % npairs], S[k % npairs], 0);
            hipLaunchKernelGGL((k_bx<WSX>), dim3(Geo<WSX>::NSTRIPX * K1), dim3(512), 0, 0, S[k % npairs], out[k % npairs]);
        });
    };
    auto team = [&](auto ws_tag, auto depth_tag, int w, int nt) {
        constexpr int WSX = decltype(ws_tag)::value, DEPTH = decltype(depth_tag)::value;
        TeamCtl *ctl; CK(hipMalloc(&ctl, sizeof(TeamCtl))); CK(hipMemset(ctl, 0, sizeof(TeamCtl)));
        unsigned launch_no = 0;
        char nm[96]; snprintf(nm, sizeof nm, "team: strips of %d columns, depth %d, %d wg/CU, nt_in=%d", WSX, DEPTH, w, nt);
        // ONE buffer set per control block (the counters are per strip): rotate over npairs control blocks instead
        time_it(nm, [&](int k) { (void)k; hipLaunchKernelGGL((k_team<WSX, DEPTH>), dim3(256 * w), dim3(512), 0, 0, in[launch_no % npairs], S[launch_no % npairs], out[launch_no % npairs], ctl, nt, launch_no + 1); ++launch_no; });
        TeamCtl h; CK(hipMemcpy(&h, ctl, sizeof h, hipMemcpyDeviceToHost));
        if (h.error) printf("  !! team kernel error %u (1: a bounded wait ran out, 2: blockIdx %% 8 is not the XCD)\n", h.error);
        {   // where the time of the LAST launch went, averaged over workgroups (10 ns ticks -> us)
            double a[6] = {0, 0, 0, 0, 0, 0};
            for (int b = 0; b < 256 * w; ++b) for (int i = 0; i < 6; ++i) a[i] += (double)h.prof[b][i] / (256 * w);
            printf("      per workgroup: A tiles %.1f us, drain+barrier+atomic %.1f, wait %.1f (%.1f polls), acquire %.1f, B tiles %.1f\n", a[0] / 100, a[1] / 100, a[2] / 100, a[5], a[3] / 100, a[4] / 100);
        }
        CK(hipFree(ctl));
    };
    using W64 = std::integral_constant<int, 64>; using W32 = std::integral_constant<int, 32>;
    using D1 = std::integral_constant<int, 1>; using D2 = std::integral_constant<int, 2>; using D0 = std::integral_constant<int, 0>;
    if (!strcmp(only, "all") || !strcmp(only, "two")) { two(W64{}, "two launches, 64-column tiles"); two(W32{}, "two launches, 32-column tiles"); }
    if (!strcmp(only, "all") || !strcmp(only, "team"))
        for (int w = 1; w <= 4; ++w)
            for (int nt : {8 + 16 + 32, 8 + 16 + 32 + 64}) {     // 8: one wave polls; 16: no acquire fence; 32: nt (L1-bypassing) loads of the intermediate; 64: teams' strips 16 apart
                team(W64{}, D1{}, w, nt);
                if (w == 2) team(W32{}, D1{}, w, nt);
            }
    {   // is "blockIdx.x % 8 = XCD" true on this box?
        unsigned *hist; CK(hipMalloc(&hist, 256)); CK(hipMemset(hist, 0, 256));
        hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, 0, hist);
        unsigned h[64]; CK(hipMemcpy(h, hist, 256, hipMemcpyDeviceToHost));
        int off = 0; for (int b = 0; b < 8; ++b) for (int x = 0; x < 8; ++x) if (b != x) off += h[b * 8 + x];
        printf("workgroups whose XCC_ID differs from blockIdx %% 8: %d of 4096\n", off);
    }
    return 0;
}
