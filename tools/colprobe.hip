// tools/colprobe.hip -- developer tool (round 3): what does HBM give the COLUMN-TILE access shape of BASELINE configs[2] part A
// (ndfft_r2c along axis 0 of 8192 x 8192 f32 -> 4097 x 8192 c32), and does it depend on the row pitch being a power of two?
//
//  part "seg":  pure segment streams.  An array of ROWS rows at a pitch of P bytes is read (or written) in segments of S bytes:
//               workgroup (x, y) moves rows [x * RPW, (x + 1) * RPW) of column tile y, dispatch order x fastest -- i.e. the whole
//               chip walks DOWN one column tile, then the next.  S = 32..1024, P = 32 KiB, 32 KiB + 128 B, + 512 B, + 4 KiB.
//  part "tile": the one-pass shape.  ONE 1024-thread workgroup owns a column tile of 8192 rows x Wi bytes of input and
//               4097 rows x Wo bytes of output (Wo = 2 Wi): all loads first (values folded into an accumulator), then all stores --
//               the traffic of a register-resident single-pass column FFT without the FFT.  Variants: pitch, tile width,
//               tile -> workgroup map (identity / tiles of one 128-byte line on one XCD), per-workgroup row rotation.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/colprobe.hip -o tools/colprobe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- part "seg" ------------------------------------------------------------------------------------------------------------
// 256 threads; thread t moves 16 bytes: segment piece (t % (S/16)), row (t / (S/16)) of the workgroup's RPW = 256 * 16 / S * U rows
template <int S, int WRITE, int U> __global__ __launch_bounds__(256) void k_seg(char *base, size_t pitch, v4f *sink) {
    constexpr int TPS = S / 16, RPI = 256 / TPS;        // threads per segment, rows per instruction
    const int piece = threadIdx.x % TPS, r0 = threadIdx.x / TPS;
    char *p = base + (size_t)blockIdx.y * S + (size_t)piece * 16 + ((size_t)blockIdx.x * RPI * U + r0) * pitch;
    v4f acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v4f *q = (v4f *)(p + (size_t)u * RPI * pitch);
        if constexpr (WRITE) { v4f v = {(float)u, 1.f, 2.f, (float)threadIdx.x}; __builtin_nontemporal_store(v, q); }
        else acc += __builtin_nontemporal_load(q);
    }
    if constexpr (!WRITE) if (acc.x == 123.456f) sink[0] = acc;
}

// ---- part "tile" -----------------------------------------------------------------------------------------------------------
// WI = input segment bytes per row (32 / 64 / 128), output segment 2 WI bytes per row, rows_in = 8192, rows_out = 4097 (the last row: one wave)
// MAP 1: the 128 / WI tiles of one 128-byte input line run back to back on ONE XCD (blocks b, b + 8, ...); ROT: workgroup w starts at row 61 w
template <int WI, int MAP, int ROT, int NTS> __global__ __launch_bounds__(1024) void k_tile(const char *in, char *out, size_t pin, size_t pout, unsigned ntiles) {
    constexpr int TI = WI / 16, RI = 1024 / TI;          // threads per input segment, input rows per instruction
    constexpr int TO = 2 * WI / 16, RO = 1024 / TO;
    constexpr int ROWS = 8192, UI = ROWS / RI, UO = 4096 / RO;
    unsigned tile = blockIdx.x;
    if constexpr (MAP == 1) {
        constexpr unsigned Sx = 128 / WI;
        if constexpr (Sx > 1) {
            const unsigned grp = 8 * Sx, g = blockIdx.x / grp, r = blockIdx.x % grp;
            if ((g + 1) * grp <= ntiles) tile = g * grp + (r & 7) * Sx + (r >> 3);
        }
    }
    const unsigned rot = ROT ? (tile * 61u) % UI : 0u;
    v4f acc = {0, 0, 0, 0};
    {
        const int piece = threadIdx.x % TI, r0 = threadIdx.x / TI;
        const char *p = in + (size_t)tile * WI + piece * 16;
#pragma unroll 16
        for (int u = 0; u < UI; ++u) {
            const unsigned uu = (u + rot) % UI;
            acc += *(const v4f *)(p + ((size_t)uu * RI + r0) * pin);
        }
    }
    {
        const int piece = threadIdx.x % TO, r0 = threadIdx.x / TO;
        char *p = out + (size_t)tile * 2 * WI + piece * 16;
#pragma unroll 16
        for (int u = 0; u < UO; ++u) {
            const unsigned uu = (u + rot) % UO;
            v4f v = acc; v.x += (float)u;
            v4f *q = (v4f *)(p + ((size_t)uu * RO + r0) * pout);
            if constexpr (NTS) __builtin_nontemporal_store(v, q); else *q = v;
        }
        if (threadIdx.x < TO) { v4f *q = (v4f *)(p + (size_t)4096 * pout); *q = acc; }   // row 4096 (Nyquist)
    }
}

struct Var { std::string name; double bytes; std::function<void()> go; };
int main(int argc, char **argv) {
    const char *part = argc > 1 ? argv[1] : "all";
    const size_t maxpitch = 32768 + 8192, maxpitch_out = 65536 + 16384;
    const size_t in_bytes = 8192 * maxpitch, out_bytes = 4097 * maxpitch_out;
    const int npairs = 4;
    std::vector<char *> a(npairs), b(npairs);
    for (int i = 0; i < npairs; ++i) { CK(hipMalloc(&a[i], in_bytes)); CK(hipMalloc(&b[i], out_bytes)); CK(hipMemset(a[i], 1 + i, in_bytes)); CK(hipMemset(b[i], 0, out_bytes)); }
    v4f *sink; CK(hipMalloc(&sink, 64));
    std::vector<Var> vs;
    static size_t cnt = 0;
    const size_t extra[] = {0, 128, 512, 4096};
    if (!strcmp(part, "seg") || !strcmp(part, "all")) {
#define SEG(S, WRITE, U) for (size_t ex : extra) { const size_t pitch = 32768 + ex; \
        vs.push_back({std::string(WRITE ? "seg write S=" : "seg read  S=") + #S + " pitch=32768+" + std::to_string(ex), 8192.0 * 32768, [=]() { \
            char *base = WRITE ? b[cnt++ % npairs] : a[cnt++ % npairs]; \
            hipLaunchKernelGGL((k_seg<S, WRITE, U>), dim3(8192 / (256 * 16 / S * U), 32768 / S), dim3(256), 0, 0, base, pitch, sink); }}); }
        SEG(32, 0, 4) SEG(64, 0, 4) SEG(128, 0, 4) SEG(256, 0, 4) SEG(512, 0, 4) SEG(1024, 0, 4)
        SEG(64, 1, 4) SEG(128, 1, 4) SEG(256, 1, 4) SEG(1024, 1, 4)
    }
    if (!strcmp(part, "tile") || !strcmp(part, "all")) {
#define TILE(WI, MAP, ROT, NTS) for (size_t ex : extra) { const size_t pin = 32768 + ex, pout = 65536 + 2 * ex; \
        vs.push_back({"tile WI=" #WI " map" #MAP " rot" #ROT " nts" #NTS " pitch=32768+" + std::to_string(ex), 8192.0 * 32768 + 4097.0 * 65536, [=]() { \
            const size_t k = cnt++ % npairs; \
            hipLaunchKernelGGL((k_tile<WI, MAP, ROT, NTS>), dim3(32768 / WI), dim3(1024), 0, 0, a[k], b[k], pin, pout, (unsigned)(32768 / WI)); }}); }
        TILE(32, 0, 0, 0) TILE(32, 1, 0, 0) TILE(32, 1, 1, 0) TILE(32, 1, 0, 1)
        TILE(64, 0, 0, 0) TILE(64, 1, 0, 0) TILE(64, 1, 1, 0) TILE(64, 0, 1, 0)
        TILE(128, 0, 0, 0) TILE(128, 0, 1, 0) TILE(128, 0, 0, 1)
    }
    const int only = argc > 2 ? atoi(argv[2]) : -1;    // run ONE variant many times (for rocprofv3 --pmc)
    if (only >= 0) {
        if (only >= (int)vs.size()) { fprintf(stderr, "variant %d of %zu\n", only, vs.size()); return 1; }
        for (int k = 0; k < 20; ++k) vs[only].go();
        CK(hipDeviceSynchronize());
        printf("ran variant %d: %s\n", only, vs[only].name.c_str());
        return 0;
    }
    std::vector<std::vector<float>> t(vs.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 5, inner = 4;
    for (int r = 0; r < rounds; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            vs[i].go();
            CK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; ++k) vs[i].go();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[i].push_back(ms * 1000.f / inner); CK(hipGetLastError());
        }
    printf("%-4s %-56s %10s %10s %8s\n", "#", "variant (4 rotating buffers, cold)", "median_us", "GB/s", "of 8T");
    for (size_t i = 0; i < vs.size(); ++i) {
        std::sort(t[i].begin(), t[i].end());
        const double med = t[i][t[i].size() / 2];
        printf("%-4zu %-56s %10.2f %10.1f %8.3f\n", i, vs[i].name.c_str(), med, vs[i].bytes / med / 1e3, vs[i].bytes / med / 1e3 / 8000.0);
    }
    return 0;
}
