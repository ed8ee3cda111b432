// tools/ldsdma_probe.hip -- developer tool (round 5): does gfx950's LDS-DMA (global_load_lds_dwordx4: global -> LDS with no VGPR
// destination) move the tile shapes of the wait-bound kernels faster than today's global_load -> VGPR -> ds_write staging?
//
// Every variant is a TILE COPY with the product kernel's access shape and nothing else: a workgroup lands one input tile (R rows of W
// bytes, rows `pitch_in` apart) in LDS as a dense image, waits, idles for `cyc` clocks (a stand-in for the butterfly passes), reads the
// image back with ds_read_b128 and stores it as the output tile (Ro rows of Wo bytes, `pitch_out` apart) with 16-byte non-temporal
// stores.  Only the LOAD mechanism differs between variants, so the difference between two rows of the table is the load mechanism:
//   reg4 / reg8 / reg16   global_load of 4 / 8 / 16 bytes per thread, 8 in flight per thread, then ds_write (today's stage_loop)
//   dma / dma-nt          one global_load_lds_dwordx4 per 1 KiB piece (per-lane SOURCE address, so a piece may span rows), all pieces
//                         of the tile in flight at once, s_waitcnt vmcnt(0) + barrier
//   pipe1 / pipe1-nt      a PERSISTENT workgroup with ONE LDS image: the tile is read into registers and the next tile's DMA is issued into the same
//                         image at once (in flight during this tile's compute stand-in and stores); LDS = image + half an image of exchange buffer
//   pipe / pipe-nt        a PERSISTENT workgroup with two LDS images: the next tile's DMA is issued before this tile's compute / store
//                         phase, retired by a counted s_waitcnt vmcnt(S) (S = this wave's stores of the previous tile, which are younger)
//                         and a raw s_barrier
// Shapes (see DESIGN.md section 3.4c / 3.5b / 3.1 for the kernels they stand for):
//   row64K   k_pow2<double,4096>: one dense 64 KiB lane per workgroup, 512 threads (cfg2)
//   csA      stage A of the column four-step on cfg3-A: 128 rows x 256 B, rows 2 MiB apart -> 64 rows x 512 B, 4 MiB apart, 512 threads
//   csB      stage B: 64 rows x 256 B, 64 KiB apart -> 64 rows x 256 B, 8 MiB apart, 256 threads
//   rfs2     second pass of the real four-step, 64 x 262144 f64: 512 rows x 128 B, 4224 B apart -> 512 rows x 128 B, 8 KiB apart, 512 threads
// Arrays are walked in `rot` replicas so that every launch reads from HBM (cold) or re-reads one replica (warm: rot = 1).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ldsdma_probe.hip -o tools/ldsdma_probe
//   run  : tools/ldsdma_probe [rounds = 5] [shape filter substring] [compute stand-in cycles]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct Shape {
    int logW, logWo;                 // tile row widths in bytes (powers of two)
    int tile_bytes;                  // R * W = Ro * Wo
    long long pitch_in, pitch_out;   // bytes between rows
    int ncol, nmid;                  // tile index = (o * nmid + m) * ncol + c
    long long col_in, mid_in, outer_in, col_out, mid_out, outer_out;   // byte offsets per index
    int ntiles;
    int cyc;
};

__device__ __forceinline__ void tile_base(const Shape &s, unsigned tile, long long &bi, long long &bo) {
    const unsigned c = tile % (unsigned)s.ncol, r = tile / (unsigned)s.ncol, m = r % (unsigned)s.nmid, o = r / (unsigned)s.nmid;
    bi = (long long)o * s.outer_in + (long long)m * s.mid_in + (long long)c * s.col_in;
    bo = (long long)o * s.outer_out + (long long)m * s.mid_out + (long long)c * s.col_out;
}
__device__ __forceinline__ void fake_compute(int cycles) {
    for (int c = 0; c < cycles; c += 512) __builtin_amdgcn_s_sleep(8);
}
template <bool NT> __device__ __forceinline__ void glds16(const char *gsrc, unsigned lds_dst) {
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// LDS byte address of a __shared__ object (what M0 takes)
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)p; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// the store phase every variant shares: LDS image (linear) -> Ro x Wo output rows, 16 bytes per thread, non-temporal
template <int THREADS> __device__ __forceinline__ void store_tile(const Shape &s, const char *img, char *out) {
    const int nv = s.tile_bytes >> 4;
    for (int f = threadIdx.x; f < nv; f += THREADS) {
        const int y = f << 4, row = y >> s.logWo, col = y & ((1 << s.logWo) - 1);
        const v4f v = *(const v4f *)(img + y);
        __builtin_nontemporal_store(v, (v4f *)(out + (long long)row * s.pitch_out + col));
    }
}

// register staging, LW bytes per load, 8 loads in flight per thread (pow2_real.h: stage_loop)
template <int THREADS, int LW, bool NT> __global__ __launch_bounds__(THREADS) void k_reg(const char *in, char *out, const Shape s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long bi, bo; tile_base(s, blockIdx.x, bi, bo);
    const char *src = in + bi;
    typedef typename std::conditional<LW == 16, v4f, typename std::conditional<LW == 8, v2f, float>::type>::type V;
    const int ne = s.tile_bytes / LW;
    constexpr int U = 8;
    int f = threadIdx.x;
    for (; f + (U - 1) * THREADS < ne; f += U * THREADS) {
        V tmp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int x = (f + u * THREADS) * LW, row = x >> s.logW, col = x & ((1 << s.logW) - 1);
            const V *p = (const V *)(src + (long long)row * s.pitch_in + col);
            tmp[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) *(V *)(smem + (f + u * THREADS) * LW) = tmp[u];
    }
    for (; f < ne; f += THREADS) {
        const int x = f * LW, row = x >> s.logW, col = x & ((1 << s.logW) - 1);
        *(V *)(smem + x) = *(const V *)(src + (long long)row * s.pitch_in + col);
    }
    __syncthreads();
    fake_compute(s.cyc);
    store_tile<THREADS>(s, smem, out + bo);
}

// LDS-DMA, one tile per workgroup: every 1 KiB piece of the image is one wave instruction; all of them in flight, then vmcnt(0) + barrier
template <int THREADS, bool NT> __global__ __launch_bounds__(THREADS) void k_dma(const char *in, char *out, const Shape s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long bi, bo; tile_base(s, blockIdx.x, bi, bo);
    const char *src = in + bi;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int npiece = s.tile_bytes >> 10;
    const unsigned lds0 = lds_addr(smem);
    for (int p = wave; p < npiece; p += THREADS / 64) {
        const int x = (p << 10) + (lane << 4), row = x >> s.logW, col = x & ((1 << s.logW) - 1);
        glds16<NT>(src + (long long)row * s.pitch_in + col, lds0 + (unsigned)(p << 10));
    }
    wait_vm<0>();
    __syncthreads();
    fake_compute(s.cyc);
    store_tile<THREADS>(s, smem, out + bo);
}

// persistent, two LDS images, counted vmcnt.  SPT = store instructions per wave per tile (tile_bytes / 16 / THREADS), a template
// parameter because s_waitcnt takes an immediate.
template <int THREADS, bool NT, int SPT> __global__ __launch_bounds__(THREADS) void k_pipe(const char *in, char *out, const Shape s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int npiece = s.tile_bytes >> 10;
    const unsigned lds0 = lds_addr(smem);
    const unsigned G = gridDim.x;
    unsigned vb = blockIdx.x;
    if (vb >= (unsigned)s.ntiles) return;
    auto issue = [&](unsigned tile, int buf) {
        long long bi, bo; tile_base(s, tile, bi, bo);
        const char *src = in + bi;
        for (int p = wave; p < npiece; p += THREADS / 64) {
            const int x = (p << 10) + (lane << 4), row = x >> s.logW, col = x & ((1 << s.logW) - 1);
            glds16<NT>(src + (long long)row * s.pitch_in + col, lds0 + (unsigned)(buf * s.tile_bytes + (p << 10)));
        }
    };
    issue(vb, 0);
    wait_vm<0>();
    int buf = 0;
    for (; vb < (unsigned)s.ntiles; vb += G, buf ^= 1) {
        // this wave's DMA of tile vb is older than its stores of the previous tile: all but the SPT youngest operations must be done
        wait_vm<SPT>();
        __builtin_amdgcn_s_barrier();        // every wave's pieces have landed, and every wave has read the other image
        if (vb + G < (unsigned)s.ntiles) issue(vb + G, buf ^ 1);
        long long bi, bo; tile_base(s, vb, bi, bo);
        fake_compute(s.cyc);
        // (the image reads below are LDS operations: lgkmcnt, counted by the compiler)
        store_tile<THREADS>(s, smem + buf * s.tile_bytes, out + bo);
    }
}

// persistent, ONE LDS image: the tile is read into registers (8 x 16 B per thread at most: tile_bytes <= 128 B x THREADS), and the NEXT tile's DMA
// is issued into the same image right away -- it is in flight during this tile's compute stand-in and stores.  The compute phase of a real
// kernel then needs its own exchange buffer (`xlds` bytes, only reserved here so that the occupancy matches).
template <int THREADS, bool NT, int SPT> __global__ __launch_bounds__(THREADS) void k_pipe1(const char *in, char *out, const Shape s) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int npiece = s.tile_bytes >> 10;
    const unsigned lds0 = lds_addr(smem);
    const unsigned G = gridDim.x;
    unsigned vb = blockIdx.x;
    if (vb >= (unsigned)s.ntiles) return;
    auto issue = [&](unsigned tile) {
        long long bi, bo; tile_base(s, tile, bi, bo);
        const char *src = in + bi;
        for (int p = wave; p < npiece; p += THREADS / 64) {
            const int x = (p << 10) + (lane << 4), row = x >> s.logW, col = x & ((1 << s.logW) - 1);
            glds16<NT>(src + (long long)row * s.pitch_in + col, lds0 + (unsigned)(p << 10));
        }
    };
    issue(vb);
    wait_vm<0>();
    for (; vb < (unsigned)s.ntiles; vb += G) {
        wait_vm<SPT>();                      // DMA(vb) is older than the previous tile's SPT stores
        __builtin_amdgcn_s_barrier();
        v4f v[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) v[k] = *(const v4f *)(smem + ((threadIdx.x + k * THREADS) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();        // every wave holds its part of the image in registers: the image is free
        if (vb + G < (unsigned)s.ntiles) issue(vb + G);
        long long bi, bo; tile_base(s, vb, bi, bo);
        fake_compute(s.cyc);
        char *o = out + bo;
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int y = (threadIdx.x + k * THREADS) << 4, row = y >> s.logWo, col = y & ((1 << s.logWo) - 1);
            __builtin_nontemporal_store(v[k], (v4f *)(o + (long long)row * s.pitch_out + col));
        }
    }
}

struct Var { std::string name; int wgcu; size_t lds; int kind; void (*launch)(const Var &, const char *, char *, const Shape &, unsigned cus); int threads; };

template <int THREADS, int LW, bool NT> static void go_reg(const Var &v, const char *i, char *o, const Shape &s, unsigned) {
    static bool once = (hipFuncSetAttribute((const void *)k_reg<THREADS, LW, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); (void)once;
    hipLaunchKernelGGL((k_reg<THREADS, LW, NT>), dim3(s.ntiles), dim3(THREADS), v.lds, 0, i, o, s);
}
template <int THREADS, bool NT> static void go_dma(const Var &v, const char *i, char *o, const Shape &s, unsigned) {
    static bool once = (hipFuncSetAttribute((const void *)k_dma<THREADS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); (void)once;
    hipLaunchKernelGGL((k_dma<THREADS, NT>), dim3(s.ntiles), dim3(THREADS), v.lds, 0, i, o, s);
}
template <int THREADS, bool NT, int SPT> static void go_pipe(const Var &v, const char *i, char *o, const Shape &s, unsigned cus) {
    static bool once = (hipFuncSetAttribute((const void *)k_pipe<THREADS, NT, SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); (void)once;
    hipLaunchKernelGGL((k_pipe<THREADS, NT, SPT>), dim3(std::min<unsigned>(s.ntiles, v.wgcu * cus)), dim3(THREADS), v.lds, 0, i, o, s);
}

template <int THREADS, bool NT, int SPT> static void go_pipe1(const Var &v, const char *i, char *o, const Shape &s, unsigned cus) {
    static bool once = (hipFuncSetAttribute((const void *)k_pipe1<THREADS, NT, SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess); (void)once;
    hipLaunchKernelGGL((k_pipe1<THREADS, NT, SPT>), dim3(std::min<unsigned>(s.ntiles, v.wgcu * cus)), dim3(THREADS), v.lds, 0, i, o, s);
}

struct Case { std::string name; Shape s; int threads; size_t bytes_in, bytes_out; int rot; };

template <int THREADS, int SPT> static void add_variants(std::vector<Var> &vs, const Shape &s, size_t extra_lds) {
    // extra_lds: what the product kernel needs besides the image (so that the occupancy matches); image + extra
    const size_t one = (size_t)s.tile_bytes + extra_lds;
    vs.push_back({"reg4", 0, one, 0, go_reg<THREADS, 4, false>, THREADS});
    vs.push_back({"reg4-nt", 0, one, 0, go_reg<THREADS, 4, true>, THREADS});
    vs.push_back({"reg8", 0, one, 0, go_reg<THREADS, 8, false>, THREADS});
    vs.push_back({"reg16", 0, one, 0, go_reg<THREADS, 16, false>, THREADS});
    vs.push_back({"reg16-nt", 0, one, 0, go_reg<THREADS, 16, true>, THREADS});
    vs.push_back({"dma", 0, one, 1, go_dma<THREADS, false>, THREADS});
    vs.push_back({"dma-nt", 0, one, 1, go_dma<THREADS, true>, THREADS});
    for (int w : {1, 2, 3, 4}) {                                   // one image + a separate exchange buffer of half the image (HALF exchange)
        const size_t l = (size_t)s.tile_bytes + (size_t)s.tile_bytes / 2 + 4096;
        if (l * w > 160 * 1024) continue;
        vs.push_back({"pipe1 " + std::to_string(w) + "wg/cu", w, l, 2, go_pipe1<THREADS, false, SPT>, THREADS});
        vs.push_back({"pipe1-nt " + std::to_string(w) + "wg/cu", w, l, 2, go_pipe1<THREADS, true, SPT>, THREADS});
    }
    for (int w : {1, 2, 3, 4}) {
        const size_t l = 2 * (size_t)s.tile_bytes + extra_lds;
        if (l * w > 160 * 1024) continue;
        vs.push_back({"pipe " + std::to_string(w) + "wg/cu", w, l, 2, go_pipe<THREADS, false, SPT>, THREADS});
        vs.push_back({"pipe-nt " + std::to_string(w) + "wg/cu", w, l, 2, go_pipe<THREADS, true, SPT>, THREADS});
    }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const char *filter = argc > 2 ? argv[2] : "";
    const int cyc_override = argc > 3 ? atoi(argv[3]) : -1;       // compute stand-in in clocks (default: per shape)
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const unsigned cus = (unsigned)pr.multiProcessorCount;
    const size_t MiB = 1ull << 20;
    std::vector<Case> cases;
    {   // row64K: 4096 lanes of 64 KiB
        Shape s{}; s.logW = 16; s.logWo = 16; s.tile_bytes = 65536; s.pitch_in = s.pitch_out = 65536; s.ncol = 1; s.nmid = 1;
        s.outer_in = s.outer_out = 65536; s.ntiles = 4096; s.cyc = 6000;
        cases.push_back({"row64K (k_pow2 f64 n=4096, cfg2)", s, 512, 256 * MiB, 256 * MiB, 6});
    }
    {   // csA: input 8192 x 8192 f32; tile (b, c): rows b + 64 a (a < 128), columns 64 c .. 64 c + 63 -> 128 x 256 B, pitch 2 MiB
        //      output [k1 < 64][b < 64][8192] c64: rows k1 of (b, c): 64 x 512 B, pitch 4 MiB
        Shape s{}; s.logW = 8; s.logWo = 9; s.tile_bytes = 32768; s.pitch_in = 64ll * 8192 * 4; s.pitch_out = 64ll * 8192 * 8;
        s.ncol = 128; s.nmid = 64; s.col_in = 256; s.mid_in = 8192 * 4; s.col_out = 512; s.mid_out = 8192 * 8; s.ntiles = 8192; s.cyc = 3000;
        cases.push_back({"csA (cfg3-A stage A: 128 x 256 B rows 2 MiB apart)", s, 512, 256 * MiB, 256 * MiB, 4});
    }
    {   // csB: input [k1 < 64][b < 64][8192] c64: tile (k1, c): rows b, 32 lanes -> 64 x 256 B, pitch 64 KiB; output rows k1 + 128 k2: 64 x 256 B, pitch 8 MiB
        Shape s{}; s.logW = 8; s.logWo = 8; s.tile_bytes = 16384; s.pitch_in = 8192 * 8; s.pitch_out = 128ll * 8192 * 8;
        s.ncol = 256; s.nmid = 64; s.col_in = 256; s.mid_in = 64ll * 8192 * 8; s.col_out = 256; s.mid_out = 8192 * 8; s.ntiles = 16384; s.cyc = 1500;
        cases.push_back({"csB (cfg3-A stage B: 64 x 256 B rows 64 KiB apart)", s, 256, 256 * MiB, 512 * MiB, 4});
    }
    {   // rfs2: 64 lanes o; s[o][n2 < 512][k1 < 264 (257 used)] c128 -> tile (o, c): 512 rows x 128 B (8 k1), pitch 4224 B; 33 tiles per o (32 counted)
        //       output X[o][k1 + 512 k2]: rows k2, 128 B, pitch 8 KiB
        Shape s{}; s.logW = 7; s.logWo = 7; s.tile_bytes = 65536; s.pitch_in = 264 * 16; s.pitch_out = 512 * 16;
        s.ncol = 32; s.nmid = 1; s.col_in = 128; s.col_out = 128; s.outer_in = 512ll * 264 * 16; s.outer_out = 512ll * 512 * 16; s.ntiles = 64 * 32; s.cyc = 6000;
        cases.push_back({"rfs2 (real four-step pass 2: 512 x 128 B rows, out 8 KiB apart)", s, 512, 64 * 512ull * 264 * 16, 64 * 512ull * 512 * 16, 6});
    }
    // pitch sweep (round 5): the second pass of the complex four-step on 256 x 65536 c128 -- tiles of 256 rows x 128 B (8 adjacent k1);
    // input = the intermediate s[n2][k1] (its pitch is OURS: 4096 B natural, or padded), output = the caller's array, rows N1 x 16 B = 4096 B apart (fixed).
    // "pout" variants with a padded OUTPUT pitch are hypothetical (what the store side would gain if the caller's pitch were not a power of two).
    for (int pin : {4096, 4096 + 128, 4096 + 512, 4096 + 1024})
        for (int pout : {4096, 4096 + 512}) {
            Shape s{}; s.logW = 7; s.logWo = 7; s.tile_bytes = 32768; s.pitch_in = pin; s.pitch_out = pout;
            s.ncol = 32; s.nmid = 1; s.col_in = 128; s.col_out = 128; s.outer_in = 256ll * pin; s.outer_out = 256ll * pout; s.ntiles = 256 * 32 * 2; s.cyc = 3000;
            cases.push_back({"fs2 pitch_in=" + std::to_string(pin) + " pitch_out=" + std::to_string(pout) + " (four-step pass 2, 256 x 128 B rows)", s, 512,
                             512ull * 256 * pin, 512ull * 256 * pout, 3});
        }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &c : cases) {
        if (filter[0] && c.name.find(filter) == std::string::npos) continue;
        if (cyc_override >= 0) c.s.cyc = cyc_override;
        std::vector<Var> vs;
        // extra LDS: the exchange buffer of the product kernel where it is not the image itself (rows: HALF exchange lives in the image)
        if (c.threads == 512 && c.s.tile_bytes == 65536) add_variants<512, 8>(vs, c.s, 0);
        else if (c.threads == 512) add_variants<512, 4>(vs, c.s, 0);
        else add_variants<256, 4>(vs, c.s, 0);
        for (int warm = 0; warm < 2; ++warm) {
            const int rot = warm ? 1 : c.rot;
            char *a, *b;
            CK(hipMalloc(&a, c.bytes_in * rot)); CK(hipMalloc(&b, c.bytes_out * rot));
            CK(hipMemset(a, 1, c.bytes_in * rot)); CK(hipMemset(b, 0, c.bytes_out * rot));
            printf("\n== %s, %s (%d replica%s), %d tiles of %d B, compute stand-in %d cycles ==\n", c.name.c_str(), warm ? "re-read" : "HBM-sourced", rot, rot > 1 ? "s" : "",
                   c.s.ntiles, c.s.tile_bytes, c.s.cyc);
            printf("%-18s %6s %9s %9s %9s %9s %12s\n", "variant", "wg/cu", "us(cyc)", "TB/s", "us(0)", "TB/s", "KiB/CU infl");
            for (auto &v : vs) {
                int occ = 0;
                double us[2];
                for (int z = 0; z < 2; ++z) {
                    Shape s = c.s; if (z) s.cyc = 0;
                    std::vector<float> t;
                    int cur = 0;
                    auto once = [&]() { v.launch(v, a + (size_t)cur * c.bytes_in, b + (size_t)cur * c.bytes_out, s, cus); cur = (cur + 1) % rot; };
                    for (int k = 0; k < 20; ++k) once();
                    CK(hipDeviceSynchronize()); CK(hipGetLastError());
                    for (int r = 0; r < rounds; ++r) {
                        const int inner = 24;
                        CK(hipEventRecord(e0, 0));
                        for (int k = 0; k < inner; ++k) once();
                        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f / inner);
                    }
                    std::sort(t.begin(), t.end()); us[z] = t[t.size() / 2];
                }
                // resident workgroups per CU by LDS and threads (2048 threads per CU), for the in-flight column
                occ = v.kind == 2 ? v.wgcu : (int)std::min<size_t>(160 * 1024 / std::max<size_t>(v.lds, 1), 2048 / v.threads);
                const double bytes = 2.0 * c.s.ntiles * (double)c.s.tile_bytes;
                const double infl = v.kind == 0 ? occ * v.threads * 8.0 * (v.name[3] == '1' ? 16 : (v.name[3] == '8' ? 8 : 4)) / 1024.0 : occ * c.s.tile_bytes / 1024.0;
                printf("%-18s %6d %9.2f %9.3f %9.2f %9.3f %12.0f\n", v.name.c_str(), occ, us[0], bytes / us[0] / 1e6, us[1], bytes / us[1] / 1e6, infl);
                fflush(stdout);
            }
            CK(hipFree(a)); CK(hipFree(b));
        }
    }
    return 0;
}
