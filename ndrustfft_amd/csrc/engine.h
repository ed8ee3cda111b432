// engine.h -- internal declarations of libndfft_mi355x (gfx950 only; not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ndfft_mi355x.h"
#include "device_common.h"
#include "switches.h"

namespace ndfft {

// ------------------------------------------------------------------------------------------
// lane geometry: where lane `l`, element `j` of an n-d view lives (offsets in ELEMENTS)
// ------------------------------------------------------------------------------------------
constexpr int kMaxBatchDims = 4;
struct LaneGeom {
    int64_t axis_stride;             // between consecutive elements of one lane
    int32_t nb;                      // batch dims in use (after merging), slowest first
    int32_t pad_;
    int64_t bshape[kMaxBatchDims];
    int64_t bstride[kMaxBatchDims];
};


constexpr int kMaxPasses = 16;
constexpr int kMaxLpb = 64;

enum IoMode : int { IO_ROW = 0, IO_COL = 1 };   // thread -> (lane, element) map for global IO

template <typename T> struct GenArgs {
    const void *in; void *out;
    LaneGeom gin, gout;
    int64_t nlanes;
    int32_t op;              // GenOp
    int32_t n;               // handler length (real length for R2C/C2R/DCT)
    int32_t n_in, n_out;     // lane lengths in elements of the in / out element type
    int32_t in_cplx, out_cplx;
    int32_t F;               // complex FFT length
    int32_t npass; int32_t radix[kMaxPasses];
    int32_t lpb;             // lanes per block
    int32_t pitch;           // LDS pitch per lane per buffer, in complex elements
    int32_t load_mode, store_mode;
    int32_t io_tpl_log, fft_tpl_log, lpb_log;   // log2 of the power-of-two thread-map factors
    int32_t inplace;         // one LDS buffer per lane: every thread owns <= 1 butterfly per pass (elementwise ops only)
    T scale;                 // normalisation scalar, applied where the reference applies it
    const cpx<T> *tw;        // tw[k] = e^{-2 pi i k/F}
    const cpx<T> *aux1, *aux2;
    // Bluestein (when F has a prime factor the radix passes do not cover)
    int32_t blue, M, npassM; int32_t radixM[kMaxPasses];
    const cpx<T> *twM, *chirp, *bhat;
};

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct HostTable { std::vector<long double> re, im; };
struct JitCfg { int n = 0, tpl = 0, e = 0, lpb = 1, vec = 1; bool partial = false; std::vector<int> radix; int row_lpb = 0; };   // row_lpb: planned lanes per row workgroup (0 = default rule)   // partial: some pass has an incomplete last round
// Rader / Good-Thomas recipe of rader_kernel.h: F = mc * p, p prime with p - 1 smooth; fft = register configuration of FFT_(p-1)
struct RaderCfg {
    int p = 0, mc = 1, mc1 = 1, mc2 = 1; JitCfg fft;   // mc = mc1 * mc2: cofactor as one butterfly (mc2 = 1) or a two-factor transform in registers
    // sym (DCT-I slot, odd cofactor > 1; round 5): the inner FFT input is built EVEN-SYMMETRIC, z[-i] = z[i] (rader_kernel.h), so the Good-Thomas rows n1 and
    // mc - n1 have mirrored spectra and only rows 0 .. (mc - 1) / 2 run Rader's convolution
    bool sym = false;
    int rows() const { return sym ? (mc + 1) / 2 : mc; }
    // ... and with cofactor 1 (F = p prime: nddct1 n = 128, 8192) the Rader sequence itself is periodic, a[q + (p-1)/2] = a[q]: a convolution of HALF the length
    bool half() const { return sym && mc == 1; }
    int conv_len() const { return half() ? (p - 1) / 2 : p - 1; }
};
}  // namespace ndfft
struct ndfft_plan;
namespace ndfft {   // built once in long double

struct FftConfig {                 // one complex-FFT-of-length-F recipe + op tables
    int F = 0;
    std::vector<int> radix;
    bool blue = false; int M = 0; std::vector<int> radixM;
    HostTable tw, twM, chirp, bhat, aux1, aux2;
    // tuned power-of-two path (register-resident Stockham), when eligible
    bool pow2 = false;             // C2C slot: pow2_kernel.h ; real-op slots: pow2_real.h
    HostTable twp;                 // per-pass transposed twiddles
    HostTable twp_rev;             // bluereg: the same for the radix list back to front (second FFT of the convolution)
    HostTable twp_col;             // C2C slot only: twiddles in the radix order of the column kernel (pow2_real.h)
    bool fs_jit = false; JitCfg fs_jitcfg; HostTable twp_fs;   // C2C, smooth non-power-of-two n <= 2048: recipe + per-pass twiddles of the hiprtc four-step passes (jit.hip: launch_jit_fourstep)
    HostTable twp_col_w;           // C2C n = 1024 only: the same for the E = 16 recipe 16.8.8 of the four-step passes (kernels_fourstep.hip: wide)
    HostTable twp_narrow;          // twiddles in the radix order of the narrow (XCD-aware) column kernel
    HostTable tinymat[4];          // MAIN slot, n = 2..16: the real-data transforms as dense real matrices (tinymat_kernel.h);
                                   // R2C plans: [0] R2C, [1] C2R; DCT plans: [0..3] DCT-I..IV; stored two reals per (re, im) entry
    HostTable wave_tw;             // C2C slot, n = 2..64 power of two: W_n^k, k < n, for the wavefront kernel (wave_kernel.h)
    // long lanes (one lane does not fit LDS): four-step F = F1 * F2 on top of the row kernels
    bool big = false; int F1 = 0, F2 = 0, logB = 0;
    bool bigblue = false;          // big && no usable split (huge prime factor): Bluestein over global memory, sub1 = C2C plan of length M
    ndfft_plan *sub1 = nullptr, *sub2 = nullptr;   // C2C sub-plans of length F1 / F2 (owned)
    HostTable twlo, twhi;          // W_F^m = twhi[m >> logB] * twlo[m & (2^logB - 1)]
    // long STRIDED lanes (pow2 n, C2C and R2C/C2R slots): column four-step n = cs_F1 * cs_F2 in two passes of
    // wide column tiles (exec.hip col_split); cs_sub1 has the kind of the owning plan, cs_sub2 is C2C
    bool cs = false; int cs_F1 = 0, cs_F2 = 0, cs_logB = 0;
    int cs_ops = 0;                // ops that take it (measured against the one-pass column tiles): 1 = C2C, 2 = R2C, 4 = C2R
    ndfft_plan *cs_sub1 = nullptr, *cs_sub2 = nullptr;
    HostTable cs_twlo, cs_twhi;    // W_n^m for m < n, split like twlo / twhi
    // long CONTIGUOUS real-data lanes (MAIN slot of R2C and DCT plans, big, n a power of two): REAL four-step n = rfs_N1 * rfs_N2
    // (exec.hip: real_fourstep) -- a real FFT of length N1 over the strided index, then complex FFTs of length N2 on the half
    // spectrum; rfs_sub1 = R2C plan of length N1, rfs_sub2 = C2C plan of length N2, twiddles W_n^m split like twlo / twhi
    bool rfs = false; int rfs_N1 = 0, rfs_N2 = 0, rfs_logB = 0;
    int rfs_ops = 0;               // ops for which it measured faster than the packed route: 1 = R2C, 2 = C2R, 4 = DCT-II, 8 = DCT-III
    ndfft_plan *rfs_sub1 = nullptr, *rfs_sub2 = nullptr;
    HostTable rfs_twlo, rfs_twhi;
    HostTable rfs_c1, rfs_c2;      // DCT plans: e^{-i pi k/(2n)} factored over k = k1 + N1 r: c1[k1] = e^{-i pi k1/(2n)} (k1 = 0..N1), c2[r] = e^{-i pi r/(2 N2)} (r < N2) -- two small
                                   // cache-resident tables instead of a stream of n/2 + 1 entries per lane beside the data (col_direct.h modes 6 / 8, pow2_real.h CS = 6)
    bool blue_reg_only = false;    // M exceeds the LDS kernel's reach: only the register kernel can run it
    bool bluereg = false;          // blue && M has a register-kernel configuration: blue_kernel.h, specialised with hiprtc
                                   // (jitcfg = configuration for M, twp = its per-pass twiddles)
    // blue && F = mc * p with p - 1 smooth: rader_kernel.h instead of Bluestein (specialised with hiprtc); rader_bhat = FFT_(p-1)(W_p^(g^-q)) / (p - 1),
    // rader_twp = per-pass twiddles of FFT_(p-1), rader_tab = g^i mod p (i < p - 1) followed by g^-i mod p
    bool rader = false; RaderCfg radercfg; HostTable rader_bhat, rader_twp, rader_twp2, rader_ctw; std::vector<int32_t> rader_tab;   // twp2: the passes in reverse order
    bool jit = false; JitCfg jitcfg;   // C2C slot: smooth non-power-of-two n -> specialised register kernel (jit.hip); twiddles in twp
    // C2C slot: the recipe of the COLUMN tiles where it differs from the rows' (f32 lanes below 256 points: the rows' re-planned recipe -- more
    // threads per lane, fewer elements each -- measured 29 % slower on column tiles, 81 x 100 x 2048 c64 56.9 vs 44.2 us, profiles/r07/r07e_*)
    bool jit_col_alt = false; JitCfg jitcfg_col; HostTable twp_jcol;
    bool unsupported = false;      // no single-kernel fit and no usable factorisation (large prime factor)
};

struct DevConfig {                 // device copies (typed by dtype) of one FftConfig
    void *tw = nullptr, *twM = nullptr, *chirp = nullptr, *bhat = nullptr, *aux1 = nullptr, *aux2 = nullptr, *twp = nullptr;
    void *twlo = nullptr, *twhi = nullptr, *twp_col = nullptr, *twp_col_w = nullptr, *twp_fs = nullptr, *twp_narrow = nullptr, *twp_jcol = nullptr;
    void *cs_twlo = nullptr, *cs_twhi = nullptr;
    void *rfs_twlo = nullptr, *rfs_twhi = nullptr, *rfs_c1 = nullptr, *rfs_c2 = nullptr;
    void *wave_tw = nullptr;
    void *tinymat[4] = {nullptr, nullptr, nullptr, nullptr};
    void *rader_bhat = nullptr, *rader_twp = nullptr, *rader_twp2 = nullptr, *rader_tab = nullptr, *twp_rev = nullptr, *rader_ctw = nullptr;   // ctw: W_mc^k of a two-factor cofactor
};

enum ConfigSlot { CFG_MAIN = 0, CFG_DCT1 = 1, CFG_DCT4 = 2, CFG_COUNT = 3 };

struct DevTables { DevConfig cfg[CFG_COUNT]; };

}  // namespace ndfft

struct ndfft_plan {
    int kind, dtype;
    size_t n;
    int refcount;
    ndfft::FftConfig cfg[ndfft::CFG_COUNT];
    bool has_cfg[ndfft::CFG_COUNT];
    std::mutex mu;
    std::map<int, ndfft::DevTables> dev;   // device id -> tables (lazily uploaded)
};

namespace ndfft {
template <typename T> struct RealArgs;

// error plumbing (thread-local message)
int fail(int code, const std::string &msg);
#define NDFFT_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ::ndfft::fail(NDFFT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// opt a kernel into more than 64 KiB of dynamic LDS once per DEVICE (the attribute is per device; a process may
// drive several through ndfft_set_device).  `mask` is a function-local static of the calling launcher.
#define NDFFT_ENSURE_LDS_ATTR(fn)                                                                                   \
    do {                                                                                                            \
        static std::atomic<unsigned long long> mask_{0};                                                            \
        int dev_ = 0;                                                                                               \
        (void)hipGetDevice(&dev_);                                                                                  \
        const unsigned long long bit_ = 1ull << (dev_ & 63);                                                        \
        if (!(mask_.load(std::memory_order_relaxed) & bit_)) {                                                      \
            (void)hipFuncSetAttribute((const void *)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
            mask_.fetch_or(bit_, std::memory_order_relaxed);                                                        \
        }                                                                                                           \
    } while (0)

void set_last_path(const char *p);
const char *last_path();
const std::string &last_err();
void clear_err();
int get_dev_tables(const ndfft_plan *plan, const DevTables **out);
// exec.hip: the argument checks of one nd* call (the reference's panics) without running it -- for shard.hip
int validate_call(const ndfft_plan *plan, int op, int ndim, const int64_t *shape_in, const int64_t *stride_in, const int64_t *shape_out,
                  const int64_t *stride_out, int axis, int norm, double scale, bool *nothing);

// kernels_generic.hip
template <typename T> int launch_generic(const GenArgs<T> &a, int threads, size_t lds_bytes, hipStream_t s);
size_t generic_lds_bytes(int lpb, int pitch, size_t csize, int nbuf = 2);
int generic_z_len(int len);
bool generic_needs_big(const int32_t *radix, int npass, const int32_t *radixM, int npassM);   // radices > 10: 512-thread class   // LDS elements a padded length-`len` complex buffer needs

// kernels_pow2.hip : register-resident Stockham for contiguous power-of-two C2C lanes
bool pow2_supported(int dtype, int n);
// layout of the per-pass transposed twiddle table for length n (host builder in plan.cpp)
void pow2_build_twiddles(int dtype, int n, HostTable &out);
int launch_pow2(int dtype, int n, const Pow2Args &a, hipStream_t s);
int xcd_chunk_for(size_t block_bytes, int64_t nblk);   // lane blocks per XCD chunk of the workgroup -> lane map (0: identity)
bool stream_loads_for(size_t in_bytes);                // input larger than the Infinity Cache: streaming (nt) loads

// kernels_wave.hip : LDS-free wavefront kernel (cross-lane shuffles) for dense C2C lanes of n = 2..64
bool wave_supported(int n);
int launch_wave(int dtype, int n, const WaveArgs &a, hipStream_t s);

// kernels_tiny.hip : one thread per lane, C2C n = 2..13, 16 (tiny_kernel.h)
bool tiny_supported(int n);
int launch_tiny(int dtype, int n, bool stage, const TinyArgs &a, hipStream_t s);

// kernels_tinymat_f32/f64.hip : one thread per lane, R2C / C2R / DCT-I..IV of n = 2..16 as a dense matrix (tinymat_kernel.h)
// shape: 0 = R2C, 1 = C2R, 2 = DCT
int launch_tinymat_f32(int n, int shape, bool stage, const TinyArgs &a, hipStream_t s);
int launch_tinymat_f64(int n, int shape, bool stage, const TinyArgs &a, hipStream_t s);

// kernels_pow2_real.hip : register-resident real-op kernels (R2C/C2R/DCT) for power-of-two inner FFT length F
bool pow2_real_supported(int F);
void pow2_real_build_twiddles(int F, HostTable &out);
template <typename T> int launch_pow2_real(int gen_op, const RealArgs<T> &a, bool col, hipStream_t s);
template <typename T> int pow2_real_col_lanes(int F, int kind);   // adjacent lanes per column tile (0: none); kind: 0 = C2C, 1 = R2C, 2 = C2R, 3 = DCT
template <typename T> int pow2_real_narrow_lanes(int F);   // lanes per XCD-aware narrow column tile (0: none)
void pow2_real_build_narrow_twiddles(int dtype, int F, HostTable &out);
template <typename T> int launch_pow2_real_narrow(int gen_op, const RealArgs<T> &a, hipStream_t s);
// column four-step, twiddled stage (kernels_colsplit.hip): cs = 1 C2C, 2 = R2C second stage, 3 = C2R first stage
bool pow2_real_config(int F, JitCfg &cfg);
template <typename T> int launch_jit_blue(int gen_op, const JitCfg &cfgM, bool col, const RealArgs<T> &a, hipStream_t s);
// rader_kernel.h (jit.hip): recipe for an inner FFT length F with one prime factor > 13 (false: none, Bluestein stays), lanes per column tile, launch
// Bluestein (jit.hip): convolution length for inner FFT length F -- the cheapest 13-smooth M in [2F - 1, m_pow2] by passes x M -- and the register recipe for it
int blue_pick_len(int dtype, int F, int m_pow2);
bool blue_plan_cfg(int dtype, int M, JitCfg &cfg);
bool rader_choose(int dtype, int F, RaderCfg &rc, bool dct1_slot = false);
template <typename T> int launch_jit_plain(int gen_op, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s);   // plain_kernel.h: odd-n real ops, smooth F
bool jit_choose_real(int dtype, int F, JitCfg &cfg);   // jit_choose for the real-op slots (rows of RealPow2Kernel): cost-model recipe
int rader_col_lanes(int dtype, const RaderCfg &rc);
template <typename T> int launch_jit_rader(int gen_op, const RaderCfg &rc, bool col, const RealArgs<T> &a, hipStream_t s);
int launch_pack_lanes(const void *strided, void *dense, const LaneGeom &g, int64_t lanes, int64_t len, int64_t pitch, int esz, int unpack, hipStream_t s);   // big.hip
int colsplit_inner_len();
int colsplit_tile_lanes();
template <typename T> int launch_colsplit(int cs, bool inverse, const RealArgs<T> &a, hipStream_t s);

// kernels_fourstep.hip : the two passes of the row four-step on the column kernels (no transpose launch)
bool jit_fourstep_choose(int dtype, int n, JitCfg &cfg);        // plan time: the recipe of those passes for a smooth non-power-of-two factor n (false: none)
bool jit_fourstep_ok(int dtype, const JitCfg &cfg);
bool jit_rfs1_ok(int dtype, const JitCfg &cfg);
bool jit_rfsi_ok(int dtype, const JitCfg &cfg);                 // ... and the inverse direction's first pass (col_direct.h modes 7 / 8; whole butterfly rounds only)                 // the real four-step's first pass (real FFT of length 2 cfg.n over the strided index) can be specialised with hiprtc             // a smooth non-power-of-two factor whose four-step passes can be specialised with hiprtc (jit.hip)
template <typename T> int launch_jit_fourstep(int pass, bool inverse, const JitCfg &cfg, const RealArgs<T> &a, hipStream_t s);
bool fourstep_supported(int F);
void fourstep_build_wide_twiddles(int F, HostTable &out);       // empty unless F has a wide (E = 16) recipe
bool fourstep_wide(int dtype, int pass, int F);                 // this pass of length F runs the wide recipe (the caller then passes twp_col_w and RealArgs::wide = 1)
template <typename T> int launch_fourstep(int pass, int F, bool inverse, const RealArgs<T> &a, hipStream_t s);
// kernels_fourstep_real.hip : the passes of the REAL four-step (stage: 1 = real column FFT, row store; 2 = twiddled column pass
// writing the half spectrum (R2C) ; 3 = the same writing DCT-II outputs ; 4 / 5 = first pass of the inverse direction, C2R / DCT-III ; 6 = second pass of the fused DCT-IV four-step ; 7 = last pass of the inverse direction, column C2R)
bool fourstep_real_supported(int N1, int N2);
template <typename T> int launch_fourstep_real(int stage, int F, const RealArgs<T> &a, hipStream_t s);

// big.hip : four-step pieces for lanes that do not fit LDS
template <typename T>
int launch_big_twiddle(cpx<T> *data, int64_t lanes, int F1, int F2, const cpx<T> *twlo, const cpx<T> *twhi, int logB, int conj,
                       T scale, hipStream_t s);
template <typename T>
int launch_blue_stage(int stage, cpx<T> *dst, int64_t pitch_dst, const cpx<T> *src, int64_t pitch_src, int64_t lanes, int F, int M,
                      const cpx<T> *chirp, const cpx<T> *bhat, int inverse, T scale, hipStream_t s);   // 0 pre, 1 mid, 2 post (big.hip)
template <typename T>
int launch_big_pre(int gen_op, const RealArgs<T> &a, cpx<T> *z, hipStream_t s, int conj_z = 0);    // raw lanes (a.in, a.pitch_in) -> z[lane][F] (conj_z: its conjugate)
template <typename T>
int launch_big_post(int gen_op, const RealArgs<T> &a, const cpx<T> *z, hipStream_t s);   // z[lane][F] -> a.out
size_t generic_max_len(size_t csize);   // longest complex FFT the single-launch LDS kernel can hold

// jit.hip : hiprtc specialisation of the register-resident kernel for smooth non-power-of-two lengths
bool jit_choose(int dtype, int n, JitCfg &cfg, bool allow_partial = false);
void shard_release_all();   // shard.hip: every shard worker frees its chunk buffers (ndfft_release_workspace)
bool jit_choose_col(int dtype, int n, const JitCfg &row_cfg, JitCfg &col_cfg);   // true: column tiles of a C2C plan should use col_cfg instead of row_cfg
void jit_build_twiddles(const JitCfg &cfg, HostTable &out);
int launch_jit_c2c(int dtype, const JitCfg &cfg, int nt, const Pow2Args &a, hipStream_t s);
bool jit_c2c_row_vec(int dtype, JitCfg &cfg);   // f32 C2C rows: true = cfg was changed to the same radix list on half the threads (twice the elements), which can use 16-byte accesses (jit.hip)
int jit_col_lanes(int dtype, const JitCfg &cfg, bool c2c = false);   // c2c: a 4-lane tile is acceptable (complex output rows)
// thread-per-lane two-factor kernels (reg_kernel.h), specialised with hiprtc
bool regfft_factor(int n, int *n1, int *n2);
int regfft_max_n(int dtype);
int launch_jit_regfft(int dtype, int n1, int n2, bool stage, const TinyArgs &a, hipStream_t s);
int launch_jit_regreal(int dtype, int gop, int n, int f1, int f2, bool stage, const RegRealArgs &a, hipStream_t s);
template <typename T> int launch_jit_real(int gen_op, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s);

// transpose.hip : batched LDS-padded 2-D transpose, elem size 4/8/16 bytes
// out[b][c][r] = in[b][r][c];  in pitch = ld_in elements per row, out pitch = ld_out
int launch_transpose(const void *in, void *out, int64_t batch, int64_t rows, int64_t cols, int64_t ld_in,
                     int64_t ld_out, int64_t bstride_in, int64_t bstride_out, int elem_bytes, hipStream_t s);

}  // namespace ndfft
