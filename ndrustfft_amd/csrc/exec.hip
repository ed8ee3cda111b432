// exec.hip -- ndfft_exec / ndfft_exec_device: the body of one nd* call.
// Replaces the reference's lane iterator (create_transform! src/lib.rs:100-167 and its _par twin
// 169-238): validation that mirrors the reference's panics, stride canonicalisation (the three
// iterator strategies collapse into "every lane along `axis`, arbitrary signed strides"), kernel
// choice, and -- for host arrays -- staging through HBM.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#if defined(__x86_64__) && !defined(NDFFT_NO_NT_COPY)
#include <immintrin.h>   // host-side streaming copy of the bounce pipeline (bulk_copy)
#endif

#include "engine.h"
#include "pow2_real.h"

namespace ndfft {

struct BatchDim { int64_t shape, sin, sout; };

struct Problem {
    const ndfft_plan *plan;
    int op;
    int64_t xlen, ylen;          // lane lengths
    int64_t xs, ys;              // axis strides
    std::vector<BatchDim> b;     // merged batch dims, slowest first
    int64_t nlanes;
    double scale;
    int keep_out = 0;            // column kernels: cache-allocating stores (the output is re-read right away, col_split)
    int stream_in = 0;           // column kernels: streaming loads (the input must not evict a cache-resident intermediate)
    int no_xcd_map = 0;          // column kernels: identity workgroup -> tile map (the stages of col_split: the map cost 7-20 us there)
    int makhoul_out = 0;         // column C2R kernels: outputs through the inverse of Makhoul's permutation (last pass of real_fourstep_inv, DCT-III)
};

static size_t real_size(int dtype) { return dtype == NDFFT_F32 ? 4 : 8; }
static bool op_in_cplx(int op) { return op == NDFFT_OP_C2C_FWD || op == NDFFT_OP_C2C_INV || op == NDFFT_OP_C2R; }
static bool op_out_cplx(int op) { return op == NDFFT_OP_C2C_FWD || op == NDFFT_OP_C2C_INV || op == NDFFT_OP_R2C; }

static int kind_of_op(int op) {
    if (op == NDFFT_OP_C2C_FWD || op == NDFFT_OP_C2C_INV) return NDFFT_KIND_C2C;
    if (op == NDFFT_OP_R2C || op == NDFFT_OP_C2R) return NDFFT_KIND_R2C;
    return NDFFT_KIND_DCT;
}

// validation shared by the host and device entry points; fills Problem. Returns 1 for "nothing to do".
static int prepare(const ndfft_plan *plan, int op, int ndim, const int64_t *shape_in, const int64_t *stride_in,
                   const int64_t *shape_out, const int64_t *stride_out, int axis, int norm, double scale,
                   Problem &P, bool &nothing) {
    nothing = false;
    if (!plan) return fail(NDFFT_ERR_INVALID_ARG, "plan is null");
    if (op < NDFFT_OP_C2C_FWD || op > NDFFT_OP_DCT4) return fail(NDFFT_ERR_INVALID_ARG, "bad op");
    if (kind_of_op(op) != plan->kind) return fail(NDFFT_ERR_INVALID_ARG, "op does not belong to this plan's handler kind");
    if (norm < NDFFT_NORM_NONE || norm > NDFFT_NORM_SCALE) return fail(NDFFT_ERR_INVALID_ARG, "bad norm");
    if (ndim < 0 || ndim > NDFFT_MAX_DIMS) return fail(NDFFT_ERR_INVALID_ARG, "ndim out of range");
    if (ndim && (!shape_in || !stride_in || !shape_out || !stride_out)) return fail(NDFFT_ERR_INVALID_ARG, "null shape/stride");
    // lib.rs:116  let n = output.shape()[axis];
    if (axis < 0 || axis >= ndim) {
        char m[96];
        snprintf(m, sizeof m, "index out of bounds: the len is %d but the index is %d", ndim, axis);
        return fail(NDFFT_ERR_AXIS, m);
    }
    // Zip::from(input.rows()).and(output.rows_mut()) needs equal producer shapes (lib.rs:120-121)
    std::vector<BatchDim> raw;
    P.nlanes = 1;
    for (int d = 0; d < ndim; ++d) {
        if (shape_in[d] < 0 || shape_out[d] < 0) return fail(NDFFT_ERR_INVALID_ARG, "negative extent");
        if (d == axis) continue;
        if (shape_in[d] != shape_out[d]) {
            char m[128];
            snprintf(m, sizeof m, "ndarray: Zip dimension mismatch on axis %d (%lld vs %lld)", d, (long long)shape_in[d],
                     (long long)shape_out[d]);
            return fail(NDFFT_ERR_SHAPE_MISMATCH, m);
        }
        P.nlanes *= shape_in[d];
        if (shape_in[d] != 1) raw.push_back({shape_in[d], stride_in[d], stride_out[d]});
    }
    P.plan = plan; P.op = op;
    P.xlen = shape_in[axis]; P.ylen = shape_out[axis];
    P.xs = stride_in[axis]; P.ys = stride_out[axis];
    if (P.nlanes == 0) { nothing = true; return NDFFT_OK; }   // the closure never runs: no size panic either
    // the lane method's asserts: data.len() first, then out.len() (lib.rs:314-315, 498-499, 507-508, 689-690)
    const int64_t want_in = (int64_t)ndfft_plan_lane_len_in(plan, op), want_out = (int64_t)ndfft_plan_lane_len_out(plan, op);
    const char *what = plan->kind == NDFFT_KIND_DCT ? "dct" : "fft";
    if (P.xlen != want_in || P.ylen != want_out) {
        char m[128];
        const bool in_bad = P.xlen != want_in;
        snprintf(m, sizeof m, "Size mismatch in %s, got %lld expected %lld", what,
                 (long long)(in_bad ? P.xlen : P.ylen), (long long)(in_bad ? want_in : want_out));
        return fail(NDFFT_ERR_SIZE_MISMATCH, m);
    }
    if (plan->n == 0) { nothing = true; return NDFFT_OK; }
    // merge adjacent batch dims that are contiguous in both views
    for (const BatchDim &d : raw) {
        if (!P.b.empty()) {
            BatchDim &p = P.b.back();
            if (p.sin == d.shape * d.sin && p.sout == d.shape * d.sout) {
                p.shape *= d.shape; p.sin = d.sin; p.sout = d.sout;
                continue;
            }
        }
        P.b.push_back(d);
    }
    // normalisation scalar at the reference's application point
    const double n = (double)plan->n;
    switch (op) {
        case NDFFT_OP_C2C_FWD: case NDFFT_OP_R2C: P.scale = 1.0; break;                         // ignored: lib.rs:313-318, 497-503
        case NDFFT_OP_C2C_INV: case NDFFT_OP_C2R:                                               // lib.rs:333-338, 525-531
            P.scale = norm == NDFFT_NORM_NONE ? 1.0 : norm == NDFFT_NORM_DEFAULT ? 1.0 / n : scale; break;
        default:                                                                                 // lib.rs:736-741
            P.scale = norm == NDFFT_NORM_NONE ? 1.0 : norm == NDFFT_NORM_DEFAULT ? 2.0 : scale; break;
    }
    return NDFFT_OK;
}

int validate_call(const ndfft_plan *plan, int op, int ndim, const int64_t *shape_in, const int64_t *stride_in, const int64_t *shape_out,
                  const int64_t *stride_out, int axis, int norm, double scale, bool *nothing) {
    Problem P;
    return prepare(plan, op, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, P, *nothing);
}

// ---------------------------------------------------------------------------------------------
// dispatch of one canonicalised problem with <= kMaxBatchDims batch dims
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dispatch_generic(const Problem &P, const void *d_in, void *d_out, const DevTables &dt, hipStream_t stream) {
    const ndfft_plan *plan = P.plan;
    const int n = (int)plan->n;
    GenArgs<T> a;
    memset(&a, 0, sizeof a);
    int slot = CFG_MAIN, gop = 0;
    switch (P.op) {
        case NDFFT_OP_C2C_FWD: gop = G_C2C_FWD; break;
        case NDFFT_OP_C2C_INV: gop = G_C2C_INV; break;
        case NDFFT_OP_R2C: gop = n % 2 ? G_R2C_ODD : G_R2C_EVEN; break;
        case NDFFT_OP_C2R: gop = n % 2 ? G_C2R_ODD : G_C2R_EVEN; break;
        case NDFFT_OP_DCT1:
            if (n == 1) { gop = G_DCT2_ODD; }              // y0 = x0/2 + x0/2 = x0 = DCT-II of length 1
            else { gop = G_DCT1; slot = CFG_DCT1; }
            break;
        case NDFFT_OP_DCT2: gop = n % 2 ? G_DCT2_ODD : G_DCT2_EVEN; break;
        case NDFFT_OP_DCT3: gop = n % 2 ? G_DCT3_ODD : G_DCT3_EVEN; break;
        default: gop = n % 2 ? G_DCT4_ODD : G_DCT4_EVEN; slot = CFG_DCT4; break;
    }
    const FftConfig &c = plan->cfg[slot];
    const DevConfig &d = dt.cfg[slot];
    a.in = d_in; a.out = d_out;
    a.nlanes = P.nlanes;
    a.op = gop; a.n = n;
    a.n_in = (int)P.xlen; a.n_out = (int)P.ylen;
    a.in_cplx = op_in_cplx(P.op); a.out_cplx = op_out_cplx(P.op);
    a.F = c.F;
    a.npass = (int)c.radix.size();
    for (int i = 0; i < a.npass; ++i) a.radix[i] = c.radix[i];
    a.scale = (T)P.scale;
    a.tw = (const cpx<T> *)d.tw; a.aux1 = (const cpx<T> *)d.aux1; a.aux2 = (const cpx<T> *)d.aux2;
    a.blue = c.blue; a.M = c.M; a.npassM = (int)c.radixM.size();
    for (int i = 0; i < a.npassM; ++i) a.radixM[i] = c.radixM[i];
    a.twM = (const cpx<T> *)d.twM; a.chirp = (const cpx<T> *)d.chirp; a.bhat = (const cpx<T> *)d.bhat;

    a.gin.axis_stride = P.xs; a.gout.axis_stride = P.ys;
    a.gin.nb = a.gout.nb = (int)P.b.size();
    for (size_t i = 0; i < P.b.size(); ++i) {
        a.gin.bshape[i] = a.gout.bshape[i] = P.b[i].shape;
        a.gin.bstride[i] = P.b[i].sin; a.gout.bstride[i] = P.b[i].sout;
    }
    // thread -> (lane, element) map for global IO: along the lane if it is unit-stride, else across
    // adjacent lanes if the fastest batch dim is unit-stride (the fused LDS transpose), else row.
    const bool last_in1 = !P.b.empty() && P.b.back().sin == 1, last_out1 = !P.b.empty() && P.b.back().sout == 1;
    a.load_mode = (P.xs == 1 || P.xlen == 1 || !last_in1) ? IO_ROW : IO_COL;
    a.store_mode = (P.ys == 1 || P.ylen == 1 || !last_out1) ? IO_ROW : IO_COL;

    // LDS pitch (complex elements): the padded FFT buffer (or Bluestein M) and the raw input lane must fit
    const int in_c = a.in_cplx ? a.n_in : (a.n_in + 1) / 2;
    const int len = std::max(c.blue ? c.M : c.F, 1);
    const int pitch = std::max(generic_z_len(len), in_c) | 1;   // odd: lanes land in different banks for the IO_COL transposes
    const size_t csize = 2 * sizeof(T);
    const size_t lds_cap = 160 * 1024;
    if (generic_lds_bytes(1, pitch, csize, 2) > lds_cap) {
        char m[160];
        snprintf(m, sizeof m, "lane of %d elements needs %zu B of LDS (> %zu): multi-pass path not built yet", n,
                 generic_lds_bytes(1, pitch, csize), lds_cap);
        return fail(NDFFT_ERR_UNSUPPORTED, m);
    }
    const bool col = a.load_mode == IO_COL || a.store_mode == IO_COL;
    // threads per lane in the FFT phases: one butterfly each in the pass with the most butterflies
    int nb_max = 1;
    {
        const std::vector<int> &rr = c.blue ? c.radixM : c.radix;
        for (int r : rr) nb_max = std::max(nb_max, len / r);
    }
    const bool elementwise = gop == G_C2C_FWD || gop == G_C2C_INV || gop == G_R2C_EVEN || gop == G_R2C_ODD;
    // (the in-place and the prime-radix kernels are compiled for <= 512 threads: 256-VGPR budget)
    const bool want_inplace = elementwise && nb_max <= 512;
    const int maxthr = (want_inplace || generic_needs_big(a.radix, a.npass, a.radixM, a.blue ? a.npassM : 0)) ? 512 : 1024;
    int fft_tpl = 1; while (fft_tpl < nb_max && fft_tpl < maxthr) fft_tpl <<= 1;
    // in place (one LDS buffer per lane) when the op is elementwise at both ends and a thread never owns more
    // than one butterfly of a pass: needs fft_tpl >= nb_max and one thread group per lane
    const bool inplace = want_inplace && fft_tpl >= nb_max;
    const int nbuf = inplace ? 1 : 2;
    // lanes per block: fill ~64 KiB of LDS (2+ blocks/CU), but never more lanes than exist
    const size_t per_lane = (size_t)nbuf * (size_t)pitch * csize;
    int lpb = (int)std::min<size_t>((64 * 1024) / per_lane, kMaxLpb);
    if (col) {
        // want >= 128 B contiguous across lanes per row of the tile; take more LDS if that is what it costs
        const int want = (int)std::min<size_t>(kMaxLpb, std::max<size_t>(128 / sizeof(T) / (a.in_cplx ? 2 : 1), 16));
        const int fit = (int)std::min<size_t>((lds_cap - 2048) / per_lane, kMaxLpb);
        lpb = std::max(lpb, std::min(want, fit));
    }
    lpb = std::max(1, lpb);
    if (inplace) lpb = std::max(1, std::min(lpb, maxthr / fft_tpl));   // one thread group per lane
    if ((int64_t)lpb > P.nlanes) lpb = (int)P.nlanes;
    // keep every CU busy: prefer >= 1024 blocks when lanes allow
    while (lpb > 1 && !col && (P.nlanes + lpb - 1) / lpb < 1024) lpb = (lpb + 1) / 2;
    int lpb_log = 0;
    if (col) {   // the across-lanes thread map needs a power of two
        while ((2 << lpb_log) <= lpb) ++lpb_log;
        lpb = 1 << lpb_log;
    }
    a.lpb = lpb; a.pitch = pitch; a.lpb_log = lpb_log; a.inplace = inplace;
    const size_t lds = generic_lds_bytes(lpb, pitch, csize, nbuf);
    int threads = 64; while (threads < lpb * fft_tpl && threads < maxthr) threads <<= 1;
    if (col) while (threads < 4 * lpb && threads < maxthr) threads <<= 1;
    if (inplace && threads < lpb * fft_tpl) return fail(NDFFT_ERR_INVALID_ARG, "internal: in-place thread map");
    fft_tpl = std::min(fft_tpl, threads);
    int io_tpl = 1; while (io_tpl < std::max(a.n_in, a.n_out) && io_tpl < threads) io_tpl <<= 1;
    a.fft_tpl_log = 0; while ((1 << a.fft_tpl_log) < fft_tpl) ++a.fft_tpl_log;
    a.io_tpl_log = 0; while ((1 << a.io_tpl_log) < io_tpl) ++a.io_tpl_log;
    set_last_path(col ? "generic_col" : (P.xs == 1 || P.xlen == 1) && (P.ys == 1 || P.ylen == 1) ? "generic_row" : "generic_strided");
    return launch_generic<T>(a, threads, lds, stream);
}

// ---------------------------------------------------------------------------------------------
// Workspace of one host thread ON ONE DEVICE: scratch arrays of the multi-pass routes (keyed by stream; they
// grow, never shrink), the staging buffers and pinned bounce buffers of ndfft_exec, and the streams / events of
// its chunk pipeline.  A thread that alternates ndfft_set_device gets one of these per device (nothing allocated
// on device 0 is ever handed to a kernel on device 1), and everything is released when the thread exits.
// ---------------------------------------------------------------------------------------------
struct Scratch { void *p = nullptr; size_t cap = 0; };
struct Staging {
    void *p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return NDFFT_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        NDFFT_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return NDFFT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct PinnedBuf {
    void *p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return NDFFT_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        NDFFT_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
        return NDFFT_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
struct Pipe {
    hipStream_t h2d = nullptr, cmp = nullptr, d2h = nullptr;
    std::vector<hipEvent_t> up, done, down;
    bool ok = false;
    void sync_all() { if (ok) { (void)hipStreamSynchronize(h2d); (void)hipStreamSynchronize(cmp); (void)hipStreamSynchronize(d2h); } }
    void release() {
        if (!ok) return;
        sync_all();
        for (auto *v : {&up, &done, &down}) { for (hipEvent_t e : *v) (void)hipEventDestroy(e); v->clear(); }
        (void)hipStreamDestroy(h2d); (void)hipStreamDestroy(cmp); (void)hipStreamDestroy(d2h);
        h2d = cmp = d2h = nullptr; ok = false;
    }
};
// What this thread's own calls imply about the 256 MiB Infinity Cache of the current device -- used for ONE decision, the load
// policy of the row / column-tile kernels that have a streaming-load form (BASELINE configs[1], [3], [4]): plain loads are up to 15 %
// faster when the input is resident in the Infinity Cache (4096 x 4096 c128: 0.86 vs 0.74 of the roofline), streaming (nt) loads 6 %
// faster when it comes from HBM (0.74 vs 0.70).  Round 2 bet on "resident" for every input <= 384 MiB; a chain of nd* calls loses that bet
// at every link (the output of a pass was written with nt stores, which bypass the cache).  The model is an LRU stack distance per buffer
// this thread has transformed:
//   * a buffer this thread WROTE as an output of more than 64 MiB is cold (nt stores) -> streaming loads when it becomes an input;
//   * a buffer it READ before (with either policy), or wrote as a small output, is worth plain loads iff the bytes this thread has moved
//     through the cache since (all inputs, small outputs) plus its own size fit ~256 MiB: a re-read input then is, or becomes, resident;
//     six rotating 256 MiB inputs never are (streaming loads for all of them);
//   * a buffer the model has never seen keeps round 2's size rule (plain loads up to 384 MiB) -- its producer is unknown.
// A stale entry (the allocator reused the addresses for something a foreign kernel produced) costs one call: after that the buffer is "read".
// ndfft_set_input_hint overrides the model per host thread.  Speed only: either policy gives the same results.
struct MallModel {
    enum State { READ = 0, OUT_SMALL = 1, OUT_COLD = 2 };
    struct Entry { uintptr_t lo, hi; uint64_t stamp; int state; };
    std::vector<Entry> e;
    uint64_t clock = 0;                          // bytes this thread has moved through the cache's address stream (inputs read, small outputs)
    static constexpr uint64_t kCap = (uint64_t)256 << 20, kSmallOut = (uint64_t)64 << 20;
    Entry *find(const void *p, size_t bytes) {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        for (auto &x : e) if (lo < x.hi && x.lo < hi) return &x;
        return nullptr;
    }
    // 1: streaming loads, 0: plain loads, -1: unknown buffer (the launcher decides by size)
    int decide(const void *in, size_t bytes) {
        const Entry *x = find(in, bytes);
        if (!x) return -1;
        if (x->state == OUT_COLD) return 1;
        // a buffer larger than the cache (BASELINE configs[2]'s 4097 x 8192 c64 is 64 KiB over): only an IMMEDIATE re-read still finds most of it
        // there (the size rule decides, as for an unknown buffer); anything else this thread has read since has pushed it out -- round 5: the
        // rotating-pairs table ran this shape with plain loads (policy 0), 95.9 us against 92 us with streaming loads
        if (bytes > kCap) return clock == x->stamp ? -1 : 1;
        return clock - x->stamp + bytes <= kCap ? 0 : 1;
    }
    void put(const void *p, size_t bytes, int state, bool through_cache) {
        const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
        for (size_t i = 0; i < e.size();) { if (lo < e[i].hi && e[i].lo < hi) e.erase(e.begin() + i); else ++i; }
        if (through_cache) clock += bytes;
        if (e.size() >= 32) e.erase(e.begin());  // oldest first
        e.push_back({lo, hi, clock, state});
    }
    void note_read(const void *p, size_t bytes) { put(p, bytes, READ, true); }
    void note_write(const void *p, size_t bytes) { if (bytes <= kSmallOut) put(p, bytes, OUT_SMALL, true); else put(p, bytes, OUT_COLD, false); }
};
struct DeviceWs {
    MallModel mall;
    std::map<hipStream_t, Scratch> scratch[8];
    Staging stage_in, stage_out;
    PinnedBuf bounce_in[3], bounce_out[3];
    Pipe pipe;
    void release() {   // the owning device must be current
        pipe.release();
        for (auto &m : scratch) { for (auto &kv : m) if (kv.second.p) (void)hipFree(kv.second.p); m.clear(); }
        stage_in.release(); stage_out.release();
        for (auto &b : bounce_in) b.release();
        for (auto &b : bounce_out) b.release();
    }
};
struct ThreadWs {
    std::map<int, DeviceWs> dev;
    void release_all() {
        if (dev.empty()) return;
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); dev.clear(); return; }   // runtime already gone (process teardown)
        for (auto &kv : dev) { if (hipSetDevice(kv.first) == hipSuccess) { (void)hipDeviceSynchronize(); kv.second.release(); } }
        (void)hipSetDevice(cur);
        dev.clear();
    }
    ~ThreadWs() { release_all(); }   // a worker thread that exits gives its device memory and streams back
};
static thread_local ThreadWs g_tws;
static int current_ws(DeviceWs **out) {
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    *out = &g_tws.dev[dev];
    return NDFFT_OK;
}
static thread_local int g_input_hint = NDFFT_INPUT_AUTO;
static thread_local int g_last_policy = -1;       // load policy the last call on this thread asked the model for (diagnostic)
static bool F_nt_ok(int F) { return F >= 64; }   // (short lanes: the staging loads are not 16-byte vectors on every path)
// load policy for the dense C2C row kernels on input `in` (Pow2Args::stream_in), and the bookkeeping for the next call
static int row_load_policy(const void *in, size_t bytes, const void *out, size_t out_bytes);
static int c2c_row_load_policy(const void *in, const void *out, size_t bytes) { return row_load_policy(in, bytes, out, bytes); }
static int row_load_policy(const void *in, size_t bytes, const void *out, size_t out_bytes) {
    const int force = sw().stream_loads;            // NDFFT_STREAM_LOADS: 0 / 1 forces a policy
    DeviceWs *ws;
    if (current_ws(&ws)) return -1;
    int pol = force >= 0 ? (force != 0) : g_input_hint == NDFFT_INPUT_CACHED ? 0 : g_input_hint == NDFFT_INPUT_COLD ? 1 : ws->mall.decide(in, bytes);
    const bool nt = pol >= 0 ? pol != 0 : stream_loads_for(bytes);
    g_last_policy = nt ? 1 : 0;
    ws->mall.note_read(in, bytes);
    // nt stores: a large output bypasses the cache (fft -> ifft on 4096 x 4096 c128: the second pass is 3-5 % faster with streaming loads); a small
    // one is still found there (1024 x 4096, 64 MiB: plain loads 2-3 % faster) -- tools/probes/chain_hint.py, profiles/r05/r05d_chain_hint.txt
    ws->mall.note_write(out, out_bytes);
    return nt ? 1 : 0;
}
static int get_scratch(int which, hipStream_t s, size_t bytes, void **out) {
    DeviceWs *ws;
    int rc = current_ws(&ws);
    if (rc) return rc;
    Scratch &sc = ws->scratch[which][s];
    if (bytes > sc.cap) {
        if (sc.p) { NDFFT_HIP(hipStreamSynchronize(s)); NDFFT_HIP(hipFree(sc.p)); sc.p = nullptr; sc.cap = 0; }
        NDFFT_HIP(hipMalloc(&sc.p, bytes));
        sc.cap = bytes;
    }
    *out = sc.p;
    return NDFFT_OK;
}

static int dispatch(const Problem &P, const void *d_in, void *d_out, hipStream_t stream);

static int transpose_batched(const void *in, void *out, int64_t batch, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                             int64_t bs_in, int64_t bs_out, int esz, hipStream_t s) {
    for (int64_t b0 = 0; b0 < batch; b0 += 32768) {   // grid.z limit
        const int64_t nb = std::min<int64_t>(32768, batch - b0);
        int rc = launch_transpose((const char *)in + b0 * bs_in * esz, (char *)out + b0 * bs_out * esz, nb, rows, cols, ld_in, ld_out,
                                  bs_in, bs_out, esz, s);
        if (rc) return rc;
    }
    return NDFFT_OK;
}


// Route switches (switches.h; parsed once, never read from the environment on the call path).  Each closes one route so that the kernel
// behind it runs: the parity tests reach every fallback kernel that way.
static bool narrow_enabled() { return true; }                              // (long strided lanes on narrow column tiles; the transpose route is the fallback)
static bool narrow_dct_enabled() { return sw().narrow_dct; }               // NDFFT_NARROW_DCT=1: long strided DCT lanes on the narrow tiles again
static bool wave_enabled() { return sw().wave; }                           // NDFFT_WAVE=0: short dense C2C lanes on the older kernels
static bool fourstep2_enabled() { return sw().fourstep2; }                 // NDFFT_FOURSTEP2=0: long power-of-two lanes on the three-pass form
// NDFFT_REAL_FOURSTEP=0 keeps long real-data lanes on the packed complex four-step with separate PRE / POST passes
// (2 = for every eligible op, also where the plan's table says the packed route is faster: parity tests)
static int real_fourstep_enabled() { return sw().real_fourstep; }
// largest handler length the thread-per-lane real-op register kernels take (raw lane + Z + outputs in registers)
static int regreal_max_n(int f64) {
    return f64 ? (int)NDFFT_DEV_INT("NDFFT_REGREAL_MAX_F64", 48)           // f64 n = 48: 0.66 vs 0.30
               : (int)NDFFT_DEV_INT("NDFFT_REGREAL_MAX_F32", 72);          // f32 n = 64: 0.66 vs 0.50; n = 96 / 100: 0.39 / 0.37 vs 0.41 / 0.49
}
static bool chunk_out_enabled() { return true; }                           // (R2C rows store whole chunks; settled in round 2)
static bool tiny_enabled() { return sw().tiny; }                           // NDFFT_TINY=0: very short lanes on the LDS kernel
static bool plain_enabled() { return sw().plain; }                         // NDFFT_PLAIN=0: odd-n real ops with a smooth inner FFT on the LDS kernel
static bool blue_enabled() { return sw().blue; }                           // NDFFT_BLUE=0: Bluestein lengths on the LDS kernel
static bool colsplit_enabled() { return sw().colsplit; }                   // NDFFT_COLSPLIT=0: long strided lanes on the narrow-tile / transpose routes

// Column four-step (pow2_real.h, CS kernels): a long STRIDED power-of-two lane, n = F1 * F2, as two passes
// of wide column tiles over a dense C-layout block [O][n][I] -- no transpose, no narrow tiles:
//   C2C      A: column C2C of length F1 over a = row / F2 (lanes (b, i): F2*I contiguous)  -> S[o][k1][b][i]
//            B: CS=1 kernel of length F2 over b, twiddle on load, rows k1 + F1 k2            -> out
//   R2C      A: column R2C of length F1 (k1 = 0..F1/2); B: CS=2 kernel (Hermitian row map)   -> out rows 0..n/2
//   C2R      A: CS=3 kernel (Hermitian gather, inverse F2, conj twiddle) -> S; B: column C2R of length F1 -> out
// Cost: two reads + two writes of the array at 60-80 % of the HBM roofline each, against one pass of
// 16-32 byte row segments at 15-25 % (8192-long f32 lanes).
template <typename T>
static int col_split(const Problem &P, const void *d_in, void *d_out, const FftConfig &c, const DevConfig &d, hipStream_t stream) {
    const int64_t I = P.b.back().shape;
    const int64_t O = P.b.size() == 2 ? P.b[0].shape : 1;
    const int64_t sin_o = P.b.size() == 2 ? P.b[0].sin : 0, sout_o = P.b.size() == 2 ? P.b[0].sout : 0;
    const int F1 = c.cs_F1, F2 = c.cs_F2;
    const bool r2c = P.op == NDFFT_OP_R2C, c2r = P.op == NDFFT_OP_C2R, inv = P.op == NDFFT_OP_C2C_INV;
    const int K1 = (r2c || c2r) ? F1 / 2 + 1 : F1;
    // Column chunks: the intermediate of one chunk (K1*F2*C complex, <= 144 MiB) is written by stage A with
    // cache-allocating stores and re-read by stage B before much of it has left the 256 MiB Infinity Cache.
    // Measured (8192x8192 f32 R2C axis 0): one chunk 248 us, two chunks (136 MiB each) 212 us, three 228 us;
    // small chunks LOSE (32 MiB: 293 us, 8 MiB: 628 us): every chunk costs two launches of a few microseconds.
    // (round 6: odd chunks on a SECOND stream with a scratch array of their own, so that one chunk's launches drain under the other's work -- measured no better at any chunk size:
    //  cfg3-A 178 us (one stream, two chunks) against 184 (two streams, two chunks) / 183 (two streams, four chunks), cfg3-A' 193 against 219 / 192; profiles/r09/r09t_cs_two_streams_negative.txt)
    int64_t C = I;
    {
        const int64_t target = (int64_t)sw().cs_chunk_mb << 20;   // NDFFT_CS_CHUNK_MB (0 = one chunk)
        const int64_t per_col = (int64_t)K1 * F2 * (int64_t)sizeof(cpx<T>);
        if (target > 0 && O == 1 && per_col * I > target) {   // equal chunks, a multiple of 64 columns each
            const int64_t nchunk = (per_col * I + target - 1) / target;
            C = std::max<int64_t>(64, (((I + nchunk - 1) / nchunk) + 63) & ~(int64_t)63);
        }
        if (C > I) C = I;
    }
    const bool chunked = C < I;
    // Row pitch of the intermediate.  It is OUR array, so its rows need not sit a power of two apart: tools/colprobe.hip
    // (profiles/r05/r05a_colprobe.txt) reads 256-byte segments 8192 rows deep at 0.55 of 8 TB/s when the rows are 32 KiB apart and
    // at 0.85 when they are 32 KiB + 512 B apart (writes 0.33 -> 0.70): a power-of-two pitch keeps every row of a column tile on
    // the same few HBM channels.  pad_bytes = padding in BYTES (0 keeps dense rows); only for one outer block (the padded
    // layout needs the (b, i) batch dimensions unmerged, and the column kernels take two).
    int64_t pad = 0;
    if (O == 1) {
        constexpr int pad_bytes = 0;   // measured on cfg3-A / cfg3-A' (profiles/r05/r05a_cs_pad_abab.txt): 0 / 256 / 512 / 1024 B all within 200-207 us: no effect, off
        pad = pad_bytes / (int64_t)sizeof(cpx<T>);
    }
    const int64_t Cp = C + pad;              // pitch of one (k1, b) row of the intermediate, in complex elements
    void *S;
    int rc = get_scratch(5, stream, (size_t)O * K1 * F2 * Cp * sizeof(cpx<T>), &S);
    if (rc) return rc;
    const DevTables *dt2;
    if ((rc = get_dev_tables(c.cs_sub2, &dt2))) return rc;
    const size_t ein = (op_in_cplx(P.op) ? 2 : 1) * sizeof(T), eout = (op_out_cplx(P.op) ? 2 : 1) * sizeof(T);
    for (int64_t c0 = 0; c0 < I; c0 += C) {
        const int64_t Cc = std::min(C, I - c0);
        const char *in_c = (const char *)d_in + (size_t)c0 * ein;
        char *out_c = (char *)d_out + (size_t)c0 * eout;
        RealArgs<T> a;
        a.nlanes = O * K1 * Cc; a.pitch_in = 0; a.pitch_out = 0; a.vec_in = 0; a.vec_out = 0; a.xcd_remap = 0; a.keep_out = 0;
        a.n = F2; a.F = F2; a.n_in = F2; a.n_out = F2; a.scale = (T)P.scale;
        a.aux1 = nullptr; a.aux2 = nullptr; a.twp = (const cpx<T> *)dt2->cfg[CFG_MAIN].twp_col;
        a.inner = Cc;
        a.cs_twlo = (const cpx<T> *)d.cs_twlo; a.cs_twhi = (const cpx<T> *)d.cs_twhi; a.cs_logB = c.cs_logB;
        a.cs_k1n = K1; a.cs_f1 = F1; a.cs_n = (int)P.plan->n; a.cs_outer_in = sin_o; a.cs_outer_out = sout_o; a.cs_pitch = I;
        Problem Q;
        Q.plan = c.cs_sub1; Q.nlanes = O * F2 * Cc; Q.scale = 1.0; Q.no_xcd_map = 1;
        const bool split_bi = chunked || pad > 0;   // (b, i) as two batch dimensions: the intermediate's rows are not back to back
        if (!c2r) {
            // A: column transform of length F1 over a = row / F2; lanes (b, i)
            Q.op = P.op; Q.xlen = F1; Q.ylen = K1; Q.xs = (int64_t)F2 * I; Q.ys = (int64_t)F2 * Cp;
            // (stream_in: the caller's array is read once and must not push the intermediate out of the Infinity Cache -- round 5, A-B-A-B on cfg3-A:
            //  198-201 -> 188.5-191 us; the C2R form's first stage below: 225 -> 220 us; profiles/r08/r08r_cs_stage_nt_abab.txt)
            Q.keep_out = chunked; Q.stream_in = (int)NDFFT_DEV_INT("NDFFT_CSA_NT", 1);
            if (split_bi) { Q.b.push_back({(int64_t)F2, I, Cp}); Q.b.push_back({Cc, 1, 1}); }
            else {
                if (O > 1) Q.b.push_back({O, sin_o, (int64_t)K1 * F2 * I});
                Q.b.push_back({(int64_t)F2 * I, 1, 1});
            }
            if ((rc = dispatch(Q, in_c, S, stream))) return rc;
            // B: twiddle on load, length F2 over b, rows k1 + F1 k2 (R2C: Hermitian row map)
            a.in = S; a.out = out_c;
            a.outer_in = (int64_t)F2 * Cp; a.outer_out = 0; a.elem_in = Cp; a.elem_out = (int64_t)F1 * I;
            if ((rc = launch_colsplit<T>(r2c ? 2 : 1, inv, a, stream))) return rc;
        } else {
            // A: Hermitian gather, inverse of length F2 over k2, conj twiddle -> S[o][k1][b][i]
            a.in = in_c; a.out = S;
            a.outer_in = 0; a.outer_out = (int64_t)F2 * Cp; a.elem_in = I; a.elem_out = Cp;
            a.stream_in = (int)NDFFT_DEV_INT("NDFFT_CS3_NT", 1);
            if ((rc = launch_colsplit<T>(3, true, a, stream))) return rc;
            a.stream_in = 0;
            // B: column C2R of length F1 over k1
            Q.op = NDFFT_OP_C2R; Q.xlen = K1; Q.ylen = F1; Q.xs = (int64_t)F2 * Cp; Q.ys = (int64_t)F2 * I;
            if (split_bi) { Q.b.push_back({(int64_t)F2, Cp, I}); Q.b.push_back({Cc, 1, 1}); }
            else {
                if (O > 1) Q.b.push_back({O, (int64_t)K1 * F2 * I, sout_o});
                Q.b.push_back({(int64_t)F2 * I, 1, 1});
            }
            if ((rc = dispatch(Q, S, out_c, stream))) return rc;
        }
    }
    set_last_path("col_split");
    return NDFFT_OK;
}

// Pitch padding of the four-step routes' intermediates (round 5).  Pass 2 reads the intermediate as column tiles: 128-byte rows one pitch apart.  With the natural
// power-of-two pitch (256 x 65536 c128: 4096 B) the rows of a tile fall on few HBM channels; 128 bytes more per row and a copy of that shape runs 6.5 % faster
// (9 % with streaming LDS-DMA loads; + 512 B: 3.7 %, + 1 KiB: slower -- tools/ldsdma_probe.hip `fs2`, profiles/r08/r08j_fs2_pitch.txt).  The real four-step's
// forward intermediate always had such a pitch (N1/2 + 1 rounded up to whole lines).  In the product the gain is small: nddct4 64 x 262144 f64 165 -> 157 us,
// 32 x 2^20 c64 291 -> 281-288 us, 256 x 65536 c128 204 -> 201-204 us (profiles/r08/r08k_longlanes_pad.txt) -- the passes already run at the copy rate of
// their tile shapes (two passes of 537 MB at 5.3-5.7 TB/s = 195 us for the c128 case).  Elements of type cpx<T>.
template <typename T> static int fs_pad_elems() { return (int)(NDFFT_DEV_INT("NDFFT_FS_PAD", 128) / (long)sizeof(cpx<T>)); }

// Four-step complex FFT of length F = F1*F2 on L lanes (zin / zout: lane pitches in elements).
// zin may equal zout.  Sub-FFTs run through dispatch() on the row kernels.
template <typename T>
static int big_fft(const FftConfig &c, const DevConfig &d, const cpx<T> *zin, int64_t pitch_in, cpx<T> *zout, int64_t pitch_out,
                   int64_t L, bool inverse, T scale, hipStream_t stream) {
    if (c.bigblue) {
        // Bluestein over global memory: two FFT_M through dispatch() (pow2 row kernel, or its own four-step for
        // M > 16384 -- which uses scratch slots 2 / 3, hence 6 / 7 here) between three elementwise stages
        const int Fl = c.F, M = c.M;
        void *a1, *a2;
        int rcb;
        if ((rcb = get_scratch(6, stream, (size_t)L * M * sizeof(cpx<T>), &a1))) return rcb;
        if ((rcb = get_scratch(7, stream, (size_t)L * M * sizeof(cpx<T>), &a2))) return rcb;
        const cpx<T> *chirp = (const cpx<T> *)d.chirp, *bhat = (const cpx<T> *)d.bhat;
        if ((rcb = launch_blue_stage<T>(0, (cpx<T> *)a1, M, zin, pitch_in, L, Fl, M, chirp, bhat, inverse ? 1 : 0, (T)1, stream))) return rcb;
        Problem Q;
        Q.plan = c.sub1; Q.op = NDFFT_OP_C2C_FWD; Q.xlen = Q.ylen = M; Q.xs = Q.ys = 1; Q.nlanes = L; Q.scale = 1.0;
        Q.b.push_back({L, (int64_t)M, (int64_t)M});
        if ((rcb = dispatch(Q, a1, a2, stream))) return rcb;
        if ((rcb = launch_blue_stage<T>(1, (cpx<T> *)a2, M, (const cpx<T> *)a2, M, L, Fl, M, chirp, bhat, 0, (T)1, stream))) return rcb;
        if ((rcb = dispatch(Q, a2, a1, stream))) return rcb;
        return launch_blue_stage<T>(2, zout, pitch_out, (const cpx<T> *)a1, M, L, Fl, M, chirp, bhat, inverse ? 1 : 0, inverse ? scale : (T)1, stream);
    }
    const int F1 = c.F1, F2 = c.F2;
    const int64_t F = (int64_t)F1 * F2;
    const int esz = (int)sizeof(cpx<T>);
    void *s1, *s2;
    int rc;
    const int64_t K1p = F1 + fs_pad_elems<T>();       // pitch of the two-pass intermediate s1[n2][k1]
    if ((rc = get_scratch(2, stream, (size_t)(L * std::max<int64_t>(F, (int64_t)F2 * K1p)) * esz, &s1))) return rc;
    // Two-pass form, no transpose launch, when both factors have a column kernel (powers of two, 64..1024):
    //   (1) length-F1 FFTs over the strided n1 axis, column load, stored TRANSPOSED as s1[n2][k1] (row store)
    //   (2) length-F2 FFTs over n2 of s1 (stride F1, adjacent k1 contiguous), twiddle W_F^(n2 k1) on load, stored at
    //       k1 + F1 k2 = natural order.   256 x 65536 c128: 317 us (three passes) -> see DESIGN.md section 3.5
    // (round 6: a smooth NON-power-of-two factor runs the same two passes on kernels specialised with hiprtc -- jit.hip: launch_jit_fourstep -- instead of the six-pass transpose route)
    const int dti = sizeof(T) == 8 ? NDFFT_F64 : NDFFT_F32;
    const FftConfig &sc1 = c.sub1->cfg[CFG_MAIN], &sc2 = c.sub2->cfg[CFG_MAIN];
    const int kind1 = (fourstep_supported(F1) && !sc1.twp_col.re.empty()) ? 1 : (sc1.fs_jit && jit_fourstep_ok(dti, sc1.fs_jitcfg)) ? 2 : 0;
    const int kind2 = (fourstep_supported(F2) && !sc2.twp_col.re.empty()) ? 1 : (sc2.fs_jit && jit_fourstep_ok(dti, sc2.fs_jitcfg)) ? 2 : 0;
    if (kind1 && kind2 && fourstep2_enabled()) {
        const DevTables *dt1, *dt2;
        if ((rc = get_dev_tables(c.sub1, &dt1)) || (rc = get_dev_tables(c.sub2, &dt2))) return rc;
        RealArgs<T> a;
        a.pitch_in = 0; a.vec_in = 0; a.vec_out = 0; a.xcd_remap = 0; a.keep_out = 0; a.stream_in = 0;
        a.aux1 = nullptr; a.aux2 = nullptr; a.chirp = nullptr; a.bhat = nullptr;
        a.cs_twlo = (const cpx<T> *)d.twlo; a.cs_twhi = (const cpx<T> *)d.twhi; a.cs_logB = c.logB;
        a.cs_k1n = 1; a.cs_f1 = F1; a.cs_n = (int)F; a.cs_outer_in = 0; a.cs_outer_out = 0; a.cs_pitch = 0;
        // (used by the half-line tiles only, F = 1024 f32 -- pow2_real.h; 32 x 2^20 c64: 403 -> 354 us, profiles/r06)
        a.xcd_chunk = (int)NDFFT_DEV_INT("NDFFT_FS_XCD_CHUNK", 32);
        // pass 1: lanes (l, n2)
        a.in = zin; a.out = s1; a.nlanes = L * F2; a.n = F1; a.F = F1; a.n_in = F1; a.n_out = F1; a.scale = (T)1;
        a.inner = F2; a.outer_in = pitch_in; a.outer_out = 0; a.elem_in = F2; a.elem_out = 0; a.pitch_out = K1p;
        const bool w1 = kind1 == 1 && fourstep_wide(dti, 1, F1) && dt1->cfg[CFG_MAIN].twp_col_w, w2 = kind2 == 1 && fourstep_wide(dti, 2, F2) && dt2->cfg[CFG_MAIN].twp_col_w;
        a.twp = (const cpx<T> *)(kind1 == 2 ? dt1->cfg[CFG_MAIN].twp_fs : w1 ? dt1->cfg[CFG_MAIN].twp_col_w : dt1->cfg[CFG_MAIN].twp_col); a.wide = w1 ? 1 : 0;
        // c128 (the lane-fastest kernels): the caller's array is read once -> streaming loads; the intermediate is re-read by pass 2 -> cache-allocating stores.
        // A-B-A-B (profiles/r08/r08s_fourstep_pass1_policy_abab.txt): 256 x 65536 203 -> 197 us, 16 x 2^20 260 -> 248 us; either one alone is neutral or worse
        // (keep alone: 213 us); c64 (staged kernels) 285 -> 291 us with the streaming loads: off there
        a.stream_in = (int)NDFFT_DEV_INT("NDFFT_FS_P1_NT", sizeof(T) == 8 ? 1 : 0); a.keep_out = (int)NDFFT_DEV_INT("NDFFT_FS_KEEP", sizeof(T) == 8 ? 1 : 0);
        if ((rc = kind1 == 2 ? launch_jit_fourstep<T>(1, inverse, sc1.fs_jitcfg, a, stream) : launch_fourstep<T>(1, F1, inverse, a, stream))) return rc;
        a.stream_in = 0; a.keep_out = 0;
        // pass 2: lanes (l, k1)
        a.in = s1; a.out = zout; a.nlanes = L * F1; a.n = F2; a.F = F2; a.n_in = F2; a.n_out = F2; a.scale = scale;
        a.inner = F1; a.outer_in = (int64_t)F2 * K1p; a.outer_out = pitch_out; a.elem_in = K1p; a.elem_out = F1; a.pitch_out = 0;
        a.twp = (const cpx<T> *)(kind2 == 2 ? dt2->cfg[CFG_MAIN].twp_fs : w2 ? dt2->cfg[CFG_MAIN].twp_col_w : dt2->cfg[CFG_MAIN].twp_col); a.wide = w2 ? 1 : 0;
        return kind2 == 2 ? launch_jit_fourstep<T>(2, inverse, sc2.fs_jitcfg, a, stream) : launch_fourstep<T>(2, F2, inverse, a, stream);
    }
    if ((rc = get_scratch(3, stream, (size_t)(L * F) * esz, &s2))) return rc;
    // Fused three-pass form when both halves run on the register kernels:
    //   (1) length-F1 FFTs IN PLACE of the layout, on the strided n1 axis, by the column-tile kernels
    //   (2) length-F2 row FFTs whose load multiplies by the four-step twiddle W_F^{n2 k1}
    //   (3) one transpose into natural order
    {
        const bool row2 = c.sub2->cfg[CFG_MAIN].pow2;
        const int col_lanes = std::is_same<T, float>::value ? pow2_real_col_lanes<float>(F1, 0) : pow2_real_col_lanes<double>(F1, 0);
        const int nar_lanes = std::is_same<T, float>::value ? pow2_real_narrow_lanes<float>(F1) : pow2_real_narrow_lanes<double>(F1);
        const bool col1 = !c.sub1->cfg[CFG_MAIN].twp_col.re.empty() && F2 >= 8 && col_lanes > 0;
        const bool nar1 = !c.sub1->cfg[CFG_MAIN].twp_narrow.re.empty() && F2 >= 64 && nar_lanes > 0 && narrow_enabled();
        if (row2 && (col1 || nar1)) {
            Problem Q;
            Q.plan = c.sub1; Q.op = inverse ? NDFFT_OP_C2C_INV : NDFFT_OP_C2C_FWD;
            Q.xlen = Q.ylen = F1; Q.xs = Q.ys = F2; Q.nlanes = L * F2; Q.scale = 1.0;
            if (L > 1) Q.b.push_back({L, pitch_in, F});
            Q.b.push_back({(int64_t)F2, 1, 1});
            if ((rc = dispatch(Q, zin, s1, stream))) return rc;
            const DevTables *dt2;
            if ((rc = get_dev_tables(c.sub2, &dt2))) return rc;
            Pow2Args a;
            a.in = s1; a.out = s2; a.nlanes = L * F1; a.pitch_in = F2; a.pitch_out = F2;
            a.inverse = inverse; a.scale = (double)scale; a.twp = dt2->cfg[CFG_MAIN].twp;
            a.twlo = d.twlo; a.twhi = d.twhi; a.logB = c.logB; a.f1 = F1;
            if ((rc = launch_pow2(c.sub2->dtype, F2, a, stream))) return rc;
            return transpose_batched(s2, zout, L, F1, F2, F2, F1, F, pitch_out, esz, stream);
        }
    }
    // x[n1][n2] -> s1[n2][n1]
    if ((rc = transpose_batched(zin, s1, L, F1, F2, F2, F1, pitch_in, F, esz, stream))) return rc;
    Problem Q;
    Q.plan = c.sub1; Q.op = inverse ? NDFFT_OP_C2C_INV : NDFFT_OP_C2C_FWD;
    Q.xlen = Q.ylen = F1; Q.xs = Q.ys = 1; Q.nlanes = L * F2; Q.scale = 1.0;
    Q.b.push_back({L * F2, (int64_t)F1, (int64_t)F1});
    if ((rc = dispatch(Q, s1, s2, stream))) return rc;
    if ((rc = launch_big_twiddle<T>((cpx<T> *)s2, L, F1, F2, (const cpx<T> *)d.twlo, (const cpx<T> *)d.twhi, c.logB, inverse ? 1 : 0, scale, stream))) return rc;
    // s2[n2][k1] -> s1[k1][n2]
    if ((rc = transpose_batched(s2, s1, L, F2, F1, F1, F2, F, F, esz, stream))) return rc;
    Q.plan = c.sub2; Q.xlen = Q.ylen = F2; Q.nlanes = L * F1;
    Q.b.clear(); Q.b.push_back({L * F1, (int64_t)F2, (int64_t)F2});
    if ((rc = dispatch(Q, s1, s2, stream))) return rc;
    // s2[k1][k2] -> out[k2][k1]  (flat index k1 + F1 k2)
    return transpose_batched(s2, zout, L, F1, F2, F2, F1, F, pitch_out, esz, stream);
}

static int gen_op_of(int op, int n, int *slot) {
    *slot = CFG_MAIN;
    switch (op) {
        case NDFFT_OP_C2C_FWD: return G_C2C_FWD;
        case NDFFT_OP_C2C_INV: return G_C2C_INV;
        case NDFFT_OP_R2C: return n % 2 ? G_R2C_ODD : G_R2C_EVEN;
        case NDFFT_OP_C2R: return n % 2 ? G_C2R_ODD : G_C2R_EVEN;
        case NDFFT_OP_DCT1: if (n == 1) return G_DCT2_ODD; *slot = CFG_DCT1; return G_DCT1;
        case NDFFT_OP_DCT2: return n % 2 ? G_DCT2_ODD : G_DCT2_EVEN;
        case NDFFT_OP_DCT3: return n % 2 ? G_DCT3_ODD : G_DCT3_EVEN;
        default: *slot = CFG_DCT4; return n % 2 ? G_DCT4_ODD : G_DCT4_EVEN;
    }
}

// REAL four-step for long contiguous real-data lanes, n = N1 * N2 a power of two (plan.hip: add_real_fourstep), R2C and DCT-II:
//   (1) real FFTs of length N1 over the strided index n1 of x[n1 N2 + n2] (DCT-II: of Makhoul's permutation of x, gathered by the load),
//       half spectrum stored transposed, s[lane][n2][k1], k1 = 0..N1/2
//   (2) complex FFTs of length N2 over n2 with the twiddle W_n^(n2 k1) fused into the load; X[k1 + N1 k2] and, for the other half of every
//       lane, conj at the mirrored index -- the half spectrum 0..n/2 exactly once; DCT-II multiplies by e^(-i pi k / 2n) and writes y[k], y[n-k]
// Two passes and an intermediate of n/2 + N2 complex per lane; the packed complex four-step needs a split pass (and a Makhoul pass) around its two.
// gop = G_DCT1 (round 5): the lane is the even extension of the caller's n = N1 N2 / 2 + 1 points (pass 1 gathers it: makhoul = 3) and pass 2 stores y[k] = Re X[k] / 2 times the
// pre-scale (src/lib.rs:688-698) for k = 0 .. n - 1 -- c / d are the DCT1 slot's tables then.
template <typename T>
static int real_fourstep(const Problem &P, int gop, const FftConfig &c, const DevConfig &d, const void *d_in, void *d_out, int64_t pin, int64_t pout, hipStream_t stream) {
    const int N1 = c.rfs_N1, N2 = c.rfs_N2;
    const bool dct1 = gop == G_DCT1;
    // k1 = 0..N1/2 of the intermediate on a pitch of whole 128-byte lines (the tiles of pass 2 are 128 bytes of adjacent k1 wide: with the
    // natural pitch N1/2 + 1 every tile row would straddle two lines shared with a tile on another XCD)
    const int K = (N1 / 2 + 1 + (int)(128 / sizeof(cpx<T>)) - 1) & ~((int)(128 / sizeof(cpx<T>)) - 1);
    const int64_t n = (int64_t)N1 * N2, B = P.nlanes;
    const DevTables *dt1, *dt2;
    int rc;
    // (round 6: a factor that is not a power of two runs its pass on a kernel specialised with hiprtc -- plan.hip: add_real_fourstep_smooth; NDFFT_ERR_UNSUPPORTED before
    //  anything is launched when that is not to be had: the caller falls back to the packed route)
    const int dti = sizeof(T) == 8 ? NDFFT_F64 : NDFFT_F32;
    const FftConfig &rc1 = c.rfs_sub1->cfg[CFG_MAIN], &rc2 = c.rfs_sub2->cfg[CFG_MAIN];
    const bool jit1 = !fourstep_supported(N1 / 2), jit2 = !fourstep_supported(N2);
    if ((jit1 && !(rc1.jit && jit_rfs1_ok(dti, rc1.jitcfg))) || (jit2 && !(rc2.fs_jit && jit_fourstep_ok(dti, rc2.fs_jitcfg)))) return NDFFT_ERR_UNSUPPORTED;
    if ((rc = get_dev_tables(c.rfs_sub1, &dt1)) || (rc = get_dev_tables(c.rfs_sub2, &dt2))) return rc;
    void *s1;
    if ((rc = get_scratch(4, stream, (size_t)(B * N2 * K) * sizeof(cpx<T>), &s1))) return rc;
    RealArgs<T> a;
    a.pitch_in = 0; a.vec_in = 0; a.vec_out = 0; a.xcd_remap = 0; a.keep_out = 0; a.stream_in = 0; a.chunk_out = 0;
    a.aux2 = nullptr; a.chirp = nullptr; a.bhat = nullptr;
    a.cs_twlo = (const cpx<T> *)d.rfs_twlo; a.cs_twhi = (const cpx<T> *)d.rfs_twhi; a.cs_logB = c.rfs_logB;
    a.cs_k1n = 1; a.cs_f1 = N1; a.cs_n = (int)n; a.cs_outer_in = 0; a.cs_outer_out = 0; a.cs_pitch = 0;
    // pass 1: lanes (l, n2), real input
    a.in = d_in; a.out = s1; a.nlanes = B * N2; a.n = N1; a.F = N1 / 2; a.n_in = N1; a.n_out = N1 / 2 + 1; a.scale = (T)1;
    a.inner = N2; a.outer_in = pin; a.outer_out = 0; a.elem_in = N2; a.elem_out = 0; a.pitch_out = K;
    a.aux1 = (const cpx<T> *)dt1->cfg[CFG_MAIN].aux1; a.twp = (const cpx<T> *)dt1->cfg[CFG_MAIN].twp;
    a.makhoul = gop == G_DCT2_EVEN ? 1 : dct1 ? 3 : 0;
    // streaming loads of the caller's lane in pass 1: R2C re-read 123.5 -> 118 us (HBM-sourced unchanged); not for DCT-II, whose mirror tiles share every line
    // (134 -> 144 us) -- profiles/r08/r08t_real_fourstep_policy_abab.txt
    // (round 6, A-B-A-B on the developer build, profiles/r09/r09l_rfs_p1_nt_abab.txt: with the lane coming from HBM streaming loads LOSE -- 122 against 119 us --, with a re-read,
    //  cache-resident lane they win -- 113 against 118 us: a resident lane read with nt loads is not re-allocated and leaves the cache to the intermediate.  So the residency
    //  model decides: streaming loads only for an input it expects IN the cache.  NDFFT_RFS_P1_NT = 0 / 1 (developer build) forces one form, 2 = the model.)
    {
        const long knob = NDFFT_DEV_INT("NDFFT_RFS_P1_NT", 2);
        const size_t es = sizeof(T);
        const bool resident = row_load_policy(d_in, (size_t)B * (dct1 ? (size_t)(n / 2 + 1) : (size_t)n) * es, d_out, (size_t)B * (size_t)(gop == G_R2C_EVEN ? (n / 2 + 1) * 2 : dct1 ? n / 2 + 1 : n) * es) == 0;
        a.stream_in = (gop == G_DCT2_EVEN || dct1) ? 0 : (knob == 2 ? (resident ? 1 : 0) : (int)knob);     // (DCT-I reads every element twice, through its own tile and the mirrored one)
    }
    if ((rc = jit1 ? launch_jit_fourstep<T>(11, false, rc1.jitcfg, a, stream) : launch_fourstep_real<T>(1, N1 / 2, a, stream))) return rc;
    a.stream_in = 0;
    // pass 2: lanes (l, k1)
    a.makhoul = dct1 ? 3 : 0;                        // (3: real outputs Re X[k] a.scale)
    a.keep_out = 1;                                  // plain stores at the lines the mirrored rows share
    a.xcd_chunk = (int)NDFFT_DEV_INT("NDFFT_RFS_XCD_CHUNK", 8);
    a.in = s1; a.out = d_out; a.nlanes = B * K; a.n = N2; a.F = N2; a.n_in = N2; a.n_out = N2; a.scale = dct1 ? (T)(0.5 * P.scale) : (T)P.scale;
    a.inner = K; a.outer_in = (int64_t)N2 * K; a.outer_out = pout; a.elem_in = K; a.elem_out = 0; a.pitch_out = 0;
    a.aux1 = nullptr; a.aux2 = (const cpx<T> *)d.aux2; a.twp = (const cpx<T> *)(jit2 ? dt2->cfg[CFG_MAIN].twp_fs : dt2->cfg[CFG_MAIN].twp_col);
    if (gop == G_DCT2_EVEN && NDFFT_DEV_INT("NDFFT_RFS_FACTORED", 1)) { a.fc1 = (const cpx<T> *)d.rfs_c1; a.fc2 = (const cpx<T> *)d.rfs_c2; }
    if (jit2) return launch_jit_fourstep<T>(gop == G_DCT2_EVEN ? 13 : 12, false, rc2.fs_jitcfg, a, stream);
    return launch_fourstep_real<T>(gop == G_DCT2_EVEN ? 3 : 2, N2, a, stream);
}

// The inverse direction of the real four-step, C2R and DCT-III:
//   (1) for k1 = 0..N1/2: the elements Xh[k1 + N1 k2] of the Hermitian extension (DCT-III: V[k] built from x[k], x[n-k]), unnormalised inverse FFT of
//       length N2 over k2, times W_n^(-n2 k1), stored as s[lane][k1][n2]
//   (2) column C2R of length N1 over k1 for every n2 (the ordinary column kernel: adjacent n2 contiguous on both sides) -> x[n1 N2 + n2]
//       (DCT-III: written through the inverse of Makhoul's permutation)
template <typename T>
static int real_fourstep_inv(const Problem &P, int gop, const FftConfig &c, const DevConfig &d, const void *d_in, void *d_out, int64_t pin, int64_t pout, hipStream_t stream) {
    const int N1 = c.rfs_N1, N2 = c.rfs_N2, Kx = N1 / 2 + 1;
    const int Kp = (Kx + (int)(128 / sizeof(cpx<T>)) - 1) & ~((int)(128 / sizeof(cpx<T>)) - 1);   // lanes per o, padded: every tile starts on a 128-byte line of the input
    const int64_t n = (int64_t)N1 * N2, B = P.nlanes;
    const DevTables *dt2;
    int rc;
    // (round 6: N2 not a power of two -> pass 1 on the hiprtc form of the lane-fastest kernel; N1 / 2 not a power of two -> pass 2 through dispatch(): the general column C2R kernel, hiprtc too)
    const int dti = sizeof(T) == 8 ? NDFFT_F64 : NDFFT_F32;
    const FftConfig &rc2 = c.rfs_sub2->cfg[CFG_MAIN];
    const bool jit2 = !fourstep_supported(N2), aot1 = fourstep_supported(N1 / 2);
    if (jit2 && !(rc2.fs_jit && jit_rfsi_ok(dti, rc2.fs_jitcfg))) return NDFFT_ERR_UNSUPPORTED;
    // DCT-III writes its outputs through the inverse of Makhoul's permutation in the LAST pass: only the column-tile kernels do that (RealArgs::makhoul), and dispatch() may
    // pick another kernel for a length that is not a power of two (a small call runs the generic kernel) -- so DCT-III needs the ahead-of-time last pass
    const FftConfig &rc1 = c.rfs_sub1->cfg[CFG_MAIN];
    const bool jit_last = gop == G_DCT3_EVEN && !aot1;       // ... or the hiprtc column tile of that length, launched HERE (not through dispatch(), which may pick another kernel for a small call)
    if (jit_last && !(rc1.jit && jit_col_lanes(dti, rc1.jitcfg, false) > 0)) return NDFFT_ERR_UNSUPPORTED;
    if ((rc = get_dev_tables(c.rfs_sub2, &dt2))) return rc;
    void *s1;
    // (no pitch padding here: with + 128 B per row ndifft_r2c 64 x 262144 f64 measured 112 -> 118 us, nddct3 unchanged -- profiles/r08/r08k_longlanes_pad.txt)
    const int64_t N2p = N2;                          // pitch of the intermediate s[k1][n2]
    if ((rc = get_scratch(4, stream, (size_t)(B * Kx * N2p) * sizeof(cpx<T>), &s1))) return rc;
    RealArgs<T> a;
    a.pitch_in = 0; a.vec_in = 0; a.vec_out = 0; a.xcd_remap = 0; a.keep_out = 0; a.stream_in = 0; a.chunk_out = 0; a.xcd_chunk = 0; a.makhoul = 0;
    a.aux1 = nullptr; a.aux2 = (const cpx<T> *)d.aux2; a.chirp = nullptr; a.bhat = nullptr;
    a.cs_twlo = (const cpx<T> *)d.rfs_twlo; a.cs_twhi = (const cpx<T> *)d.rfs_twhi; a.cs_logB = c.rfs_logB;
    a.cs_k1n = Kx; a.cs_f1 = N1; a.cs_n = (int)n; a.cs_outer_in = 0; a.cs_outer_out = 0; a.cs_pitch = 0;
    a.in = d_in; a.out = s1; a.nlanes = B * Kp; a.n = N2; a.F = N2; a.n_in = N2; a.n_out = N2; a.scale = (T)P.scale;
    a.inner = Kp; a.outer_in = pin; a.outer_out = 0; a.elem_in = N1; a.elem_out = 0; a.pitch_out = N2p;
    a.twp = (const cpx<T> *)(jit2 ? dt2->cfg[CFG_MAIN].twp_fs : dt2->cfg[CFG_MAIN].twp_col);
    // runs of consecutive tiles per XCD: the mirrored index N1 - k1 is shifted by one element against the tile grid (and DCT-III's real rows are
    // half lines), so neighbouring tiles share every line
    a.xcd_chunk = (int)NDFFT_DEV_INT("NDFFT_RFS_XCD_CHUNK", 8);
    a.stream_in = (int)NDFFT_DEV_INT("NDFFT_RFSI_P1_NT", 0);      // (measured: ndifft_r2c re-read 112 -> 125 us with streaming loads of the half spectrum: off)
    if (gop == G_DCT3_EVEN && NDFFT_DEV_INT("NDFFT_RFS_FACTORED", 1)) { a.fc1 = (const cpx<T> *)d.rfs_c1; a.fc2 = (const cpx<T> *)d.rfs_c2; }
    if ((rc = jit2 ? launch_jit_fourstep<T>(gop == G_DCT3_EVEN ? 15 : 14, false, rc2.fs_jitcfg, a, stream) : launch_fourstep_real<T>(gop == G_DCT3_EVEN ? 5 : 4, N2, a, stream))) return rc;
    a.stream_in = 0;
    if ((sw().rfs_c2r_tile && aot1) || jit_last) {   // the column C2R kernel on 128-byte tiles (0: the general column kernel through dispatch())
        const DevTables *dt1;
        if ((rc = get_dev_tables(c.rfs_sub1, &dt1))) return rc;
        a.xcd_chunk = 0; a.keep_out = 0; a.fc1 = nullptr; a.fc2 = nullptr;
        a.in = s1; a.out = d_out; a.nlanes = B * N2; a.n = N1; a.F = N1 / 2; a.n_in = Kx; a.n_out = N1; a.scale = (T)1;
        a.inner = N2; a.outer_in = (int64_t)Kx * N2p; a.outer_out = pout; a.elem_in = N2p; a.elem_out = N2; a.pitch_in = 0; a.pitch_out = 0;
        a.aux1 = (const cpx<T> *)dt1->cfg[CFG_MAIN].aux1; a.twp = (const cpx<T> *)dt1->cfg[CFG_MAIN].twp;
        a.makhoul = gop == G_DCT3_EVEN ? 1 : 0;
        if (jit_last) return launch_jit_real<T>(G_C2R_EVEN, rc1.jitcfg, true, a, stream);
        return launch_fourstep_real<T>(7, N1 / 2, a, stream);
    }
    Problem Q;
    Q.plan = c.rfs_sub1; Q.op = NDFFT_OP_C2R; Q.xlen = Kx; Q.ylen = N1; Q.xs = N2p; Q.ys = N2; Q.nlanes = B * N2; Q.scale = 1.0;
    if (B > 1) Q.b.push_back({B, (int64_t)Kx * N2p, pout});
    Q.b.push_back({(int64_t)N2, 1, 1});
    Q.no_xcd_map = 1; Q.makhoul_out = gop == G_DCT3_EVEN ? 1 : 0;
    return dispatch(Q, s1, d_out, stream);
}

// DCT-IV of a long even lane, n = 2 F, F = F1 * F2 powers of two: the complex four-step of length F with the fold z[j] = (x[2j] + i x[n-1-2j]) s w_j built by
// pass 1's load and the outputs y[2k] = Re(Z[k] c_k), y[n-1-2k] = -Im(Z[k] c_k) written by pass 2's store -- two passes instead of four.
template <typename T>
static int dct4_fourstep(const Problem &P, const FftConfig &c, const DevConfig &d, const void *d_in, void *d_out, int64_t pin, int64_t pout, hipStream_t stream, bool jit1 = false, bool jit2 = false) {
    const int F1 = c.F1, F2 = c.F2;
    const int64_t F = (int64_t)F1 * F2, B = P.nlanes;
    const DevTables *dt1, *dt2;
    int rc;
    if ((rc = get_dev_tables(c.sub1, &dt1)) || (rc = get_dev_tables(c.sub2, &dt2))) return rc;
    void *s1;
    const int64_t K1p = F1 + fs_pad_elems<T>();      // pitch of the intermediate s1[n2][k1] (fs_pad_elems)
    if ((rc = get_scratch(2, stream, (size_t)(B * F2 * K1p) * sizeof(cpx<T>), &s1))) return rc;
    RealArgs<T> a;
    a.pitch_in = 0; a.vec_in = 0; a.vec_out = 0; a.xcd_remap = 0; a.keep_out = 1; a.stream_in = 0; a.chunk_out = 0; a.xcd_chunk = 0;
    a.aux1 = (const cpx<T> *)d.aux1; a.aux2 = (const cpx<T> *)d.aux2; a.chirp = nullptr; a.bhat = nullptr;
    a.cs_twlo = (const cpx<T> *)d.twlo; a.cs_twhi = (const cpx<T> *)d.twhi; a.cs_logB = c.logB;
    a.cs_k1n = 1; a.cs_f1 = F1; a.cs_n = (int)(2 * F); a.cs_outer_in = 0; a.cs_outer_out = 0; a.cs_pitch = 0;
    // pass 1: lanes (l, n2) of the REAL input
    a.in = d_in; a.out = s1; a.nlanes = B * F2; a.n = F1; a.F = F1; a.n_in = F1; a.n_out = F1; a.scale = (T)P.scale;
    a.inner = F2; a.outer_in = pin; a.outer_out = 0; a.elem_in = F2; a.elem_out = 0; a.pitch_out = K1p;
    a.twp = (const cpx<T> *)(jit1 ? dt1->cfg[CFG_MAIN].twp_fs : dt1->cfg[CFG_MAIN].twp_col); a.makhoul = 2;
    // pass 1: streaming loads of the caller's lane, cache-allocating stores of the intermediate (the staged ROWOUT store ignored keep_out until round 5):
    // nddct4 64 x 262144 f64 157.6 -> 153.3 (stores) -> 148-150 us (both)
    a.stream_in = (int)NDFFT_DEV_INT("NDFFT_DCT4_P1_NT", 1); a.keep_out = (int)NDFFT_DEV_INT("NDFFT_DCT4_KEEP", 1);
    if ((rc = jit1 ? launch_jit_fourstep<T>(1, false, c.sub1->cfg[CFG_MAIN].fs_jitcfg, a, stream) : launch_fourstep<T>(1, F1, false, a, stream))) return rc;
    a.stream_in = 0;
    // pass 2: lanes (l, k1), real output
    a.makhoul = 0; a.keep_out = 0;
    a.in = s1; a.out = d_out; a.nlanes = B * F1; a.n = F2; a.F = F2; a.n_in = F2; a.n_out = F2; a.scale = (T)1;
    a.inner = F1; a.outer_in = (int64_t)F2 * K1p; a.outer_out = pout; a.elem_in = K1p; a.elem_out = 0; a.pitch_out = 0;
    a.twp = (const cpx<T> *)(jit2 ? dt2->cfg[CFG_MAIN].twp_fs : dt2->cfg[CFG_MAIN].twp_col);
    if (jit2) return launch_jit_fourstep<T>(16, false, c.sub2->cfg[CFG_MAIN].fs_jitcfg, a, stream);
    return launch_fourstep_real<T>(6, F2, a, stream);
}

// contiguous lanes whose inner FFT does not fit one workgroup's LDS
template <typename T>
static int dispatch_big(const Problem &P, const void *d_in, void *d_out, const DevTables &dt, hipStream_t stream) {
    const ndfft_plan *plan = P.plan;
    int slot;
    const int gop = gen_op_of(P.op, (int)plan->n, &slot);
    const FftConfig &c = plan->cfg[slot];
    const DevConfig &d = dt.cfg[slot];
    const int64_t pin = P.b.empty() ? P.xlen : P.b[0].sin, pout = P.b.empty() ? P.ylen : P.b[0].sout;
    if (gop == G_C2C_FWD || gop == G_C2C_INV) {
        int rc0 = big_fft<T>(c, d, (const cpx<T> *)d_in, pin, (cpx<T> *)d_out, pout, P.nlanes, gop == G_C2C_INV, (T)P.scale, stream);
        set_last_path(c.bigblue ? "blue_global" : "four_step");
        return rc0;
    }
    // (a plan whose factors are not powers of two has the forward ops only, and its passes need hiprtc: NDFFT_ERR_UNSUPPORTED from real_fourstep = nothing launched, packed route below)
    const bool rfs_smooth = c.rfs && (((c.rfs_N1 & (c.rfs_N1 - 1)) != 0) || ((c.rfs_N2 & (c.rfs_N2 - 1)) != 0));
    const int rfs_on = real_fourstep_enabled(), rfs_ops = (rfs_on == 2 ? (c.rfs_ops & 16 ? 16 : 15) : (rfs_on ? c.rfs_ops : 0)) & (rfs_smooth ? c.rfs_ops : 31);
    if (c.rfs && gop == G_DCT1 && (rfs_ops & 16)) {
        const int rc0 = real_fourstep<T>(P, gop, c, d, d_in, d_out, pin, pout, stream);
        if (!(rfs_smooth && rc0 == NDFFT_ERR_UNSUPPORTED)) { set_last_path("real_four_step"); return rc0; }
    }
    if (c.rfs && ((gop == G_DCT2_EVEN && (rfs_ops & 4)) || (gop == G_R2C_EVEN && P.scale == 1.0 && (rfs_ops & 1)))) {
        const int rc0 = real_fourstep<T>(P, gop, c, d, d_in, d_out, pin, pout, stream);
        if (!(rfs_smooth && rc0 == NDFFT_ERR_UNSUPPORTED)) { set_last_path("real_four_step"); return rc0; }
    }
    // (round 6: a factor that is not a power of two on the hiprtc forms of the same two kernels -- pass 1 the staged ROWOUT kernel, pass 2 the lane-fastest one, whole rounds)
    const int dti4 = sizeof(T) == 8 ? NDFFT_F64 : NDFFT_F32;
    auto d4_kind1 = [&]() { const FftConfig &s1c = c.sub1->cfg[CFG_MAIN]; return (fourstep_supported(c.F1) && !s1c.twp_col.re.empty()) ? 1 : (s1c.fs_jit && jit_fourstep_ok(dti4, s1c.fs_jitcfg)) ? 2 : 0; };
    auto d4_kind2 = [&]() { const FftConfig &s2c = c.sub2->cfg[CFG_MAIN]; return (fourstep_supported(c.F2) && !s2c.twp_col.re.empty()) ? 1 : (s2c.fs_jit && jit_rfsi_ok(dti4, s2c.fs_jitcfg)) ? 2 : 0; };
    if (gop == G_DCT4_EVEN && rfs_on && c.big && !c.bigblue && fourstep2_enabled() && d4_kind1() && d4_kind2()) {
        const int rc0 = dct4_fourstep<T>(P, c, d, d_in, d_out, pin, pout, stream, d4_kind1() == 2, d4_kind2() == 2);
        set_last_path("real_four_step");
        return rc0;
    }
    if (c.rfs && ((gop == G_C2R_EVEN && (rfs_ops & 2)) || (gop == G_DCT3_EVEN && (rfs_ops & 8)))) {
        const int rc0 = real_fourstep_inv<T>(P, gop, c, d, d_in, d_out, pin, pout, stream);
        if (!(rfs_smooth && rc0 == NDFFT_ERR_UNSUPPORTED)) { set_last_path("real_four_step"); return rc0; }
    }
    RealArgs<T> a;
    a.in = d_in; a.out = d_out; a.nlanes = P.nlanes; a.pitch_in = pin; a.pitch_out = pout;
    a.n = (int)plan->n; a.F = c.F; a.n_in = (int)P.xlen; a.n_out = (int)P.ylen; a.scale = (T)P.scale;
    a.aux1 = (const cpx<T> *)d.aux1; a.aux2 = (const cpx<T> *)d.aux2; a.twp = nullptr;
    void *z;
    int rc;
    if ((rc = get_scratch(4, stream, (size_t)(P.nlanes * c.F) * sizeof(cpx<T>), &z))) return rc;
    // Two of the ops need only ONE of the elementwise passes over global memory (round 3):
    //   R2C, even n: the PRE fold is z[i] = (x[2i], x[2i+1]) -- the raw real lane read as complex.  The FFT takes the input array itself.
    //   C2R, even n: the POST is x[2k] = Re, x[2k+1] = -Im of the forward FFT of conj(Zt) -- i.e. the inverse-by-conjugation of Zt written
    //                into the real output lane read as complex.  PRE emits Zt, the FFT runs as an inverse and stores the result itself.
    const bool cplx_view_ok = [&] {   // a real lane can be addressed as complex elements: even pitch, complex-aligned base
        const void *p = gop == G_R2C_EVEN ? d_in : (const void *)d_out;
        const int64_t pitch = gop == G_R2C_EVEN ? pin : pout;
        return pitch % 2 == 0 && (uintptr_t)p % sizeof(cpx<T>) == 0;
    }();
    if (gop == G_R2C_EVEN && cplx_view_ok) {
        if ((rc = big_fft<T>(c, d, (const cpx<T> *)d_in, pin / 2, (cpx<T> *)z, c.F, P.nlanes, false, (T)1, stream))) return rc;
        rc = launch_big_post<T>(gop, a, (const cpx<T> *)z, stream);
    } else if (gop == G_C2R_EVEN && cplx_view_ok) {
        if ((rc = launch_big_pre<T>(gop, a, (cpx<T> *)z, stream, 1))) return rc;
        rc = big_fft<T>(c, d, (const cpx<T> *)z, c.F, (cpx<T> *)d_out, pout / 2, P.nlanes, true, (T)1, stream);
    } else {
        if ((rc = launch_big_pre<T>(gop, a, (cpx<T> *)z, stream))) return rc;
        if ((rc = big_fft<T>(c, d, (const cpx<T> *)z, c.F, (cpx<T> *)z, c.F, P.nlanes, false, (T)1, stream))) return rc;
        rc = launch_big_post<T>(gop, a, (const cpx<T> *)z, stream);
    }
    set_last_path(c.bigblue ? "blue_global" : "four_step");
    return rc;
}

// Long lanes on a non-contiguous axis of a C-layout array, viewed as (outer, n, inner):
// transpose -> row transform on contiguous lanes -> transpose back.
static int dispatch_transposed(const Problem &P, const void *d_in, void *d_out, hipStream_t stream, int64_t outer,
                               int64_t inner, size_t ein, size_t eout) {
    const int64_t n_in = P.xlen, n_out = P.ylen;
    // scratch lanes are pitched to a multiple of 16 bytes so both transposes can use 16-byte accesses
    const int64_t p_in = (n_in + 3) & ~(int64_t)3, p_out = (n_out + 3) & ~(int64_t)3;
    void *s1, *s2;
    int rc;
    if ((rc = get_scratch(0, stream, (size_t)(outer * inner * p_in) * ein, &s1))) return rc;
    if ((rc = get_scratch(1, stream, (size_t)(outer * inner * p_out) * eout, &s2))) return rc;
    // in[o][j][i] -> s1[o][i][j]
    if ((rc = transpose_batched(d_in, s1, outer, n_in, inner, inner, p_in, n_in * inner, inner * p_in, (int)ein, stream))) return rc;
    Problem Q = P;
    Q.xs = Q.ys = 1;
    Q.b.clear();
    Q.b.push_back({outer * inner, p_in, p_out});
    if ((rc = dispatch(Q, s1, s2, stream))) return rc;
    // s2[o][i][k] -> out[o][k][i]
    return transpose_batched(s2, d_out, outer, inner, n_out, p_out, inner, inner * p_out, n_out * inner, (int)eout, stream);
}

static int dispatch(const Problem &P, const void *d_in, void *d_out, hipStream_t stream) {
    const DevTables *dt;
    int rc = get_dev_tables(P.plan, &dt);
    if (rc) return rc;
    const ndfft_plan *plan = P.plan;
    // short dense power-of-two C2C lanes (n = 2..64): the LDS-free wavefront kernel -- coalesced 16-byte accesses and
    // cross-lane shuffles (wave_kernel.h).  NDFFT_WAVE=0 keeps the older paths (parity tests cover both).
    if (plan->kind == NDFFT_KIND_C2C && wave_supported((int)plan->n) && dt->cfg[CFG_MAIN].wave_tw && P.xs == 1 && P.ys == 1 && P.b.size() <= 1 &&
        (P.b.empty() || (P.b[0].sin == (int64_t)plan->n && P.b[0].sout == (int64_t)plan->n)) &&
        (uintptr_t)d_in % 16 == 0 && (uintptr_t)d_out % 16 == 0 && wave_enabled()) {
        WaveArgs a;
        a.in = d_in; a.out = d_out; a.total = P.nlanes * (int64_t)plan->n;
        a.inverse = P.op == NDFFT_OP_C2C_INV; a.scale = P.scale; a.tw = dt->cfg[CFG_MAIN].wave_tw; a.xcd_chunk = 0;
        set_last_path("wave_reg");
        return launch_wave(plan->dtype, (int)plan->n, a, stream);
    }
    // very short C2C lanes (n = 2..13, 16) that the wavefront kernel did not take: one thread per lane (tiny_kernel.h).
    // Column layouts (adjacent lanes contiguous) are coalesced as they are; dense rows are staged through LDS.
    if (plan->kind == NDFFT_KIND_C2C && tiny_supported((int)plan->n) && P.b.size() <= 2 && tiny_enabled()) {
        const int n = (int)plan->n;
        const bool rows = P.xs == 1 && P.ys == 1 && P.b.size() <= 1;
        const bool dense = rows && (P.b.empty() || (P.b[0].sin == n && P.b[0].sout == n));
        const bool cols = !P.b.empty() && P.b.back().sin == 1 && P.b.back().sout == 1;
        {
            TinyArgs a;
            a.in = d_in; a.out = d_out; a.nlanes = P.nlanes; a.inverse = P.op == NDFFT_OP_C2C_INV; a.scale = P.scale;
            a.elem_in = P.xs; a.elem_out = P.ys;
            a.inner = P.b.empty() ? 1 : P.b.back().shape;
            a.lane_in = P.b.empty() ? 0 : P.b.back().sin; a.lane_out = P.b.empty() ? 0 : P.b.back().sout;
            a.outer_in = P.b.size() == 2 ? P.b[0].sin : 0; a.outer_out = P.b.size() == 2 ? P.b[0].sout : 0;
            const bool stage = dense;   // 256 lanes x (n | 1) elements of LDS: at most 68 KiB (n = 16, f64)
            set_last_path(stage ? "tiny_row" : cols ? "tiny_col" : "tiny_strided");
            return launch_tiny(plan->dtype, n, stage, a, stream);
        }
    }
    // C2C lanes of 14 .. 64 (f64) / 96 (f32) points that factor into two butterflies: one thread per lane, two passes in
    // registers (reg_kernel.h), specialised with hiprtc; only worth a compile when there is real work
    if (plan->kind == NDFFT_KIND_C2C && plan->n >= 14 && (int)plan->n <= regfft_max_n(plan->dtype) && dt->cfg[CFG_MAIN].wave_tw &&
        P.b.size() <= 2 && P.nlanes * (int64_t)plan->n >= (1 << 16) && tiny_enabled() && !tiny_supported((int)plan->n)) {
        int n1, n2;
        if (regfft_factor((int)plan->n, &n1, &n2)) {
            const int n = (int)plan->n;
            const bool rows = P.xs == 1 && P.ys == 1 && P.b.size() <= 1;
            const bool dense = rows && (P.b.empty() || (P.b[0].sin == n && P.b[0].sout == n));
            const bool cols = !P.b.empty() && P.b.back().sin == 1 && P.b.back().sout == 1;
            if ((dense && n <= 63) || (cols && !dense)) {          // rows beyond 63 points: the general register kernel is faster
                TinyArgs a;
                a.in = d_in; a.out = d_out; a.nlanes = P.nlanes; a.inverse = P.op == NDFFT_OP_C2C_INV; a.scale = P.scale;
                a.mat = dt->cfg[CFG_MAIN].wave_tw;
                a.elem_in = P.xs; a.elem_out = P.ys;
                a.inner = P.b.empty() ? 1 : P.b.back().shape;
                a.lane_in = P.b.empty() ? 0 : P.b.back().sin; a.lane_out = P.b.empty() ? 0 : P.b.back().sout;
                a.outer_in = P.b.size() == 2 ? P.b[0].sin : 0; a.outer_out = P.b.size() == 2 ? P.b[0].sout : 0;
                const int rcj = launch_jit_regfft(plan->dtype, n1, n2, dense, a, stream);
                if (rcj == NDFFT_OK) { set_last_path(dense ? "reg_row" : "reg_col"); return NDFFT_OK; }
                if (rcj != NDFFT_ERR_UNSUPPORTED) return rcj;
            }
        }
    }
    // the real-data transforms on lanes of 12 .. 48 (f64) / 72 (f32) points (from 12 up the butterflies beat the dense matrix of the tiny kernel: n = 16 0.49-0.59 -> see DESIGN 3.0c) whose inner FFT factors into butterflies:
    // one thread per lane, everything in registers (reg_kernel.h: RegReal), specialised with hiprtc
    if (plan->kind != NDFFT_KIND_C2C && plan->n >= 12 && (int)plan->n <= (plan->dtype == NDFFT_F32 ? regreal_max_n(0) : regreal_max_n(1)) &&
        P.b.size() <= 2 && P.nlanes * (int64_t)plan->n >= (1 << 16) && tiny_enabled()) {
        const int n = (int)plan->n;
        int slot;
        const int gop = gen_op_of(P.op, n, &slot);
        const FftConfig &c = plan->cfg[slot];
        const DevConfig &d = dt->cfg[slot];
        int f1, f2;
        if (!c.big && d.wave_tw && regfft_factor(c.F, &f1, &f2)) {   // (c.blue does not matter: primes 17..31 have their own butterfly here)
            const bool rows = P.xs == 1 && P.ys == 1 && P.b.size() <= 1;
            const bool dense = rows && (P.b.empty() || (P.b[0].sin == P.xlen && P.b[0].sout == P.ylen));
            const bool cols = !P.b.empty() && P.b.back().sin == 1 && P.b.back().sout == 1;
            if (dense || cols) {
                RegRealArgs ra;
                TinyArgs &a = ra.t;
                a.in = d_in; a.out = d_out; a.nlanes = P.nlanes; a.inverse = 0; a.scale = P.scale; a.mat = d.wave_tw;
                a.elem_in = P.xs; a.elem_out = P.ys;
                a.inner = P.b.empty() ? 1 : P.b.back().shape;
                a.lane_in = P.b.empty() ? 0 : P.b.back().sin; a.lane_out = P.b.empty() ? 0 : P.b.back().sout;
                a.outer_in = P.b.size() == 2 ? P.b[0].sin : 0; a.outer_out = P.b.size() == 2 ? P.b[0].sout : 0;
                ra.aux1 = d.aux1; ra.aux2 = d.aux2;
                const int rcj = launch_jit_regreal(plan->dtype, gop, n, f1, f2, dense, ra, stream);
                if (rcj == NDFFT_OK) { set_last_path(dense ? "regreal_row" : "regreal_col"); return NDFFT_OK; }
                if (rcj != NDFFT_ERR_UNSUPPORTED) return rcj;
            }
        }
    }
    // the real-data transforms on very short lanes (n = 2..16): one thread per lane, the transform as a dense matrix
    if (plan->kind != NDFFT_KIND_C2C && plan->n >= 2 && plan->n <= 16 && P.b.size() <= 2 && tiny_enabled()) {
        const int n = (int)plan->n;
        const int q = plan->kind == NDFFT_KIND_R2C ? (P.op == NDFFT_OP_R2C ? 0 : 1) : P.op - NDFFT_OP_DCT1;
        const void *mat = dt->cfg[CFG_MAIN].tinymat[q];
        if (mat) {
            const bool rows = P.xs == 1 && P.ys == 1 && P.b.size() <= 1;
            const bool dense = rows && (P.b.empty() || (P.b[0].sin == P.xlen && P.b[0].sout == P.ylen));
            const bool cols = !P.b.empty() && P.b.back().sin == 1 && P.b.back().sout == 1;
            TinyArgs a;
            a.in = d_in; a.out = d_out; a.nlanes = P.nlanes; a.inverse = 0; a.scale = P.scale; a.mat = mat;
            a.elem_in = P.xs; a.elem_out = P.ys;
            a.inner = P.b.empty() ? 1 : P.b.back().shape;
            a.lane_in = P.b.empty() ? 0 : P.b.back().sin; a.lane_out = P.b.empty() ? 0 : P.b.back().sout;
            a.outer_in = P.b.size() == 2 ? P.b[0].sin : 0; a.outer_out = P.b.size() == 2 ? P.b[0].sout : 0;
            set_last_path(dense ? "tinymat_row" : cols ? "tinymat_col" : "tinymat_strided");
            const int shape = plan->kind == NDFFT_KIND_R2C ? (P.op == NDFFT_OP_R2C ? 0 : 1) : 2;
            return plan->dtype == NDFFT_F32 ? launch_tinymat_f32(n, shape, dense, a, stream) : launch_tinymat_f64(n, shape, dense, a, stream);
        }
    }
    // tuned path: contiguous power-of-two C2C lanes at a uniform pitch
    if (plan->kind == NDFFT_KIND_C2C && plan->cfg[CFG_MAIN].pow2 && P.xs == 1 && P.ys == 1 && P.b.size() <= 1) {
        Pow2Args a;
        a.in = d_in; a.out = d_out; a.nlanes = P.nlanes;
        a.pitch_in = P.b.empty() ? (int64_t)plan->n : P.b[0].sin;
        a.pitch_out = P.b.empty() ? (int64_t)plan->n : P.b[0].sout;
        a.inverse = P.op == NDFFT_OP_C2C_INV;
        a.scale = P.scale;
        a.twp = dt->cfg[CFG_MAIN].twp;
        if (a.pitch_in == (int64_t)plan->n && a.pitch_out == (int64_t)plan->n)
            a.stream_in = c2c_row_load_policy(d_in, d_out, (size_t)P.nlanes * plan->n * 2 * real_size(plan->dtype));
        set_last_path("pow2_reg");
        return launch_pow2(plan->dtype, (int)plan->n, a, stream);
    }
    // smooth non-power-of-two C2C lanes: the same register-resident kernel, specialised at first use (jit.hip);
    // only worth a compile when there is real work
    if (plan->kind == NDFFT_KIND_C2C && plan->cfg[CFG_MAIN].jit && P.xs == 1 && P.ys == 1 && P.b.size() <= 1 &&
        P.nlanes * (int64_t)plan->n >= (1 << 17)) {
        Pow2Args a;
        a.in = d_in; a.out = d_out; a.nlanes = P.nlanes;
        a.pitch_in = P.b.empty() ? (int64_t)plan->n : P.b[0].sin;
        a.pitch_out = P.b.empty() ? (int64_t)plan->n : P.b[0].sout;
        a.inverse = P.op == NDFFT_OP_C2C_INV;
        a.scale = P.scale;
        a.twp = dt->cfg[CFG_MAIN].twp;
        const size_t bytes_c = (size_t)P.nlanes * plan->n * 2 * real_size(plan->dtype);
        const int pol = (a.pitch_in == (int64_t)plan->n && a.pitch_out == (int64_t)plan->n) ? c2c_row_load_policy(d_in, d_out, bytes_c) : -1;
        const int nt = (pol >= 0 ? pol != 0 : stream_loads_for(bytes_c)) ? 3 : 1;
        const int rcj = launch_jit_c2c(plan->dtype, plan->cfg[CFG_MAIN].jitcfg, nt, a, stream);
        if (rcj == NDFFT_OK) { set_last_path("jit_reg"); return NDFFT_OK; }
        if (rcj != NDFFT_ERR_UNSUPPORTED) return rcj;   // a real HIP error; UNSUPPORTED = no hiprtc / compile failed -> LDS kernel
        if (sw().jit_verbose) fprintf(stderr, "ndfft: jit_reg declined n = %zu\n", plan->n);
    }
    // tuned paths on the register-resident real-op engine (pow2_real.h), power-of-two inner FFT:
    //   row: R2C / C2R / DCT on contiguous lanes;  col: the same ops AND C2C on a strided axis whose
    //   adjacent lanes are contiguous (strategy ii), through an LDS tile of adjacent lanes
    {
        const int n = (int)plan->n;
        int slot;
        const int gop = gen_op_of(P.op, n, &slot);
        const FftConfig &c = plan->cfg[slot];
        const DevConfig &d = dt->cfg[slot];
        const bool is_c2c = plan->kind == NDFFT_KIND_C2C;
        const bool odd_variant = gop == G_R2C_ODD || gop == G_C2R_ODD || gop == G_DCT2_ODD || gop == G_DCT3_ODD || gop == G_DCT4_ODD;
        // (2^15 points since round 5: ndfft_r2c axis 0 of the reference's 264 x 264 bench shape is 264 lanes x 132 points -- generic_col 11.6 us, jit_col 6-7 us in a graph;
        //  its code object ships in jit_prebuilt/)
        const bool use_jit = c.jit && !c.pow2 && P.nlanes * (int64_t)c.F >= (1 << 15);
        // Bluestein lengths on the register kernel (blue_kernel.h), every op incl. the odd-n variants and row C2C
        // Rader / Good-Thomas (rader_kernel.h) wherever the plan has a recipe -- also for lanes beyond Bluestein's single-launch reach
        // (F > 4096: M' = 2^k >= 2F - 1 no longer fits, Rader's F complex elements of LDS do); Bluestein stays the fallback where it exists
        // (no length rule any more: with the short-lane recipe weights of jit.hip a scan of n = 34..260 has nddct2 on Rader at a median 1.59x over Bluestein with two lengths
        //  7 % slower, C2C at 1.7x with none, profiles/r04/r04zd_rader_short_real.txt)
        const bool use_rader = c.rader && P.nlanes * (int64_t)c.F >= (1 << 16) && blue_enabled();
        const bool use_blue = use_rader || (c.bluereg && ((P.nlanes * (int64_t)c.M >= (1 << 16) && blue_enabled()) || c.blue_reg_only));
        const bool use_plain = odd_variant && use_jit && !use_blue && plain_enabled();         // odd-n real ops with a smooth inner FFT: plain_kernel.h
        const bool have_tw = use_jit || use_blue || (is_c2c ? !c.twp_col.re.empty() : c.pow2);
        // column tiles of a C2C plan may have their own recipe (FftConfig::jit_col_alt)
        const bool col_alt = use_jit && !use_blue && !use_plain && is_c2c && c.jit_col_alt;
        const JitCfg &jcol = col_alt ? c.jitcfg_col : c.jitcfg;
        const bool row = (!is_c2c || use_blue) && P.xs == 1 && P.ys == 1 && P.b.size() <= 1;
        bool col = false, narrow = false;
        const int col_kind = is_c2c ? 0 : (P.op == NDFFT_OP_R2C ? 1 : (P.op == NDFFT_OP_C2R ? 2 : 3));   // which sides of a column tile are real lanes (kernels_pow2_real.hip: ColGeom)
        if (!row && (!odd_variant || use_blue || use_plain) && P.xlen > 1 && !P.b.empty() && P.b.size() <= 2 && P.b.back().sin == 1 && P.b.back().sout == 1) {
            if (have_tw && P.b.back().shape >= 8) {
                const int lanes = (use_jit || use_blue) ? std::max((use_jit || c.bluereg) ? jit_col_lanes(plan->dtype, use_blue ? c.jitcfg : jcol, is_c2c && use_jit && !use_blue) : 0, use_rader ? rader_col_lanes(plan->dtype, c.radercfg) : 0)
                                          : plan->dtype == NDFFT_F32 ? pow2_real_col_lanes<float>(c.F, col_kind) : pow2_real_col_lanes<double>(c.F, col_kind);
                col = lanes > 0;
            }
            // long lanes: XCD-aware narrow tiles (one HBM pass) instead of the three-pass transpose route
            // (not for the DCTs since round 3: transpose -> row kernel -> transpose measured faster -- nddct2 axis 0 of 4096 x 4096 / 8192 x 2048 f32 166 / 172 -> 100 / 103 us,
            //  f64 174 / 283 -> 164 / 190 us, profiles/r06/r06z_*; C2R f64 n = 4096 stays: 126 vs 174 us)
            if (!col && (col_kind != 3 || narrow_dct_enabled()) && !c.twp_narrow.re.empty() && P.b.back().shape >= 64 && narrow_enabled()) {
                const int lanes = plan->dtype == NDFFT_F32 ? pow2_real_narrow_lanes<float>(c.F) : pow2_real_narrow_lanes<double>(c.F);
                narrow = lanes > 0;
            }
        }
        // long strided power-of-two lanes on a dense C-layout block: column four-step (two wide-tile passes)
        if (c.cs && slot == CFG_MAIN && colsplit_enabled() && !row && !P.b.empty() && P.b.size() <= 2 && P.b.back().sin == 1 &&
            P.b.back().sout == 1 && P.xs == P.b.back().shape && P.ys == P.b.back().shape && P.b.back().shape >= 16 &&
            (((P.op == NDFFT_OP_C2C_FWD || P.op == NDFFT_OP_C2C_INV) && (c.cs_ops & 1)) || (P.op == NDFFT_OP_R2C && (c.cs_ops & 2)) || (P.op == NDFFT_OP_C2R && (c.cs_ops & 4)))) {
            return plan->dtype == NDFFT_F32 ? col_split<float>(P, d_in, d_out, c, d, stream) : col_split<double>(P, d_in, d_out, c, d, stream);
        }
        if (narrow) {
            auto fill = [&](auto &a) {
                a.in = d_in; a.out = d_out; a.nlanes = P.nlanes;
                a.pitch_in = 0; a.pitch_out = 0; a.vec_in = 0; a.vec_out = 0;
                a.xcd_remap = 1;
                a.keep_out = 0;
                a.n = n; a.F = c.F; a.n_in = (int)P.xlen; a.n_out = (int)P.ylen;
                a.inner = P.b.back().shape;
                a.outer_in = P.b.size() == 2 ? P.b[0].sin : 0;
                a.outer_out = P.b.size() == 2 ? P.b[0].sout : 0;
                a.elem_in = P.xs; a.elem_out = P.ys;
            };
            int rc2;
            if (plan->dtype == NDFFT_F32) {
                RealArgs<float> a; fill(a); a.scale = (float)P.scale;
                a.aux1 = (const cpx<float> *)d.aux1; a.aux2 = (const cpx<float> *)d.aux2; a.twp = (const cpx<float> *)d.twp_narrow;
                rc2 = launch_pow2_real_narrow<float>(gop, a, stream);
            } else {
                RealArgs<double> a; fill(a); a.scale = P.scale;
                a.aux1 = (const double2 *)d.aux1; a.aux2 = (const double2 *)d.aux2; a.twp = (const double2 *)d.twp_narrow;
                rc2 = launch_pow2_real_narrow<double>(gop, a, stream);
            }
            set_last_path("pow2_col_xcd");
            return rc2;
        }
        if (have_tw && (!odd_variant || use_blue || use_plain) && (row || col)) {
            auto fill = [&](auto &a) {
                a.in = d_in; a.out = d_out; a.nlanes = P.nlanes;
                a.pitch_in = P.b.empty() ? P.xlen : P.b[0].sin;
                a.pitch_out = P.b.empty() ? P.ylen : P.b[0].sout;
                a.n = n; a.F = c.F; a.n_in = (int)P.xlen; a.n_out = (int)P.ylen;
                a.inner = col ? P.b.back().shape : 1;
                a.outer_in = col && P.b.size() == 2 ? P.b[0].sin : 0;
                a.outer_out = col && P.b.size() == 2 ? P.b[0].sout : 0;
                a.elem_in = P.xs; a.elem_out = P.ys;
                const size_t es_in = (op_in_cplx(P.op) ? 2 : 1) * real_size(plan->dtype);
                a.vec_in = !col && ((uintptr_t)d_in % 16 == 0) && ((size_t)a.pitch_in * es_in) % 16 == 0;
                a.xcd_remap = 0; a.keep_out = P.keep_out; a.stream_in = P.stream_in; a.xcd_chunk = P.no_xcd_map ? 0 : -1;
                a.makhoul = col ? P.makhoul_out : 0;
                const size_t es_out = (op_out_cplx(P.op) ? 2 : 1) * real_size(plan->dtype);
                // dense rows of the ahead-of-time real-op kernels (BASELINE configs[3]): load policy from the Infinity-Cache model, as for the C2C rows
                // (round 5: also the real-input rows of the Rader kernel)
                if (!col && ((!use_jit && !use_blue && !use_plain) || (use_rader && !op_in_cplx(P.op))) && a.pitch_in == P.xlen && a.pitch_out == P.ylen && F_nt_ok(c.F))
                    a.stream_in = row_load_policy(d_in, (size_t)P.nlanes * P.xlen * es_in, d_out, (size_t)P.nlanes * P.ylen * es_out) == 1;
                // column tiles of a caller's array (not the stages of col_split / the four-step, which set their own policy): the same model
                if (col && !P.no_xcd_map && !P.stream_in && !P.keep_out)
                    a.stream_in = row_load_policy(d_in, (size_t)P.nlanes * P.xlen * es_in, d_out, (size_t)P.nlanes * P.ylen * es_out) == 1;
                a.vec_out = !col && ((uintptr_t)d_out % 16 == 0) && ((size_t)a.pitch_out * es_out) % 16 == 0;
                // R2C rows with dense output lanes: the workgroup stores its lanes as one contiguous chunk (pow2_real.h: chunk_out)
                // Measured (profiles/r03j, 2^24 points f32): n = 96 / 100 48 -> 37 / 32 -> 30 us, powers of two 128..1024 +4 %; n = 500 / 1000 and
                // n >= 2048 lose 2-8 % (fewer, longer lanes per workgroup: the per-lane stores are already long runs), so short lanes only.
                const bool short_lane = c.F <= 64 || (c.F <= 512 && (c.F & (c.F - 1)) == 0);
                a.chunk_out = !col && gop == G_R2C_EVEN && short_lane && ((uintptr_t)d_out % 16 == 0) && a.pitch_out == P.ylen && chunk_out_enabled();
                if (a.chunk_out) a.xcd_chunk = 0;
            };
            int rc2;
            if (plan->dtype == NDFFT_F32) {
                RealArgs<float> a; fill(a); a.scale = (float)P.scale;
                a.aux1 = (const cpx<float> *)d.aux1; a.aux2 = (const cpx<float> *)d.aux2; a.twp = (const cpx<float> *)((is_c2c && !use_jit && !use_blue) ? d.twp_col : d.twp);
                a.chirp = (const cpx<float> *)d.chirp; a.bhat = (const cpx<float> *)d.bhat; a.twp_rev = (const cpx<float> *)d.twp_rev;
                if (use_rader) {
                    RealArgs<float> r = a;
                    r.twp = (const cpx<float> *)d.rader_twp; r.twp_rev = (const cpx<float> *)d.rader_twp2; r.chirp = (const cpx<float> *)d.rader_ctw; r.bhat = (const cpx<float> *)d.rader_bhat; r.rader_tab = (const int32_t *)d.rader_tab;
                    const int rr = launch_jit_rader<float>(gop, c.radercfg, col, r, stream);
                    if (rr != NDFFT_ERR_UNSUPPORTED) { set_last_path(col ? "rader_col" : "rader_reg"); return rr; }
                }
                rc2 = use_blue ? (c.bluereg ? launch_jit_blue<float>(gop, c.jitcfg, col, a, stream) : NDFFT_ERR_UNSUPPORTED)
                      : use_plain ? launch_jit_plain<float>(gop, c.jitcfg, col, a, stream)
                      : use_jit ? ((col && col_alt) ? (a.twp = (const cpx<float> *)d.twp_jcol, launch_jit_real<float>(gop, jcol, col, a, stream)) : launch_jit_real<float>(gop, c.jitcfg, col, a, stream))
                      : launch_pow2_real<float>(gop, a, col, stream);
            } else {
                RealArgs<double> a; fill(a); a.scale = P.scale;
                a.aux1 = (const double2 *)d.aux1; a.aux2 = (const double2 *)d.aux2; a.twp = (const double2 *)((is_c2c && !use_jit && !use_blue) ? d.twp_col : d.twp);
                a.chirp = (const double2 *)d.chirp; a.bhat = (const double2 *)d.bhat; a.twp_rev = (const double2 *)d.twp_rev;
                if (use_rader) {
                    RealArgs<double> r = a;
                    r.twp = (const double2 *)d.rader_twp; r.twp_rev = (const double2 *)d.rader_twp2; r.chirp = (const double2 *)d.rader_ctw; r.bhat = (const double2 *)d.rader_bhat; r.rader_tab = (const int32_t *)d.rader_tab;
                    const int rr = launch_jit_rader<double>(gop, c.radercfg, col, r, stream);
                    if (rr != NDFFT_ERR_UNSUPPORTED) { set_last_path(col ? "rader_col" : "rader_reg"); return rr; }
                }
                rc2 = use_blue ? (c.bluereg ? launch_jit_blue<double>(gop, c.jitcfg, col, a, stream) : NDFFT_ERR_UNSUPPORTED)
                      : use_plain ? launch_jit_plain<double>(gop, c.jitcfg, col, a, stream)
                      : use_jit ? ((col && col_alt) ? (a.twp = (const double2 *)d.twp_jcol, launch_jit_real<double>(gop, jcol, col, a, stream)) : launch_jit_real<double>(gop, c.jitcfg, col, a, stream))
                      : launch_pow2_real<double>(gop, a, col, stream);
            }
            if (!((use_jit || use_blue) && rc2 == NDFFT_ERR_UNSUPPORTED)) {   // UNSUPPORTED from the JIT = no hiprtc / compile failed: fall through to the LDS kernel
                set_last_path(use_blue ? (col ? "blue_col" : "blue_reg") : use_plain ? (col ? "plain_col" : "plain_real") : use_jit ? (col ? "jit_col" : "jit_real") : (col ? "pow2_col" : "pow2_real"));
                return rc2;
            }
        }
    }
    {   // long lanes: four-step on the row kernels (contiguous lanes) -- strided ones reach here via the transpose route
        int slot;
        (void)gen_op_of(P.op, (int)plan->n, &slot);
        const FftConfig &c = plan->cfg[slot];
        if (c.unsupported)
            return fail(NDFFT_ERR_UNSUPPORTED, "lane length has a prime factor too large for the single-launch Bluestein and no usable "
                                               "four-step split (DESIGN.md section 9)");
        if (c.big && P.xs == 1 && P.ys == 1 && P.b.size() <= 1)
            return plan->dtype == NDFFT_F32 ? dispatch_big<float>(P, d_in, d_out, *dt, stream) : dispatch_big<double>(P, d_in, d_out, *dt, stream);
    }
    // strided axis of a C-layout array whose lanes are too long for a useful LDS tile of adjacent
    // lanes (< 128 B contiguous per tile row): go through the batched transpose
    if (P.xs != 1 && P.ys != 1 && P.xlen > 1 && !P.b.empty() && P.b.back().sin == 1 && P.b.back().sout == 1 && P.b.size() <= 2) {
        const int64_t inner = P.b.back().shape;
        const int64_t outer = P.b.size() == 2 ? P.b[0].shape : 1;
        const size_t r = real_size(plan->dtype);
        const size_t ein = op_in_cplx(P.op) ? 2 * r : r, eout = op_out_cplx(P.op) ? 2 * r : r;
        const bool c_layout = P.xs == inner && P.ys == inner &&
                              (P.b.size() == 1 || (P.b[0].sin == P.xlen * inner && P.b[0].sout == P.ylen * inner));
        // LDS bytes one lane needs in the generic kernel (two padded complex buffers)
        const int slot = P.op == NDFFT_OP_DCT1 && plan->n > 1 ? CFG_DCT1 : P.op == NDFFT_OP_DCT4 ? CFG_DCT4 : CFG_MAIN;
        const FftConfig &c = plan->cfg[slot];
        const size_t per_lane = 2 * (size_t)generic_z_len(std::max(c.blue ? c.M : c.F, 1)) * 2 * r;
        const size_t fit = c.big ? 0 : (160 * 1024 - 2048) / std::max<size_t>(per_lane, 1);
        const size_t row_bytes = std::min<size_t>(fit, (size_t)inner) * std::min(ein, eout);
        if (c_layout && row_bytes < 128 && (inner >= 16 || c.big)) {
            int rc2 = dispatch_transposed(P, d_in, d_out, stream, outer, inner, ein, eout);
            if (!rc2) {
                static thread_local std::string path;
                path = std::string("transpose+") + last_path();
                set_last_path(path.c_str());
            }
            return rc2;
        }
    }
    {
        int slot;
        (void)gen_op_of(P.op, (int)plan->n, &slot);
        if (plan->cfg[slot].big) {
            // long lanes in an arbitrary strided layout: pack -> row path on dense lanes -> unpack
            const size_t r = real_size(plan->dtype);
            const size_t ein = op_in_cplx(P.op) ? 2 * r : r, eout = op_out_cplx(P.op) ? 2 * r : r;
            void *s1, *s2;
            int rc2;
            if ((rc2 = get_scratch(0, stream, (size_t)(P.nlanes * P.xlen) * ein, &s1))) return rc2;
            if ((rc2 = get_scratch(1, stream, (size_t)(P.nlanes * P.ylen) * eout, &s2))) return rc2;
            LaneGeom gi, go;
            gi.axis_stride = P.xs; go.axis_stride = P.ys; gi.nb = go.nb = (int32_t)P.b.size(); gi.pad_ = go.pad_ = 0;
            for (size_t k = 0; k < P.b.size(); ++k) { gi.bshape[k] = go.bshape[k] = P.b[k].shape; gi.bstride[k] = P.b[k].sin; go.bstride[k] = P.b[k].sout; }
            if ((rc2 = launch_pack_lanes(d_in, s1, gi, P.nlanes, P.xlen, P.xlen, (int)ein, 0, stream))) return rc2;
            Problem Q = P;
            Q.xs = Q.ys = 1; Q.b.clear(); Q.b.push_back({P.nlanes, P.xlen, P.ylen});
            if ((rc2 = dispatch(Q, s1, s2, stream))) return rc2;
            if ((rc2 = launch_pack_lanes(d_out, s2, go, P.nlanes, P.ylen, P.ylen, (int)eout, 1, stream))) return rc2;
            static thread_local std::string path;
            path = std::string("pack+") + last_path();
            set_last_path(path.c_str());
            return NDFFT_OK;
        }
    }
    return plan->dtype == NDFFT_F32 ? dispatch_generic<float>(P, d_in, d_out, *dt, stream)
                                    : dispatch_generic<double>(P, d_in, d_out, *dt, stream);
}

// more than kMaxBatchDims un-mergeable batch dims: peel the slowest ones on the host
static int dispatch_peeled(Problem &P, const char *d_in, char *d_out, size_t ein, size_t eout, hipStream_t stream) {
    if (P.b.size() <= (size_t)kMaxBatchDims) return dispatch(P, d_in, d_out, stream);
    BatchDim outer = P.b.front();
    Problem Q = P;
    Q.b.erase(Q.b.begin());
    Q.nlanes = P.nlanes / outer.shape;
    for (int64_t i = 0; i < outer.shape; ++i) {
        int rc = dispatch_peeled(Q, d_in + i * outer.sin * (int64_t)ein, d_out + i * outer.sout * (int64_t)eout, ein, eout, stream);
        if (rc) return rc;
    }
    return NDFFT_OK;
}

// element range [lo, hi] (inclusive, relative to element 0) touched by a view
static void view_range(int ndim, const int64_t *shape, const int64_t *stride, int64_t &lo, int64_t &hi, int64_t &count) {
    lo = hi = 0; count = 1;
    for (int d = 0; d < ndim; ++d) {
        count *= shape[d];
        if (shape[d] <= 0) continue;
        const int64_t ext = (shape[d] - 1) * stride[d];
        if (ext < 0) lo += ext; else hi += ext;
    }
}

// Copies exactly the elements of an n-d view between two byte images of the same address range (`dst` and `src`
// both point at the image of element 0).  Used for output views with holes: the device result comes back as an
// image of the view's whole span, and only the elements the view OWNS may be written to the caller's memory --
// Rust's `&mut ArrayViewMut` guarantees exclusivity of those elements only (two threads may hold interleaved
// views of one allocation, e.g. even / odd columns from multi_slice_mut).
static void copy_view_elements(char *dst, const char *src, int ndim, const int64_t *shape, const int64_t *stride, size_t esz) {
    struct D { int64_t n, s; };
    std::vector<D> d;
    for (int k = 0; k < ndim; ++k) {
        if (shape[k] == 0) return;
        if (shape[k] > 1 && stride[k] != 0) d.push_back({shape[k], stride[k]});
    }
    std::sort(d.begin(), d.end(), [](const D &a, const D &b) { return std::llabs(a.s) < std::llabs(b.s); });
    // innermost contiguous run (|stride| == 1), merged with outer dims that continue it
    int64_t run = 1, run_off = 0;   // run_off: offset of the run's lowest element relative to the index-0 element
    size_t first = 0;
    if (!d.empty() && std::llabs(d[0].s) == 1) {
        run = d[0].n; run_off = d[0].s < 0 ? -(d[0].n - 1) : 0; first = 1;
        while (first < d.size() && d[first].s == run && run_off == 0) { run *= d[first].n; ++first; }
    }
    std::vector<D> o(d.begin() + first, d.end());
    std::vector<int64_t> idx(o.size(), 0);
    int64_t off = 0;
    for (;;) {
        memcpy(dst + (off + run_off) * (int64_t)esz, src + (off + run_off) * (int64_t)esz, (size_t)run * esz);
        size_t k = 0;
        for (; k < o.size(); ++k) {
            off += o[k].s;
            if (++idx[k] < o[k].n) break;
            off -= o[k].s * o[k].n; idx[k] = 0;
        }
        if (k == o.size()) break;
    }
}

}  // namespace ndfft

using namespace ndfft;

namespace {
// ---- pinned host arrays: H2D || kernel || D2H over row chunks --------------------------------------------
// Pageable host memory is staged by the runtime and the two PCIe directions do not overlap (tools/h2d_bench.hip:
// 9.7 ms for 2 x 256 MiB whatever the threading); arrays allocated with ndfft_host_alloc are pinned, their copies
// are real DMA and the directions overlap (5.6 ms).  A dense C-layout call whose slowest dimension is a batch
// dimension is therefore split into row chunks: chunk c+1 uploads while chunk c transforms and chunk c-1 downloads.
bool is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// ---- registration cache for the caller's own (pageable) arrays ---------------------------------------------------------
// The reference's signature hands over ndarrays in ordinary host memory (src/lib.rs:105-115).  Pinned memory moves at the PCIe duplex
// rate (2 x 256 MiB: 6.2 ms through the chunk pipeline) but hipHostRegister costs ~22 ms per 512 MiB, pageable memory goes through bounce
// buffers (8-10 ms: the host memcpys bound it on a 16-CPU quota).  A caller that transforms the SAME arrays again and again -- a time
// stepper, the reference's own benches -- should pay the registration once: the SECOND time a range is seen it is registered and kept in an
// LRU, from then on its calls run the pinned pipeline.  One-shot arrays never pay.
// OPT-IN (ndfft_host_reg_cache / NDFFT_HOST_REG_CACHE_MB, default 0 = off), because a registration outlives the array: when the caller
// frees a registered array and the allocator hands the addresses out again, HIP still treats them as the old pinned object and EVERY copy
// from or to them -- this library's, torch's, the caller's own -- fails with "invalid argument" or aborts inside the HIP runtime (both seen
// on the MI355X with numpy arrays in the first version, which had the cache on by default).  This library recovers from the error form
// (it forgets the range and retries through the bounce buffers) but cannot protect other code nor survive the abort, so only a caller that
// owns its arrays' lifetimes should switch it on, and it must call ndfft_host_forget before freeing them.
class HostRegCache {
  public:
    static HostRegCache &get() { static HostRegCache *c = new HostRegCache; return *c; }
    // true: [p, p + bytes) lies inside a registration THIS CACHE owns, and is held until release(): its LRU stamp is fresh and neither
    // eviction, forget() nor set_limit(0) will unregister it while the call's copies are in flight.  Asked BEFORE is_pinned() (round 4: a
    // registered array looks like any pinned one to hipPointerGetAttributes, and the steady-state calls used to bypass the cache -- the
    // hottest arrays were evicted first, nothing held them during the DMA, and a stale registration was not retried).
    // Two steps (round 5, advisor): lookup() only consults the registrations this cache owns; sight() records a sighting of a range and registers it
    // on the second one.  HostPin calls sight() only for memory that is NOT pinned already: an ndfft_host_alloc / hipHostMalloc array that missed the
    // lookup is the caller's own pinned memory -- registering it again would either fail every time or leave the cache owning (and later
    // unregistering) a registration over memory the caller frees with hipHostFree.
    bool lookup(const void *p, size_t bytes) {
        const uintptr_t lo = (uintptr_t)p, hi = lo + std::max<size_t>(bytes, 1);
        std::lock_guard<std::mutex> g(mu_);
        ++tick_;
        for (R &r : v_) if (r.registered && r.lo <= lo && hi <= r.hi) { r.last = tick_; ++r.inuse; return true; }   // any size: sub-views of a registered array too
        return false;
    }
    bool wants(size_t bytes) { std::lock_guard<std::mutex> g(mu_); return limit_ && bytes >= ((size_t)8 << 20); }
    bool sight(const void *p, size_t bytes) {
        const uintptr_t lo = (uintptr_t)p, hi = lo + std::max<size_t>(bytes, 1);
        std::lock_guard<std::mutex> g(mu_);
        ++tick_;
        if (!limit_ || bytes < ((size_t)8 << 20)) return false;
        for (size_t i = 0; i < v_.size();) {
            R &r = v_[i];
            if (!(r.lo == lo && r.hi == hi) && lo < r.hi && r.lo < hi) {   // overlaps another range: the caller's allocation changed (an unregistered sighting
                if (r.inuse) return false;                                //   of a larger, older array must not be what gets pinned -- only the range of THIS call is)
                drop(i);
                continue;
            }
            ++i;
        }
        R *hit = nullptr;                                 // (looked up after the erasures above: they move entries)
        for (R &r : v_) if (r.lo == lo && r.hi == hi) { hit = &r; break; }
        if (hit) {
            hit->last = tick_;
            if (++hit->seen < 2) return false;            // (after a failed registration `seen` restarts at -8: a bounded back-off, not a ban)
            if (hipHostRegister((void *)lo, hi - lo, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); hit->seen = -8; return false; }
            hit->registered = true; hit->inuse = 1; reg_bytes_ += hi - lo;
            evict(lo);                                    // (evict() may move entries: `hit` is dead from here)
            return true;
        }
        if (v_.size() >= 256) {                           // forget the oldest unregistered sighting
            size_t o = v_.size();
            for (size_t i = 0; i < v_.size(); ++i) if (!v_[i].registered && (o == v_.size() || v_[i].last < v_[o].last)) o = i;
            if (o < v_.size()) v_.erase(v_.begin() + o);
        }
        v_.push_back({lo, hi, tick_, 1, false, 0});
        return false;
    }
    void release(const void *p) {
        const uintptr_t a = (uintptr_t)p;
        std::lock_guard<std::mutex> g(mu_);
        for (R &r : v_) if (r.registered && r.lo <= a && a < r.hi && r.inuse > 0) { --r.inuse; return; }
    }
    // p == nullptr: everything.  Returns the number of registrations given back.
    int forget(const void *p) {
        const uintptr_t a = (uintptr_t)p;
        std::lock_guard<std::mutex> g(mu_);
        int n = 0;
        for (size_t i = 0; i < v_.size();) {
            if ((!p || (v_[i].lo <= a && a < v_[i].hi)) && !v_[i].inuse) { n += v_[i].registered; drop(i); } else ++i;
        }
        return n;
    }
  private:
    struct R { uintptr_t lo, hi; uint64_t last; int seen; bool registered; int inuse; };
    HostRegCache() { limit_ = (size_t)std::max(0L, sw().host_reg_cache_mb) << 20; }
  public:
    void set_limit(size_t bytes) {
        std::lock_guard<std::mutex> g(mu_);
        limit_ = bytes;
        if (!bytes) { for (size_t i = 0; i < v_.size();) { if (!v_[i].inuse) drop(i); else ++i; } }
        else evict(0);
    }
  private:
    void drop(size_t i) {
        if (v_[i].registered) { (void)hipHostUnregister((void *)v_[i].lo); (void)hipGetLastError(); reg_bytes_ -= v_[i].hi - v_[i].lo; }
        v_.erase(v_.begin() + i);
    }
    void evict(uintptr_t keep) {
        while (reg_bytes_ > limit_) {
            size_t o = v_.size();
            for (size_t i = 0; i < v_.size(); ++i)
                if (v_[i].registered && !v_[i].inuse && v_[i].lo != keep && (o == v_.size() || v_[i].last < v_[o].last)) o = i;
            if (o == v_.size()) return;
            drop(o);
        }
    }
    std::mutex mu_;
    std::vector<R> v_;
    uint64_t tick_ = 0;
    size_t reg_bytes_ = 0, limit_ = 0;
};
struct HostPin {       // one side of a call: registered for the duration of the call if the cache says so
    const void *p = nullptr; bool held = false;
    HostPin(const void *ptr, size_t bytes) : p(ptr) {
        HostRegCache &c = HostRegCache::get();
        held = c.lookup(ptr, bytes);
        if (!held && c.wants(bytes) && !is_pinned(ptr)) held = c.sight(ptr, bytes);
    }
    ~HostPin() { if (held) HostRegCache::get().release(p); }
};
int pipe_init(Pipe &p, int chunks) {
    if (!p.ok) {
        NDFFT_HIP(hipStreamCreate(&p.h2d)); NDFFT_HIP(hipStreamCreate(&p.cmp)); NDFFT_HIP(hipStreamCreate(&p.d2h));
        p.ok = true;
    }
    while ((int)p.up.size() < chunks) {
        hipEvent_t a, b, c;
        NDFFT_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming)); NDFFT_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        NDFFT_HIP(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        p.up.push_back(a); p.done.push_back(b); p.down.push_back(c);
    }
    return NDFFT_OK;
}
// span (in elements) of one index of dimension 0, i.e. of the sub-view shape[1:], or -1 if it has negative strides
int64_t inner_span(int ndim, const int64_t *shape, const int64_t *stride) {
    int64_t hi = 0;
    for (int d = 1; d < ndim; ++d) {
        if (shape[d] <= 0) return 0;
        if (stride[d] < 0) return -1;
        hi += (shape[d] - 1) * stride[d];
    }
    return hi + 1;
}
}  // namespace

static int exec_pinned_pipeline_body(DeviceWs &ws, const ndfft_plan *plan, int op, const char *hin, char *hout, int ndim, const int64_t *shape_in,
                                     const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                                     double scale, size_t ein, size_t eout, int chunks) {
    int rc;
    Pipe &pp = ws.pipe;
    if ((rc = pipe_init(pp, chunks))) return rc;
    const int64_t R = shape_in[0];
    const int64_t isp = inner_span(ndim, shape_in, stride_in), osp = inner_span(ndim, shape_out, stride_out);
    std::vector<int64_t> si(shape_in, shape_in + ndim), so(shape_out, shape_out + ndim);
    for (int c = 0; c < chunks; ++c) {
        const int64_t r0 = R * c / chunks, r1 = R * (c + 1) / chunks;
        if (r1 <= r0) continue;
        si[0] = so[0] = r1 - r0;
        const size_t off_in = (size_t)(r0 * stride_in[0]) * ein, off_out = (size_t)(r0 * stride_out[0]) * eout;
        const size_t bytes_in = (size_t)((r1 - r0 - 1) * stride_in[0] + isp) * ein, bytes_out = (size_t)((r1 - r0 - 1) * stride_out[0] + osp) * eout;
        NDFFT_HIP(hipMemcpyAsync((char *)ws.stage_in.p + off_in, hin + off_in, bytes_in, hipMemcpyHostToDevice, pp.h2d));
        NDFFT_HIP(hipEventRecord(pp.up[c], pp.h2d));
        NDFFT_HIP(hipStreamWaitEvent(pp.cmp, pp.up[c], 0));
        Problem P;
        bool nothing;
        if ((rc = prepare(plan, op, ndim, si.data(), stride_in, so.data(), stride_out, axis, norm, scale, P, nothing))) return rc;
        if (!nothing && (rc = dispatch_peeled(P, (const char *)ws.stage_in.p + off_in, (char *)ws.stage_out.p + off_out, ein, eout, pp.cmp))) return rc;
        NDFFT_HIP(hipEventRecord(pp.done[c], pp.cmp));
        NDFFT_HIP(hipStreamWaitEvent(pp.d2h, pp.done[c], 0));
        NDFFT_HIP(hipMemcpyAsync(hout + off_out, (const char *)ws.stage_out.p + off_out, bytes_out, hipMemcpyDeviceToHost, pp.d2h));
    }
    NDFFT_HIP(hipStreamSynchronize(pp.d2h));
    NDFFT_HIP(hipStreamSynchronize(pp.cmp));
    return NDFFT_OK;
}
static int exec_pinned_pipeline(DeviceWs &ws, const ndfft_plan *plan, int op, const char *hin, char *hout, int ndim, const int64_t *shape_in,
                                const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                                double scale, size_t ein, size_t eout, int chunks) {
    const int rc = exec_pinned_pipeline_body(ws, plan, op, hin, hout, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, ein, eout, chunks);
    // on an error some chunks' asynchronous copies into the caller's arrays (and into the staging buffers, which the
    // next call may regrow) are still in flight: never return before they have drained
    if (rc) ws.pipe.sync_all();
    return rc;
}

// ---- pageable host arrays: the same chunk pipeline through pinned bounce buffers --------------------------------
// hipMemcpy from / to pageable memory is staged by the runtime on one thread, and the two PCIe directions never overlap
// (tools/h2d_bench.hip: 9.7 ms for 2 x 256 MiB).  Here the staging is ours: a small pool of host threads copies row
// chunks between the caller's arrays and three pinned slots per direction while the DMA engines move the previous
// chunks, so upload, transform and download overlap for ANY host array (ndarray allocates pageable memory).
namespace {
// Bulk host copy with streaming (non-temporal) stores: the pieces the pool moves (a few MiB each) are below glibc's own non-temporal threshold, so plain
// memcpy reads the destination lines before overwriting them (three memory transfers per byte instead of two).  AVX2 only where the CPU has it;
// (-DNDFFT_NO_NT_COPY keeps memcpy).  Host code only.
#if defined(__x86_64__) && !defined(NDFFT_NO_NT_COPY)
__attribute__((target("avx2"))) void copy_nt_avx2(char *d, const char *s, size_t n) {
    while (n && ((uintptr_t)d & 31)) { *d++ = *s++; --n; }
    size_t k = n / 128;
    for (; k; --k, d += 128, s += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)s), b = _mm256_loadu_si256((const __m256i *)(s + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
        _mm256_stream_si256((__m256i *)d, a); _mm256_stream_si256((__m256i *)(d + 32), b);
        _mm256_stream_si256((__m256i *)(d + 64), c); _mm256_stream_si256((__m256i *)(d + 96), e);
    }
    _mm_sfence();
    n &= 127;
    if (n) memcpy(d, s, n);
}
void bulk_copy(char *d, const char *s, size_t n) {
    static const bool nt = __builtin_cpu_supports("avx2");
    if (nt && n >= ((size_t)256 << 10)) copy_nt_avx2(d, s, n); else memcpy(d, s, n);
}
#else
void bulk_copy(char *d, const char *s, size_t n) { memcpy(d, s, n); }
#endif
struct CopyGroup { std::atomic<int> left{0}; std::mutex m; std::condition_variable cv; };
class CopyPool {
  public:
    static CopyPool &get() { static CopyPool *p = new CopyPool; return *p; }   // never destroyed: detached workers
    int threads() const { return nthreads_; }
    // copies `bytes` in `pieces` slices on the pool; returns immediately
    void copy_async(CopyGroup &g, char *dst, const char *src, size_t bytes, int pieces) {
        pieces = (int)std::max<size_t>(1, std::min<size_t>((size_t)pieces, bytes / (256 << 10) + 1));
        g.left.store(pieces);
        const size_t per = (bytes / pieces + 63) & ~(size_t)63;
        std::lock_guard<std::mutex> lk(mu_);
        for (int i = 0; i < pieces; ++i) {
            const size_t o = std::min(bytes, (size_t)i * per), e = i + 1 == pieces ? bytes : std::min(bytes, (size_t)(i + 1) * per);
            q_.push_back({dst + o, src + o, e - o, &g});
        }
        cv_.notify_all();
    }
    static void wait(CopyGroup &g) {
        std::unique_lock<std::mutex> lk(g.m);
        g.cv.wait(lk, [&] { return g.left.load() == 0; });
    }
  private:
    struct Piece { char *d; const char *s; size_t n; CopyGroup *g; };
    CopyPool() {
        // threads: three quarters of the CPUs this process may use (affinity / hardware count capped by the cgroup quota:
        // the MI355X boxes show 256 CPUs and grant 16), between 2 and 12.  Measured on 4096 x 4096 c128 (2 x 256 MiB,
        // plain path 9.8 ms): 4 threads 9.5-10.6 ms, 8 threads 8.8 ms, 12 threads 7.9-8.4 ms -- the host copies, not PCIe, bound it
        const int forced = sw().copy_threads;            // NDFFT_COPY_THREADS
        long hw = (long)std::thread::hardware_concurrency();
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0}; long per = 0;
            if (fscanf(f, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) hw = std::min(hw, std::max(1L, (atol(q) + per - 1) / per));
            fclose(f);
        }
        nthreads_ = forced > 0 ? forced : (int)std::max(2L, std::min(12L, hw * 3 / 4));
        if (nthreads_ < 1) nthreads_ = 1;
        for (int i = 0; i < nthreads_; ++i) std::thread([this] { loop(); }).detach();
    }
    void loop() {
        for (;;) {
            Piece p;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !q_.empty(); });
                p = q_.front(); q_.pop_front();
            }
            bulk_copy(p.d, p.s, p.n);
            // the count changes only under the group's mutex: a waiter (whose CopyGroup lives on its stack) cannot see zero, return and
            // destroy the group while this thread is still about to lock it
            { std::lock_guard<std::mutex> lk(p.g->m); if (p.g->left.fetch_sub(1) == 1) p.g->cv.notify_all(); }
        }
    }
    int nthreads_ = 1;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Piece> q_;
};
}  // namespace

static int exec_bounce_pipeline_body(DeviceWs &ws, const ndfft_plan *plan, int op, const char *hin, char *hout, int ndim, const int64_t *shape_in,
                                     const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                                     double scale, size_t ein, size_t eout, int chunks) {
    int rc;
    Pipe &pp = ws.pipe;
    if ((rc = pipe_init(pp, chunks))) return rc;
    CopyPool &pool = CopyPool::get();
    const int64_t R = shape_in[0];
    const int64_t isp = inner_span(ndim, shape_in, stride_in), osp = inner_span(ndim, shape_out, stride_out);
    auto r0_of = [&](int c) { return R * c / chunks; };
    auto off_in = [&](int c) { return (size_t)(r0_of(c) * stride_in[0]) * ein; };
    auto off_out = [&](int c) { return (size_t)(r0_of(c) * stride_out[0]) * eout; };
    auto bytes_in = [&](int c) { const int64_t n = r0_of(c + 1) - r0_of(c); return n <= 0 ? (size_t)0 : (size_t)((n - 1) * stride_in[0] + isp) * ein; };
    auto bytes_out = [&](int c) { const int64_t n = r0_of(c + 1) - r0_of(c); return n <= 0 ? (size_t)0 : (size_t)((n - 1) * stride_out[0] + osp) * eout; };
    size_t max_in = 0, max_out = 0;
    for (int c = 0; c < chunks; ++c) { max_in = std::max(max_in, bytes_in(c)); max_out = std::max(max_out, bytes_out(c)); }
    for (int k = 0; k < 3; ++k) { if ((rc = ws.bounce_in[k].reserve(max_in)) || (rc = ws.bounce_out[k].reserve(max_out))) return rc; }
    std::vector<int64_t> si(shape_in, shape_in + ndim), so(shape_out, shape_out + ndim);
    const int half = std::max(1, pool.threads() / 2);
    CopyGroup gu, gd;
    for (int it = 0; it < chunks + 2; ++it) {
        const int cu = it, cd = it - 2;
        const bool up = cu < chunks && bytes_in(cu) > 0, dn = cd >= 0 && bytes_out(cd) > 0;
        // (both waits BEFORE any pool copy is submitted: an error return must never leave the pool working on gu / gd)
        if (up && cu >= 3) NDFFT_HIP(hipEventSynchronize(pp.up[cu - 3]));      // the slot's previous upload has left it
        if (dn) NDFFT_HIP(hipEventSynchronize(pp.down[cd]));                    // chunk cd has arrived in its slot
        if (up) pool.copy_async(gu, (char *)ws.bounce_in[cu % 3].p, hin + off_in(cu), bytes_in(cu), half);
        if (dn) pool.copy_async(gd, hout + off_out(cd), (const char *)ws.bounce_out[cd % 3].p, bytes_out(cd), half);
        if (up) CopyPool::wait(gu);
        if (dn) CopyPool::wait(gd);
        if (!up) continue;
        NDFFT_HIP(hipMemcpyAsync((char *)ws.stage_in.p + off_in(cu), ws.bounce_in[cu % 3].p, bytes_in(cu), hipMemcpyHostToDevice, pp.h2d));
        NDFFT_HIP(hipEventRecord(pp.up[cu], pp.h2d));
        NDFFT_HIP(hipStreamWaitEvent(pp.cmp, pp.up[cu], 0));
        si[0] = so[0] = r0_of(cu + 1) - r0_of(cu);
        Problem P;
        bool nothing;
        if ((rc = prepare(plan, op, ndim, si.data(), stride_in, so.data(), stride_out, axis, norm, scale, P, nothing))) return rc;
        if (!nothing && (rc = dispatch_peeled(P, (const char *)ws.stage_in.p + off_in(cu), (char *)ws.stage_out.p + off_out(cu), ein, eout, pp.cmp))) return rc;
        NDFFT_HIP(hipEventRecord(pp.done[cu], pp.cmp));
        NDFFT_HIP(hipStreamWaitEvent(pp.d2h, pp.done[cu], 0));
        NDFFT_HIP(hipMemcpyAsync(ws.bounce_out[cu % 3].p, (const char *)ws.stage_out.p + off_out(cu), bytes_out(cu), hipMemcpyDeviceToHost, pp.d2h));
        NDFFT_HIP(hipEventRecord(pp.down[cu], pp.d2h));
    }
    return NDFFT_OK;
}
static int exec_bounce_pipeline(DeviceWs &ws, const ndfft_plan *plan, int op, const char *hin, char *hout, int ndim, const int64_t *shape_in,
                                const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                                double scale, size_t ein, size_t eout, int chunks) {
    const int rc = exec_bounce_pipeline_body(ws, plan, op, hin, hout, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, ein, eout, chunks);
    if (rc) ws.pipe.sync_all();   // (pool copies are always waited for inside the body; only device work can be in flight)
    return rc;
}

extern "C" {

int ndfft_exec_device(const ndfft_plan *plan, int op, const void *d_in, void *d_out, int ndim,
                      const int64_t *shape_in, const int64_t *stride_in, const int64_t *shape_out,
                      const int64_t *stride_out, int axis, int norm, double scale, void *stream) {
    clear_err();
    g_last_policy = -1;
    Problem P;
    bool nothing;
    int rc = prepare(plan, op, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, P, nothing);
    if (rc || nothing) return rc;
    if (!d_in || !d_out) return fail(NDFFT_ERR_INVALID_ARG, "null array pointer");
    const size_t r = real_size(plan->dtype);
    return dispatch_peeled(P, (const char *)d_in, (char *)d_out, op_in_cplx(op) ? 2 * r : r, op_out_cplx(op) ? 2 * r : r,
                           (hipStream_t)stream);
}

int ndfft_host_alloc(void **h_ptr, size_t bytes) {
    clear_err();
    if (!h_ptr) return fail(NDFFT_ERR_INVALID_ARG, "h_ptr is null");
    NDFFT_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return NDFFT_OK;
}
int ndfft_host_free(void *h_ptr) {
    clear_err();
    if (h_ptr) NDFFT_HIP(hipHostFree(h_ptr));
    return NDFFT_OK;
}

int ndfft_exec(const ndfft_plan *plan, int op, const void *in, void *out, int ndim, const int64_t *shape_in,
               const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
               double scale) {
    clear_err();
    Problem P;
    bool nothing;
    int rc = prepare(plan, op, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, P, nothing);
    if (rc || nothing) return rc;
    if (!in || !out) return fail(NDFFT_ERR_INVALID_ARG, "null array pointer");
    DeviceWs *wsp;
    if ((rc = current_ws(&wsp))) return rc;
    DeviceWs &ws = *wsp;
    const size_t r = real_size(plan->dtype);
    const size_t ein = op_in_cplx(op) ? 2 * r : r, eout = op_out_cplx(op) ? 2 * r : r;
    int64_t ilo, ihi, icnt, olo, ohi, ocnt;
    view_range(ndim, shape_in, stride_in, ilo, ihi, icnt);
    view_range(ndim, shape_out, stride_out, olo, ohi, ocnt);
    const size_t ibytes = (size_t)(ihi - ilo + 1) * ein, obytes = (size_t)(ohi - olo + 1) * eout;
    const char *hin = (const char *)in + ilo * (int64_t)ein;
    char *hout = (char *)out + olo * (int64_t)eout;
    const bool out_dense = (int64_t)(ohi - olo + 1) == ocnt;
    // Small calls (the reference's own bench shapes: benches/ndrustfft.rs:6-7, n x n with n = 128 ... 264): no DMA at all.  The kernels read the
    // input straight from a pinned, device-mapped bounce buffer over PCIe and write the output into another one: two host memcpys and ONE stream
    // synchronisation are the whole call (the plain path below pays two synchronous hipMemcpy of pageable memory, ~15-20 us each whatever the size).
    const size_t small_limit = (size_t)NDFFT_DEV_INT("NDFFT_HOST_SMALL_KB", 2048) << 10;
    if (ibytes + obytes <= small_limit) {
        if ((rc = ws.bounce_in[0].reserve(std::max(ibytes, small_limit))) || (rc = ws.bounce_out[0].reserve(std::max(obytes, small_limit)))) return rc;
        memcpy(ws.bounce_in[0].p, hin, ibytes);
        const char *din = (const char *)ws.bounce_in[0].p - ilo * (int64_t)ein;
        char *dout = (char *)ws.bounce_out[0].p - olo * (int64_t)eout;
        rc = dispatch_peeled(P, din, dout, ein, eout, (hipStream_t) nullptr);
        const hipError_t se = hipStreamSynchronize(nullptr);
        if (rc) return rc;
        if (se != hipSuccess) return fail(NDFFT_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(se));
        if (out_dense) memcpy(hout, ws.bounce_out[0].p, obytes);
        else copy_view_elements((char *)out, (const char *)ws.bounce_out[0].p - olo * (int64_t)eout, ndim, shape_out, stride_out, eout);   // holes belong to the caller
        return NDFFT_OK;
    }
    if ((rc = ws.stage_in.reserve(ibytes))) return rc;
    if ((rc = ws.stage_out.reserve(obytes))) return rc;
    // dense, C-ordered in dimension 0, transform along another axis: pipelined row chunks -- straight DMA for pinned arrays
    // (ndfft_host_alloc), through pinned bounce buffers filled by the copy pool for pageable ones.  NDFFT_HOST_PIPE=0: plain path.
    if (ndim >= 2 && axis != 0 && ilo == 0 && olo == 0 && shape_in[0] == shape_out[0] && shape_in[0] >= 16 &&
        out_dense && ibytes + obytes >= ((size_t)8 << 20)) {
        const int64_t isp = inner_span(ndim, shape_in, stride_in), osp = inner_span(ndim, shape_out, stride_out);
        if (isp > 0 && osp > 0 && stride_in[0] >= isp && stride_out[0] >= osp) {
            // a chunk must stay a real problem: kernel choice depends on the size of a call (hiprtc specialisation from 2^16-2^17
            // points), so never cut below 2^18 points per chunk
            const int64_t max_chunks = std::max<int64_t>(1, (P.nlanes * std::max(P.xlen, P.ylen)) >> 18);
            // the caller's own arrays, seen before: registered once, DMA straight from / to them from then on.  The cache is asked FIRST
            // (held = a registration it owns, kept alive for this call); only an array it does not own can be the caller's pinned memory
            bool retry_unpinned = false;
            {
                HostPin pin_in(hin, ibytes), pin_out(hout, obytes);
                const bool own_in = !pin_in.held && is_pinned(in), own_out = !pin_out.held && is_pinned(out);
                if ((own_in || pin_in.held) && (own_out || pin_out.held)) {
                    const int chunks = (int)std::min<int64_t>(std::min<int64_t>(shape_in[0], max_chunks), 8);
                    const int rcp = exec_pinned_pipeline(ws, plan, op, hin, hout, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, ein, eout, chunks);
                    // A registration made by the cache can be stale: the caller freed the array and the allocator handed the same addresses out
                    // again (seen on the MI355X with numpy arrays: hipMemcpyAsync then fails with "invalid argument" -- before anything has been
                    // written to the caller's output).  Forget both ranges and run this call through the bounce buffers instead.
                    if (rcp == NDFFT_OK || rcp != NDFFT_ERR_HIP || !(pin_in.held || pin_out.held)) return rcp;
                    retry_unpinned = true;
                }
            }
            if (retry_unpinned) {
                (void)hipGetLastError();
                HostRegCache::get().forget(hin); HostRegCache::get().forget(hout);
                clear_err();
            }
            const int hp = sw().host_pipe;                    // NDFFT_HOST_PIPE
            const bool force = hp == 1;                       // tests: pipeline small calls too
            if (force || (hp != 0 && max_chunks >= 4 && ibytes + obytes >= ((size_t)32 << 20))) {   // small calls: the plain path
                // chunks of ~32 MiB per direction (at least 4, at most 64)
                const int64_t want = std::max<int64_t>(4, std::min<int64_t>(64, (int64_t)(std::max(ibytes, obytes) >> 25)));
                const int chunks = (int)std::min<int64_t>(std::min<int64_t>(shape_in[0], force ? 64 : max_chunks), want);
                return exec_bounce_pipeline(ws, plan, op, hin, hout, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, ein, eout, chunks);
            }
        }
    }
    if (!out_dense && (rc = ws.bounce_out[0].reserve(obytes))) return rc;   // before anything is in flight
    // (a copy that fails may have met a stale cached registration over the caller's array -- see HostRegCache: forget it and try once more)
    auto copy_host = [](void *dst, const void *src, size_t bytes, hipMemcpyKind kind, const void *host_side) -> int {
        if (hipMemcpy(dst, src, bytes, kind) == hipSuccess) return NDFFT_OK;
        (void)hipGetLastError();
        if (!HostRegCache::get().forget(host_side)) return fail(NDFFT_ERR_HIP, "hipMemcpy between the caller's array and the device failed");
        NDFFT_HIP(hipMemcpy(dst, src, bytes, kind));
        return NDFFT_OK;
    };
    if ((rc = copy_host(ws.stage_in.p, hin, ibytes, hipMemcpyHostToDevice, hin))) return rc;
    const char *din = (const char *)ws.stage_in.p - ilo * (int64_t)ein;
    char *dout = (char *)ws.stage_out.p - olo * (int64_t)eout;
    rc = dispatch_peeled(P, din, dout, ein, eout, (hipStream_t) nullptr);
    if (rc) { (void)hipStreamSynchronize(nullptr); return rc; }
    if (out_dense) {
        if ((rc = copy_host(hout, ws.stage_out.p, obytes, hipMemcpyDeviceToHost, hout))) return rc;   // synchronises with the kernel
    } else {
        // The output view has holes.  They belong to the caller (possibly to ANOTHER thread's &mut view of the same
        // allocation), so they are neither read nor written: the span comes back into a private pinned image and only
        // the view's own elements are copied out of it.
        NDFFT_HIP(hipMemcpy(ws.bounce_out[0].p, ws.stage_out.p, obytes, hipMemcpyDeviceToHost));
        copy_view_elements((char *)out, (const char *)ws.bounce_out[0].p - olo * (int64_t)eout, ndim, shape_out, stride_out, eout);
    }
    return NDFFT_OK;
}

int ndfft_host_reg_cache(size_t max_bytes) {
    clear_err();
    HostRegCache::get().set_limit(max_bytes);
    return NDFFT_OK;
}

int ndfft_host_forget(const void *h_ptr) {
    clear_err();
    (void)HostRegCache::get().forget(h_ptr);
    return NDFFT_OK;
}

int ndfft_last_input_policy(void) { return g_last_policy; }

int ndfft_set_input_hint(int hint) {
    clear_err();
    if (hint < NDFFT_INPUT_AUTO || hint > NDFFT_INPUT_COLD) return fail(NDFFT_ERR_INVALID_ARG, "bad input hint");
    g_input_hint = hint;
    return NDFFT_OK;
}

int ndfft_release_workspace(void) {
    clear_err();
    g_tws.release_all();   // every device this thread has used; each synchronised under its own hipSetDevice
    shard_release_all();   // and the chunk buffers of the multi-device workers (up to 4 x 64 MiB per worker)
    return NDFFT_OK;
}

}  // extern "C"
