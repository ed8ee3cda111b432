// shard.hip -- ndfft_exec_sharded / ndfft_exec_sharded_device: one nd* call spread over several MI355X from ONE
// host process (no torch, no MPI): the native counterpart of the reference's `_par` functions
// (create_transform_par!, src/lib.rs:169-238), which hand the independent lanes to rayon's workers
// (src/lib.rs:187-194) -- here the workers are GPUs.
//
// Lanes are independent, so the array is cut into contiguous blocks along its outermost non-transform dimension,
// one block per device, and every device runs the ordinary single-device call on its block: no collective, no
// exchange.  Host arrays go up and down each device's own PCIe link concurrently; a device-resident array is
// scattered from / gathered to the device that holds it with hipMemcpyPeerAsync over xGMI (host-less).
// Each device id has one persistent worker thread bound to it (hipSetDevice once), so its workspace -- staging
// buffers, pinned bounce buffers, scratch, twiddle tables, JIT modules -- lives as long as the process.
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <future>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace ndfft {
namespace {

struct TaskResult { int rc = NDFFT_OK; std::string err, path; };

class Worker {
  public:
    explicit Worker(int device) : device_(device), th_([this] { loop(); }) { th_.detach(); }
    std::future<TaskResult> submit(std::function<TaskResult()> fn) {
        std::packaged_task<TaskResult()> t([this, fn]() {
            if (set_err_ != hipSuccess) {
                TaskResult r; r.rc = NDFFT_ERR_HIP;
                r.err = "hipSetDevice(" + std::to_string(device_) + "): " + hipGetErrorString(set_err_);
                return r;
            }
            return fn();
        });
        std::future<TaskResult> f = t.get_future();
        { std::lock_guard<std::mutex> g(mu_); q_.push_back(std::move(t)); }
        cv_.notify_one();
        return f;
    }
  private:
    void loop() {
        set_err_ = hipSetDevice(device_);
        for (;;) {
            std::packaged_task<TaskResult()> t;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !q_.empty(); });
                t = std::move(q_.front()); q_.pop_front();
            }
            t();
        }
    }
    hipError_t set_err_ = hipSuccess;
    int device_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<TaskResult()>> q_;
    std::thread th_;
};

// never destroyed: the workers are detached and outlive static destruction
std::mutex &pool_mu() { static std::mutex *m = new std::mutex; return *m; }
std::map<int, Worker *> &pool_ref() { static std::map<int, Worker *> *pool = new std::map<int, Worker *>; return *pool; }
Worker &worker_for(int device) {
    std::map<int, Worker *> *pool = &pool_ref();
    std::lock_guard<std::mutex> g(pool_mu());
    auto it = pool->find(device);
    if (it == pool->end()) it = pool->emplace(device, new Worker(device)).first;
    return *it->second;
}

size_t elem_size(int dtype, bool cplx) { return (dtype == NDFFT_F32 ? 4 : 8) * (cplx ? 2 : 1); }
bool in_cplx(int op) { return op == NDFFT_OP_C2C_FWD || op == NDFFT_OP_C2C_INV || op == NDFFT_OP_C2R; }
bool out_cplx(int op) { return op == NDFFT_OP_C2C_FWD || op == NDFFT_OP_C2C_INV || op == NDFFT_OP_R2C; }

// the dimension to cut: not the transform axis, extent >= 2, outermost in the OUTPUT's memory (largest |stride|),
// so that a block is as close to one contiguous byte range as the layout allows; -1: a single lane
int split_dim(int ndim, const int64_t *shape, const int64_t *stride_in, const int64_t *stride_out, int axis) {
    int best = -1;
    for (int d = 0; d < ndim; ++d) {
        if (d == axis || shape[d] < 2) continue;
        if (best < 0) { best = d; continue; }
        const int64_t a = std::llabs(stride_out[d]), b = std::llabs(stride_out[best]);
        if (a > b || (a == b && std::llabs(stride_in[d]) > std::llabs(stride_in[best]))) best = d;
    }
    return best;
}

struct Block { int device; int64_t lo, hi; };
int plan_blocks(int ndim, const int64_t *shape_in, const int64_t *stride_in, const int64_t *stride_out, int axis, int n_devices,
                const int *device_ids, int *dim_out, std::vector<Block> &blocks) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
    if (n_devices < 1 || !device_ids) return fail(NDFFT_ERR_INVALID_ARG, "sharded exec needs n_devices >= 1 and a device_ids array");
    for (int g = 0; g < n_devices; ++g)
        if (device_ids[g] < 0 || device_ids[g] >= ndev)
            return fail(NDFFT_ERR_INVALID_ARG, "device id " + std::to_string(device_ids[g]) + " out of range (" + std::to_string(ndev) + " visible)");
    const int d = split_dim(ndim, shape_in, stride_in, stride_out, axis);
    *dim_out = d;
    if (d < 0) { blocks.push_back({device_ids[0], 0, 1}); return NDFFT_OK; }
    const int64_t ext = shape_in[d];
    const int64_t G = std::min<int64_t>(n_devices, ext);
    for (int64_t g = 0; g < G; ++g) blocks.push_back({device_ids[g], ext * g / G, ext * (g + 1) / G});
    return NDFFT_OK;
}

int collect(std::vector<std::future<TaskResult>> &fs) {
    int rc = NDFFT_OK; std::string err, path;
    for (size_t g = 0; g < fs.size(); ++g) {
        TaskResult r = fs[g].get();                 // wait for EVERY device before returning, also after a failure
        if (g == 0) path = r.path;
        if (r.rc != NDFFT_OK && rc == NDFFT_OK) { rc = r.rc; err = r.err; }
    }
    if (rc != NDFFT_OK) return fail(rc, err);
    if (!path.empty()) { static thread_local std::string p; p = "sharded:" + path; set_last_path(p.c_str()); }
    return NDFFT_OK;
}

// lowest / highest element offset of a view relative to its element 0
void span_of(int ndim, const int64_t *shape, const int64_t *stride, int64_t &lo, int64_t &hi, int64_t &count) {
    lo = hi = 0; count = 1;
    for (int d = 0; d < ndim; ++d) {
        count *= shape[d];
        if (shape[d] <= 0) continue;
        const int64_t e = (shape[d] - 1) * stride[d];
        if (e < 0) lo += e; else hi += e;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident arrays: what a worker thread (bound to device `dev`) needs to move a block between the device
// that holds the array (`root`) and its own device without ever touching an element the block does not own
// ---------------------------------------------------------------------------------------------------------------
struct ShardBuf {
    void *p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {   // the owning device must be current
        if (bytes <= cap) return NDFFT_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        NDFFT_HIP(hipMalloc(&p, bytes ? bytes : 1));
        cap = bytes;
        return NDFFT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
constexpr int kSlots = 2;   // chunk c+1 is scattered while chunk c is transformed and chunk c-1 gathered
struct ShardStreams {       // streams and events of ONE device, created by the thread that uses them
    hipStream_t s[3] = {nullptr, nullptr, nullptr};   // dev: 0 scatter, 1 transform, 2 gather; root: 0 pack / unpack
    hipEvent_t ev[4][kSlots] = {};                    // per slot: 0 packed (root), 1 scattered, 2 transformed, 3 gathered
    bool ok = false;
    int init() {
        if (ok) return NDFFT_OK;
        for (auto &x : s) NDFFT_HIP(hipStreamCreate(&x));
        for (auto &row : ev) for (auto &e : row) NDFFT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ok = true;
        return NDFFT_OK;
    }
};
struct ShardWs {
    std::map<int, ShardStreams> streams;                         // device -> streams / events
    std::map<int, ShardBuf> rin[kSlots], rout[kSlots];           // root-side dense images (packed input, result to unpack), by root device
    ShardBuf din[kSlots], dout[kSlots];                          // this worker's device
    ~ShardWs() {   // a worker never exits; a caller thread that used the root path may
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
        for (int k = 0; k < kSlots; ++k) {
            for (auto *m : {&rin[k], &rout[k]}) for (auto &kv : *m) if (hipSetDevice(kv.first) == hipSuccess) kv.second.release();
        }
        (void)hipSetDevice(cur);
    }
};
thread_local ShardWs t_sws;

// ndfft_release_workspace: every worker gives back its chunk buffers (its own device's and the root-side images it reserved), after its streams drained
void release_this_threads_shard_buffers() {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    for (auto &kv : t_sws.streams) if (kv.second.ok && hipSetDevice(kv.first) == hipSuccess) for (auto &x : kv.second.s) (void)hipStreamSynchronize(x);
    for (int k = 0; k < kSlots; ++k) {
        for (auto *m : {&t_sws.rin[k], &t_sws.rout[k]}) for (auto &kv : *m) if (hipSetDevice(kv.first) == hipSuccess) kv.second.release();
    }
    (void)hipSetDevice(cur);
    for (int k = 0; k < kSlots; ++k) { t_sws.din[k].release(); t_sws.dout[k].release(); }
    (void)hipGetLastError();
}
}  // namespace
void shard_release_all() {
    std::vector<Worker *> ws;
    { std::lock_guard<std::mutex> g(pool_mu()); for (auto &kv : pool_ref()) ws.push_back(kv.second); }
    std::vector<std::future<TaskResult>> fs;
    for (Worker *w : ws) fs.push_back(w->submit([]() { release_this_threads_shard_buffers(); return TaskResult(); }));
    for (auto &f : fs) (void)f.get();
    release_this_threads_shard_buffers();          // a caller thread that ran a root block itself
}
namespace {

// a view as the device copy kernel sees it: extent-1 dimensions dropped, C order (last dimension fastest)
struct ViewDesc { int32_t nd; int32_t pad_; int64_t shape[NDFFT_MAX_DIMS]; int64_t stride[NDFFT_MAX_DIMS]; };
ViewDesc make_desc(int ndim, const int64_t *shape, const int64_t *stride) {
    ViewDesc v; v.nd = 0; v.pad_ = 0;
    for (int d = 0; d < ndim; ++d) if (shape[d] != 1) { v.shape[v.nd] = shape[d]; v.stride[v.nd] = stride[d]; ++v.nd; }
    return v;
}
// dense[i] <-> view[multi-index of i in C order]; unpack = 1 writes the view's OWN elements and nothing else
template <typename E>
__global__ __launch_bounds__(256) void k_view_copy(const E *view, E *dense, ViewDesc v, int64_t total, int unpack) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i, off = 0;
        for (int d = v.nd - 1; d >= 0; --d) { const int64_t c = r % v.shape[d]; r /= v.shape[d]; off += c * v.stride[d]; }
        if (unpack) ((E *)view)[off] = dense[i]; else dense[i] = view[off];
    }
}
int launch_view_copy(const void *view, void *dense, const ViewDesc &v, int64_t total, size_t esz, int unpack, hipStream_t s) {
    if (total <= 0) return NDFFT_OK;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 16384);
    if (esz == 4) hipLaunchKernelGGL(k_view_copy<float>, dim3(grid), dim3(256), 0, s, (const float *)view, (float *)dense, v, total, unpack);
    else if (esz == 8) hipLaunchKernelGGL(k_view_copy<double>, dim3(grid), dim3(256), 0, s, (const double *)view, (double *)dense, v, total, unpack);
    else hipLaunchKernelGGL(k_view_copy<double2>, dim3(grid), dim3(256), 0, s, (const double2 *)view, (double2 *)dense, v, total, unpack);
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}
void c_strides(int ndim, const int64_t *shape, std::vector<int64_t> &st) {
    st.assign(ndim, 1);
    for (int d = ndim - 2; d >= 0; --d) st[d] = st[d + 1] * std::max<int64_t>(shape[d + 1], 1);
}

// xGMI peer access, both directions, once per ordered pair and process.  Without it hipMemcpyPeerAsync still works but is
// staged through host memory; a pair that cannot be mapped (no link) is left on that slow path rather than refused.
int ensure_peer_access(int a, int b) {
    if (a == b) return NDFFT_OK;
    static std::mutex *mu = new std::mutex;
    static std::map<std::pair<int, int>, bool> *done = new std::map<std::pair<int, int>, bool>;
    std::lock_guard<std::mutex> g(*mu);
    int cur = 0;
    NDFFT_HIP(hipGetDevice(&cur));
    for (int pass = 0; pass < 2; ++pass) {
        const int from = pass ? b : a, to = pass ? a : b;
        if (done->count({from, to})) continue;
        int can = 0;
        NDFFT_HIP(hipDeviceCanAccessPeer(&can, from, to));
        if (can) {
            NDFFT_HIP(hipSetDevice(from));
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipSetDevice(cur); return fail(NDFFT_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)); }
            (void)hipGetLastError();
        }
        (*done)[{from, to}] = can != 0;
    }
    NDFFT_HIP(hipSetDevice(cur));
    return NDFFT_OK;
}

size_t shard_chunk_bytes() {   // developer / test switch: bytes of input per pipelined chunk of a block (default 64 MiB)
    const long kb = sw().shard_chunk_kb;             // NDFFT_SHARD_CHUNK_KB
    return kb > 0 ? (size_t)kb << 10 : (size_t)64 << 20;
}

struct RemoteBlock {
    const ndfft_plan *plan; int op, ndim, axis, norm, d, root, dev; double scale; size_t ein, eout;
    std::vector<int64_t> si, so, sti, sto;
    const char *pin; char *pout;
};

// the block already lives on the device that transforms it: the ordinary call on the ORIGINAL views (it writes only
// the block's own elements), on this worker's own stream
TaskResult run_root_block(const RemoteBlock &b) {
    TaskResult r;
    ShardStreams &S = t_sws.streams[b.dev];
    if ((r.rc = S.init())) { r.err = ndfft_last_error(); return r; }
    r.rc = ndfft_exec_device(b.plan, b.op, b.pin, b.pout, b.ndim, b.si.data(), b.sti.data(), b.so.data(), b.sto.data(), b.axis, b.norm, b.scale, S.s[1]);
    if (r.rc) r.err = ndfft_last_error(); else r.path = ndfft_last_path();
    const hipError_t e = hipStreamSynchronize(S.s[1]);
    if (!r.rc && e != hipSuccess) { r.rc = NDFFT_ERR_HIP; r.err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); }
    return r;
}

// A block for another device, as a pipeline of chunks along the split dimension:
//   [pack on root] -> xGMI scatter -> transform on `dev` -> xGMI gather -> [unpack on root]
// A view whose elements fill its address span exactly travels as that span; any other view (holes: the split
// dimension is not the outermost one in memory, stepped views, ...) is packed into a dense C-order image by a copy
// kernel on root and, on the way back, unpacked by one that writes the block's OWN elements and nothing else --
// other devices are writing their blocks of the same array at the same time.
TaskResult run_remote_block(const RemoteBlock &b) {
    TaskResult r;
    auto bad = [&r](int rc) { r.rc = rc; r.err = ndfft_last_error(); return r; };
#define SH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { r.rc = NDFFT_ERR_HIP; r.err = std::string(#call) + ": " + hipGetErrorString(e_); goto done; } } while (0)
    const int ndim = b.ndim, d = b.d, root = b.root, dev = b.dev;
    const int64_t ext = d >= 0 ? b.si[d] : 1;
    int64_t in_elems = 1;
    for (int q = 0; q < ndim; ++q) in_elems *= b.si[q];
    const int64_t K = std::max<int64_t>(1, std::min<int64_t>({ext, (int64_t)64, (int64_t)((in_elems * b.ein + shard_chunk_bytes() - 1) / shard_chunk_bytes())}));
    ShardStreams &SD = t_sws.streams[dev];
    if (int rc = SD.init()) return bad(rc);
    if (hipSetDevice(root) != hipSuccess) { r.rc = NDFFT_ERR_HIP; r.err = "hipSetDevice(root)"; return r; }
    ShardStreams &SR = t_sws.streams[root];
    int rc0 = SR.init();
    // geometry of a chunk of extent e along d (all chunks have extent floor or ceil of ext / K)
    struct Side { bool packed; int64_t lo, span, count; };
    auto side_of = [&](const std::vector<int64_t> &shape, const std::vector<int64_t> &stride, int64_t e) {
        std::vector<int64_t> sh(shape);
        if (d >= 0) sh[d] = e;
        Side s; int64_t hi;
        span_of(ndim, sh.data(), stride.data(), s.lo, hi, s.count);
        s.span = hi - s.lo + 1;
        s.packed = s.span > s.count;
        return s;
    };
    const int64_t emax = (ext + K - 1) / K;
    {   // every buffer is sized once for the largest chunk, in whichever form (span or dense image) is larger
        const Side i = side_of(b.si, b.sti, emax), o = side_of(b.so, b.sto, emax);
        const size_t ib = (size_t)std::max(i.span, i.count) * b.ein, ob = (size_t)std::max(o.span, o.count) * b.eout;
        for (int k = 0; k < kSlots && !rc0; ++k) {   // root is current; root-side images only for the sides that travel packed (a span is copied directly)
            if (i.packed && (rc0 = t_sws.rin[k][root].reserve(ib))) break;
            if (o.packed) rc0 = t_sws.rout[k][root].reserve(ob);
        }
        if (hipSetDevice(dev) != hipSuccess) { r.rc = NDFFT_ERR_HIP; r.err = "hipSetDevice(dev)"; return r; }
        if (rc0) return bad(rc0);
        for (int k = 0; k < kSlots; ++k) {
            if (int rc = t_sws.din[k].reserve(ib)) return bad(rc);
            if (int rc = t_sws.dout[k].reserve(ob)) return bad(rc);
        }
    }
    {
        for (int64_t k = 0; k < K; ++k) {
            const int slot = (int)(k % kSlots);
            const int64_t lo = ext * k / K, hi = ext * (k + 1) / K;
            std::vector<int64_t> si(b.si), so(b.so);
            const char *pin = b.pin; char *pout = b.pout;
            if (d >= 0) {
                si[d] = so[d] = hi - lo;
                pin += lo * b.sti[d] * (int64_t)b.ein; pout += lo * b.sto[d] * (int64_t)b.eout;
            }
            const Side I = side_of(b.si, b.sti, hi - lo), O = side_of(b.so, b.sto, hi - lo);
            std::vector<int64_t> ci, co;
            c_strides(ndim, si.data(), ci); c_strides(ndim, so.data(), co);
            void *rin = t_sws.rin[slot][root].p, *rout = t_sws.rout[slot][root].p, *din = t_sws.din[slot].p, *dout = t_sws.dout[slot].p;
            // -- root: pack (after the scatter of the chunk that used this slot before has read the image)
            if (I.packed) {
                SH_HIP(hipSetDevice(root));
                if (k >= kSlots) SH_HIP(hipStreamWaitEvent(SR.s[0], SD.ev[1][slot], 0));
                const int rc = launch_view_copy(pin, rin, make_desc(ndim, si.data(), b.sti.data()), I.count, b.ein, 0, SR.s[0]);
                if (rc) { r.rc = rc; r.err = ndfft_last_error(); (void)hipSetDevice(dev); goto done; }
                SH_HIP(hipEventRecord(SR.ev[0][slot], SR.s[0]));
                SH_HIP(hipSetDevice(dev));
            }
            // -- scatter over xGMI (after the transform that read this slot's input image)
            if (I.packed) SH_HIP(hipStreamWaitEvent(SD.s[0], SR.ev[0][slot], 0));
            if (k >= kSlots) SH_HIP(hipStreamWaitEvent(SD.s[0], SD.ev[2][slot], 0));
            SH_HIP(hipMemcpyPeerAsync(din, dev, I.packed ? (const char *)rin : pin + I.lo * (int64_t)b.ein, root,
                                      (size_t)(I.packed ? I.count : I.span) * b.ein, SD.s[0]));
            SH_HIP(hipEventRecord(SD.ev[1][slot], SD.s[0]));
            // -- transform (after the gather that read this slot's output image)
            SH_HIP(hipStreamWaitEvent(SD.s[1], SD.ev[1][slot], 0));
            if (k >= kSlots) SH_HIP(hipStreamWaitEvent(SD.s[1], SD.ev[3][slot], 0));
            {
                const char *xin = I.packed ? (const char *)din : (const char *)din - I.lo * (int64_t)b.ein;
                char *xout = O.packed ? (char *)dout : (char *)dout - O.lo * (int64_t)b.eout;
                const int rc = ndfft_exec_device(b.plan, b.op, xin, xout, ndim, si.data(), I.packed ? ci.data() : b.sti.data(), so.data(),
                                                 O.packed ? co.data() : b.sto.data(), b.axis, b.norm, b.scale, SD.s[1]);
                if (rc) { r.rc = rc; r.err = ndfft_last_error(); goto done; }
                if (k == 0) r.path = ndfft_last_path();
            }
            SH_HIP(hipEventRecord(SD.ev[2][slot], SD.s[1]));
            // -- gather over xGMI: a hole-free span straight into the array, anything else into root's image of this slot
            SH_HIP(hipStreamWaitEvent(SD.s[2], SD.ev[2][slot], 0));
            if (O.packed && k >= kSlots) SH_HIP(hipStreamWaitEvent(SD.s[2], SR.ev[3][slot], 0));   // the unpack that read it
            SH_HIP(hipMemcpyPeerAsync(O.packed ? (char *)rout : pout + O.lo * (int64_t)b.eout, root, dout, dev,
                                      (size_t)(O.packed ? O.count : O.span) * b.eout, SD.s[2]));
            SH_HIP(hipEventRecord(SD.ev[3][slot], SD.s[2]));
            // -- root: unpack, element by element of the block's own view
            if (O.packed) {
                SH_HIP(hipSetDevice(root));
                SH_HIP(hipStreamWaitEvent(SR.s[0], SD.ev[3][slot], 0));
                const int rc = launch_view_copy(pout, rout, make_desc(ndim, so.data(), b.sto.data()), O.count, b.eout, 1, SR.s[0]);
                if (rc) { r.rc = rc; r.err = ndfft_last_error(); (void)hipSetDevice(dev); goto done; }
                SH_HIP(hipEventRecord(SR.ev[3][slot], SR.s[0]));
                SH_HIP(hipSetDevice(dev));
            }
        }
    }
done:
    (void)hipSetDevice(dev);
    for (hipStream_t s : SD.s) { const hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess && !r.rc) { r.rc = NDFFT_ERR_HIP; r.err = std::string("hipStreamSynchronize: ") + hipGetErrorString(e); } }
    { const hipError_t e = hipStreamSynchronize(SR.s[0]); if (e != hipSuccess && !r.rc) { r.rc = NDFFT_ERR_HIP; r.err = std::string("hipStreamSynchronize(root): ") + hipGetErrorString(e); } }
#undef SH_HIP
    return r;
}

}  // namespace
}  // namespace ndfft

using namespace ndfft;

extern "C" {

int ndfft_exec_sharded(const ndfft_plan *plan, int op, const void *in, void *out, int ndim, const int64_t *shape_in,
                       const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                       double scale, int n_devices, const int *device_ids) {
    clear_err();
    // the same checks -- and the same panic texts -- as the single-device call, BEFORE anything is started
    bool nothing = false;
    int rc = validate_call(plan, op, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, &nothing);
    if (rc || nothing) return rc;
    if (!in || !out) return fail(NDFFT_ERR_INVALID_ARG, "null array pointer");
    int d;
    std::vector<Block> blocks;
    if ((rc = plan_blocks(ndim, shape_in, stride_in, stride_out, axis, n_devices, device_ids, &d, blocks))) return rc;
    const size_t ein = elem_size(plan->dtype, in_cplx(op)), eout = elem_size(plan->dtype, out_cplx(op));
    std::vector<std::future<TaskResult>> fs;
    for (const Block &b : blocks) {
        std::vector<int64_t> si(shape_in, shape_in + ndim), so(shape_out, shape_out + ndim), sti(stride_in, stride_in + ndim), sto(stride_out, stride_out + ndim);
        const char *pin = (const char *)in; char *pout = (char *)out;
        if (d >= 0) {
            si[d] = so[d] = b.hi - b.lo;
            pin += b.lo * stride_in[d] * (int64_t)ein; pout += b.lo * stride_out[d] * (int64_t)eout;
        }
        fs.push_back(worker_for(b.device).submit([=]() {
            TaskResult r;
            r.rc = ndfft_exec(plan, op, pin, pout, ndim, si.data(), sti.data(), so.data(), sto.data(), axis, norm, scale);
            if (r.rc) r.err = ndfft_last_error(); else r.path = ndfft_last_path();
            return r;
        }));
    }
    return collect(fs);
}

int ndfft_exec_sharded_device(const ndfft_plan *plan, int op, const void *d_in, void *d_out, int ndim, const int64_t *shape_in,
                              const int64_t *stride_in, const int64_t *shape_out, const int64_t *stride_out, int axis, int norm,
                              double scale, int n_devices, const int *device_ids, void *stream) {
    clear_err();
    bool nothing = false;
    int rc = validate_call(plan, op, ndim, shape_in, stride_in, shape_out, stride_out, axis, norm, scale, &nothing);
    if (rc || nothing) return rc;
    if (!d_in || !d_out) return fail(NDFFT_ERR_INVALID_ARG, "null array pointer");
    int root = 0;
    {
        hipPointerAttribute_t a;
        NDFFT_HIP(hipPointerGetAttributes(&a, d_out));
        root = a.device;
    }
    int d;
    std::vector<Block> blocks;
    if ((rc = plan_blocks(ndim, shape_in, stride_in, stride_out, axis, n_devices, device_ids, &d, blocks))) return rc;
    for (const Block &b : blocks) if ((rc = ensure_peer_access(root, b.device))) return rc;
    NDFFT_HIP(hipStreamSynchronize((hipStream_t)stream));   // the input's producers on the caller's stream have finished
    const size_t ein = elem_size(plan->dtype, in_cplx(op)), eout = elem_size(plan->dtype, out_cplx(op));
    std::vector<std::future<TaskResult>> fs;
    for (const Block &b : blocks) {
        RemoteBlock rb;
        rb.plan = plan; rb.op = op; rb.ndim = ndim; rb.axis = axis; rb.norm = norm; rb.scale = scale; rb.d = d; rb.root = root; rb.dev = b.device;
        rb.ein = ein; rb.eout = eout;
        rb.si.assign(shape_in, shape_in + ndim); rb.so.assign(shape_out, shape_out + ndim);
        rb.sti.assign(stride_in, stride_in + ndim); rb.sto.assign(stride_out, stride_out + ndim);
        rb.pin = (const char *)d_in; rb.pout = (char *)d_out;
        if (d >= 0) {
            rb.si[d] = rb.so[d] = b.hi - b.lo;
            rb.pin += b.lo * stride_in[d] * (int64_t)ein; rb.pout += b.lo * stride_out[d] * (int64_t)eout;
        }
        // developer / test switch: NDFFT_SHARD_FORCE_REMOTE=1 sends the root's own blocks through the scatter / gather pipeline too, so that a
        // one-GPU box exercises the pack / unpack kernels, the streams and the events (a peer copy to the same device is a device copy)
        const bool remote = rb.dev != rb.root || sw().shard_force_remote;
        fs.push_back(worker_for(b.device).submit([rb, remote]() { return remote ? run_remote_block(rb) : run_root_block(rb); }));
    }
    return collect(fs);
}

}  // extern "C"
