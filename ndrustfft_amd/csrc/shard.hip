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
Worker &worker_for(int device) {
    static std::map<int, Worker *> *pool = new std::map<int, Worker *>;
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

// per-worker-thread (= per device) buffers for the blocks of a device-resident array
struct ShardBuf {
    void *p = nullptr; size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return NDFFT_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        NDFFT_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return NDFFT_OK;
    }
    ~ShardBuf() { if (p) (void)hipFree(p); }
};
thread_local ShardBuf t_shard_in, t_shard_out;

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
    NDFFT_HIP(hipStreamSynchronize((hipStream_t)stream));   // the input's producers on the caller's stream have finished
    const size_t ein = elem_size(plan->dtype, in_cplx(op)), eout = elem_size(plan->dtype, out_cplx(op));
    std::vector<std::future<TaskResult>> fs;
    for (const Block &b : blocks) {
        std::vector<int64_t> si(shape_in, shape_in + ndim), so(shape_out, shape_out + ndim), sti(stride_in, stride_in + ndim), sto(stride_out, stride_out + ndim);
        const char *pin = (const char *)d_in; char *pout = (char *)d_out;
        if (d >= 0) {
            si[d] = so[d] = b.hi - b.lo;
            pin += b.lo * stride_in[d] * (int64_t)ein; pout += b.lo * stride_out[d] * (int64_t)eout;
        }
        const int dev = b.device;
        fs.push_back(worker_for(dev).submit([=]() {
            TaskResult r;
            auto hipfail = [&r](hipError_t e, const char *what) { r.rc = NDFFT_ERR_HIP; r.err = std::string(what) + ": " + hipGetErrorString(e); return r; };
            if (dev == root) {   // the block is already where it runs
                r.rc = ndfft_exec_device(plan, op, pin, pout, ndim, si.data(), sti.data(), so.data(), sto.data(), axis, norm, scale, nullptr);
                if (!r.rc) { const hipError_t e = hipStreamSynchronize(nullptr); if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize"); }
                if (r.rc) r.err = ndfft_last_error(); else r.path = ndfft_last_path();
                return r;
            }
            // xGMI scatter of the block's span -> transform on this device -> xGMI gather of the result span
            int64_t ilo, ihi, icnt, olo, ohi, ocnt;
            span_of(ndim, si.data(), sti.data(), ilo, ihi, icnt);
            span_of(ndim, so.data(), sto.data(), olo, ohi, ocnt);
            const size_t ibytes = (size_t)(ihi - ilo + 1) * ein, obytes = (size_t)(ohi - olo + 1) * eout;
            if ((r.rc = t_shard_in.reserve(ibytes)) || (r.rc = t_shard_out.reserve(obytes))) { r.err = ndfft_last_error(); return r; }
            hipError_t e = hipMemcpyPeerAsync(t_shard_in.p, dev, pin + ilo * (int64_t)ein, root, ibytes, nullptr);
            if (e != hipSuccess) return hipfail(e, "hipMemcpyPeerAsync (scatter)");
            if ((int64_t)(ohi - olo + 1) != ocnt) {   // an output view with holes: carry the holes through the round trip
                e = hipMemcpyPeerAsync(t_shard_out.p, dev, pout + olo * (int64_t)eout, root, obytes, nullptr);
                if (e != hipSuccess) return hipfail(e, "hipMemcpyPeerAsync (holes)");
            }
            r.rc = ndfft_exec_device(plan, op, (const char *)t_shard_in.p - ilo * (int64_t)ein, (char *)t_shard_out.p - olo * (int64_t)eout, ndim,
                                     si.data(), sti.data(), so.data(), sto.data(), axis, norm, scale, nullptr);
            if (r.rc) { r.err = ndfft_last_error(); (void)hipStreamSynchronize(nullptr); return r; }
            r.path = ndfft_last_path();
            e = hipMemcpyPeerAsync(pout + olo * (int64_t)eout, root, t_shard_out.p, dev, obytes, nullptr);
            if (e != hipSuccess) return hipfail(e, "hipMemcpyPeerAsync (gather)");
            e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return hipfail(e, "hipStreamSynchronize");
            return r;
        }));
    }
    return collect(fs);
}

}  // extern "C"
