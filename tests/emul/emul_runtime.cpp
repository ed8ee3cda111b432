// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emul/hip/hip_runtime.h.
#include <ucontext.h>

#include <atomic>

#include <cstdio>
#include <map>
#include <mutex>
#include <vector>

#include "hip/hip_runtime.h"

emul_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local int emul_cur_device = 0;
int emul_device_count() {
    static const int n = [] { const char *e = getenv("EMUL_DEVICES"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    return n;
}

namespace ndfft { alignas(16) char smem[160 * 1024 + 64]; }   // `extern __shared__ char smem[]` of the kernels

namespace {
struct Fiber { ucontext_t ctx; std::vector<char> stack; bool done = false; };
std::vector<Fiber> g_f;
ucontext_t g_main;
int g_cur = -1;
void (*g_fn)(void *);
void *g_arg;

void entry() {
    g_fn(g_arg);
    g_f[g_cur].done = true;
    swapcontext(&g_f[g_cur].ctx, &g_main);
}
}  // namespace

// ---- guarded "device" memory: kernels must not write outside what the host allocated ----------------------
namespace {
constexpr size_t kGuard = 256;
// (never destroyed: plans may be freed from Python finalisers after static destructors have run)
std::map<char *, size_t> &g_allocs = *new std::map<char *, size_t>;      // user pointer -> user size
std::map<char *, int> &g_alloc_dev = *new std::map<char *, int>;         // user pointer -> device it was allocated on
std::mutex &g_launch_mu = *new std::mutex;                               // the fiber scheduler is single-threaded
std::mutex &g_alloc_mu = *new std::mutex;
void check_guards(const char *when) {
    std::lock_guard<std::mutex> g(g_alloc_mu);
    for (auto &kv : g_allocs) {
        const unsigned char *lo = (const unsigned char *)kv.first - kGuard, *hi = (const unsigned char *)kv.first + kv.second;
        for (size_t i = 0; i < kGuard; ++i)
            if (lo[i] != 0xC3 || hi[i] != 0xC3) {
                fprintf(stderr, "emul: %s: write outside a device allocation of %zu bytes (%s guard, offset %zu)\n", when, kv.second,
                        lo[i] != 0xC3 ? "lower" : "upper", i);
                abort();
            }
    }
}
}  // namespace
hipError_t hipMalloc(void **p, size_t n) {
    char *raw = (char *)malloc(n + 2 * kGuard);
    if (!raw) return 2;
    memset(raw, 0xC3, kGuard); memset(raw + kGuard + n, 0xC3, kGuard);
    std::lock_guard<std::mutex> g(g_alloc_mu);
    g_allocs[raw + kGuard] = n;
    g_alloc_dev[raw + kGuard] = emul_cur_device;
    *p = raw + kGuard;
    return 0;
}
hipError_t hipFree(void *p) {
    if (!p) return 0;
    check_guards("hipFree");
    std::lock_guard<std::mutex> g(g_alloc_mu);
    g_allocs.erase((char *)p);
    g_alloc_dev.erase((char *)p);
    free((char *)p - kGuard);
    return 0;
}

int emul_device_of(const void *p) {
    std::lock_guard<std::mutex> g(g_alloc_mu);
    auto it = g_allocs.upper_bound((char *)p);
    if (it == g_allocs.begin()) return -1;
    --it;
    if ((const char *)p >= it->first + it->second + (it->second == 0)) return -1;
    return g_alloc_dev[it->first];
}
void emul_check_current(const void *devptr, const char *what) {
    const int d = emul_device_of(devptr);
    if (d >= 0 && d != emul_cur_device) {
        fprintf(stderr, "emul: %s: device memory of device %d used while device %d is current\n", what, d, emul_cur_device);
        abort();
    }
}
namespace { std::map<std::pair<int, int>, bool> &g_peer = *new std::map<std::pair<int, int>, bool>; std::mutex &g_peer_mu = *new std::mutex; }
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer) {
    if (dev < 0 || peer < 0 || dev >= emul_device_count() || peer >= emul_device_count()) return 101;
    *can = dev != peer;
    return 0;
}
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) {
    if (peer < 0 || peer >= emul_device_count() || peer == emul_cur_device) return 101;
    std::lock_guard<std::mutex> g(g_peer_mu);
    bool &on = g_peer[{emul_cur_device, peer}];
    if (on) return hipErrorPeerAccessAlreadyEnabled;
    on = true;
    return 0;
}
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t) {
    if (dst_dev != src_dev) {
        std::lock_guard<std::mutex> g(g_peer_mu);
        if (!g_peer[{dst_dev, src_dev}] || !g_peer[{src_dev, dst_dev}]) {
            fprintf(stderr, "emul: hipMemcpyPeerAsync between devices %d and %d without hipDeviceEnablePeerAccess in both directions\n", dst_dev, src_dev);
            abort();
        }
    }
    if (dst_dev != emul_cur_device && src_dev != emul_cur_device) {
        fprintf(stderr, "emul: hipMemcpyPeerAsync: neither end (%d, %d) is the current device %d\n", dst_dev, src_dev, emul_cur_device);
        abort();
    }
    if (emul_device_of(dst) != dst_dev || emul_device_of(src) != src_dev) {
        fprintf(stderr, "emul: hipMemcpyPeerAsync: pointer is not on the device it is claimed to be on (dst %d vs %d, src %d vs %d)\n",
                emul_device_of(dst), dst_dev, emul_device_of(src), src_dev);
        abort();
    }
    memcpy(dst, src, n);
    return 0;
}

// ---- hipHostRegister: ranges of ordinary host memory the library has "pinned" ------------------------------------------------
namespace { std::map<char *, size_t> &g_reg = *new std::map<char *, size_t>; std::mutex &g_reg_mu = *new std::mutex; }
hipError_t hipHostRegister(void *p, size_t n, unsigned) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (auto &kv : g_reg) if ((char *)p < kv.first + kv.second && kv.first < (char *)p + n) return 712;   // hipErrorHostMemoryAlreadyRegistered
    g_reg[(char *)p] = n;
    return 0;
}
hipError_t hipHostUnregister(void *p) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    return g_reg.erase((char *)p) ? 0 : 713;
}
// test hook: the next n hipMemcpyAsync calls that touch a registered host range fail with "invalid argument" -- what the MI355X runtime does when a
// registration has outlived its array (tests of the registration cache's retry through the bounce buffers)
namespace { std::atomic<int> g_fail_reg_copies{0}; }
extern "C" void emul_fail_registered_copies(int n) { g_fail_reg_copies = n; }
bool emul_copy_should_fail(const void *a, const void *b) {
    if (g_fail_reg_copies.load() <= 0) return false;
    if (!emul_host_registered(a) && !emul_host_registered(b)) return false;
    return g_fail_reg_copies.fetch_sub(1) > 0;
}
bool emul_host_registered(const void *p) {
    std::lock_guard<std::mutex> g(g_reg_mu);
    for (auto &kv : g_reg) if ((const char *)p >= kv.first && (const char *)p < kv.first + kv.second) return true;
    return false;
}
extern "C" int emul_is_registered(const void *p) { return emul_host_registered(p) ? 1 : 0; }
extern "C" size_t emul_host_registered_bytes() {
    std::lock_guard<std::mutex> g(g_reg_mu);
    size_t b = 0;
    for (auto &kv : g_reg) b += kv.second;
    return b;
}

void __syncthreads() {
    // yield; the scheduler resumes fibers round-robin, so returning here means every live fiber
    // of the block has reached a barrier (or finished) since we left
    swapcontext(&g_f[g_cur].ctx, &g_main);
}

// wave_kernel.h's bit swap between thread bit `tb` and a register pair (lo, hi), emulated with two block barriers:
// threads with the bit clear end with (own lo, partner's lo), threads with it set with (partner's hi, own hi)
namespace ndfft { void emul_wave_swap(unsigned &lo, unsigned &hi, int tb); }
void ndfft::emul_wave_swap(unsigned &lo, unsigned &hi, int tb) {
    static unsigned slo[1024], shi[1024];
    const unsigned t = threadIdx.x;
    slo[t] = lo; shi[t] = hi;
    __syncthreads();
    const unsigned p = t ^ (1u << tb);
    if ((t >> tb) & 1u) lo = shi[p]; else hi = slo[p];
    __syncthreads();
}

void emul::launch(void (*fn)(void *), void *arg, dim3 grid, dim3 block, size_t lds_bytes) {
    std::lock_guard<std::mutex> launch_guard(g_launch_mu);   // host threads (shared handlers, sharded exec) take turns
    if (block.x > 1024 || block.x == 0) { fprintf(stderr, "emul: invalid workgroup size %u (a gfx950 workgroup has at most 1024 threads)\n", block.x); abort(); }
    if (lds_bytes > 160 * 1024) { fprintf(stderr, "emul: LDS request %zu > 160 KiB\n", lds_bytes); abort(); }
    g_fn = fn; g_arg = arg;
    blockDim = {block.x, 1, 1}; gridDim = {grid.x, grid.y, grid.z};
    const size_t kStack = 256 * 1024;
    if (g_f.size() < block.x) g_f.resize(block.x);
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned b = 0; b < grid.x; ++b) {
        memset(ndfft::smem, 0xA5, sizeof ndfft::smem);   // poison: uninitialised LDS reads show up
        for (unsigned t = 0; t < block.x; ++t) {
            Fiber &f = g_f[t];
            if (f.stack.size() != kStack) f.stack.resize(kStack);
            f.done = false;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack.data();
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = &g_main;
            makecontext(&f.ctx, entry, 0);
        }
        bool any = true;
        while (any) {
            any = false;
            for (unsigned t = 0; t < block.x; ++t) {
                if (g_f[t].done) continue;
                any = true;
                g_cur = (int)t;
                threadIdx = {t, 0, 0}; blockIdx = {b, by, bz};
                swapcontext(&g_main, &g_f[t].ctx);
            }
        }
        // a workgroup may only touch the dynamic LDS it asked for: everything past it must still be poison
        for (size_t i = lds_bytes; i < sizeof ndfft::smem; ++i)
            if ((unsigned char)ndfft::smem[i] != 0xA5) {
                fprintf(stderr, "emul: block (%u,%u,%u) wrote LDS byte %zu, beyond the %zu bytes requested at launch\n", b, by, bz, i, lds_bytes);
                abort();
            }
    }
    check_guards("kernel launch");
}
