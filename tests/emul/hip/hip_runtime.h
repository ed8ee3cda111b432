// TEST INFRASTRUCTURE ONLY -- a single-threaded, fiber-based emulation of the handful of HIP
// constructs the kernels in ndrustfft_amd/csrc use, so that the UNMODIFIED .hip sources can be
// compiled with g++ (include path trick: -I tests/emul shadows <hip/hip_runtime.h>) and their index
// arithmetic debugged / sanitised on the CPU-only build container.  Every workgroup runs as
// blockDim.x fibers on one OS thread; __syncthreads() yields to the next fiber.
// Nothing in the product (ndrustfft_amd/, include/) ever includes or loads this: the emulated
// library is built into tests/emul/_build and bound only by tests/test_emul_*.py.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <algorithm>

// wave_kernel.h's cross-lane bit swap, emulated with block barriers (emul_runtime.cpp)
namespace ndfft { void emul_wave_swap(unsigned &lo, unsigned &hi, int tb); }
#define NDFFT_WAVE_SWAP_OVERRIDE(lo, hi, tb) ::ndfft::emul_wave_swap(lo, hi, tb)
#define NDFFT_LDS_BARRIER_OVERRIDE() __syncthreads()
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__
#define __constant__
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct float2 { float x, y; };
struct double2 { double x, y; };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct emul_idx { unsigned x, y, z; };
extern emul_idx threadIdx, blockIdx, blockDim, gridDim;
using std::min; using std::max;

typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };

inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return 0; }
// several fake devices (EMUL_DEVICES, default 1): the current device is per host thread, as in HIP
extern thread_local int emul_cur_device;
int emul_device_count();
inline hipError_t hipGetDevice(int *d) { *d = emul_cur_device; return 0; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= emul_device_count()) return 101; emul_cur_device = d; return 0; }
inline hipError_t hipGetDeviceCount(int *n) { *n = emul_device_count(); return 0; }
// "device" allocations carry 256-byte guard zones that are checked after every kernel launch (emul_runtime.cpp)
hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
// host <-> device copies must name memory of the CURRENT device: the library keeps one workspace per device, and a
// staging buffer of device 0 showing up while device 1 is current is exactly the bug class this catches
void emul_check_current(const void *devptr, const char *what);
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) {
    if (k == hipMemcpyHostToDevice) emul_check_current(d, "hipMemcpy H2D");
    if (k == hipMemcpyDeviceToHost) emul_check_current(s, "hipMemcpy D2H");
    memcpy(d, s, n); return 0;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
// pinned-host pipeline of ndfft_exec: never taken in the emulation (no pointer is ever "pinned"), stubs only
typedef void *hipEvent_t;
enum { hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipMemoryTypeHost = 1 };
struct hipPointerAttribute_t { int type; int device; };
int emul_device_of(const void *p);   // device an address was hipMalloc'd on, -1 if not device memory
bool emul_host_registered(const void *p);     // inside a range registered with hipHostRegister (emul_runtime.cpp)
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    const int d = emul_device_of(p);
    if (d < 0 && emul_host_registered(p)) { a->type = hipMemoryTypeHost; a->device = 0; return 0; }
    if (d < 0) return 1;                       // plain host memory: not "pinned" unless registered
    a->type = 2; a->device = d;
    return 0;
}
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
// peer copies check that each pointer lies in an allocation made on the device it is claimed to be on
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t);
int emul_device_of(const void *p);   // device an address was hipMalloc'd on, -1 if not device memory
// peer access: every fake device can map every other one, but a peer copy ABORTS unless access was enabled in both
// directions first (on the MI355X such a copy would silently be staged through host memory instead of xGMI)
enum { hipErrorPeerAccessAlreadyEnabled = 704 };
hipError_t hipDeviceCanAccessPeer(int *can, int dev, int peer);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
bool emul_copy_should_fail(const void *a, const void *b);   // emul_runtime.cpp: injected failures of copies on registered host ranges
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { if (emul_copy_should_fail(d, s)) return 1; return hipMemcpy(d, s, n, k); }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void *p) { free(p); return 0; }
enum { hipHostRegisterDefault = 0 };
// registrations are tracked (emul_runtime.cpp) so that the registration cache of ndfft_exec and the pinned pipeline behind it run on the CPU container;
// overlapping an existing registration fails, as on the real runtime
hipError_t hipHostRegister(void *p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void *p);
extern "C" size_t emul_host_registered_bytes();
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return 0; }

void __syncthreads();
// bit-pattern casts used by the wavefront kernel
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __double2loint(double d) { unsigned long long u; memcpy(&u, &d, 8); return (int)(unsigned)(u & 0xffffffffu); }
inline int __double2hiint(double d) { unsigned long long u; memcpy(&u, &d, 8); return (int)(unsigned)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double d; memcpy(&d, &u, 8); return d; }

namespace emul {
// runs fn(arg) as grid x block fibers, with lds_bytes of "LDS" per block
void launch(void (*fn)(void *), void *arg, dim3 grid, dim3 block, size_t lds_bytes);
}  // namespace emul

#include <tuple>
#include <utility>
template <typename... KA, typename... A>
inline void hipLaunchKernelGGL(void (*k)(KA...), dim3 grid, dim3 block, size_t lds, hipStream_t, A... a) {
    struct Pack { void (*k)(KA...); std::tuple<typename std::decay<KA>::type...> args; };
    Pack p{k, std::tuple<typename std::decay<KA>::type...>(a...)};
    emul::launch([](void *q) { Pack *s = (Pack *)q; std::apply(s->k, s->args); }, &p, grid, block, lds);
}
