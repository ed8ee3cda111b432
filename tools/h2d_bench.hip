// How fast can 256 MiB go host->device and 256 MiB device->host from PAGEABLE memory, and do the two
// directions overlap?  (design input for the host-array path of ndfft_exec)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 256u << 20;
    char *hin = (char *)malloc(N), *hout = (char *)malloc(N);
    memset(hin, 1, N); memset(hout, 2, N);
    char *din, *dout; CK(hipMalloc(&din, N)); CK(hipMalloc(&dout, N));
    CK(hipMemset(dout, 3, N)); CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy(din, hin, N, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(hout, dout, N, hipMemcpyDeviceToHost)); double t2 = now();
        printf("A sequential pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9, (t2 - t1) * 1e3, N / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {   // two host threads, one per direction
        double t0 = now();
        std::thread a([&] { CK(hipSetDevice(0)); CK(hipMemcpy(din, hin, N, hipMemcpyHostToDevice)); });
        std::thread b([&] { CK(hipSetDevice(0)); CK(hipMemcpy(hout, dout, N, hipMemcpyDeviceToHost)); });
        a.join(); b.join();
        double t1 = now();
        printf("C two threads, both directions at once: %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, 2.0 * N / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {   // chunks from two threads (what a pipelined exec would do)
        const int C = 8; const size_t cs = N / C;
        double t0 = now();
        std::thread a([&] { CK(hipSetDevice(0)); for (int c = 0; c < C; ++c) CK(hipMemcpy(din + c * cs, hin + c * cs, cs, hipMemcpyHostToDevice)); });
        std::thread b([&] { CK(hipSetDevice(0)); for (int c = 0; c < C; ++c) CK(hipMemcpy(hout + c * cs, dout + c * cs, cs, hipMemcpyDeviceToHost)); });
        a.join(); b.join();
        double t1 = now();
        printf("C' two threads x 8 chunks: %.2f ms (%.1f GB/s aggregate)\n", (t1 - t0) * 1e3, 2.0 * N / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); CK(hipHostRegister(hin, N, hipHostRegisterDefault)); CK(hipHostRegister(hout, N, hipHostRegisterDefault)); double t1 = now();
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        double t2 = now();
        CK(hipMemcpyAsync(din, hin, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(hout, dout, N, hipMemcpyDeviceToHost, s2));
        CK(hipDeviceSynchronize()); double t3 = now();
        CK(hipHostUnregister(hin)); CK(hipHostUnregister(hout)); double t4 = now();
        printf("B register %.2f ms, both directions async %.2f ms (%.1f GB/s aggregate), unregister %.2f ms\n", (t1 - t0) * 1e3, (t3 - t2) * 1e3, 2.0 * N / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
        CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
    }
    return 0;
}
