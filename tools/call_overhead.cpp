// tools/call_overhead.cpp -- what one nd* call costs a COMPILED host (the reference's callers are Rust): wall time per ndfft_exec_device call on the reference's
// smallest bench shape (128 x 128 Complex<f64>, axis 0, benches/ndrustfft.rs:6) and on 1024 x 1024, queued back to back, through the C ABI only.
//   build: g++ -O2 -std=c++17 -Iinclude tools/call_overhead.cpp -o tools/call_overhead -Lndrustfft_amd/csrc -lndfft_mi355x -Wl,-rpath,$PWD/ndrustfft_amd/csrc -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "ndfft_mi355x.h"

int main() {
    for (int n : {128, 264, 1024}) {
        ndfft_plan *plan = nullptr;
        if (ndfft_plan_create(NDFFT_KIND_C2C, NDFFT_F64, (size_t)n, &plan)) { printf("plan: %s\n", ndfft_last_error()); return 1; }
        void *x = nullptr, *y = nullptr;
        const size_t bytes = (size_t)n * n * 16;
        ndfft_dev_alloc(&x, bytes); ndfft_dev_alloc(&y, bytes);
        std::vector<double> h((size_t)n * n * 2, 1.0);
        ndfft_dev_upload(x, h.data(), bytes);
        const int64_t shape[2] = {n, n}, stride[2] = {n, 1};
        for (int axis = 0; axis < 2; ++axis) {
            for (int i = 0; i < 50; ++i) ndfft_exec_device(plan, NDFFT_OP_C2C_FWD, x, y, 2, shape, stride, shape, stride, axis, NDFFT_NORM_DEFAULT, 0.0, nullptr);
            ndfft_dev_sync(nullptr);
            const int reps = 2000;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) ndfft_exec_device(plan, NDFFT_OP_C2C_FWD, x, y, 2, shape, stride, shape, stride, axis, NDFFT_NORM_DEFAULT, 0.0, nullptr);
            auto t1 = std::chrono::steady_clock::now();
            ndfft_dev_sync(nullptr);
            auto t2 = std::chrono::steady_clock::now();
            const double issue = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps, total = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
            printf("ndfft %4d x %-4d c128 axis %d: host issues a call every %5.2f us; %5.2f us per call including the GPU (path %s)\n", n, n, axis, issue, total, ndfft_last_path());
        }
        ndfft_dev_free(x); ndfft_dev_free(y); ndfft_plan_destroy(plan);
    }
    return 0;
}
