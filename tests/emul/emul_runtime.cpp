// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tests/emul/hip/hip_runtime.h.
#include <ucontext.h>

#include <cstdio>
#include <vector>

#include "hip/hip_runtime.h"

emul_idx threadIdx, blockIdx, blockDim, gridDim;

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

void __syncthreads() {
    // yield; the scheduler resumes fibers round-robin, so returning here means every live fiber
    // of the block has reached a barrier (or finished) since we left
    swapcontext(&g_f[g_cur].ctx, &g_main);
}

void emul::launch(void (*fn)(void *), void *arg, dim3 grid, dim3 block, size_t lds_bytes) {
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
    }
}
