// TEST INFRASTRUCTURE ONLY: the CPU emulation build has no hiprtc; report "no specialised kernel" --
// except for the Bluestein register kernel (blue_kernel.h), which is instantiated ahead of time here for
// M = 64 and 256 so that its index arithmetic is exercised on the CPU too (the product specialises it with
// hiprtc for any M; see ndrustfft_amd/csrc/jit.hip: launch_jit_blue).
#include "engine.h"
#include "blue_kernel.h"
namespace ndfft {
bool jit_choose(int, int, JitCfg &) { return false; }
void jit_build_twiddles(const JitCfg &, HostTable &) {}
int launch_jit_c2c(int, const JitCfg &, int, const Pow2Args &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
int jit_col_lanes(int, const JitCfg &cfg) { return (cfg.n == 64 || cfg.n == 256) ? 8 : 0; }
template <typename T> int launch_jit_real(int, const JitCfg &, bool, const RealArgs<T> &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
template int launch_jit_real<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_real<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

template <typename K, typename T> __global__ void k_blue_emul(const RealArgs<T> a) { K::run(a); }

template <typename T, int M, int TPL, typename RL, int OP> static int blue_one(bool col, const RealArgs<T> &a, hipStream_t s) {
    if (col) {
        using K = BlueKernel<T, M, TPL, 8, RL, OP, true>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else {
        constexpr int LPB = 256 / TPL;
        using K = BlueKernel<T, M, TPL, LPB, RL, OP, false>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + LPB - 1) / LPB)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    }
    return NDFFT_OK;
}
template <typename T, int M, int TPL, typename RL> static int blue_M(int gop, bool col, const RealArgs<T> &a, hipStream_t s) {
    switch (gop) {
#define B(OP_) case OP_: return blue_one<T, M, TPL, RL, OP_>(col, a, s);
        B(G_C2C_FWD) B(G_C2C_INV) B(G_R2C_EVEN) B(G_R2C_ODD) B(G_C2R_EVEN) B(G_C2R_ODD) B(G_DCT1)
        B(G_DCT2_EVEN) B(G_DCT2_ODD) B(G_DCT3_EVEN) B(G_DCT3_ODD) B(G_DCT4_EVEN) B(G_DCT4_ODD)
#undef B
        default: return NDFFT_ERR_UNSUPPORTED;
    }
}
template <typename T> int launch_jit_blue(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    if (cfg.n == 64) return blue_M<T, 64, 8, RadixList<8, 8>>(gop, col, a, s);        // = RealCfg<64> / RealCfg<256> (kernels_pow2_real.hip)
    if (cfg.n == 256) return blue_M<T, 256, 32, RadixList<8, 8, 4>>(gop, col, a, s);
    return NDFFT_ERR_UNSUPPORTED;
}
template int launch_jit_blue<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_blue<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);
}
