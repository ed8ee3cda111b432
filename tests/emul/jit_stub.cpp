// TEST INFRASTRUCTURE ONLY: the CPU emulation build has no hiprtc; report "no specialised kernel" --
// except for the Bluestein register kernel (blue_kernel.h), which is instantiated ahead of time here for
// M = 64 and 256 so that its index arithmetic is exercised on the CPU too (the product specialises it with
// hiprtc for any M; see ndrustfft_amd/csrc/jit.hip: launch_jit_blue).
#include "engine.h"
#include "blue_kernel.h"
#include "reg_kernel.h"
#include "rader_kernel.h"
#include "plain_kernel.h"
namespace ndfft {
// two PARTIAL-round configurations of the C2C row kernel (pow2_kernel.h: slots / full) instantiated ahead of
// time, so that the predicated passes are exercised on the CPU: 264 = 11.8.3 on 12 threads, 210 = 7.6.5 on 14
using PRL264 = RadixList<11, 8, 3>;
using PRL210 = RadixList<7, 6, 5>;
using PRL45 = RadixList<9, 5>;
bool jit_choose(int, int n, JitCfg &cfg, bool allow_partial) {
    if (allow_partial && n == 90) { cfg.n = 90; cfg.partial = true; cfg.vec = 1; cfg.tpl = 9; cfg.e = 18; cfg.radix = {10, 9}; cfg.lpb = 7; return true; }   // DCT-IV with n = 45: inner FFT 2n
    if (allow_partial && n == 45) { cfg.n = 45; cfg.partial = true; cfg.vec = 1; cfg.tpl = 5; cfg.e = 10; cfg.radix = {9, 5}; cfg.lpb = 12; return true; }   // odd-n real ops (plain_kernel.h)
    if (!allow_partial || (n != 264 && n != 210)) return false;
    cfg.n = n; cfg.partial = true; cfg.vec = 1;
    if (n == 264) { cfg.tpl = 12; cfg.radix = {11, 8, 3}; cfg.lpb = 21; } else { cfg.tpl = 14; cfg.radix = {7, 6, 5}; cfg.lpb = 18; }
    return true;
}
bool jit_choose_real(int dtype, int F, JitCfg &cfg) { return jit_choose(dtype, F, cfg, true); }
bool jit_choose_col(int, int, const JitCfg &, JitCfg &) { return false; }   // (no hiprtc in the emulation: no separate column recipe)
bool jit_c2c_row_vec(int, JitCfg &) { return false; }
// the CPU build keeps Bluestein on powers of two (M = 64 and 256 are instantiated below) except for F = 263, which gets the smooth length 550 = 11.10.5 on 55 threads
// (a PARTIAL first round: 50 butterflies of radix 11) like the product's blue_pick_len would choose
int blue_pick_len(int, int F, int m_pow2) { return F == 263 ? 550 : m_pow2; }
bool blue_plan_cfg(int, int M, JitCfg &cfg) {
    if (M != 550) return false;
    cfg = JitCfg(); cfg.n = 550; cfg.tpl = 55; cfg.e = 11; cfg.radix = {11, 10, 5}; cfg.partial = true; cfg.vec = 1; cfg.row_lpb = 4;
    return true;
}
void jit_build_twiddles(const JitCfg &cfg, HostTable &out) { if (cfg.n == 264) build_tw<PRL264>(out); else if (cfg.n == 210) build_tw<PRL210>(out); else if (cfg.n == 45) build_tw<PRL45>(out); else if (cfg.n == 90) build_tw<RadixList<10, 9>>(out); }
template <typename K> __global__ void k_c2c_emul(const Pow2Args a) { K::run(a); }
template <typename T, int N, int TPL, int LPB, typename RL> static int c2c_one(const Pow2Args &a, hipStream_t s) {
    using K = Pow2Kernel<T, N, TPL, LPB, true, RL, 0, 1, 1, 1>;
    hipLaunchKernelGGL((k_c2c_emul<K>), dim3((unsigned)((a.nlanes + LPB - 1) / LPB)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    return NDFFT_OK;
}
int launch_jit_c2c(int dtype, const JitCfg &cfg, int, const Pow2Args &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    if (cfg.n == 264) return dtype == NDFFT_F32 ? c2c_one<float, 264, 12, 21, PRL264>(a, s) : c2c_one<double, 264, 12, 21, PRL264>(a, s);
    if (cfg.n == 210) return dtype == NDFFT_F32 ? c2c_one<float, 210, 14, 18, PRL210>(a, s) : c2c_one<double, 210, 14, 18, PRL210>(a, s);
    return NDFFT_ERR_UNSUPPORTED;
}
int jit_col_lanes(int, const JitCfg &cfg, bool) { return (cfg.n == 64 || cfg.n == 256 || cfg.n == 264 || cfg.n == 210 || cfg.n == 45 || cfg.n == 550 || cfg.n == 90) ? 8 : 0; }
// the same two partial-round configurations on the real-op / column kernel (pow2_real.h)
template <typename K, typename T> __global__ void k_real_emul(const RealArgs<T> a) { K::run(a); }
template <typename T, int F, int TPL, int LPBR, typename RL, int OP> static int real_one(bool col, const RealArgs<T> &a, hipStream_t s) {
    if (col) {
        using K = RealPow2Kernel<T, F, TPL, 8, RL, OP, true>;
        hipLaunchKernelGGL((k_real_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else if constexpr (OP != G_C2C_FWD && OP != G_C2C_INV) {
        using K = RealPow2Kernel<T, F, TPL, LPBR, RL, OP, false>;
        hipLaunchKernelGGL((k_real_emul<K, T>), dim3((unsigned)((a.nlanes + LPBR - 1) / LPBR)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else {
        return NDFFT_ERR_UNSUPPORTED;
    }
    return NDFFT_OK;
}
template <typename T, int F, int TPL, int LPBR, typename RL> static int real_F(int gop, bool col, const RealArgs<T> &a, hipStream_t s) {
    switch (gop) {
#define B(OP_) case OP_: return real_one<T, F, TPL, LPBR, RL, OP_>(col, a, s);
        B(G_C2C_FWD) B(G_C2C_INV) B(G_R2C_EVEN) B(G_C2R_EVEN) B(G_DCT1) B(G_DCT2_EVEN) B(G_DCT3_EVEN) B(G_DCT4_EVEN)
#undef B
        default: return NDFFT_ERR_UNSUPPORTED;
    }
}
template <typename T> int launch_jit_real(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    if (cfg.n == 264) return real_F<T, 264, 12, 21, PRL264>(gop, col, a, s);
    if (cfg.n == 210) return real_F<T, 210, 14, 18, PRL210>(gop, col, a, s);
    return NDFFT_ERR_UNSUPPORTED;
}
template int launch_jit_real<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_real<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

template <typename K, typename T> __global__ void k_blue_emul(const RealArgs<T> a) { K::run(a); }

// odd-n real ops with the smooth inner FFT 45 = 9.5 (plain_kernel.h; the product specialises it with hiprtc, jit.hip: launch_jit_plain)
template <typename T, int OP> static int plain_one(bool col, const RealArgs<T> &a, hipStream_t s) {
    if (col) {
        using K = PlainRealKernel<T, 45, 5, 8, PRL45, OP, true>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else {
        using K = PlainRealKernel<T, 45, 5, 12, PRL45, OP, false>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 11) / 12)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    }
    return NDFFT_OK;
}
template <typename T> static int plain_dct4(bool col, const RealArgs<T> &a, hipStream_t s) {
    if (col) {
        using K = PlainRealKernel<T, 90, 9, 8, RadixList<10, 9>, G_DCT4_ODD, true>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else {
        using K = PlainRealKernel<T, 90, 9, 7, RadixList<10, 9>, G_DCT4_ODD, false>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 6) / 7)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    }
    return NDFFT_OK;
}
template <typename T> int launch_jit_plain(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    if (cfg.n == 90 && gop == G_DCT4_ODD) return plain_dct4<T>(col, a, s);
    if (cfg.n != 45) return NDFFT_ERR_UNSUPPORTED;
    switch (gop) {
        case G_R2C_ODD: return plain_one<T, G_R2C_ODD>(col, a, s);
        case G_C2R_ODD: return plain_one<T, G_C2R_ODD>(col, a, s);
        case G_DCT2_ODD: return plain_one<T, G_DCT2_ODD>(col, a, s);
        case G_DCT3_ODD: return plain_one<T, G_DCT3_ODD>(col, a, s);
        default: return NDFFT_ERR_UNSUPPORTED;
    }
}
template int launch_jit_plain<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_plain<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

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
    if (cfg.n == 550) return blue_M<T, 550, 55, RadixList<11, 10, 5>>(gop, col, a, s);
    return NDFFT_ERR_UNSUPPORTED;
}
template int launch_jit_blue<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_blue<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

// Rader / Good-Thomas kernel (rader_kernel.h; the product specialises it with hiprtc, jit.hip: launch_jit_rader), ahead of time for
// F = 31 and 97 (primes), 62 = 2 x 31 and 511 = 7 x 73 (cofactor butterflies), 306 = (6 x 3) x 17 (two-factor cofactor, one-pass FFT_16), 103 (102 = 17 x 6: a radix-17 pass);
// FFT_30 runs with a PARTIAL second pass (6.5 on 5 threads)
bool rader_choose(int, int F, RaderCfg &rc, bool dct1_slot) {
    rc.fft.vec = 1; rc.fft.lpb = 1;
    rc.sym = dct1_slot && (F == 511 || F == 31);      // DCT-I with an odd cofactor: the symmetric form (4 of the 7 Rader transforms); F = 31 prime: the half-length convolution
    if (F == 31 && rc.sym) { rc.p = 31; rc.mc = 1; rc.mc1 = 1; rc.fft.n = 15; rc.fft.tpl = 3; rc.fft.e = 6; rc.fft.radix = {5, 3}; rc.fft.partial = true; return true; }
    if (F == 31 || F == 62) { rc.p = 31; rc.mc = F / 31; rc.mc1 = rc.mc; rc.fft.n = 30; rc.fft.tpl = 5; rc.fft.e = 10; rc.fft.radix = {6, 5}; rc.fft.partial = true; return true; }
    if (F == 97) { rc.p = 97; rc.mc = 1; rc.fft.n = 96; rc.fft.tpl = 8; rc.fft.e = 12; rc.fft.radix = {6, 4, 4}; return true; }
    if (F == 511) { rc.p = 73; rc.mc = 7; rc.mc1 = 7; rc.fft.n = 72; rc.fft.tpl = 6; rc.fft.e = 12; rc.fft.radix = {6, 4, 3}; return true; }
    if (F == 103) { rc.p = 103; rc.mc = 1; rc.fft.n = 102; rc.fft.tpl = 6; rc.fft.e = 18; rc.fft.radix = {17, 6}; rc.fft.partial = true; return true; }   // p - 1 = 17 x 6: a radix-17 pass
    if (F == 306) { rc.p = 17; rc.mc = 18; rc.mc1 = 6; rc.mc2 = 3; rc.fft.n = 16; rc.fft.tpl = 1; rc.fft.e = 16; rc.fft.radix = {16}; return true; }   // two-factor cofactor, one-pass FFT_16
    return false;
}
int rader_col_lanes(int, const RaderCfg &) { return 8; }
template <typename T, int P, int MC1, int MC2, int TPL, typename RL, int OP> static int rader_one(bool col, const RealArgs<T> &a, hipStream_t s) {
    constexpr int MC = MC1 * MC2;
    if (col) {
        using K = RaderKernel<T, P, MC1, MC2, TPL, 8, RL, OP, true>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    } else {
        constexpr int LPB = 256 / (TPL * MC);
        using K = RaderKernel<T, P, MC1, MC2, TPL, LPB, RL, OP, false>;
        hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + LPB - 1) / LPB)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    }
    return NDFFT_OK;
}
template <typename T, int P, int MC1, int MC2, int TPL, typename RL> static int rader_P(int gop, bool col, const RealArgs<T> &a, hipStream_t s) {
    switch (gop) {
#define B(OP_) case OP_: return rader_one<T, P, MC1, MC2, TPL, RL, OP_>(col, a, s);
        B(G_C2C_FWD) B(G_C2C_INV) B(G_R2C_EVEN) B(G_R2C_ODD) B(G_C2R_EVEN) B(G_C2R_ODD) B(G_DCT1)
        B(G_DCT2_EVEN) B(G_DCT2_ODD) B(G_DCT3_EVEN) B(G_DCT3_ODD) B(G_DCT4_EVEN) B(G_DCT4_ODD)
#undef B
        default: return NDFFT_ERR_UNSUPPORTED;
    }
}
template <typename T> int launch_jit_rader(int gop, const RaderCfg &rc, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    { const char *e = getenv("NDFFT_RADER"); if (e && e[0] == '0') return NDFFT_ERR_UNSUPPORTED; }
    if (rc.sym && rc.p == 31) {
        if (gop != G_DCT1 || rc.mc != 1) return NDFFT_ERR_UNSUPPORTED;
        if (col) {
            using K = RaderKernel<T, 31, 1, 1, 3, 8, RadixList<5, 3>, G_DCT1, true, true>;
            hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
        } else {
            using K = RaderKernel<T, 31, 1, 1, 3, 16, RadixList<5, 3>, G_DCT1, false, true>;
            hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 15) / 16)), dim3(K::THREADS), K::LDS_BYTES, s, a);
        }
        return NDFFT_OK;
    }
    if (rc.sym) {
        if (gop != G_DCT1 || rc.p != 73 || rc.mc != 7) return NDFFT_ERR_UNSUPPORTED;
        if (col) {
            using K = RaderKernel<T, 73, 7, 1, 6, 8, RadixList<6, 4, 3>, G_DCT1, true, true>;
            hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 7) / 8)), dim3(K::THREADS), K::LDS_BYTES, s, a);
        } else {
            using K = RaderKernel<T, 73, 7, 1, 6, 10, RadixList<6, 4, 3>, G_DCT1, false, true>;
            hipLaunchKernelGGL((k_blue_emul<K, T>), dim3((unsigned)((a.nlanes + 9) / 10)), dim3(K::THREADS), K::LDS_BYTES, s, a);
        }
        return NDFFT_OK;
    }
    if (rc.p == 31 && rc.mc == 1) return rader_P<T, 31, 1, 1, 5, RadixList<6, 5>>(gop, col, a, s);
    if (rc.p == 31 && rc.mc == 2) return rader_P<T, 31, 2, 1, 5, RadixList<6, 5>>(gop, col, a, s);
    if (rc.p == 97 && rc.mc == 1) return rader_P<T, 97, 1, 1, 8, RadixList<6, 4, 4>>(gop, col, a, s);
    if (rc.p == 73 && rc.mc == 7) return rader_P<T, 73, 7, 1, 6, RadixList<6, 4, 3>>(gop, col, a, s);
    if (rc.p == 17 && rc.mc == 18) return rader_P<T, 17, 6, 3, 1, RadixList<16>>(gop, col, a, s);
    if (rc.p == 103 && rc.mc == 1) return rader_P<T, 103, 1, 1, 6, RadixList<17, 6>>(gop, col, a, s);
    return NDFFT_ERR_UNSUPPORTED;
}
template int launch_jit_rader<float>(int, const RaderCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_rader<double>(int, const RaderCfg &, bool, const RealArgs<double> &, hipStream_t);

// thread-per-lane kernels (reg_kernel.h): 18 = 6 x 3, 30 = 6 x 5, 40 = 8 x 5 and the prime 23 instantiated ahead of time for the CPU tests
bool regfft_factor(int n, int *n1, int *n2) {
    if (n == 18) { *n1 = 6; *n2 = 3; return true; }
    if (n == 30) { *n1 = 6; *n2 = 5; return true; }
    if (n == 40) { *n1 = 8; *n2 = 5; return true; }
    if (n == 23) { *n1 = 23; *n2 = 1; return true; }
    if (n == 9 || n == 17) { *n1 = n; *n2 = 1; return true; }        // inner FFTs of the real-op kernels below
    if (n == 21) { *n1 = 7; *n2 = 3; return true; }
    if (n == 20) { *n1 = 5; *n2 = 4; return true; }
    if (n == 42) { *n1 = 7; *n2 = 6; return true; }
    return false;
}
int regfft_max_n(int) { return 64; }
template <typename K> __global__ void k_reg_emul(const TinyArgs a) { K::run(a); }
template <typename T, int N1, int N2, int LANES, bool STAGE> static int reg_one(const TinyArgs &a, hipStream_t s) {
    using K = RegFft2<T, N1, N2, LANES, STAGE>;
    hipLaunchKernelGGL((k_reg_emul<K>), dim3((unsigned)((a.nlanes + LANES - 1) / LANES)), dim3(K::THREADS), K::LDS_BYTES, s, a);
    return NDFFT_OK;
}
template <typename T, int N1, int N2> static int reg_n(bool stage, const TinyArgs &a, hipStream_t s) {
    if (!stage) return reg_one<T, N1, N2, 256, false>(a, s);
    int lanes = 256;
    while (lanes > 64 && (size_t)lanes * ((N1 * N2) | 1) * sizeof(cpx<T>) > (size_t)64 * 1024) lanes >>= 1;
    return lanes == 256 ? reg_one<T, N1, N2, 256, true>(a, s) : lanes == 128 ? reg_one<T, N1, N2, 128, true>(a, s) : reg_one<T, N1, N2, 64, true>(a, s);
}
int launch_jit_regfft(int dtype, int n1, int n2, bool stage, const TinyArgs &a, hipStream_t s) {
    if (a.nlanes <= 0) return NDFFT_OK;
    if (n1 == 6 && n2 == 3) return dtype == NDFFT_F32 ? reg_n<float, 6, 3>(stage, a, s) : reg_n<double, 6, 3>(stage, a, s);
    if (n1 == 6 && n2 == 5) return dtype == NDFFT_F32 ? reg_n<float, 6, 5>(stage, a, s) : reg_n<double, 6, 5>(stage, a, s);
    if (n1 == 8 && n2 == 5) return dtype == NDFFT_F32 ? reg_n<float, 8, 5>(stage, a, s) : reg_n<double, 8, 5>(stage, a, s);
    if (n1 == 23 && n2 == 1) return dtype == NDFFT_F32 ? reg_n<float, 23, 1>(stage, a, s) : reg_n<double, 23, 1>(stage, a, s);
    return NDFFT_ERR_UNSUPPORTED;
}

// RegReal (reg_kernel.h): handler lengths 18 (even ops, inner F = 9; DCT-I F = 17) and 21 (odd ops, F = 21; DCT-I F = 20;
// DCT-IV F = 42) instantiated ahead of time so that every PRE / POST formula runs on the CPU in its register form
template <typename K> __global__ void k_regreal_emul(const RegRealArgs a) { K::run(a); }
template <typename T, int OP, int N, int F1, int F2> static int regreal_one(bool stage, const RegRealArgs &a, hipStream_t s) {
    if (stage) {
        using K = RegReal<T, OP, N, F1, F2, 64, true>;
        hipLaunchKernelGGL((k_regreal_emul<K>), dim3((unsigned)((a.t.nlanes + 63) / 64)), dim3(64), K::LDS_BYTES, s, a);
    } else {
        using K = RegReal<T, OP, N, F1, F2, 256, false>;
        hipLaunchKernelGGL((k_regreal_emul<K>), dim3((unsigned)((a.t.nlanes + 255) / 256)), dim3(256), 0, s, a);
    }
    return NDFFT_OK;
}
template <typename T> static int regreal_T(int gop, int n, int f1, int f2, bool stage, const RegRealArgs &a, hipStream_t s) {
#define RR(OP_, N_, F1_, F2_) if (gop == OP_ && n == N_ && f1 == F1_ && f2 == F2_) return regreal_one<T, OP_, N_, F1_, F2_>(stage, a, s);
    RR(G_R2C_EVEN, 18, 9, 1) RR(G_C2R_EVEN, 18, 9, 1) RR(G_DCT1, 18, 17, 1) RR(G_DCT2_EVEN, 18, 9, 1) RR(G_DCT3_EVEN, 18, 9, 1) RR(G_DCT4_EVEN, 18, 9, 1)
    RR(G_R2C_ODD, 21, 7, 3) RR(G_C2R_ODD, 21, 7, 3) RR(G_DCT1, 21, 5, 4) RR(G_DCT2_ODD, 21, 7, 3) RR(G_DCT3_ODD, 21, 7, 3) RR(G_DCT4_ODD, 21, 7, 6)
#undef RR
    return NDFFT_ERR_UNSUPPORTED;
}
int launch_jit_regreal(int dtype, int gop, int n, int f1, int f2, bool stage, const RegRealArgs &a, hipStream_t s) {
    if (a.t.nlanes <= 0) return NDFFT_OK;
    return dtype == NDFFT_F32 ? regreal_T<float>(gop, n, f1, f2, stage, a, s) : regreal_T<double>(gop, n, f1, f2, stage, a, s);
}
}
namespace ndfft {
bool jit_fourstep_ok(int, const JitCfg &) { return false; }
bool jit_fourstep_choose(int, int, JitCfg &) { return false; }
bool jit_rfs1_ok(int, const JitCfg &) { return false; }
bool jit_rfsi_ok(int, const JitCfg &) { return false; }      // (no hiprtc in the emulation: non-power-of-two four-step factors keep the transpose route)
template <typename T> int launch_jit_fourstep(int, bool, const JitCfg &, const RealArgs<T> &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
template int launch_jit_fourstep<float>(int, bool, const JitCfg &, const RealArgs<float> &, hipStream_t);
template int launch_jit_fourstep<double>(int, bool, const JitCfg &, const RealArgs<double> &, hipStream_t);
}
// (no hiprtc in the emulation: the build step of the real library has nothing to do here)
extern "C" int ndfft_jit_prebuild(const char *, const char *, int, int, int *built, int *present, int *failed) {
    if (built) *built = 0;
    if (present) *present = 0;
    if (failed) *failed = 0;
    return NDFFT_ERR_UNSUPPORTED;
}
