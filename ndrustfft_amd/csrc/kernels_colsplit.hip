// kernels_colsplit.hip -- the twiddled stage of the column four-step (pow2_real.h, CS kernels):
// column C2C kernels of length F2 = 64 with the W_N^(b k1) twiddle fused into the load or the store and
// the k1 + F1 k2 output row order (plus the Hermitian row maps of the R2C / C2R forms).
// The other stage is an ordinary column kernel of length F1 = N / F2 launched through dispatch().
#include "pow2_real.h"

namespace ndfft {

// must match RealCfg<64> in kernels_pow2_real.hip: the per-pass twiddles come from the length-64 C2C
// plan's twp_col table
using CsRL64 = RadixList<8, 8>;
#ifndef NDFFT_CS_TPL      // (overridable for variant builds, tools/build_variant.sh)
#define NDFFT_CS_TPL 8
#endif
#ifndef NDFFT_CS_LPB
#define NDFFT_CS_LPB 32
#endif
static constexpr int kCsTPL = NDFFT_CS_TPL, kCsLPB = NDFFT_CS_LPB;

template <typename T, int OP, int CS> static int launch_cs(const RealArgs<T> &a, hipStream_t s) {
    using K = RealPow2Kernel<T, 64, kCsTPL, kCsLPB, CsRL64, OP, true, false, CS>;
    const int64_t nblk = (a.nlanes + kCsLPB - 1) / kCsLPB;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return fail(NDFFT_ERR_UNSUPPORTED, "too many lanes for one launch");
    // lanes are (o, k1, i): when a row of i is whole tiles the grid is (tiles of i, K1, O) and the kernel reads its position off blockIdx
    // instead of dividing a flat lane index (pow2_real.h: cs_pos); otherwise the flat 1-D grid
    const int64_t rows = a.inner > 0 ? a.nlanes / a.inner : 0, O = a.cs_k1n > 0 ? rows / a.cs_k1n : 0;
    if (NDFFT_DEV_INT("NDFFT_CS_GRID3", 1) && a.inner > 0 && a.inner % kCsLPB == 0 && a.cs_k1n <= 65535 && O >= 1 && O <= 65535 &&
        O * a.cs_k1n * a.inner == a.nlanes && a.inner / kCsLPB <= 0x7fffffffLL) {
        RealArgs<T> b = a;
        b.cs_grid3 = 1;
        hipLaunchKernelGGL((k_pow2_real<K, T>), dim3((unsigned)(a.inner / kCsLPB), (unsigned)a.cs_k1n, (unsigned)O), dim3(K::THREADS), K::LDS_BYTES, s, b);
    } else {
        hipLaunchKernelGGL((k_pow2_real<K, T>), dim3((unsigned)nblk), dim3(K::THREADS), K::LDS_BYTES, s, a);
    }
    NDFFT_HIP(hipGetLastError());
    return NDFFT_OK;
}

int colsplit_inner_len() { return 64; }
int colsplit_tile_lanes() { return kCsLPB; }

template <typename T> int launch_colsplit(int cs, bool inverse, const RealArgs<T> &a, hipStream_t s) {
    switch (cs) {
        case 1: return inverse ? launch_cs<T, G_C2C_INV, 1>(a, s) : launch_cs<T, G_C2C_FWD, 1>(a, s);
        case 2: return launch_cs<T, G_C2C_FWD, 2>(a, s);
        case 3: return launch_cs<T, G_C2C_INV, 3>(a, s);
        default: return fail(NDFFT_ERR_INVALID_ARG, "column four-step: bad stage kind");
    }
}
template int launch_colsplit<float>(int, bool, const RealArgs<float> &, hipStream_t);
template int launch_colsplit<double>(int, bool, const RealArgs<double> &, hipStream_t);

}  // namespace ndfft
