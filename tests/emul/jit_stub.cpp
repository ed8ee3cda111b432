// TEST INFRASTRUCTURE ONLY: the CPU emulation build has no hiprtc; report "no specialised kernel".
#include "engine.h"
#include "pow2_real.h"
namespace ndfft {
bool jit_choose(int, int, JitCfg &) { return false; }
void jit_build_twiddles(const JitCfg &, HostTable &) {}
int launch_jit_c2c(int, const JitCfg &, int, const Pow2Args &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
int jit_col_lanes(int, const JitCfg &) { return 0; }
template <typename T> int launch_jit_real(int, const JitCfg &, bool, const RealArgs<T> &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
template int launch_jit_real<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_real<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);
}
