// TEST INFRASTRUCTURE ONLY: the CPU emulation build has no hiprtc; report "no specialised kernel".
#include "engine.h"
namespace ndfft {
bool jit_choose(int, int, JitCfg &) { return false; }
void jit_build_twiddles(const JitCfg &, HostTable &) {}
int launch_jit_c2c(int, const JitCfg &, int, const Pow2Args &, hipStream_t) { return NDFFT_ERR_UNSUPPORTED; }
}
