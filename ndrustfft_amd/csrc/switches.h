// switches.h -- every environment switch of the library, parsed ONCE into one struct (plan.hip: sw()).
//
// Nothing on the call path reads the environment: ndfft_exec / ndfft_exec_device / plan creation look at `sw()`, a pointer load.
// The struct is filled on first use and again only by ndfft_reload_switches() (a test hook: the parity tests flip a route switch,
// reload, run a case, flip it back -- never concurrently with transforms on other threads).
//
// START-UP ONLY (captured once in a static the first time they are used; ndfft_reload_switches() does not reach them): NDFFT_JIT=0 against
// an already loaded hiprtc (a process started with NDFFT_JIT=0 never loads it, and a reload to 1 does not either), NDFFT_HOST_REG_CACHE_MB
// (the initial budget; ndfft_host_reg_cache() changes it at run time) and NDFFT_COPY_THREADS (the copy pool is created once).  Numeric
// switches are clamped to their valid range by the parser; text that is not a number keeps the default.
//
// Two classes:
//  * DOCUMENTED switches (the fields below; INTEGRATION.md section "Environment switches" lists every one, tests/test_switches.py
//    keeps the two lists identical): where code objects are cached, which kernel routes are allowed (so that every fallback kernel
//    can be reached by a test), host-path and multi-GPU chunking.
//  * developer knobs (NDFFT_DEV_INT / NDFFT_DEV_STR at their point of use): tuning parameters whose A/B is recorded in DESIGN.md.
//    They are compile-time constants in the product build; `make DEV=1` (-DNDFFT_DEV_KNOBS) turns each into an environment value
//    that is read AT EVERY USE (per plan / per call, no caching), so that a sweep may change a knob between plans inside one
//    process (tools/probes/rader_tune.py, jit_col_lanes.py).
#pragma once
#include <cstdlib>
#include <string>

namespace ndfft {

struct Switches {
    // ---- run-time specialisation (jit.hip)
    int jit = 1;                         // NDFFT_JIT: 0 = none (as on a host without libhiprtc), "cached" (2) = cached / prebuilt code objects only, never compile
    bool jit_cache_set = false;          // NDFFT_JIT_CACHE: directory of the user's code-object cache ("" or 0 = no cache)
    std::string jit_cache;
    bool jit_prebuilt_set = false;       // NDFFT_JIT_PREBUILT: read-only directory looked up after the cache ("" or 0 = none; default: <library dir>/jit_prebuilt)
    std::string jit_prebuilt;
    bool jit_verbose = false;            // NDFFT_JIT_VERBOSE: say on stderr why a specialisation was declined
    std::string jit_dump_src;            // NDFFT_JIT_DUMP_SRC: file that every specialised kernel's source text is appended to (tools/prebuild_jit.py writes the manifest of jit_prebuilt/ with it)
    std::string xdg_cache_home, home;    // (not ours: $XDG_CACHE_HOME / $HOME, where the cache lives when NDFFT_JIT_CACHE is unset)
    // ---- kernel routes (exec.hip, jit.hip): 0 closes a route so that the kernel behind it runs (parity tests of every fallback)
    bool wave = true;                    // NDFFT_WAVE: LDS-free wavefront kernel for short dense C2C lanes
    bool tiny = true;                    // NDFFT_TINY: thread-per-lane kernels for n <= 16
    bool plain = true;                   // NDFFT_PLAIN: odd-n real ops on the plain complex kernel
    bool blue = true;                    // NDFFT_BLUE: Bluestein register kernel
    bool rader = true;                   // NDFFT_RADER: Rader / Good-Thomas register kernel
    bool colsplit = true;                // NDFFT_COLSPLIT: column four-step for long strided power-of-two lanes
    bool fourstep2 = true;               // NDFFT_FOURSTEP2: two-pass four-step for long lanes (0 = three passes)
    int real_fourstep = 1;               // NDFFT_REAL_FOURSTEP: 0 = packed route, 1 = where the plan's table says so, 2 = every eligible op
    int fs_direct = -1;                  // NDFFT_FS_DIRECT: -1 = by size, 0 / 1 = tile / lane-fastest kernel of the four-step's first pass
    bool narrow_dct = false;             // NDFFT_NARROW_DCT: 1 = long strided DCT lanes on the narrow column tiles again
    bool rfs_c2r_tile = true;            // NDFFT_RFS_C2R_TILE: 0 = the real four-step's inverse through the general column kernel
    int rfs_logn1 = 0;                   // NDFFT_RFS_LOGN1: 7..11 forces the real four-step's split n = 2^a * N2
    int cs_chunk_mb = 144;               // NDFFT_CS_CHUNK_MB: column four-step chunk (0 = one chunk)
    int stream_loads = -1;               // NDFFT_STREAM_LOADS: 0 / 1 forces the load policy of the dense C2C row kernels (default: the residency model)
    // ---- host arrays (exec.hip)
    int host_pipe = -1;                  // NDFFT_HOST_PIPE: 0 = never chunk-pipeline host calls, 1 = always (default: from 32 MiB)
    long host_reg_cache_mb = 0;          // NDFFT_HOST_REG_CACHE_MB: initial budget of the opt-in registration cache (ndfft_host_reg_cache)
    int copy_threads = 0;                // NDFFT_COPY_THREADS: size of the host copy pool (default: 3/4 of the usable CPUs, 2..12)
    // ---- multi-GPU (shard.hip)
    long shard_chunk_kb = 0;             // NDFFT_SHARD_CHUNK_KB: chunk of the scatter | transform | gather pipeline (default 64 MiB)
    bool shard_force_remote = false;     // NDFFT_SHARD_FORCE_REMOTE: 1 = the root's own block takes the remote path too (one-GPU test of the pipeline)
};

// the names above, for ndfft_explain_switches() and tests/test_switches.py
#define NDFFT_DOCUMENTED_SWITCHES                                                                                                  \
    "NDFFT_JIT", "NDFFT_JIT_CACHE", "NDFFT_JIT_PREBUILT", "NDFFT_JIT_VERBOSE", "NDFFT_JIT_DUMP_SRC", "NDFFT_WAVE", "NDFFT_TINY", "NDFFT_PLAIN", "NDFFT_BLUE",  \
    "NDFFT_RADER", "NDFFT_COLSPLIT", "NDFFT_FOURSTEP2", "NDFFT_REAL_FOURSTEP", "NDFFT_FS_DIRECT", "NDFFT_NARROW_DCT",                  \
    "NDFFT_RFS_C2R_TILE", "NDFFT_RFS_LOGN1", "NDFFT_CS_CHUNK_MB", "NDFFT_STREAM_LOADS", "NDFFT_HOST_PIPE",            \
    "NDFFT_HOST_REG_CACHE_MB", "NDFFT_COPY_THREADS", "NDFFT_SHARD_CHUNK_KB", "NDFFT_SHARD_FORCE_REMOTE"

const Switches &sw();            // plan.hip
void reload_switches();          // plan.hip (ndfft_reload_switches)

// developer knobs: constants in the product build; under -DNDFFT_DEV_KNOBS environment values read at every use (the developer build
// is the only one that calls getenv outside parse_switches)
#ifdef NDFFT_DEV_KNOBS
#define NDFFT_DEV_INT(NAME, DEF) ([]() -> long { const char *e_ = getenv(NAME); return e_ ? atol(e_) : (long)(DEF); }())
#define NDFFT_DEV_STR(NAME) ((const char *)getenv(NAME))
#else
#define NDFFT_DEV_INT(NAME, DEF) ((long)(DEF))
#define NDFFT_DEV_STR(NAME) ((const char *)nullptr)
#endif

}  // namespace ndfft
