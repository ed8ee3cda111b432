// jit.hip -- plan-time specialisation of the register-resident Stockham kernel (pow2_kernel.h) for smooth
// NON-power-of-two lengths.  The hand-written kernel template is radix- and size-generic but needs n, the
// threads per lane and the radix list as compile-time constants; power-of-two lengths are instantiated
// ahead of time (kernels_pow2.hip), every other 2-3-5-7-smooth length is compiled on first use with hiprtc
// from the SAME header text (embedded at build time: _build/jit_sources.inc) and cached per process.
// libhiprtc is loaded lazily with dlopen; if it is missing, or a compile fails, the caller falls back to the
// LDS kernel (generic_kernel.h) -- still on the GPU.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <utime.h>
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"
#include "pow2_real.h"
#include "jit_sources.inc"
// the kernel headers embedded for hiprtc (Makefile: _build/jit_sources.inc), in one place: names as the kernels #include them, and their text
static const char *const kJitHdrNames[] = {"device_common.h", "butterflies.h", "pow2_kernel.h", "realops.h", "pow2_real.h", "blue_kernel.h", "reg_kernel.h", "rader_kernel.h", "plain_kernel.h", "col_direct.h"};
static const char *const kJitHdrSrc[] = {jit_src_device_common_h, jit_src_butterflies_h, jit_src_pow2_kernel_h, jit_src_realops_h, jit_src_pow2_real_h, jit_src_blue_kernel_h, jit_src_reg_kernel_h, jit_src_rader_kernel_h, jit_src_plain_kernel_h, jit_src_col_direct_h};
static constexpr int kNJitHdr = 10;

namespace ndfft {

namespace {
typedef struct _hiprtcProgram *rtcProgram;
struct Rtc {
    void *lib = nullptr;
    int (*create)(rtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(rtcProgram, int, const char **) = nullptr;
    int (*log_size)(rtcProgram, size_t *) = nullptr;
    int (*log)(rtcProgram, char *) = nullptr;
    int (*code_size)(rtcProgram, size_t *) = nullptr;
    int (*code)(rtcProgram, char *) = nullptr;
    int (*destroy)(rtcProgram *) = nullptr;
    bool ok = false;
};
Rtc &rtc() {
    static Rtc r = [] {
        Rtc x;
        if (sw().jit == 0) return x;   // NDFFT_JIT=0: behave like a host without libhiprtc
        const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
        for (const char *n : names) if ((x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!x.lib) return x;
        x.create = (decltype(x.create))dlsym(x.lib, "hiprtcCreateProgram");
        x.compile = (decltype(x.compile))dlsym(x.lib, "hiprtcCompileProgram");
        x.log_size = (decltype(x.log_size))dlsym(x.lib, "hiprtcGetProgramLogSize");
        x.log = (decltype(x.log))dlsym(x.lib, "hiprtcGetProgramLog");
        x.code_size = (decltype(x.code_size))dlsym(x.lib, "hiprtcGetCodeSize");
        x.code = (decltype(x.code))dlsym(x.lib, "hiprtcGetCode");
        x.destroy = (decltype(x.destroy))dlsym(x.lib, "hiprtcDestroyProgram");
        x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
        return x;
    }();
    return r;
}

struct Entry { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; bool failed = false; };
// key -> entry; `ready == false` marks a compile in flight on some thread: a second thread that wants the SAME kernel
// waits on g_cv, threads that want other kernels (or cached ones) are never held up by a ~0.5 s hiprtc compile
struct Slot { Entry e; bool ready = false; };
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, Slot> g_cache;

bool jit_disabled() {
    return sw().jit == 0;
}
}  // namespace

// Picks threads-per-lane and a radix list for a smooth length: every radix divides E = n / TPL (so each
// thread owns whole butterflies in every pass), fewest passes first, then the smallest E.
// Partial-round configurations (pow2_kernel.h: slots / full): any radix list whose product is n, any TPL.
// Cost ~ passes x (work incl. idle threads of partial rounds): minimise NP / utilisation; ties -> E nearest 16.
static size_t jit_lds_limit();
static int jit_full_min() { return 256; }
static bool jit_choose_partial(int dtype, int n, JitCfg &cfg, int emax_arg = 0) {
    const int emax = emax_arg > 0 ? emax_arg : dtype == NDFFT_F32 ? 32 : 30;
    const int cand[] = {16, 13, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    std::vector<int> cur, best;
    int best_tpl = 0, best_e = 0;
    double best_cost = 1e30;
    auto eval = [&]() {
        const int np = (int)cur.size();
        for (int tpl = std::max(1, (n + emax - 1) / emax / 2); tpl <= std::min(1024, n / 2); ++tpl) {
            int e = 0; double work = 0;
            for (int r : cur) {
                const int nb = n / r, sl = (nb + tpl - 1) / tpl;
                e = std::max(e, sl * r);
                work += (double)sl * tpl * r;
            }
            if (e > emax) continue;
            const double cost = work / n + 0.02 * std::abs(e - 16) / 16.0 + (tpl % 4 ? 0.05 : 0.0);   // work/n = NP / utilisation
            (void)np;
            if (cost < best_cost) { best_cost = cost; best = cur; best_tpl = tpl; best_e = e; }
        }
    };
    std::function<void(int, int)> rec = [&](int m, int maxr) {
        if (m == 1) { eval(); return; }
        if (cur.size() >= 6) return;
        for (int c : cand) {
            if (c > maxr || m % c) continue;
            cur.push_back(c);
            rec(m / c, c);
            cur.pop_back();
        }
    };
    rec(n, 16);
    if (best.empty()) return false;
    cfg.n = n; cfg.e = best_e; cfg.tpl = best_tpl; cfg.radix = best;
    cfg.lpb = cfg.tpl >= 256 ? 1 : std::max(1, 256 / cfg.tpl);
    cfg.vec = 1; cfg.partial = true;
    return true;
}

// what the planner is asked for beyond (M, mc): the Rader kernel's caps (an explicit argument since round 6 -- four thread_local flags before)
struct FftPlanOpts {
    bool rader = false;        // planned for the Rader kernel: f64 cap of elements per thread 21 instead of 18
    bool sym = false;          // ... its symmetric DCT-I form: f64 cap 16 (nddct1 n = 512, FFT_72 on 4 rows: 9.8 on 5 threads, e = 18, 222 us; on 8 threads, e = 16, two lanes
                               //     per wave, 142 us -- profiles/r08/r08e_dct1_sym_tune_512.txt)
    bool half = false;         // ... its half-length form (cofactor 1): f64 cap 18 -- nddct1 n = 1010 f64 87 us at e = 21, 71 us at e = 14; n = 8192 177 us at e = 18, 190 us at e = 15
    bool full_waves = false;   // ... only recipes whose lane is a divisor or a multiple of one wave (the cost model's own picks, 9.8 on 5 threads and 12.6 on 6, measured 222 and > 190 us)
};
static bool plan_fft_by_cost(int dtype, int M, int mc, size_t lane_bytes, JitCfg &out, int wide = 0, double *cost_out = nullptr, const FftPlanOpts &opts = FftPlanOpts());
static bool jit_choose_default(int dtype, int n, JitCfg &cfg, bool allow_partial, int emax_override = 0);
// The default recipe ("fewest passes, every radix divides E") gives some lengths 20-30 elements per thread on a handful of threads
// (F = 48: 8.6 on 2 threads, e = 24; 3000 = 10.10.10.3, e = 30).  Those measure badly -- f64 from e > 18, f32 from e > 24
// (profiles/r04/r04j_realplan_ab.txt: nddct2 / ndfft_r2c n = 96, 120, 300, 360, 1200, 3000, 6000 gain 1.3-2.9x; recipes with e <= 12 were as good
// or better than the model's pick) -- and are replaced by the pick of the cost model fitted for the Rader kernel (plan_fft_by_cost).
static bool jit_choose_impl(int dtype, int n, JitCfg &cfg, bool allow_partial, bool short_too);
bool jit_choose(int dtype, int n, JitCfg &cfg, bool allow_partial) { return jit_choose_impl(dtype, n, cfg, allow_partial, true); }
// The short-lane f32 re-plan below was measured on ROWS (+30 % at n = 80 / 96 / 160); on column tiles the default recipe is the faster one
// (round 4, A-B-A-B against the round-2 library and with the recipe forced: 81 x 100 x 2048 c64 axis 1 56.9 -> 44.2 us; rows of n = 100 the same
// either way).  A C2C plan therefore keeps a second recipe for its column tiles where the two differ.
bool jit_choose_col(int dtype, int n, const JitCfg &row_cfg, JitCfg &col_cfg) {
    if (dtype != NDFFT_F32 || n >= 256 || !NDFFT_DEV_INT("NDFFT_JIT_COL_ALT", 1)) return false;
    if (!jit_choose_impl(dtype, n, col_cfg, true, false)) return false;
    if (col_cfg.tpl == row_cfg.tpl && col_cfg.radix == row_cfg.radix) return false;
    // Sweep of all 23 lengths in 97..255 where the two recipes differ, column tiles of (k, n, 2048) and (k, n, 64) c64 arrays, two alternating processes per
    // setting (tools/sweep_col_recipes.py, profiles/r07/r07f_col_recipe_sweep_c64.jsonl): the default recipe wins 13-41 % on all ten lengths where it
    // keeps <= 18 elements per thread in no more passes than the rows' recipe (98, 99, 100, 110, 121, 143, 144, 156, 162, 220) and loses 3-32 % on nine
    // of the other thirteen (126, 132, 135, 160, 176, 189, 192, 225, 242; 140 / 147 / 154 within 2 %; only 231 would have gained, 12 %).
    return col_cfg.e <= 18 && col_cfg.radix.size() <= row_cfg.radix.size();
}
static bool jit_choose_impl(int dtype, int n, JitCfg &cfg, bool allow_partial, bool short_too) {
    constexpr bool on = true;
    constexpr int nmax = 8192;
    if (!jit_choose_default(dtype, n, cfg, allow_partial)) return false;
    if (const char *e = NDFFT_DEV_STR("NDFFT_JIT_CFG")) {       // developer knob: "n:tpl:r0.r1.r2[:lanes]" replaces the recipe of length n (read per plan)
        if (atoi(e) == n && allow_partial) {
            const char *q = strchr(e, ':');
            JitCfg c; c.n = n; c.tpl = q ? atoi(q + 1) : 0; c.vec = 1;
            q = q ? strchr(q + 1, ':') : nullptr;
            int prod = 1;
            while (q && *q && *q != '\0') { const int r = atoi(q + 1); if (r < 2) break; c.radix.push_back(r); prod *= r; const char *d = strchr(q + 1, '.'), *cl = strchr(q + 1, ':'); if (cl && (!d || cl < d)) { c.row_lpb = atoi(cl + 1); break; } q = d; }
            if (c.tpl >= 1 && prod == n) {
                for (int r : c.radix) { const int nb = n / r, sl = (nb + c.tpl - 1) / c.tpl; c.e = std::max(c.e, sl * r); if (nb % c.tpl) c.partial = true; }
                c.lpb = c.row_lpb > 0 ? c.row_lpb : (c.tpl >= 64 ? 1 : std::max(1, 64 / c.tpl));
                if (c.row_lpb == 0) c.row_lpb = c.lpb;
                if (dtype == NDFFT_F32 && !c.partial && (c.e / c.radix.front()) % 2 == 0 && (c.e / c.radix.back()) % 2 == 0) c.vec = 2;
                cfg = c;
                return true;
            }
        }
    }
    // f32 lanes below 256 points are re-planned from e > 8 (A-B-A-B, profiles/r04/r04zg_abab_shortplan.txt: c64 n = 80 / 96 / 160 51.5 / 50.5 / 49.7 -> 39.6 / 38.9 / 38.2 us,
    // ndfft_r2c f32 n = 160 / 192 / 320 +10 %, the rest within 3 %; in f64 the same rule was a wash: c128 n = 96 -6 %, nddct2 n = 192 +5 %)
    const bool bad_e = cfg.e > (dtype == NDFFT_F32 ? 24 : 18) || (short_too && dtype == NDFFT_F32 && n < 256 && cfg.e > 8);
    if (!on || !allow_partial || n > nmax || !bad_e) return true;
    const size_t lane = (size_t)((n + (n >> 4) + 3) & ~1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
    JitCfg alt;
    if (plan_fft_by_cost(dtype, n, 1, lane, alt)) { alt.vec = 1; alt.row_lpb = alt.lpb; cfg = alt; }
    return true;
}
static bool jit_choose_default(int dtype, int n, JitCfg &cfg, bool allow_partial, int emax_override) {
    // one lane's half exchange (n reals, padded) must fit the 160 KiB of LDS: n <= 19274 (f64) / 32768 (f32, capped)
    const size_t lane_lds = ((size_t)n + ((size_t)n >> 4) + 1) * (dtype == NDFFT_F32 ? 4 : 8);
    if (jit_disabled() || n < 12 || n > 32768 || lane_lds > jit_lds_limit() || pow2_supported(dtype, n)) return false;
    {   int m = n; for (int p : {2, 3, 5, 7, 11, 13}) while (m % p == 0) m /= p; if (m != 1) return false; }
    // E complex registers per thread: 2E (f32) / 4E (f64) VGPRs of data.  Mixed 2-3-5 lengths need E = 30.
    const int emax = emax_override > 0 ? emax_override : dtype == NDFFT_F32 ? 32 : 30;
    const int cand[] = {16, 13, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    std::vector<int> best, cur;
    int best_e = 0;
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    std::function<void(int, int, int)> rec = [&](int m, int maxr, int e) {
        if (e > emax || n % e) return;
        if (m == 1) {
            if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && e < best_e)) { best = cur; best_e = e; }
            return;
        }
        if (!best.empty() && cur.size() + 1 > best.size()) return;
        for (int c : cand) {
            if (c > maxr || m % c) continue;
            cur.push_back(c);
            rec(m / c, c, e / gcd(e, c) * c);
            cur.pop_back();
        }
    };
    rec(n, 16, 1);
    if (best.empty() || n / best_e > 1024) return allow_partial && jit_choose_partial(dtype, n, cfg);
    cfg.n = n; cfg.e = best_e; cfg.tpl = n / best_e;
    cfg.radix = best;   // non-increasing: the largest radix first (most loads in flight on the global read)
    cfg.lpb = cfg.tpl >= 256 ? 1 : std::max(1, 256 / cfg.tpl);
    // f32: 16-byte (two-element) global accesses need an even number of butterflies per thread in the
    // first and the last pass
    cfg.vec = (dtype == NDFFT_F32 && (best_e / best.front()) % 2 == 0 && (best_e / best.back()) % 2 == 0) ? 2 : 1;
    return true;
}

// dynamic LDS a module (hiprtc) function may be launched with: the full 160 KiB of a gfx950 CU -- no
// opt-in attribute is needed for module functions (checked on the MI355X: a 136 KiB launch runs and is correct)
static size_t jit_lds_limit() {
    return (size_t)160 * 1024;
}

void jit_build_twiddles(const JitCfg &cfg, HostTable &out) {
    const long double kPiL = 3.14159265358979323846264338327950288L;
    unsigned long long Ns = 1;
    for (size_t p = 0; p < cfg.radix.size(); ++p) {
        const unsigned long long R = (unsigned long long)cfg.radix[p];
        if (p > 0)
            for (unsigned long long r = 1; r < R; ++r)
                for (unsigned long long k = 0; k < Ns; ++k) {
                    const unsigned long long num = (r * k) % (Ns * R);
                    const long double ang = 2.0L * kPiL * (long double)num / (long double)(Ns * R);
                    out.re.push_back(cosl(ang)); out.im.push_back(-sinl(ang));
                }
        Ns *= R;
    }
}

namespace {
// compiles `src` (which defines extern "C" kernel k_jit) once per key; returns the cached entry
// ---- on-disk cache of the compiled code objects ------------------------------------------------------
// $NDFFT_JIT_CACHE (a directory; "0" disables), else $XDG_CACHE_HOME/ndfft_mi355x, else ~/.cache/ndfft_mi355x.
// File name = FNV-1a hash of the kernel source, every embedded header and the compile options, so a rebuilt
// library with different kernel text never picks up a stale object.  Written to a temp file and renamed.
const char *const kJitOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast"};
std::string cache_dir() {
    const Switches &S = sw();                       // NDFFT_JIT_CACHE
    std::string d;
    if (S.jit_cache_set) { if (S.jit_cache.empty() || S.jit_cache == "0") return ""; d = S.jit_cache; }
    else if (!S.xdg_cache_home.empty()) d = S.xdg_cache_home + "/ndfft_mi355x";
    else if (!S.home.empty()) { const std::string c = S.home + "/.cache"; (void)mkdir(c.c_str(), 0755); d = c + "/ndfft_mi355x"; }
    else return "";
    if (mkdir(d.c_str(), 0755) != 0 && errno != EEXIST) return "";
    return d;
}
uint64_t fnv1a(uint64_t h, const char *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } return h; }
std::string cache_name(const std::string &src, const char *const *hs, int nh);
std::string cache_path(const std::string &src, const char *const *hs, int nh) {
    const std::string d = cache_dir();
    if (d.empty()) return "";
    return d + cache_name(src, hs, nh);
}
std::string cache_name(const std::string &src, const char *const *hs, int nh) {
    uint64_t h = 1469598103934665603ull;
    h = fnv1a(h, src.data(), src.size());
    for (int i = 0; i < nh; ++i) h = fnv1a(h, hs[i], strlen(hs[i]));
    for (const char *o : kJitOpts) h = fnv1a(h, o, strlen(o));
    char name[40];
    snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
    return name;
}
// A second, READ-ONLY location is looked up after the user's cache: <directory of this library>/jit_prebuilt (NDFFT_JIT_PREBUILT
// overrides; "0" disables).  It ships code objects for the reference's own test / bench / example lengths (3, 6, 264, 129, 265, 513,
// 1025 -- tools/prebuild_jit.py writes it on an MI355X), so that a first nd* call on those never waits ~0.5 s for hiprtc.  Same file
// names (hash of source + headers + options): an object built from other kernel text is simply not found.
std::string prebuilt_path(const std::string &cache_file_path_or_name) {
    const std::string dir = [] {
        if (sw().jit_prebuilt_set) return (sw().jit_prebuilt.empty() || sw().jit_prebuilt == "0") ? std::string() : sw().jit_prebuilt;   // NDFFT_JIT_PREBUILT
        Dl_info info;
        if (!dladdr((const void *)&fnv1a, &info) || !info.dli_fname) return std::string();
        std::string p = info.dli_fname;
        const size_t k = p.find_last_of('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/jit_prebuilt";
    }();
    if (dir.empty()) return "";
    const size_t k = cache_file_path_or_name.find_last_of('/');
    return dir + "/" + (k == std::string::npos ? cache_file_path_or_name : cache_file_path_or_name.substr(k + 1));
}
bool read_file(const std::string &path, std::string &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    bool ok = n > 0;
    if (ok) { out.resize((size_t)n); ok = fread(&out[0], 1, (size_t)n, f) == (size_t)n; }
    fclose(f);
    return ok;
}
void write_file_atomic(const std::string &path, const std::string &data) {
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}

// ---- the shipped code objects as BUILD products (round 6) ---------------------------------------------------------------------
// jit_prebuilt/ used to hold ~160 tracked .hsaco files that only a GPU box could regenerate and that every edit of a kernel header made stale.  Now the tree tracks a
// MANIFEST of the specialised kernels' (few-line) source texts -- jit_prebuilt/manifest.txt, appended to by any run with NDFFT_JIT_DUMP_SRC=<file> (tools/prebuild_jit.py
// on an MI355X: which kernels the reference's own lengths ask for is an exec-time choice) -- and ndfft_jit_prebuild() compiles every entry with hiprtc against the headers
// embedded in THIS library, under the file names compile_entry() will look up.  hiprtc needs no device: __graft_entry__.build() runs it on the CPU-only build box.
const char kManifestSep[] = "\n=====NDFFT-JIT-ENTRY=====\n";
static void dump_source(const std::string &src) {
    const std::string &fs = sw().jit_dump_src;                // NDFFT_JIT_DUMP_SRC (parsed once, switches.h)
    if (fs.empty()) return;
    const char *f = fs.c_str();
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    FILE *fp = fopen(f, "ab");
    if (!fp) return;
    fwrite(src.data(), 1, src.size(), fp); fwrite(kManifestSep, 1, sizeof kManifestSep - 1, fp);
    fclose(fp);
}
// compiles `src` with hiprtc (no device needed); returns the code object, empty on failure
static std::string rtc_compile(const std::string &src, const std::string &what) {
    Rtc &r = rtc();
    const char **hn = (const char **)kJitHdrNames, **hs = (const char **)kJitHdrSrc;
    rtcProgram prog = nullptr;
    bool ok = r.ok && r.create(&prog, src.c_str(), "k_jit.hip", kNJitHdr, hs, hn) == 0;
    if (ok) {
        ok = r.compile(prog, 4, (const char **)kJitOpts) == 0;
        if (!ok && sw().jit_verbose) {
            size_t ls = 0; r.log_size(prog, &ls);
            std::string log(ls, '\0'); r.log(prog, &log[0]);
            fprintf(stderr, "ndfft jit: compile of %s failed:\n%s\n", what.c_str(), log.c_str());
        }
    }
    std::string code;
    if (ok) { size_t cs = 0; ok = r.code_size(prog, &cs) == 0 && cs > 0; if (ok) { code.resize(cs); ok = r.code(prog, &code[0]) == 0; } }
    if (prog) r.destroy(&prog);
    if (!ok) code.clear();
    return code;
}

static Entry compile_entry(const std::string &src, const std::string &what) {
    Rtc &r = rtc();
    Entry ne;
    dump_source(src);
    {   // a code object compiled by an earlier process?
        const char *const *hs0 = kJitHdrSrc;
        const std::string path = cache_path(src, hs0, kNJitHdr);
        std::string code;
        if (!path.empty() && read_file(path, code) && hipModuleLoadData(&ne.mod, code.data()) == hipSuccess &&
            hipModuleGetFunction(&ne.fn, ne.mod, "k_jit") == hipSuccess)
            return ne;
        (void)hipGetLastError();
        ne = Entry();
        const std::string pre = prebuilt_path(cache_name(src, hs0, kNJitHdr));    // shipped with the library
        if (!pre.empty() && read_file(pre, code) && hipModuleLoadData(&ne.mod, code.data()) == hipSuccess &&
            hipModuleGetFunction(&ne.fn, ne.mod, "k_jit") == hipSuccess)
            return ne;
        (void)hipGetLastError();
        ne = Entry();
        if (sw().jit == 2) { ne.failed = true; return ne; }      // NDFFT_JIT=cached: cached / prebuilt code objects only
    }
    const char *const *hs = kJitHdrSrc;
    const std::string code = rtc_compile(src, what);
    bool ok = !code.empty();
    if (ok) {
        const hipError_t e1 = hipModuleLoadData(&ne.mod, code.data());
        const hipError_t e2 = e1 == hipSuccess ? hipModuleGetFunction(&ne.fn, ne.mod, "k_jit") : e1;
        ok = e2 == hipSuccess;
        if (!ok && sw().jit_verbose) fprintf(stderr, "ndfft jit: loading %s failed: %s\n", what.c_str(), hipGetErrorString(e2));
    } else if (sw().jit_verbose) {
        fprintf(stderr, "ndfft jit: no code object for %s (hiprtc %s)\n", what.c_str(), r.ok ? "present" : "missing");
    }
    if (ok) { const std::string path = cache_path(src, hs, kNJitHdr); if (!path.empty()) write_file_atomic(path, code); }
    if (!ok) { (void)hipGetLastError(); ne.failed = true; }
    return ne;
}

Entry get_or_compile(const std::string &key, const std::string &src, const std::string &what) {
    {
        std::unique_lock<std::mutex> g(g_mu);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) {
            g_cv.wait(g, [&] { return g_cache[key].ready; });   // someone else is compiling this very kernel
            return g_cache[key].e;
        }
        g_cache.emplace(key, Slot());                            // ours to compile; the lock is NOT held meanwhile
    }
    const Entry ne = compile_entry(src, what);
    {
        std::lock_guard<std::mutex> g(g_mu);
        Slot &sl = g_cache[key];
        sl.e = ne; sl.ready = true;
    }
    g_cv.notify_all();
    return ne;
}
// bytes of scratch (register spill) per work-item of a compiled kernel; -1 if the runtime cannot tell.  Cached per function handle.
int entry_scratch_bytes(const Entry &e) {
    static std::mutex mu;
    static std::map<hipFunction_t, int> cache;
    if (e.failed || !e.fn) return -1;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(e.fn);
        if (it != cache.end()) return it->second;
    }
    int v = -1;
    if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, e.fn) != hipSuccess) { (void)hipGetLastError(); v = -1; }
    std::lock_guard<std::mutex> g(mu);
    cache[e.fn] = v;
    return v;
}
std::string radix_list(const JitCfg &cfg) {
    std::string rl;
    for (size_t i = 0; i < cfg.radix.size(); ++i) rl += (i ? ", " : "") + std::to_string(cfg.radix[i]);
    return rl;
}
}  // namespace

// f32 C2C rows whose default recipe ("fewest passes, smallest E") leaves ONE butterfly per thread in the first or the last pass cannot use the 16-byte
// (two-element) global accesses (Pow2Kernel: VEC = 2) and sat at 0.55 of the roofline beside 0.70-0.72 for the power-of-two lengths.  The same radix list on HALF
// the threads with twice the elements (whole rounds stay whole, the twiddle tables depend on the radix list only) has an even number everywhere:
// c64 n = 1000 (10.10.10: 100 threads x 10 -> 50 x 20) 61.7 -> 52.3 us (profiles/r09/r09y_c64_vec_knob.txt).  Not for the re-planned / partial-round recipes.
bool jit_c2c_row_vec(int dtype, JitCfg &cfg) {
    if (dtype != NDFFT_F32 || cfg.vec != 1 || cfg.partial || cfg.row_lpb != 0 || cfg.tpl < 2 || (cfg.tpl & 1) || 2 * cfg.e > 24) return false;
    if (!NDFFT_DEV_INT("NDFFT_JIT_ROW_VEC", 1)) return false;
    for (int r : cfg.radix) if (cfg.e % r) return false;      // (whole rounds in every pass)
    cfg.e *= 2; cfg.tpl /= 2; cfg.vec = 2;
    return true;
}

// returns NDFFT_OK and launches, or NDFFT_ERR_UNSUPPORTED if no specialised kernel can be had (caller falls back)
int launch_jit_c2c(int dtype, const JitCfg &cfg_plan, int nt, const Pow2Args &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const bool aligned = a.pitch_in % 2 == 0 && a.pitch_out % 2 == 0 && ((uintptr_t)a.in % 16) == 0 && ((uintptr_t)a.out % 16) == 0;
    JitCfg cfg_in = cfg_plan;
    if (aligned) { JitCfg cv = cfg_plan; if (jit_c2c_row_vec(dtype, cv)) cfg_in = cv; }
    const bool vec_ok = cfg_in.vec == 2 && aligned;
    const int vec = vec_ok ? 2 : 1;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = dtype == NDFFT_F32 ? "float" : "double";
    // one-wave workgroups where a lane needs <= 64 threads: +1-4 % in alternating A-B-A-B runs (profiles/r04/r04s_abab_c2c_row.txt: 1000 / 264 / 1331 / 96 c128,
    // 1000 c64); 0 = the recipe's own lanes (256 threads)
    const int c2c_thr = (int)NDFFT_DEV_INT("NDFFT_JIT_C2C_ROW_THREADS", 64);
    JitCfg cfg = cfg_in;
    if (c2c_thr > 0 && cfg.row_lpb == 0) cfg.lpb = cfg.tpl >= c2c_thr ? 1 : std::max(1, c2c_thr / cfg.tpl);
    const int threads = cfg.tpl * cfg.lpb;
    // f32 lanes exchange whole complex elements (64-bit LDS accesses, half the LDS instructions) as long as two
    // workgroups still fit a CU, like the ahead-of-time f32 configurations (kernels_pow2.hip: Pow2Half).  Measured on
    // 2^25 points (tools/probes/jit_f32_full.py): n = 1000 102 -> 96 us, 3000 141 -> 133, 6000 177 -> 137; but
    // n = 10000 (85 KiB: one workgroup per CU) 149 -> 172, hence the 80 KiB bound.
    const bool half = !(dtype == NDFFT_F32 && cfg.n >= jit_full_min() &&
                        (size_t)cfg.lpb * (size_t)(cfg.n + (cfg.n >> 4) + 1) * 8 <= (size_t)80 * 1024);
    const std::string inst = std::string("Pow2Kernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " +
                             std::to_string(cfg.lpb) + (half ? ", true" : ", false") + ", RadixList<" + radix_list(cfg) + ">, " + (a.twlo ? "8" : "0") + ", 1, " + std::to_string(nt) + ", " + std::to_string(vec) + ">";
    const std::string src = "#include \"pow2_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(" +
                            std::to_string(threads) + ") void k_jit(const Pow2Args a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) { if (sw().jit_verbose) fprintf(stderr, "ndfft jit: %s unavailable\n", inst.c_str()); return NDFFT_ERR_UNSUPPORTED; }
    const size_t esz = dtype == NDFFT_F32 ? 4 : 8;
    const size_t lds = (size_t)cfg.lpb * (size_t)(cfg.n + (cfg.n >> 4) + 1) * esz * (half ? 1 : 2);   // Pow2Kernel::LDS_BYTES
    if (lds > jit_lds_limit()) { if (sw().jit_verbose) fprintf(stderr, "ndfft jit: %s needs %zu B of LDS\n", inst.c_str(), lds); return NDFFT_ERR_UNSUPPORTED; }
    const int64_t nblk = (a.nlanes + cfg.lpb - 1) / cfg.lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    Pow2Args arg = a;
    arg.xcd_chunk = xcd_chunk_for((size_t)cfg.lpb * cfg.n * 2 * esz, nblk);
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}

// ---- thread-per-lane two-factor kernels (reg_kernel.h) ---------------------------------------------------------------
// n = n1 * n2 with both factors in {2..13, 16}; the most balanced pair (fewest multiply-adds).  false: no such pair.
bool regfft_factor(int n, int *n1, int *n2) {
    auto ok = [](int r) { return (r >= 2 && r <= 13) || r == 16 || r == 17 || r == 19 || r == 23 || r == 29 || r == 31; };
    if (ok(n)) { *n1 = n; *n2 = 1; return true; }                // a length with its own butterfly (primes 17..31; short inner FFTs of the real ops)
    int best = 0;
    for (int a = 2; a * a <= n; ++a)
        if (n % a == 0 && ok(a) && ok(n / a)) best = a;
    if (!best) return false;
    *n1 = n / best; *n2 = best;          // n1 >= n2
    return true;
}
// largest n whose lane fits the registers of one thread with room for the butterflies (2 / 4 VGPRs per complex element)
int regfft_max_n(int dtype) {
    // measured (profiles/r03c_reg_kernel_sweep.txt): dense rows win up to n = 63 in both precisions (0.80 -> 0.50 of 8 TB/s; the
    // general register kernel takes over from n = 72: 0.65-0.86); strided axes up to 63 (f64: 0.73-0.86, n = 64 has its tile
    // kernel) and 96 (f32: 0.70-0.85 against 0.49-0.75)
    const int f32 = (int)NDFFT_DEV_INT("NDFFT_REG_MAX_F32", 96), f64 = (int)NDFFT_DEV_INT("NDFFT_REG_MAX_F64", 63);
    return dtype == NDFFT_F32 ? f32 : f64;
}
int launch_jit_regfft(int dtype, int n1, int n2, bool stage, const TinyArgs &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const int n = n1 * n2;
    const size_t esz = dtype == NDFFT_F32 ? 8 : 16;
    int lanes = 256;
    if (stage) while (lanes > 64 && (size_t)lanes * (size_t)(n | 1) * esz > (size_t)64 * 1024) lanes >>= 1;
    if (stage && (size_t)lanes * (size_t)(n | 1) * esz > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = dtype == NDFFT_F32 ? "float" : "double";
    const std::string inst = std::string("RegFft2<") + tn + ", " + std::to_string(n1) + ", " + std::to_string(n2) + ", " + std::to_string(lanes) + ", " + (stage ? "true" : "false") + ">";
    const std::string src = std::string("#include \"reg_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") +
                            std::to_string(lanes) + ") void k_jit(const TinyArgs a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.nlanes + lanes - 1) / lanes;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    const size_t lds = stage ? (size_t)lanes * (size_t)(n | 1) * esz : 0;
    TinyArgs arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)lanes, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}

// RegReal (reg_kernel.h): R2C / C2R / DCT-I..IV with the whole lane in one thread's registers, inner FFT F = f1 * f2
int launch_jit_regreal(int dtype, int gop, int n, int f1, int f2, bool stage, const RegRealArgs &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const size_t rsz = dtype == NDFFT_F32 ? 4 : 8;
    const bool in_c = gop == G_C2R_EVEN || gop == G_C2R_ODD, out_c = gop == G_R2C_EVEN || gop == G_R2C_ODD;
    const int m2 = 2 * (n / 2 + 1), ni = in_c ? m2 : n, no = out_c ? m2 : n, pmax = std::max(ni | 1, no | 1);
    int lanes = 256;
    if (stage) while (lanes > 64 && (size_t)lanes * (size_t)pmax * rsz > (size_t)64 * 1024) lanes >>= 1;
    if (stage && (size_t)lanes * (size_t)pmax * rsz > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = dtype == NDFFT_F32 ? "float" : "double";
    const std::string inst = std::string("RegReal<") + tn + ", " + std::to_string(gop) + ", " + std::to_string(n) + ", " + std::to_string(f1) + ", " +
                             std::to_string(f2) + ", " + std::to_string(lanes) + ", " + (stage ? "true" : "false") + ">";
    // developer switch (read per call): NDFFT_REPRO_MASKED_TAIL=1 builds the predicated-tail form of the staging loads -- see reg_kernel.h
    const bool repro = NDFFT_DEV_INT("NDFFT_REPRO_MASKED_TAIL", 0) == 1;   // (developer build only: tools/repro_masked_tail.py)
    const std::string src = std::string(repro ? "#define NDFFT_REPRO_MASKED_TAIL 1\n" : "") + "#include \"reg_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(" +
                            std::to_string(lanes) + ") void k_jit(const RegRealArgs a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst + (repro ? ":masked-tail" : ""), src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.t.nlanes + lanes - 1) / lanes;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    const size_t lds = stage ? (size_t)lanes * (size_t)pmax * rsz : 0;
    RegRealArgs arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)lanes, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}

// lanes per column tile of the specialised real-op / column kernel (0: no useful tile)
int jit_col_lanes(int dtype, const JitCfg &cfg, bool c2c) {
    if (const char *e = NDFFT_DEV_STR("NDFFT_JIT_COL_LPB")) { const int l = atoi(e); return l * cfg.tpl <= 1024 ? l : 0; }   // developer knob
    // 8 adjacent lanes (f32: 16 where they fit): rows of 64-128 bytes that start on a 64-byte boundary.  Measured on 2^24 points (profiles/r04/
    // r04i_jit_col_lanes.txt): the former "as many as fit" gave 9 lanes for 1000x16384 c128 (144-byte rows) = 257 us, 8 lanes 137 us;
    // nddct2 1000x16384 f64 150 -> 80 us; 264x65536 c128 143 -> 127 us; n = 96 unchanged.
    const size_t lane = (size_t)((cfg.n + (cfg.n >> 4) + 2) | 1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
    for (int l : {dtype == NDFFT_F32 ? 16 : 8, 8})
        if (l * cfg.tpl <= 1024 && (size_t)l * lane <= jit_lds_limit()) return l;
    // C2C lanes too long for an 8-lane tile: 4 lanes (64-byte rows for c128, 32 for c64) still beat the transpose route -- ndfft axis 0 of 1500 x 11184 / 2000 x 8384 c128
    // 326 / 309 -> 209 / 203 us, 2000 x 8384 c64 188 -> 149 us; real OUTPUT rows of 4 lanes are ruinous (nddct2 / ndifft_r2c 2-4 x slower): C2C only (profiles/r06/r06y_*)
    if (c2c && 4 * cfg.tpl <= 1024 && (size_t)4 * lane <= jit_lds_limit()) return 4;
    return 0;
}

// RealPow2Kernel (pow2_real.h) specialised for a smooth inner FFT length cfg.n: R2C / C2R / DCT rows, and
// every op incl. C2C on column tiles
template <typename T> int launch_jit_real(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    // rows: one-wave workgroups where a lane needs <= 64 threads (alternating A-B-A-B runs, profiles/r04/r04s_abab_jit_row.txt: nddct2 f64 n = 100..2000 +5-19 %,
    // ndfft_r2c f32 +2-6 % against 256-thread workgroups)
    const int row_thr = (int)NDFFT_DEV_INT("NDFFT_JIT_ROW_THREADS", 64);
    const int lpb = col ? jit_col_lanes(dtype, cfg, gop == G_C2C_FWD || gop == G_C2C_INV) : cfg.row_lpb > 0 ? cfg.row_lpb : (cfg.tpl >= row_thr ? 1 : std::max(1, row_thr / cfg.tpl));
    if (lpb <= 0) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const int threads = cfg.tpl * lpb;
    const std::string inst = std::string("RealPow2Kernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " +
                             std::to_string(lpb) + ", RadixList<" + radix_list(cfg) + ">, " + std::to_string(gop) + ", " + (col ? "true" : "false") + ", false>";
    // f32 kernels of >= 512 threads: floor of 8 waves per SIMD = 64 VGPRs (kernels_pow2_real.hip: RealAotWaves) -- but only where the specialised kernel FITS 64 registers:
    // the compiled code object is asked for its scratch size, and a recipe that spills falls back to the plain form (n = 1500: 2-3 x slower with the floor, n = 1000 / 2000:
    // 10-20 % faster, profiles/r06/r06zr_*).  NDFFT_JIT_F32_MIN_WAVES overrides the floor in the developer build (1 = none).
    const int f32_floor = (int)NDFFT_DEV_INT("NDFFT_JIT_F32_MIN_WAVES", 8);
    const int floor_w = (sizeof(T) == 4 && threads >= 512 && f32_floor > 1 && gop != G_DCT3_EVEN) ? f32_floor : 1;
    auto make_src = [&](int w) {
        return std::string("#include \"pow2_real.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") + std::to_string(threads) +
               (w > 1 ? ", " + std::to_string(w) : std::string()) + ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    };
    Entry e;
    bool plain = floor_w <= 1;
    if (!plain) {
        e = get_or_compile("dev" + std::to_string(dev) + ":" + inst + "/w" + std::to_string(floor_w), make_src(floor_w), inst);
        if (e.failed || entry_scratch_bytes(e) != 0) plain = true;
    }
    if (plain) e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, make_src(1), inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const int F = cfg.n;
    const size_t lane_lds = col ? (size_t)((F + (F >> 4) + 2) | 1) : (size_t)((F + (F >> 4) + 3) & ~1);
    const size_t lanes_lds = (size_t)lpb * lane_lds * 2 * sizeof(T);
    const size_t lds = lanes_lds + (col ? col_post_table_bytes(F, 2 * sizeof(T), gop, lanes_lds) : 0);     // = RealPow2Kernel::LDS_BYTES (pow2_real.h: the column tiles' POST tables)
    if (lds > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    if (col) real_args_set_inner_shift(arg, lpb);
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
// The two passes of the ROW four-step (exec.hip: big_fft) for a smooth NON-power-of-two factor cfg.n (round 6): the column kernels of pow2_real.h specialised like every other smooth
// length -- pass 1 = column load / ROW store (ROWOUT), pass 2 = four-step twiddle on load / column store (CS = 4) -- so that such lanes take two passes over HBM like the powers of two
// instead of the six of the transpose route (ndfft 85 x 196608 c128: 562 -> 238 us).  Whole or partial butterfly rounds, tiles of 128-byte rows.
static int jit_fourstep_lanes(int dtype, const JitCfg &cfg) {
    for (int l : {dtype == NDFFT_F32 ? 16 : 8, 8})
        if (l * cfg.tpl <= 1024) return l;
    return 0;
}
static size_t jit_fourstep_lds(int dtype, const JitCfg &cfg, int lpb, int pass) {
    const size_t csz = dtype == NDFFT_F32 ? 8 : 16;
    const size_t lanes = (size_t)lpb * (size_t)((cfg.n + (cfg.n >> 4) + 2) | 1) * csz;
    const int r0 = cfg.radix.empty() ? 1 : cfg.radix[0], e0 = ((cfg.n / r0 + cfg.tpl - 1) / cfg.tpl) * r0;      // RealPow2Kernel::CS4_E0
    return lanes + (pass == 2 ? (size_t)e0 * lpb * csz : 0);      // = RealPow2Kernel::LDS_BYTES (CS = 4: + the tile's step twiddles)
}
bool jit_fourstep_ok(int dtype, const JitCfg &cfg) {
    if (!rtc().ok || jit_disabled() || !NDFFT_DEV_INT("NDFFT_JIT_FOURSTEP", 1)) return false;
    if (cfg.n < 16 || cfg.tpl < 1 || cfg.radix.empty()) return false;
    const int lpb = jit_fourstep_lanes(dtype, cfg);
    return lpb > 0 && jit_fourstep_lds(dtype, cfg, lpb, 2) <= jit_lds_limit();
}
// Plan time: the recipe of a smooth non-power-of-two factor n for those passes -- whole butterfly rounds, and few enough elements per thread that a tile still brings a useful number
// of waves (a tile is 8 / 16 lanes x n / E threads): E <= 16 (f64) / 24 (f32) where such a recipe exists, else the default whole-round recipe
bool jit_fourstep_choose(int dtype, int n, JitCfg &cfg) {
    if (jit_disabled() || n < 48 || n > 2048 || pow2_supported(dtype, n)) return false;
    // fewest passes at any E first; then the smallest cap on E whose recipe needs at most one pass more (320 f64: 10.4.4.2 at E = 20, not the six passes E <= 16 would force;
    // 384 f64: 6.4.4.4 at E = 12 rather than 8.8.6 at E = 24 on 16 threads per lane)
    JitCfg cmin, c;
    if (jit_choose_default(dtype, n, cmin, false) && !cmin.partial && cmin.e <= (dtype == NDFFT_F32 ? 32 : 24)) {
        c = cmin;
        for (int cap : {10, 12, 16, 20}) {
            if (cap >= cmin.e) break;
            JitCfg t;
            if (jit_choose_default(dtype, n, t, false, cap) && !t.partial && t.radix.size() <= cmin.radix.size() + 1) { c = t; break; }
        }
    } else {
        // no whole-round recipe with a bearable E (factors 7 / 11 / 13, mixed 2-3-5): the rows' recipe with partial rounds (the planner's pick by cost)
        if (!jit_choose(dtype, n, c, true) || c.e > (dtype == NDFFT_F32 ? 32 : 24)) return false;
    }
    const int lpb = jit_fourstep_lanes(dtype, c);
    if (lpb <= 0 || jit_fourstep_lds(dtype, c, lpb, 2) > jit_lds_limit()) return false;
    cfg = c;
    return true;
}
// the first pass of the REAL four-step (exec.hip: real_fourstep) for a real FFT length N1 = 2 cfg.n that is not a power of two: lanes of 128-byte REAL rows, halved like RfsGeom
static int jit_rfs1_lanes(int dtype, const JitCfg &cfg) {
    const size_t lane = (size_t)((cfg.n + (cfg.n >> 4) + 2) | 1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
    int l = dtype == NDFFT_F32 ? 32 : 16;
    while (l > 8 && (l * cfg.tpl > 1024 || (size_t)l * lane > 80 * 1024)) l /= 2;
    return (l * cfg.tpl <= 1024 && (size_t)l * lane <= jit_lds_limit()) ? l : 0;
}
bool jit_rfs1_ok(int dtype, const JitCfg &cfg) {
    if (!rtc().ok || jit_disabled() || !NDFFT_DEV_INT("NDFFT_JIT_FOURSTEP", 1)) return false;
    return cfg.n >= 16 && cfg.tpl >= 1 && !cfg.radix.empty() && jit_rfs1_lanes(dtype, cfg) > 0;
}
// the first pass of the INVERSE real four-step (col_direct.h modes 7 / 8: C2R, DCT-III) for a complex length cfg.n that is not a power of two: the lane-fastest kernel, whole rounds only
static size_t jit_rfsi_lds(int dtype, const JitCfg &cfg, int lpb) {
    const size_t rsz = dtype == NDFFT_F32 ? 4 : 8;
    return (size_t)lpb * (size_t)(cfg.n + (cfg.n >> NDFFT_PHI_SHIFT) + 1) * rsz + (size_t)cfg.e * lpb * 2 * rsz;      // ColDirectKernel::LDS_BYTES: half exchange + step twiddles
}
bool jit_rfsi_ok(int dtype, const JitCfg &cfg) {
    if (!rtc().ok || jit_disabled() || !NDFFT_DEV_INT("NDFFT_JIT_FOURSTEP", 1)) return false;
    if (cfg.n < 16 || cfg.tpl < 1 || cfg.radix.empty() || cfg.partial || cfg.e * cfg.tpl != cfg.n) return false;
    const int lpb = jit_fourstep_lanes(dtype, cfg);
    return lpb > 0 && jit_rfsi_lds(dtype, cfg, lpb) <= jit_lds_limit();
}
template <typename T> static int launch_jit_rfsi(int mode, const JitCfg &cfg, const RealArgs<T> &a, hipStream_t s) {
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    if (!jit_rfsi_ok(dtype, cfg)) return NDFFT_ERR_UNSUPPORTED;
    const int lpb = jit_fourstep_lanes(dtype, cfg), threads = cfg.tpl * lpb;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const std::string inst = std::string("ColDirectKernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " + std::to_string(lpb) + ", RadixList<" + radix_list(cfg) + ">, " +
                             std::to_string(G_C2C_FWD) + ", " + std::to_string(mode) + ">";
    const std::string src = std::string("#include \"col_direct.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") + std::to_string(threads) +
                            ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const size_t lds = jit_rfsi_lds(dtype, cfg, lpb);
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
// kind: 1 = complex pass 1 (ROWOUT), 2 = complex pass 2 (CS 4), 11 = real pass 1 (R2C, ROWOUT), 12 / 13 = real pass 2 (CS 5: R2C / DCT-I, CS 6: DCT-II),
//       14 / 15 = inverse real pass 1 (col_direct.h modes 7 / 8: C2R / DCT-III)
template <typename T> int launch_jit_fourstep(int kind, bool inverse, const JitCfg &cfg, const RealArgs<T> &a, hipStream_t s) {
    if (kind == 14 || kind == 15 || kind == 16) return launch_jit_rfsi<T>(kind == 14 ? 7 : kind == 15 ? 8 : 9, cfg, a, s);      // (16: second pass of the fused DCT-IV four-step, mode 9)
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    const bool real1 = kind == 11;
    if (real1 ? !jit_rfs1_ok(dtype, cfg) : !jit_fourstep_ok(dtype, cfg)) return NDFFT_ERR_UNSUPPORTED;
    const int lpb = real1 ? jit_rfs1_lanes(dtype, cfg) : jit_fourstep_lanes(dtype, cfg), threads = cfg.tpl * lpb;
    const int pass = (kind == 1 || kind == 11) ? 1 : 2;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const std::string inst = std::string("RealPow2Kernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " + std::to_string(lpb) + ", RadixList<" + radix_list(cfg) + ">, " +
                             std::to_string(real1 ? G_R2C_EVEN : inverse ? G_C2C_INV : G_C2C_FWD) + ", true, false, " +
                             (pass == 1 ? std::string("0, true") : std::to_string(kind == 12 ? 5 : kind == 13 ? 6 : 4) + ", false") + ">";
    const std::string src = std::string("#include \"pow2_real.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") + std::to_string(threads) +
                            ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const size_t lds = jit_fourstep_lds(dtype, cfg, lpb, pass);      // (real pass 1: the lane region of F complex = the raw real lane; no POST table with ROWOUT)
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    real_args_set_inner_shift(arg, lpb);
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
template int launch_jit_fourstep<float>(int, bool, const JitCfg &, const RealArgs<float> &, hipStream_t);
template int launch_jit_fourstep<double>(int, bool, const JitCfg &, const RealArgs<double> &, hipStream_t);

// PlainRealKernel (plain_kernel.h): the odd-n forms of the real-data ops for a smooth inner FFT length cfg.n, rows and column tiles
template <typename T> int launch_jit_plain(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    const int lpb = col ? jit_col_lanes(dtype, cfg) : cfg.row_lpb > 0 ? cfg.row_lpb : (cfg.tpl >= 64 ? 1 : std::max(1, 64 / cfg.tpl));
    if (lpb <= 0) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const int threads = cfg.tpl * lpb;
    const std::string inst = std::string("PlainRealKernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " +
                             std::to_string(lpb) + ", RadixList<" + radix_list(cfg) + ">, " + std::to_string(gop) + ", " + (col ? "true" : "false") + ">";
    const std::string src = std::string("#include \"plain_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") +
                            std::to_string(threads) + ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const int F = cfg.n;
    const size_t lane_lds = col ? (size_t)((F + (F >> 4) + 3) | 1) : (size_t)((F + (F >> 4) + 4) & ~1);     // PlainRealKernel::LANE_LDS
    const size_t lds = (size_t)lpb * lane_lds * 2 * sizeof(T);
    if (lds > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
template int launch_jit_plain<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_plain<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

// BlueKernel (blue_kernel.h) for Bluestein length cfgM.n = M: every op, rows and column tiles
template <typename T> int launch_jit_blue(int gop, const JitCfg &cfg, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (!rtc().ok) return NDFFT_ERR_UNSUPPORTED;
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    const int row_thr = (int)NDFFT_DEV_INT("NDFFT_BLUE_ROW_THREADS", 256);
    const int lpb = col ? jit_col_lanes(dtype, cfg) : cfg.row_lpb > 0 ? cfg.row_lpb : (cfg.tpl >= row_thr ? 1 : std::max(1, row_thr / cfg.tpl));
    if (lpb <= 0) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const int threads = cfg.tpl * lpb;
    const std::string inst = std::string("BlueKernel<") + tn + ", " + std::to_string(cfg.n) + ", " + std::to_string(cfg.tpl) + ", " +
                             std::to_string(lpb) + ", RadixList<" + radix_list(cfg) + ">, " + std::to_string(gop) + ", " + (col ? "true" : "false") + ">";
    const std::string src = std::string("#include \"blue_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") +
                            std::to_string(threads) + ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const int M = cfg.n;
    const size_t lane_lds = col ? (size_t)((M + (M >> 4) + 2) | 1) : (size_t)((M + (M >> 4) + 3) & ~1);
    const size_t lds = (size_t)lpb * lane_lds * 2 * sizeof(T);
    if (lds > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
// ---- Rader / Good-Thomas kernel (rader_kernel.h) ---------------------------------------------------------------------
static bool rader_enabled() { return sw().rader; }   // NDFFT_RADER=0 keeps every such length on Bluestein
static size_t rader_lane_lds(const RaderCfg &rc, bool col) {   // complex elements per lane = RaderKernel::LANE_LDS
    const size_t M = (size_t)rc.conv_len(), F = (size_t)rc.p * rc.mc;
    const size_t zlen = rc.sym ? (size_t)((rc.p + 1) / 2) * rc.mc : F;                  // RaderKernel::ZLEN / ZRAW
    const size_t zraw = std::max(zlen + (zlen >> 4) + 3, rc.sym ? (F + 2) / 2 : (size_t)0);
    const size_t sub = M + (M >> 4) + 2, lane = std::max((size_t)rc.rows() * sub, zraw);
    (void)col;
    return lane | 1;
}
// lanes per workgroup of the row kernel.  Measured (profiles/r04/r04d_rader_tune_lpb.txt): workgroups of ONE wave win wherever a lane needs
// <= 64 threads (no real barriers: 127 c128 150 -> 121 us, 511 c128 125 -> 110 us), otherwise the fullest waves with the fewest of them.
static int row_lanes_by_fill(int lt, size_t lane, int forced, double *util_out);
static int rader_row_lanes_for(int dtype, const RaderCfg &rc, double *util_out) {
    const int forced = (int)NDFFT_DEV_INT("NDFFT_RADER_LPB", 0);
    return row_lanes_by_fill(rc.fft.tpl * rc.rows(), rader_lane_lds(rc, false) * 2 * (dtype == NDFFT_F32 ? 4 : 8), forced, util_out);
}
static int rader_row_lanes(int dtype, const RaderCfg &rc) { return rader_row_lanes_for(dtype, rc, nullptr); }

// Configuration of FFT_(p-1): radix list (any multiset of 2..13, 16), threads per transform, partial rounds allowed.
// cost = passes x (work incl. idle threads of partial rounds) / (fill of the workgroup's waves), 13 % / 5 % off for one- / two-wave
// workgroups, plus a penalty for many elements per thread (f64: e = 21 costs 5-20 %, e = 24 twice the time) -- fitted to the sweeps
// under profiles/r04/r04c_rader_tune.txt and r04d_rader_tune_lpb.txt (tools/probes/rader_tune.py).
static bool rader_plan_fft(int dtype, int M, RaderCfg &rc, int wide) {
    FftPlanOpts o; o.rader = true; o.sym = rc.sym; o.half = rc.half();
    const size_t lane = rader_lane_lds(rc, false) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
    bool ok = false;
    if (rc.sym && !rc.half()) {      // (cofactor 1: one row per lane, the ordinary planner -- the full-wave rule put nddct1 n = 128 f32 on 4 threads per lane: 120 us against 69 us)
        o.full_waves = true;
        ok = plan_fft_by_cost(dtype, M, rc.rows(), lane, rc.fft, wide, nullptr, o);
        o.full_waves = false;
    }
    if (!ok) ok = plan_fft_by_cost(dtype, M, rc.rows(), lane, rc.fft, wide, nullptr, o);
    return ok;
}
// lanes per workgroup for `lt` threads per lane and `lane` bytes of LDS per lane: one wave where a lane needs <= 64 threads, else the
// fullest waves with the fewest of them (see rader_row_lanes_for)
static int row_lanes_by_fill(int lt, size_t lane, int forced, double *util_out) {
    int best = 0; double best_util = 0.0;
    // one wave where a lane needs <= 64 threads (as many lanes as fill it); else up to 256 threads: among the lane counts whose waves are at least 85 % as full as the
    // fullest, the one that lets the most lanes share a CU's LDS, the smallest on a tie (measured: 1008 points on 84 threads 3 lanes 156 us / 2 lanes 160 us;
    // 2016 points on 126 threads 1 lane 144-150 us / 2 lanes 166 us -- 4 lanes per CU either way, smaller workgroups win; profiles/r04/r04zj_abab_rader_lanes.txt)
    if (forced > 0 || lt <= 64) {
        const int want = forced > 0 ? forced : 64 / lt;
        for (int l = want; l >= 1; --l) {
            const int thr = l * lt, waves = (thr + 63) / 64;
            if (thr > 1024 || (size_t)l * lane > jit_lds_limit()) { if (forced > 0) break; continue; }
            best = l; best_util = (double)thr / (64.0 * waves);
            break;
        }
    } else {
        const int lmax = std::max(1, 256 / lt);
        double umax = 0.0;
        for (int l = 1; l <= lmax; ++l) if ((size_t)l * lane <= jit_lds_limit()) umax = std::max(umax, (double)(l * lt) / (64.0 * ((l * lt + 63) / 64)));
        size_t best_cu = 0;
        for (int l = 1; l <= lmax; ++l) {
            if ((size_t)l * lane > jit_lds_limit()) break;
            const double util = (double)(l * lt) / (64.0 * ((l * lt + 63) / 64));
            if (util < 0.85 * umax) continue;
            const size_t cu = jit_lds_limit() / ((size_t)l * lane) * l;
            if (cu > best_cu) { best_cu = cu; best = l; best_util = util; }
        }
    }
    if (util_out) *util_out = best_util;
    return best;
}
static bool plan_fft_by_cost(int dtype, int M, int mc, size_t lane_bytes, JitCfg &out, int wide, double *cost_out, const FftPlanOpts &opts) {
    // f64 cap: 18 for the row kernels (1500 = 10.6.5.5 at e = 20 lost 10 %), 21 for the Rader kernel (mc > 0 marks it: 2016 = 16.9.7.2 on 126 threads, e = 21, 146 us
    // against 185 us for 12.12.7.2 on 168 threads, e = 14 -- two full waves against three at 7/8)
    const bool rader_call = opts.rader;
    const int emax = dtype == NDFFT_F32 ? 32 : (wide ? std::max(wide, rader_call ? 21 : 18) : (opts.half ? 18 : opts.sym ? 16 : rader_call ? 21 : 18)), esoft = dtype == NDFFT_F32 ? 21 : 18;
    const double eslope = dtype == NDFFT_F32 ? 0.05 : 0.1;
    // wide: M has one factor 17 or 19 (f32 also 23, 29, 31; Rader for primes like 103, 137, 191, 47, 59): that radix joins the list
    std::vector<int> cand = {16, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    if (wide) cand.insert(cand.begin(), wide);
    std::vector<int> cur;
    JitCfg best; double best_cost = 1e30;
    auto eval = [&]() {
        std::vector<int> tpls;
        for (int r : cur) for (int sdiv = 1; sdiv <= 4; ++sdiv) tpls.push_back((M / r + sdiv - 1) / sdiv);
        if (M < 200) for (int r : cur) tpls.push_back(2 * (M / r));      // short lanes: a pass may leave half its threads idle if that buys threads for the rest
        std::sort(tpls.begin(), tpls.end()); tpls.erase(std::unique(tpls.begin(), tpls.end()), tpls.end());
        for (int tpl : tpls) {
            if (tpl < 1 || tpl * mc > 1024) continue;
            if (opts.full_waves && (tpl * mc) % 64 != 0 && 64 % (tpl * mc) != 0) continue;
            int e = 0; double work = 0; bool partial = false;
            for (int r : cur) { const int nb = M / r, sl = (nb + tpl - 1) / tpl; e = std::max(e, sl * r); work += (double)sl * tpl * r; if (nb % tpl) partial = true; }
            if (e > emax) continue;
            double util = 0.0;
            const int lpb = row_lanes_by_fill(tpl * mc, lane_bytes, 0, &util);
            if (lpb <= 0) continue;
            const int waves = (lpb * tpl * mc + 63) / 64;
            // + elements of the lane per thread beyond 4: the stage / gather / POST loops run on the lane's threads, and with few threads per lane few waves fit a
            // CU's LDS (profiles/r04/r04ze_rader_tune_short.txt: F = 34 on 2 threads 152 us, on 8 threads 76 us; F = 31 on 5 / 10 threads 118 / 90 us)
            // (short lanes only: the weights fade out between 128 and 256 points -- 1008 points on 84 threads and 2016 on 168 measured best, r04c / r04d;
            //  weights fitted to that sweep by grid search: on short lanes the FFT work counts 0.3x, each lane element per thread beyond 2 costs 0.2 -- worst pick 26 % / mean 6 % off
            //  the best configuration over its 11 cases)
            const double sw = std::min(1.0, std::max(0.0, (256.0 - (double)(M + 1) * mc) / 128.0));   // by lane length: 1 up to 128 elements, 0 from 256 (the sweep's range)
            const double pe = (double)(M + 1) / tpl;
            const double cost = (1.0 - 0.7 * sw) * work / M / util * (waves == 1 ? 0.87 : waves == 2 ? 0.95 : 1.0) + (e > esoft ? eslope * (e - esoft) : 0.0) + 0.01 * e +
                                0.2 * sw * std::max(0.0, pe - (4.0 - 2.0 * sw));
            if (cost < best_cost) { best_cost = cost; best = JitCfg(); best.n = M; best.tpl = tpl; best.e = e; best.radix = cur; best.partial = partial; best.lpb = lpb; }
        }
    };
    std::function<void(int, int)> rec = [&](int m, int maxr) {
        if (m == 1) { eval(); return; }
        if (cur.size() >= 6) return;
        for (int c : cand) {
            if (c > maxr || m % c) continue;
            cur.push_back(c);
            rec(m / c, c);
            cur.pop_back();
        }
    };
    rec(M, wide ? wide : 16);
    if (best.radix.empty()) return false;
    out = best;
    if (cost_out) *cost_out = best_cost;
    return true;
}

// real-op slots: inner FFTs below 128 points are re-planned from e > 8 in f64 as well -- the PRE / POST loops run on the lane's threads, so short lanes want
// more of them (nddct2 f64 n = 63: 9.7 on 9 threads 49 us, 7.3.3 on 21 threads 39 us; the even-n forms n = 144..240 moved by -1..+5 % under the same rule,
// profiles/r04/r04zg_abab_shortplan.txt)
bool jit_choose_real(int dtype, int F, JitCfg &cfg) {
    if (!jit_choose(dtype, F, cfg, true)) return false;
    if (!NDFFT_DEV_STR("NDFFT_JIT_CFG") && F < 128 && cfg.e > 8 && cfg.row_lpb == 0) {
        const size_t lane = (size_t)((F + (F >> 4) + 3) & ~1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
        JitCfg alt;
        if (plan_fft_by_cost(dtype, F, 1, lane, alt)) { alt.vec = 1; alt.row_lpb = alt.lpb; cfg = alt; }
    }
    return true;
}

// ---- Bluestein on a smooth convolution length ---------------------------------------------------------------------------
// fewest radices (<= 16, incl. 12) whose product is m; 99 if m is not 13-smooth
static int min_passes(int m) {
    static const int cand[] = {16, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    if (m == 1) return 0;
    int best = 99;
    for (int c : cand) if (m % c == 0) { const int r = min_passes(m / c); if (r + 1 < best) best = r + 1; if (best <= 1) break; }
    return best;
}
int blue_pick_len(int dtype, int F, int m_pow2) {
    if (jit_disabled()) return m_pow2;
    // shortlist by passes x length, then the register planner's own cost (passes x work / wave fill + penalties) x length decides; the power of two
    // competes with its E = 8 recipe: passes x length (+ 5 %: a smooth length must win clearly)
    // Measured (profiles/r04/r04y_abab_blue_smooth.txt): per point the mixed-radix passes cost about 1.3x the power-of-two E = 8 recipe, so a smooth length only
    // pays when it is much shorter -- F = 263: M 1024 -> 550, 325 -> 211 us; F = 1283: 4096 -> 2592, 369 -> 279 us; but F = 227: 512 -> 480, 183 -> 228 us --
    // F = 83: 256 -> 169 (0.66), 217 -> 228 us -- hence only lengths up to 0.65 x the power of two are considered.
    std::vector<std::pair<long, int>> cands;
    for (int m = 2 * F - 1; m < m_pow2 && 100 * (long)m <= 65 * (long)m_pow2; ++m) {
        int q = m; for (int f : {2, 3, 5, 7, 11, 13}) while (q % f == 0) q /= f;
        if (q != 1) continue;
        const int np = min_passes(m);
        if (np < 99) cands.push_back({(long)m * np, m});
    }
    std::sort(cands.begin(), cands.end());
    JitCfg p2;
    const double pow2_cost = (pow2_real_config(m_pow2, p2) ? (double)p2.radix.size() : (double)min_passes(m_pow2)) * m_pow2;
    int best = m_pow2; double best_cost = pow2_cost * 0.95;
    for (size_t i = 0; i < cands.size() && i < 6; ++i) {
        const int m = cands[i].second;
        JitCfg c; double cost = 0.0;
        const size_t lane = (size_t)((m + (m >> 4) + 3) & ~1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
        if (lane > jit_lds_limit() || !plan_fft_by_cost(dtype, m, 1, lane, c, false, &cost)) continue;
        if (cost * m < best_cost) { best_cost = cost * m; best = m; }
    }
    return best;
}
bool blue_plan_cfg(int dtype, int M, JitCfg &cfg) {
    if (jit_disabled()) return false;
    const size_t lane = (size_t)((M + (M >> 4) + 3) & ~1) * 2 * (dtype == NDFFT_F32 ? 4 : 8);      // BlueKernel::LANE_LDS (rows)
    if (lane > jit_lds_limit()) return false;
    if (!plan_fft_by_cost(dtype, M, 1, lane, cfg)) return false;
    cfg.vec = 1; cfg.row_lpb = cfg.lpb;
    return true;
}

bool rader_choose(int dtype, int F, RaderCfg &rc, bool dct1_slot) {
    if (jit_disabled() || F < 17) return false;
    int p = 1, m = F;
    for (int f = 2; (int64_t)f * f <= m; ++f) while (m % f == 0) { p = f; m /= f; }
    if (m > 1) p = m;                                   // largest prime factor
    const int mc = F / p;
    if (p <= 13 || mc % p == 0) return false;
    // cofactor: one butterfly (2..16) or two butterfly factors held in one thread's registers (w[.][mc] next to the E elements of the FFT)
    rc.mc1 = mc; rc.mc2 = 1;
    if (mc > 16) {
        int n1 = 0, n2 = 0;
        // (f64: 32; 33 = 11 x 3 in the DCT-I slot only, where it is nddct1 n = 1024: F = 1023 = 33 x 31, otherwise Bluestein at 392 us -- round 5)
        if (mc > (dtype == NDFFT_F32 ? 48 : (dct1_slot && (mc & 1) ? 33 : 32)) || !regfft_factor(mc, &n1, &n2) || n1 > ((dct1_slot && (mc & 1)) ? NDFFT_DEV_INT("NDFFT_RADER_DCT1_MC", 23) : 16) || n2 > 16) return false;   // (DCT-I slot, ODD cofactor = the symmetric form only: a prime cofactor 17 / 19 / 23 as ONE butterfly -- nddct1 n = 2048: F = 2047 = 23 x 89; even cofactors 34 / 38 / 46 would build the full form with 34-46 complex registers per column, never measured: they stay on Bluestein)
        rc.mc1 = n1; rc.mc2 = n2;
    }
    // p - 1: 13-smooth, or with ONE factor 17 / 19 / 23 / 29 / 31 -- a pass of that radix, E >= that many complex registers
    int wide = 0;
    { int q = p - 1; for (int f : {2, 3, 5, 7, 11, 13}) while (q % f == 0) q /= f;
      // (f64 too since round 6 -- 31 complex doubles are 124 VGPRs, the kernel runs one or two waves per SIMD and still beats Bluestein: ndfft c128 2^24 points n = 139 253 -> 185 us,
      //  278 251 -> 148, 311 289 -> 231, 622 304 -> 240, 1244 277 -> 246, 233 unchanged at 205 us -- profiles/r09/r09n_rader_wide_f64.txt; NDFFT_RADER_WIDE_F64=0: developer build, A/B)
      if (q == 17 || q == 19 || ((dtype == NDFFT_F32 || NDFFT_DEV_INT("NDFFT_RADER_WIDE_F64", 1)) && (q == 23 || q == 29 || q == 31))) wide = q; else if (q != 1) return false;
      // (at least 6 butterflies of the wide radix per lane, over all cofactor rows: 47 = 23 x 2 + 1 alone would run on 2 threads per lane -- 300 us against
      //  Bluestein's 118 us for 2^24 points c64; 235 = 5 x 47: 91 against 119 us, 139 = 23 x 6 + 1: 105 against 145 us, 590 = 10 x 59: 82 against 173 us,
      //  profiles/r04/r04za_rader_f32_wide.txt)
      if (wide && mc * ((p - 1) / wide) < 6) return false; }
    rc.p = p; rc.mc = mc;
    // DCT-I with an odd cofactor > 1 (nddct1 n = 512: F = 511 = 7 x 73): even-symmetric inner FFT input, (mc + 1) / 2 of the mc Rader transforms
    // (developer build: NDFFT_RADER_SYM=0 keeps the full form for A/B runs)
    // (cofactor 1, F prime: the half-length convolution, RaderCfg::half)
    rc.sym = dct1_slot && (mc & 1) && p > 3 && NDFFT_DEV_INT("NDFFT_RADER_SYM", 1) != 0 && (mc > 1 || NDFFT_DEV_INT("NDFFT_RADER_HALF", 1) != 0);
    if (rader_lane_lds(rc, false) * 2 * (dtype == NDFFT_F32 ? 4 : 8) > jit_lds_limit()) return false;
    if (const char *e = NDFFT_DEV_STR("NDFFT_RADER_CFG")) {     // developer knob (tools/probes/rader_tune.py): "tpl:r0.r1.r2" for FFT_(p-1), read per plan
        JitCfg &c = rc.fft;
        c = JitCfg(); c.n = rc.conv_len(); c.tpl = atoi(e);
        const char *q = strchr(e, ':');
        int prod = 1;
        while (q && *q) { const int r = atoi(q + 1); if (r < 2) break; c.radix.push_back(r); prod *= r; q = strchr(q + 1, '.'); }
        if (c.tpl < 1 || prod != rc.conv_len() || c.tpl * rc.rows() > 1024) return false;
        for (int r : c.radix) { const int nb = c.n / r, sl = (nb + c.tpl - 1) / c.tpl; c.e = std::max(c.e, sl * r); if (nb % c.tpl) c.partial = true; }
        return true;
    }
    if (rader_plan_fft(dtype, rc.conv_len(), rc, wide)) return true;
    // no recipe under the symmetric form's caps: the full form (cap 21) may still exist -- before round 6 such a length dropped to Bluestein without trying it
    if (rc.sym && rc.mc1 <= 16) {
        rc.sym = false;
        if (rader_lane_lds(rc, false) * 2 * (dtype == NDFFT_F32 ? 4 : 8) <= jit_lds_limit() && rader_plan_fft(dtype, rc.conv_len(), rc, wide)) return true;
    }
    return false;
}
int rader_col_lanes(int dtype, const RaderCfg &rc) {
    const int lt = rc.fft.tpl * rc.rows();
    if (const char *e = NDFFT_DEV_STR("NDFFT_RADER_COL_LPB")) { const int l = atoi(e); return l * lt <= 1024 ? l : 0; }   // developer knob
    // whole multiples of 8 adjacent lanes (64-byte rows in f64): measured 512x65536 f64 DCT-I 16 lanes 260 us (8: 284, 12: 375), 1009x16384 c128
    // 8 lanes 207 us (9: 321), 127x131072 c64 24 lanes 87 us (8: 93, 16: 100, 32: 96)
    const size_t lane = rader_lane_lds(rc, true) * 2 * (dtype == NDFFT_F32 ? 4 : 8);
    for (int l : {24, 16, 8})
        if (l * lt <= 1024 && (size_t)l * lane <= jit_lds_limit()) return l;
    return 0;
}
template <typename T> int launch_jit_rader(int gop, const RaderCfg &rc, bool col, const RealArgs<T> &a, hipStream_t s) {
    if (!rtc().ok || !rader_enabled()) return NDFFT_ERR_UNSUPPORTED;
    const int dtype = sizeof(T) == 4 ? NDFFT_F32 : NDFFT_F64;
    const int lt = rc.fft.tpl * rc.rows();
    if (rc.sym && gop != G_DCT1) return NDFFT_ERR_UNSUPPORTED;       // (the symmetric recipe exists in the DCT-I slot only)
    const int lpb = col ? rader_col_lanes(dtype, rc) : rader_row_lanes(dtype, rc);
    if (lpb <= 0) return NDFFT_ERR_UNSUPPORTED;
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    const char *tn = sizeof(T) == 4 ? "float" : "double";
    const int threads = lt * lpb;
    const std::string inst = std::string("RaderKernel<") + tn + ", " + std::to_string(rc.p) + ", " + std::to_string(rc.mc1) + ", " + std::to_string(rc.mc2) + ", " + std::to_string(rc.fft.tpl) + ", " +
                             std::to_string(lpb) + ", RadixList<" + radix_list(rc.fft) + ">, " + std::to_string(gop) + ", " + (col ? "true" : "false") + ", " + (rc.sym ? "true" : "false") + ">";
    const std::string src = std::string("#include \"rader_kernel.h\"\nusing namespace ndfft;\nextern \"C\" __global__ __launch_bounds__(") +
                            std::to_string(threads) + ") void k_jit(const RealArgs<" + tn + "> a) { " + inst + "::run(a); }\n";
    const Entry e = get_or_compile("dev" + std::to_string(dev) + ":" + inst, src, inst);
    if (e.failed) return NDFFT_ERR_UNSUPPORTED;
    const size_t lds = (size_t)lpb * rader_lane_lds(rc, col) * 2 * sizeof(T);
    if (lds > jit_lds_limit()) return NDFFT_ERR_UNSUPPORTED;
    const int64_t nblk = (a.nlanes + lpb - 1) / lpb;
    if (nblk <= 0) return NDFFT_OK;
    if (nblk > 0x7fffffffLL) return NDFFT_ERR_UNSUPPORTED;
    RealArgs<T> arg = a;
    void *params[] = {(void *)&arg};
    NDFFT_HIP(hipModuleLaunchKernel(e.fn, (unsigned)nblk, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds, s, params, nullptr));
    return NDFFT_OK;
}
template int launch_jit_rader<float>(int, const RaderCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_rader<double>(int, const RaderCfg &, bool, const RealArgs<double> &, hipStream_t);

template int launch_jit_blue<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_blue<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

template int launch_jit_real<float>(int, const JitCfg &, bool, const RealArgs<float> &, hipStream_t);
template int launch_jit_real<double>(int, const JitCfg &, bool, const RealArgs<double> &, hipStream_t);

// ndfft_jit_prebuild (see "the shipped code objects as BUILD products" above): entries [first, first + stride, ...] of the manifest, so that several processes can share it
int jit_prebuild(const char *manifest, const char *out_dir, int first, int stride, int *built, int *present, int *failed) {
    *built = *present = *failed = 0;
    std::string all;
    if (!manifest || !out_dir || !read_file(manifest, all)) return fail(NDFFT_ERR_INVALID_ARG, "ndfft_jit_prebuild: cannot read the manifest");
    if (!rtc().ok) return fail(NDFFT_ERR_UNSUPPORTED, "ndfft_jit_prebuild: libhiprtc not available");
    if (mkdir(out_dir, 0755) != 0 && errno != EEXIST) return fail(NDFFT_ERR_INVALID_ARG, "ndfft_jit_prebuild: cannot create the output directory");
    const char *const *hs = kJitHdrSrc;
    const std::string sep(kManifestSep);
    std::vector<std::string> srcs;
    for (size_t pos = 0; pos < all.size();) {
        const size_t e = all.find(sep, pos);
        std::string one = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        pos = e == std::string::npos ? all.size() : e + sep.size();
        if (one.find("k_jit") == std::string::npos) continue;
        if (std::find(srcs.begin(), srcs.end(), one) == srcs.end()) srcs.push_back(one);
    }
    if (stride < 1) stride = 1;
    for (size_t i = first < 0 ? 0 : (size_t)first; i < srcs.size(); i += (size_t)stride) {
        const std::string path = std::string(out_dir) + cache_name(srcs[i], hs, kNJitHdr);
        std::string have;
        if (read_file(path, have)) { (void)utime(path.c_str(), nullptr); ++*present; continue; }   // (touched: the caller removes objects older than its pass = of older kernel text)
        const std::string code = rtc_compile(srcs[i], "manifest entry " + std::to_string(i));
        if (code.empty()) { ++*failed; continue; }
        write_file_atomic(path, code);
        ++*built;
    }
    return NDFFT_OK;
}

}  // namespace ndfft

extern "C" int ndfft_jit_prebuild(const char *manifest, const char *out_dir, int first, int stride, int *built, int *present, int *failed) {
    int b = 0, p = 0, f = 0;
    const int rc = ndfft::jit_prebuild(manifest, out_dir, first, stride, &b, &p, &f);
    if (built) *built = b;
    if (present) *present = p;
    if (failed) *failed = f;
    return rc;
}
