// plan.hip -- ndfft_plan: factorisation, long-double twiddle tables, lazy per-device upload.
// Stands for FftHandler::new / R2cFftHandler::new / DctHandler::new (src/lib.rs:294, 477, 665):
// plans are built eagerly (DctHandler plans all four types, lib.rs:666-670) and are immutable
// afterwards, so one plan can be shared by any number of host threads (lib.rs:192-194).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <tuple>

#include "engine.h"

namespace ndfft {

static thread_local std::string g_err;
static thread_local const char *g_path = "";

int fail(int code, const std::string &msg) { g_err = msg; return code; }
void set_last_path(const char *p) { g_path = p; }
const char *last_path() { return g_path; }
const std::string &last_err() { return g_err; }
void clear_err() { g_err.clear(); }

static const long double kPiL = 3.14159265358979323846264338327950288L;

// e^{-2 pi i num/den}, argument reduced exactly in integers first
static void unit(HostTable &t, unsigned long long num, unsigned long long den) {
    num %= den;
    long double ang = 2.0L * kPiL * (long double)num / (long double)den;
    t.re.push_back(cosl(ang));
    t.im.push_back(-sinl(ang));
}

// radix list for the LDS Stockham kernel.  Allowed radices: 2..10 (6, 9, 10 are composite butterflies that
// save passes; 12 and 16 cost more in registers than they save: measured) and the primes 11, 13.
// Among all factorisations the search keeps the one with the fewest passes, then the largest minimum
// radix (balanced passes keep every thread busy: 96 = 6*4*4, not 8*6*2), then the smallest maximum.
static void factor_search(int m, int max_r, std::vector<int> &cur, std::vector<int> &best) {
    if (m == 1) {
        auto key = [](const std::vector<int> &v) {
            int mn = 1 << 30, mx = 0;
            for (int r : v) { mn = std::min(mn, r); mx = std::max(mx, r); }
            return std::make_tuple((int)v.size(), -mn, mx);
        };
        if (best.empty() || key(cur) < key(best)) best = cur;
        return;
    }
    if (!best.empty() && cur.size() + 1 > best.size()) return;
    static const int cand[] = {13, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
    for (int c : cand) {
        if (c > max_r || m % c) continue;
        cur.push_back(c);
        factor_search(m / c, c, cur, best);   // non-increasing order: each multiset is visited once
        cur.pop_back();
    }
}
static bool factorize(int F, std::vector<int> &radix) {
    radix.clear();
    if (F <= 1) return true;
    int m = F;   // reject early if a prime factor > 13 remains
    for (int p : {2, 3, 5, 7, 11, 13}) while (m % p == 0) m /= p;
    if (m != 1) return false;
    std::vector<int> cur;
    factor_search(F, 13, cur, radix);
    // largest radix first: the first pass reads global memory directly with R independent loads in flight
    std::sort(radix.begin(), radix.end(), [](int x, int y) { return x > y; });
    return !radix.empty() && radix.size() <= (size_t)kMaxPasses;
}

static ndfft_plan *make_plan(int kind, int dtype, size_t n);
static void add_narrow_tables(ndfft_plan *p);
static void add_colsplit(ndfft_plan *p);
static void add_real_fourstep(ndfft_plan *p);

// per-pass transposed twiddles of the LDS Stockham kernel: for pass p (radix R, Ns = product of the
// earlier radices) the block  tw_p[(r-1) Ns + k] = e^{-2 pi i r k/(Ns R)},  r in [1,R), k in [0,Ns);
// blocks are concatenated, pass 0 (Ns = 1, all ones) is skipped.  Consecutive butterflies have consecutive
// k, so a wave reads them coalesced.
static void build_pass_twiddles(HostTable &t, const std::vector<int> &radix, int len) {
    (void)len;
    unsigned long long Ns = 1;
    for (size_t p = 0; p < radix.size(); ++p) {
        const unsigned long long R = (unsigned long long)radix[p];
        if (p > 0)
            for (unsigned long long r = 1; r < R; ++r)
                for (unsigned long long k = 0; k < Ns; ++k) unit(t, r * k, Ns * R);
        Ns *= R;
    }
    if (t.re.empty()) unit(t, 0, 1);   // single-pass transforms: keep the table non-empty
}

static int blue_len(int F) { int M = 1; while (M < 2 * F - 1) M <<= 1; return M; }

// can ONE launch of the row kernels transform a lane of this length?
static bool single_kernel_ok(int len, int dtype) {
    const size_t maxlen = generic_max_len(dtype == NDFFT_F32 ? 8 : 16);
    if (len <= 1 || pow2_supported(dtype, len)) return true;
    std::vector<int> r;
    if (factorize(len, r)) return (size_t)len <= maxlen;
    return (size_t)blue_len(len) <= maxlen;
}

// chirp[j] = e^{-i pi j^2/F} (j < F) and bhat = FFT_M(conj chirp, wrapped) / M, computed in long double by an
// iterative radix-2 FFT (per-stage twiddle tables) so that the device tables are correctly rounded
static void dft_ld(std::vector<long double> &re, std::vector<long double> &im);
static void build_bluestein_tables(FftConfig &c, int F, int M) {
    if (M & (M - 1)) {      // smooth non-power-of-two M: the general long-double DFT below
        for (int j = 0; j < F; ++j) unit(c.chirp, ((unsigned long long)j * j) % (2ull * F), 2ull * F);
        std::vector<long double> br(M, 0.0L), bi(M, 0.0L);
        for (int j = 0; j < F; ++j) {
            br[j] = c.chirp.re[j]; bi[j] = -c.chirp.im[j];
            if (j) { br[M - j] = br[j]; bi[M - j] = bi[j]; }
        }
        dft_ld(br, bi);
        c.bhat.re.reserve(M); c.bhat.im.reserve(M);
        for (int k = 0; k < M; ++k) { c.bhat.re.push_back(br[k] / M); c.bhat.im.push_back(bi[k] / M); }
        return;
    }
    for (int j = 0; j < F; ++j) unit(c.chirp, ((unsigned long long)j * j) % (2ull * F), 2ull * F);
    std::vector<long double> br(M, 0.0L), bi(M, 0.0L);
    for (int j = 0; j < F; ++j) {
        br[j] = c.chirp.re[j]; bi[j] = -c.chirp.im[j];
        if (j) { br[M - j] = br[j]; bi[M - j] = bi[j]; }
    }
    for (int i = 1, j = 0; i < M; ++i) {
        int bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
    }
    std::vector<long double> wr, wi;
    for (int len = 2; len <= M; len <<= 1) {
        const int h = len / 2;
        wr.resize(h); wi.resize(h);
        for (int k = 0; k < h; ++k) { const long double ang = -2.0L * kPiL * (long double)k / (long double)len; wr[k] = cosl(ang); wi[k] = sinl(ang); }
        for (int i = 0; i < M; i += len)
            for (int k = 0; k < h; ++k) {
                const int a = i + k, b = i + k + h;
                const long double tr = br[b] * wr[k] - bi[b] * wi[k], ti = br[b] * wi[k] + bi[b] * wr[k];
                br[b] = br[a] - tr; bi[b] = bi[a] - ti; br[a] += tr; bi[a] += ti;
            }
    }
    c.bhat.re.reserve(M); c.bhat.im.reserve(M);
    for (int k = 0; k < M; ++k) { c.bhat.re.push_back(br[k] / M); c.bhat.im.push_back(bi[k] / M); }
}

// ---- Rader / Good-Thomas tables (rader_kernel.h) ------------------------------------------------------------------------
// forward DFT of any length in long double: recursive decimation in time over the smallest prime factor (plan time only)
static void dft_ld(std::vector<long double> &re, std::vector<long double> &im) {
    const size_t n = re.size();
    if (n <= 1) return;
    size_t r = n;
    for (size_t f = 2; f * f <= n; ++f) if (n % f == 0) { r = f; break; }
    const size_t m = n / r;
    std::vector<std::vector<long double>> sr(r, std::vector<long double>(m)), si(r, std::vector<long double>(m));
    for (size_t j = 0; j < r; ++j)
        for (size_t i = 0; i < m; ++i) { sr[j][i] = re[i * r + j]; si[j][i] = im[i * r + j]; }
    if (m > 1) for (size_t j = 0; j < r; ++j) dft_ld(sr[j], si[j]);
    std::vector<long double> wr(n), wi(n);
    for (size_t k = 0; k < n; ++k) { const long double ang = -2.0L * kPiL * (long double)k / (long double)n; wr[k] = cosl(ang); wi[k] = sinl(ang); }
    for (size_t k = 0; k < n; ++k) {
        long double ar = 0.0L, ai = 0.0L;
        for (size_t j = 0; j < r; ++j) {
            const size_t w = (j * k) % n;
            const long double xr = sr[j][k % m], xi = si[j][k % m];
            ar += xr * wr[w] - xi * wi[w]; ai += xr * wi[w] + xi * wr[w];
        }
        re[k] = ar; im[k] = ai;
    }
}
static void build_rader_tables(FftConfig &c) {
    const int p = c.radercfg.p, M = p - 1;
    auto powmod = [&](unsigned long long b, unsigned long long e) { unsigned long long r = 1; b %= p; while (e) { if (e & 1) r = r * b % p; b = b * b % p; e >>= 1; } return r; };
    std::vector<int> pf;
    { int m = M; for (int f = 2; f * f <= m; ++f) if (m % f == 0) { pf.push_back(f); while (m % f == 0) m /= f; } if (m > 1) pf.push_back(m); }
    int g = 2;
    for (;; ++g) { bool ok = true; for (int f : pf) if (powmod(g, M / f) == 1) { ok = false; break; } if (ok) break; }   // smallest primitive root
    c.rader_tab.assign(2 * (size_t)M, 0);
    unsigned long long x = 1;
    for (int i = 0; i < M; ++i) { c.rader_tab[i] = (int32_t)x; c.rader_tab[M + (M - i) % M] = (int32_t)x; x = x * g % p; }   // g^i; g^-j = g^(M - j)
    std::vector<long double> br(M), bi(M);
    for (int q = 0; q < M; ++q) { const long double ang = -2.0L * kPiL * (long double)c.rader_tab[M + q] / (long double)p; br[q] = cosl(ang); bi[q] = sinl(ang); }
    // half-length form (RaderCfg::half: DCT-I with F = p prime, even input): a is periodic with period MC = M / 2, so only the even bins of FFT_M(a) are non-zero and
    // (a (*) b)[t] = IFFT_MC(FFT_MC(a[0..MC)) . bh)[t], bh = FFT_MC(b[q] + b[q + MC]) / MC   (b[q] + b[q + MC] = 2 cos(2 pi g^-q / p): the table is built from the sum)
    const int MC = c.radercfg.conv_len();
    if (c.radercfg.half()) {
        for (int q = 0; q < MC; ++q) { br[q] += br[q + MC]; bi[q] += bi[q + MC]; }
        br.resize(MC); bi.resize(MC);
    }
    dft_ld(br, bi);
    c.rader_bhat = HostTable();
    for (int k = 0; k < MC; ++k) { c.rader_bhat.re.push_back(br[k] / MC); c.rader_bhat.im.push_back(bi[k] / MC); }
    c.rader_twp = HostTable();
    build_pass_twiddles(c.rader_twp, c.radercfg.fft.radix, MC);
    c.rader_ctw = HostTable();
    for (int k = 0; k < c.radercfg.mc; ++k) unit(c.rader_ctw, (unsigned long long)k, (unsigned long long)c.radercfg.mc);   // W_mc^k: inner twiddles of a two-factor cofactor
    c.rader_twp2 = HostTable();
    build_pass_twiddles(c.rader_twp2, std::vector<int>(c.radercfg.fft.radix.rbegin(), c.radercfg.fft.radix.rend()), MC);
}

// fills F, radix / Bluestein, tw (and twM, chirp, bhat); or the four-step split for long lanes
static void build_fft(FftConfig &c, int F, int dtype) {
    c.F = F;
    if (F <= 1) return;
    const size_t maxlen = generic_max_len(dtype == NDFFT_F32 ? 8 : 16);
    const bool smooth = factorize(F, c.radix);
    if (smooth && (size_t)F <= maxlen) {
        build_pass_twiddles(c.tw, c.radix, F);
        return;
    }
    // Bluestein needs M = blue_len(F) in ONE launch: the LDS kernel up to `maxlen`, the register kernel
    // (blue_kernel.h, hiprtc) wherever M has a register configuration (M <= 8192)
    JitCfg regcfg;
    const bool blue_lds = (size_t)blue_len(F) <= maxlen, blue_reg = pow2_real_config(blue_len(F), regcfg);
    if (smooth || !(blue_lds || blue_reg)) {
        // long lane: F = F1 * F2 with both halves inside one launch, as square as possible
        c.radix.clear();
        // (round 6: among the splits, the most square one whose BOTH factors can run the two-pass form -- a power of two with a column kernel, or a smooth length with a
        //  whole-round hiprtc recipe, jit.hip: jit_fourstep_choose -- wins over a squarer one that would take the six-pass transpose route)
        int best = 0, best2 = 0;
        auto two_pass_factor = [&](int f) { JitCfg t; return fourstep_supported(f) || jit_fourstep_choose(dtype, f, t); };
        for (int d = 2; (int64_t)d * d <= F; ++d)
            if (F % d == 0 && single_kernel_ok(d, dtype) && single_kernel_ok(F / d, dtype)) {
                best = d;
                if (two_pass_factor(d) && two_pass_factor(F / d)) best2 = d;
            }
        if (best2) best = best2;
        if (!best) {
            // a prime factor too large for any single launch: Bluestein over global memory -- chirp multiply,
            // FFT_M through the power-of-two row path (its own four-step when M > 16384), * bhat, FFT_M, chirp
            const int M = blue_len(F);
            if (M > (1 << 21)) { c.unsupported = true; return; }
            c.big = true; c.bigblue = true; c.M = M;
            c.sub1 = make_plan(NDFFT_KIND_C2C, dtype, (size_t)M);
            build_bluestein_tables(c, F, M);
            return;
        }
        c.big = true; c.F2 = best; c.F1 = F / best;           // F1 >= F2
        c.sub1 = make_plan(NDFFT_KIND_C2C, dtype, (size_t)c.F1);
        c.sub2 = make_plan(NDFFT_KIND_C2C, dtype, (size_t)c.F2);
        int logF = 0; while ((1ll << logF) < F) ++logF;
        c.logB = (logF + 1) / 2;
        const int64_t B = 1ll << c.logB;
        for (int64_t k = 0; k < B && k < F; ++k) unit(c.twlo, k, F);
        for (int64_t k = 0; k * B < F; ++k) unit(c.twhi, k * B, F);
        return;
    }
    // Bluestein: chirp[j] = e^{-i pi j^2/F}; bhat = FFT_M(conj chirp wrapped)/M computed here in
    // long double by a direct radix-2 recursion so the device table is correctly rounded.
    c.blue = true;
    c.blue_reg_only = !blue_lds;
    c.radix.clear();
    // M: the next power of two, or -- where the LDS kernel can hold it too -- a cheaper 13-smooth length >= 2F - 1 (jit.hip: blue_pick_len)
    int M = blue_len(F);
    if (blue_lds) { const int Ms = blue_pick_len(dtype, F, M); std::vector<int> rm; if (Ms >= 2 * F - 1 && Ms < M && factorize(Ms, rm)) M = Ms; }
    c.M = M;
    factorize(M, c.radixM);
    build_pass_twiddles(c.twM, c.radixM, M);
    build_bluestein_tables(c, F, M);
    if (c.blue_reg_only) {   // without hiprtc the register kernel is not available: Bluestein over global memory instead
        c.big = true; c.bigblue = true;
        c.sub1 = make_plan(NDFFT_KIND_C2C, dtype, (size_t)M);
    }
}

static void build_plan_tables(ndfft_plan *p) {
    const int n = (int)p->n;
    for (int i = 0; i < CFG_COUNT; ++i) p->has_cfg[i] = false;
    if (n == 0) return;
    FftConfig &m = p->cfg[CFG_MAIN];
    if (p->kind == NDFFT_KIND_C2C) {
        build_fft(m, n, p->dtype);
        if (pow2_supported(p->dtype, n)) { m.pow2 = true; pow2_build_twiddles(p->dtype, n, m.twp); }
        if (pow2_real_supported(n)) pow2_real_build_twiddles(n, m.twp_col);
        fourstep_build_wide_twiddles(n, m.twp_col_w);
        if (!m.pow2 && jit_fourstep_choose(p->dtype, n, m.fs_jitcfg)) { m.fs_jit = true; jit_build_twiddles(m.fs_jitcfg, m.twp_fs); }
        if (wave_supported(n) || n <= 128) for (int k = 0; k < n; ++k) unit(m.wave_tw, k, n);   // W_n^k: wavefront and thread-per-lane kernels
        if (!m.pow2 && !m.blue && jit_choose(p->dtype, n, m.jitcfg, true)) {
            m.jit = true; jit_build_twiddles(m.jitcfg, m.twp);
            if (jit_choose_col(p->dtype, n, m.jitcfg, m.jitcfg_col)) { m.jit_col_alt = true; jit_build_twiddles(m.jitcfg_col, m.twp_jcol); }
        }
        p->has_cfg[CFG_MAIN] = true;
    } else if (p->kind == NDFFT_KIND_R2C) {
        if (n % 2 == 0) {
            build_fft(m, n / 2, p->dtype);
            for (int k = 0; k <= n / 2; ++k) unit(m.aux1, k, n);          // W_n^k
            if (pow2_real_supported(n / 2)) { m.pow2 = true; pow2_real_build_twiddles(n / 2, m.twp); }
        } else {
            build_fft(m, n, p->dtype);
        }
        p->has_cfg[CFG_MAIN] = true;
    } else {
        // DCT-II / DCT-III (Makhoul through a real FFT of length n)
        if (n % 2 == 0) {
            build_fft(m, n / 2, p->dtype);
            for (int k = 0; k <= n / 2; ++k) unit(m.aux1, k, n);          // W_n^k
            if (pow2_real_supported(n / 2)) { m.pow2 = true; pow2_real_build_twiddles(n / 2, m.twp); }
        } else {
            build_fft(m, n, p->dtype);
        }
        for (int k = 0; k < n; ++k) unit(m.aux2, k, 4ull * n);            // e^{-i pi k/(2n)}
        p->has_cfg[CFG_MAIN] = true;
        // DCT-I: real FFT of the even extension, length 2(n-1)
        if (n >= 2) {
            FftConfig &d1 = p->cfg[CFG_DCT1];
            build_fft(d1, n - 1, p->dtype);
            for (int k = 0; k <= n - 1; ++k) unit(d1.aux1, k, 2ull * (n - 1));
            if (pow2_real_supported(n - 1)) { d1.pow2 = true; pow2_real_build_twiddles(n - 1, d1.twp); }
            p->has_cfg[CFG_DCT1] = true;
        }
        // DCT-IV
        FftConfig &d4 = p->cfg[CFG_DCT4];
        if (n % 2 == 0) {
            build_fft(d4, n / 2, p->dtype);
            for (int j = 0; j < n / 2; ++j) unit(d4.aux1, 4ull * j + 1, 8ull * n);   // e^{-i pi (4j+1)/(4n)}
            for (int k = 0; k < n / 2; ++k) unit(d4.aux2, k, 2ull * n);              // e^{-i pi k/n}
            if (pow2_real_supported(n / 2)) { d4.pow2 = true; pow2_real_build_twiddles(n / 2, d4.twp); }
        } else {
            build_fft(d4, 2 * n, p->dtype);
            for (int j = 0; j < n; ++j) unit(d4.aux1, j, 4ull * n);                  // e^{-i pi j/(2n)}
            for (int k = 0; k < n; ++k) unit(d4.aux2, 2ull * k + 1, 8ull * n);       // e^{-i pi (2k+1)/(4n)}
        }
        p->has_cfg[CFG_DCT4] = true;
    }
}

// The real-data transforms of a very short lane as dense real matrices (tinymat_kernel.h), NO x NI row-major, in long double.
// Unnormalised, exactly the lane methods' definitions; the scalar of the normalisation is applied by the kernel.
static void push_real_matrix(HostTable &t, const std::vector<long double> &m) {
    for (size_t i = 0; i < m.size(); i += 2) { t.re.push_back(m[i]); t.im.push_back(i + 1 < m.size() ? m[i + 1] : 0.0L); }
}
static void build_tiny_mats(ndfft_plan *p) {
    const int n = (int)p->n;
    if (n < 2 || n > 16 || p->kind == NDFFT_KIND_C2C) return;
    FftConfig &c = p->cfg[CFG_MAIN];
    auto cs = [&](unsigned long long num, unsigned long long den, bool sine) {   // cos / sin of 2 pi num / den, argument reduced in integers
        num %= den;
        const long double ang = 2.0L * kPiL * (long double)num / (long double)den;
        return sine ? sinl(ang) : cosl(ang);
    };
    if (p->kind == NDFFT_KIND_R2C) {
        const int m = n / 2 + 1;
        std::vector<long double> f((size_t)2 * m * n), b((size_t)n * 2 * m, 0.0L);
        for (int k = 0; k < m; ++k)
            for (int j = 0; j < n; ++j) {
                const long double co = cs((unsigned long long)j * k, n, false), si = cs((unsigned long long)j * k, n, true);
                f[(size_t)(2 * k) * n + j] = co;                       // X[k] = sum x[j] e^{-2 pi i jk/n}   (lib.rs:497-503)
                f[(size_t)(2 * k + 1) * n + j] = -si;
                // x[j] = sum over the Hermitian-extended spectrum; imaginary parts of DC and Nyquist are dropped (lib.rs:516-521)
                const bool edge = k == 0 || (n % 2 == 0 && k == n / 2);
                const long double w = edge ? 1.0L : 2.0L;
                b[(size_t)j * 2 * m + 2 * k] = w * co;
                b[(size_t)j * 2 * m + 2 * k + 1] = edge ? 0.0L : -w * si;
            }
        push_real_matrix(c.tinymat[0], f); push_real_matrix(c.tinymat[1], b);
        return;
    }
    std::vector<long double> d1((size_t)n * n), d2((size_t)n * n), d3((size_t)n * n), d4((size_t)n * n);
    for (int k = 0; k < n; ++k)
        for (int j = 0; j < n; ++j) {
            // DCT-I: y[k] = x[0]/2 + (-1)^k x[n-1]/2 + sum_{0<j<n-1} x[j] cos(pi jk/(n-1))            (lib.rs:688-698)
            if (j == 0) d1[(size_t)k * n + j] = 0.5L;
            else if (j == n - 1) d1[(size_t)k * n + j] = (k % 2 ? -0.5L : 0.5L);
            else d1[(size_t)k * n + j] = cs((unsigned long long)j * k, 2ull * (n - 1), false);
            d2[(size_t)k * n + j] = cs((unsigned long long)k * (2 * j + 1), 4ull * n, false);            // cos(pi k(2j+1)/(2n))
            d3[(size_t)k * n + j] = j == 0 ? 0.5L : cs((unsigned long long)j * (2 * k + 1), 4ull * n, false);
            d4[(size_t)k * n + j] = cs((unsigned long long)(2 * j + 1) * (2 * k + 1), 8ull * n, false);  // cos(pi(2j+1)(2k+1)/(4n))
        }
    push_real_matrix(c.tinymat[0], d1); push_real_matrix(c.tinymat[1], d2); push_real_matrix(c.tinymat[2], d3); push_real_matrix(c.tinymat[3], d4);
}

static ndfft_plan *make_plan(int kind, int dtype, size_t n) {
    ndfft_plan *p = new ndfft_plan();
    p->kind = kind; p->dtype = dtype; p->n = n; p->refcount = 1;
    build_plan_tables(p);
    // W_F^k of every short inner FFT (F <= 128): twiddles of the thread-per-lane register kernels (reg_kernel.h)
    for (int i = 0; i < CFG_COUNT; ++i)
        if (p->has_cfg[i] && p->cfg[i].F >= 2 && p->cfg[i].F <= 128 && p->cfg[i].wave_tw.re.empty())
            for (int k = 0; k < p->cfg[i].F; ++k) unit(p->cfg[i].wave_tw, k, p->cfg[i].F);
    build_tiny_mats(p);
    add_narrow_tables(p);
    add_colsplit(p);
    add_real_fourstep(p);
    return p;
}

// Column four-step for long strided power-of-two lanes (C2C n >= 4096, R2C n >= 8192): n = F1 * 64.
// Only the MAIN slot of C2C / R2C plans; F1 (or its inner real FFT F1/2) must have a wide column kernel.
static void add_colsplit(ndfft_plan *p) {
    if (p->kind != NDFFT_KIND_C2C && p->kind != NDFFT_KIND_R2C) return;
    const size_t n = p->n;
    if (n == 0 || (n & (n - 1))) return;
    const int F2 = colsplit_inner_len();
    if (n % F2) return;
    const size_t F1 = n / F2;
    // shortest F1 per op, measured against the one-pass column tiles on 2^24-point arrays (profiles/r06/r06x_*):
    //   f64: C2C from n = 4096, R2C / C2R from n = 8192 (below: 4-lane / narrow column tiles win or tie)
    //   f32: C2C from n = 2048 (narrow tiles 135 us -> 110 us), R2C from n = 4096 (137 -> 57 us), C2R from n = 2048 (95 -> 55 us; n = 4096: 98 -> 58 us)
    const bool f32 = p->dtype == NDFFT_F32, c2c = p->kind == NDFFT_KIND_C2C;
    size_t lo_fwd = c2c ? (f32 ? 32 : 64) : (f32 ? 64 : 128), lo_inv = c2c ? lo_fwd : (f32 ? 32 : 128);
    if (const char *e = c2c ? NDFFT_DEV_STR("NDFFT_CS_LO_C2C") : NDFFT_DEV_STR("NDFFT_CS_LO_R2C")) lo_fwd = lo_inv = (size_t)atoi(e);   // developer knob
    const size_t lo = std::min(lo_fwd, lo_inv), hi = c2c ? 1024 : 2048;
    if (F1 < lo || F1 > hi) return;
    FftConfig &c = p->cfg[CFG_MAIN];
    c.cs_ops = c2c ? (F1 >= lo_fwd ? 1 : 0) : ((F1 >= lo_fwd ? 2 : 0) | (F1 >= lo_inv ? 4 : 0));
    c.cs = true; c.cs_F1 = (int)F1; c.cs_F2 = F2;
    c.cs_sub1 = make_plan(p->kind, p->dtype, F1);
    c.cs_sub2 = make_plan(NDFFT_KIND_C2C, p->dtype, (size_t)F2);
    int logn = 0; while (((size_t)1 << logn) < n) ++logn;
    c.cs_logB = (logn + 1) / 2;
    const int64_t B = 1ll << c.cs_logB;
    for (int64_t k = 0; k < B && k < (int64_t)n; ++k) unit(c.cs_twlo, k, n);
    for (int64_t k = 0; k * B < (int64_t)n; ++k) unit(c.cs_twhi, k * B, n);
}

// REAL four-step for long contiguous real-data lanes (R2C and DCT plans, MAIN slot, n = 2^e too long for one launch):
// n = N1 * N2, a real FFT of length N1 = 2^a over the strided index n1 (inner complex FFT N1/2 = 64..1024), then complex FFTs of length
// N2 = 2^b (64..1024) over n2 for k1 = 0..N1/2 only.  Half the intermediate of the complex four-step on the packed lane, and no
// separate split pass: two passes over global memory instead of three (R2C) or four (DCT-II).
static void add_real_fourstep_slot(ndfft_plan *p, FftConfig &c, size_t n, bool dct1);
static void add_real_fourstep(ndfft_plan *p) {
    if (p->kind != NDFFT_KIND_R2C && p->kind != NDFFT_KIND_DCT) return;
    if (p->has_cfg[CFG_MAIN]) add_real_fourstep_slot(p, p->cfg[CFG_MAIN], p->n, false);
    // DCT-I (round 5): the real-even DFT of length 2 (n - 1) -- where that is a power of two too long for one launch, the same two passes run on the even
    // extension (gathered by pass 1's load) and pass 2 stores Re X[k] / 2 for k = 0 .. n - 1 instead of the half spectrum: two passes instead of the packed
    // route's four (PRE, two four-step passes, POST)
    if (p->kind == NDFFT_KIND_DCT && p->has_cfg[CFG_DCT1] && p->n >= 2) add_real_fourstep_slot(p, p->cfg[CFG_DCT1], 2 * (p->n - 1), true);
}
// ... and for a smooth lane length that is NOT a power of two (round 6; R2C, DCT-II, DCT-I, and C2R / DCT-III where N2 allows): n = N1 N2 with N1 and N2 even (the real FFT over n1 runs
// through a complex FFT of N1 / 2; pass 2 splits the mirrored half of its outputs at N2 / 2), each factor a power of two with ahead-of-time passes or a smooth length
// whose pass can be specialised with hiprtc (jit.hip); the most square such pair.  The inverse ops keep the packed route there.
static bool add_real_fourstep_smooth(ndfft_plan *p, FftConfig &c, size_t n, bool dct1) {
    if (n % 4 || n > ((size_t)1 << 24) || !NDFFT_DEV_INT("NDFFT_RFS_SMOOTH", 1)) return false;
    { size_t m = n; for (int q : {2, 3, 5, 7, 11, 13}) while (m % q == 0) m /= q; if (m != 1) return false; }
    auto aot = [](size_t f) { return f == 64 || f == 128 || f == 256 || f == 512 || f == 1024; };
    auto ok1 = [&](size_t N1) { JitCfg t; return N1 % 2 == 0 && (aot(N1 / 2) || (N1 / 2 >= 48 && N1 / 2 <= 2048 && jit_choose_real(p->dtype, (int)(N1 / 2), t) && t.e <= (p->dtype == NDFFT_F32 ? 32 : 24))); };
    auto ok2 = [&](size_t N2) { JitCfg t; return N2 % 2 == 0 && (aot(N2) || jit_fourstep_choose(p->dtype, (int)N2, t)); };
    size_t best1 = 0, best2 = 0; double bestd = 1e30;
    for (size_t N2 = 64; N2 <= 2048 && N2 * 64 <= n; N2 += 2) {
        if (n % N2) continue;
        const size_t N1 = n / N2;
        if (N1 < 128 || N1 > 4096 || !ok2(N2) || !ok1(N1)) continue;
        // f64: pass 2 of a power-of-two N2 is the lane-fastest kernel (col_direct.h), of any other N2 the staged column kernel -- measured nddct2 85 x 196608 f64 191 us as 512 x 384
        // against 144 us for 102 x 163840 as 320 x 512: a power-of-two N2 is worth a less square split
        const double dd = std::fabs(std::log((double)N1 / (double)N2)) + ((p->dtype == NDFFT_F64 && !aot(N2)) ? 0.7 : 0.0) + (!aot(N1 / 2) ? 0.1 : 0.0);
        if (dd < bestd) { bestd = dd; best1 = N1; best2 = N2; }
    }
    if (!best1) return false;
    int e = 0; while (((size_t)1 << e) < n) ++e;
    // inverse ops (C2R, DCT-III) too where pass 1 of that direction exists for N2: the lane-fastest kernel, ahead of time for a power of two, hiprtc for a whole-round recipe
    bool inv_ok = aot(best2);
    if (!inv_ok) { JitCfg t; inv_ok = jit_fourstep_choose(p->dtype, (int)best2, t) && !t.partial && t.e * t.tpl == (int)best2; }
    c.rfs_ops = dct1 ? 16 : ((1 | 4) | (inv_ok ? (2 | 8) : 0));
    c.rfs = true; c.rfs_N1 = (int)best1; c.rfs_N2 = (int)best2;
    c.rfs_sub1 = make_plan(NDFFT_KIND_R2C, p->dtype, best1);
    c.rfs_sub2 = make_plan(NDFFT_KIND_C2C, p->dtype, best2);
    c.rfs_logB = (e + 1) / 2;
    const int64_t B = 1ll << c.rfs_logB;
    for (int64_t k = 0; k < B && k < (int64_t)n; ++k) unit(c.rfs_twlo, k, n);
    for (int64_t k = 0; k * B < (int64_t)n; ++k) unit(c.rfs_twhi, k * B, n);
    if (p->kind == NDFFT_KIND_DCT && !dct1) {
        for (int k1 = 0; k1 <= c.rfs_N1; ++k1) unit(c.rfs_c1, (unsigned long long)k1, 4ull * n);
        for (int r = 0; r < c.rfs_N2; ++r) unit(c.rfs_c2, (unsigned long long)r, 4ull * (unsigned long long)c.rfs_N2);
    }
    return true;
}
static void add_real_fourstep_slot(ndfft_plan *p, FftConfig &c, size_t n, bool dct1) {
    if (n == 0 || !c.big || c.bigblue) return;
    if (n & (n - 1)) { (void)add_real_fourstep_smooth(p, c, n, dct1); return; }
    int e = 0; while (((size_t)1 << e) < n) ++e;
    // The split and the ops that take this route, from the sweep over n = 2^16..2^21 at 2^24 points per array (profiles/r06/r06l_*, r06m_*):
    //   f64: N1 = 2^ceil(e/2), but N1 = 2048 rather than N2 = 1024 at e = 20; faster than the packed route for every op and length (1.02-1.7 x)
    //   f32: N1 = 1024 from e = 17 up (2048 at e = 21); R2C only up to e = 19 (re-swept after the f32 wave floors, r06zu_*), C2R and DCT-II up to e = 20, DCT-III always (the packed route's
    //        complex passes run at twice f64's element rate, so its extra pass costs less)
    const bool f32 = p->dtype == NDFFT_F32;
    int a = f32 ? std::min(10, (e + 1) / 2 + 1) : (e + 1) / 2;
    if (e == 20 && !f32) a = 11;
    a = std::min(11, std::max(7, a));
    if (e - a > 10) a = e - 10;
    if (e - a < 6) a = e - 6;
    if (dct1 && f32 && e >= 21) return;      // (measured: f32 n = 2^20 + 1 210 vs 206 us on the packed route -- profiles/r08/r08z_dct1_long_lanes.txt)
    c.rfs_ops = dct1 ? 16 : f32 ? ((e <= 19 ? 1 : 0) | (e <= 20 ? 2 | 4 : 0) | 8) : 15;      // bit 4: the DCT-I form (DCT1 slot only)
    if (const int v = sw().rfs_logn1; v >= 7 && v <= 11) a = v;   // NDFFT_RFS_LOGN1 (parity tests of every split)
    const int b = e - a;
    if (b < 6 || b > 10 || !fourstep_real_supported(1 << a, 1 << b)) return;
    c.rfs = true; c.rfs_N1 = 1 << a; c.rfs_N2 = 1 << b;
    c.rfs_sub1 = make_plan(NDFFT_KIND_R2C, p->dtype, (size_t)c.rfs_N1);
    c.rfs_sub2 = make_plan(NDFFT_KIND_C2C, p->dtype, (size_t)c.rfs_N2);
    c.rfs_logB = (e + 1) / 2;
    const int64_t B = 1ll << c.rfs_logB;
    for (int64_t k = 0; k < B && k < (int64_t)n; ++k) unit(c.rfs_twlo, k, n);
    for (int64_t k = 0; k * B < (int64_t)n; ++k) unit(c.rfs_twhi, k * B, n);
    if (p->kind == NDFFT_KIND_DCT && !dct1) {    // DCT-II / DCT-III post- / pre-twiddle e^{-i pi k/(2n)}, k = k1 + N1 r (see engine.h)
        for (int k1 = 0; k1 <= c.rfs_N1; ++k1) unit(c.rfs_c1, (unsigned long long)k1, 4ull * n);
        for (int r = 0; r < c.rfs_N2; ++r) unit(c.rfs_c2, (unsigned long long)r, 4ull * (unsigned long long)c.rfs_N2);
    }
}

static void add_narrow_tables(ndfft_plan *p) {
    // Bluestein lengths: the register-kernel form (blue_kernel.h) when M has an E = 8 configuration
    for (int i = 0; i < CFG_COUNT; ++i) {
        FftConfig &c = p->cfg[i];
        const bool m_pow2 = (c.M & (c.M - 1)) == 0;
        if (p->has_cfg[i] && c.blue && (m_pow2 ? pow2_real_config(c.M, c.jitcfg) : blue_plan_cfg(p->dtype, c.M, c.jitcfg))) {
            c.bluereg = true; c.twp = HostTable();
            if (m_pow2) pow2_real_build_twiddles(c.M, c.twp); else build_pass_twiddles(c.twp, c.jitcfg.radix, c.M);
            c.twp_rev = HostTable(); build_pass_twiddles(c.twp_rev, std::vector<int>(c.jitcfg.radix.rbegin(), c.jitcfg.radix.rend()), c.M);
        }
        // ... and, where F = (small cofactor) x (prime p with p - 1 smooth), Rader's convolution of length p - 1 instead (rader_kernel.h)
        //     (also beyond Bluestein's single-launch reach: lanes of up to ~9600 (f64) / ~19000 (f32) elements that otherwise take the multi-pass routes)
        if (p->has_cfg[i] && (c.blue || c.big) && c.F >= 17 && rader_choose(p->dtype, c.F, c.radercfg, p->kind == NDFFT_KIND_DCT && i == CFG_DCT1)) { c.rader = true; build_rader_tables(c); }
    }
    // specialised (hiprtc) register kernels for the real-data ops with a smooth non-power-of-two inner FFT
    if (p->kind != NDFFT_KIND_C2C)
        for (int i = 0; i < CFG_COUNT; ++i) {
            FftConfig &c = p->cfg[i];
            if (p->has_cfg[i] && !c.pow2 && !c.blue && c.F > 1 && jit_choose_real(p->dtype, c.F, c.jitcfg)) { c.jit = true; jit_build_twiddles(c.jitcfg, c.twp); }
        }
    for (int i = 0; i < CFG_COUNT; ++i)
        if (p->has_cfg[i] && !p->cfg[i].blue && !p->cfg[i].big) pow2_real_build_narrow_twiddles(p->dtype, p->cfg[i].F, p->cfg[i].twp_narrow);
}

template <typename T> static int upload(const HostTable &t, void **dptr) {
    *dptr = nullptr;
    const size_t cnt = t.re.size();
    if (!cnt) return NDFFT_OK;
    std::vector<T> h(2 * cnt);
    for (size_t i = 0; i < cnt; ++i) { h[2 * i] = (T)t.re[i]; h[2 * i + 1] = (T)t.im[i]; }
    NDFFT_HIP(hipMalloc(dptr, h.size() * sizeof(T)));
    NDFFT_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return NDFFT_OK;
}

static int upload_any(int dtype, const HostTable &t, void **dptr) {
    return dtype == NDFFT_F32 ? upload<float>(t, dptr) : upload<double>(t, dptr);
}

// device tables of `plan` on the CURRENT device (uploaded on first use, then cached)
int get_dev_tables(const ndfft_plan *cplan, const DevTables **out) {
    ndfft_plan *plan = const_cast<ndfft_plan *>(cplan);
    int dev = 0;
    NDFFT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> g(plan->mu);
    auto it = plan->dev.find(dev);
    if (it != plan->dev.end()) { *out = &it->second; return NDFFT_OK; }
    DevTables t;
    for (int i = 0; i < CFG_COUNT; ++i) {
        if (!plan->has_cfg[i]) continue;
        const FftConfig &c = plan->cfg[i];
        DevConfig &d = t.cfg[i];
        int rc;
        if ((rc = upload_any(plan->dtype, c.tw, &d.tw))) return rc;
        if ((rc = upload_any(plan->dtype, c.twM, &d.twM))) return rc;
        if ((rc = upload_any(plan->dtype, c.chirp, &d.chirp))) return rc;
        if ((rc = upload_any(plan->dtype, c.bhat, &d.bhat))) return rc;
        if ((rc = upload_any(plan->dtype, c.aux1, &d.aux1))) return rc;
        if ((rc = upload_any(plan->dtype, c.aux2, &d.aux2))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp, &d.twp))) return rc;
        if ((rc = upload_any(plan->dtype, c.twlo, &d.twlo))) return rc;
        if ((rc = upload_any(plan->dtype, c.twhi, &d.twhi))) return rc;
        if ((rc = upload_any(plan->dtype, c.cs_twlo, &d.cs_twlo))) return rc;
        if ((rc = upload_any(plan->dtype, c.cs_twhi, &d.cs_twhi))) return rc;
        if ((rc = upload_any(plan->dtype, c.rfs_twlo, &d.rfs_twlo))) return rc;
        if ((rc = upload_any(plan->dtype, c.rfs_twhi, &d.rfs_twhi))) return rc;
        if ((rc = upload_any(plan->dtype, c.rfs_c1, &d.rfs_c1))) return rc;
        if ((rc = upload_any(plan->dtype, c.rfs_c2, &d.rfs_c2))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_col, &d.twp_col))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_col_w, &d.twp_col_w))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_fs, &d.twp_fs))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_jcol, &d.twp_jcol))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_narrow, &d.twp_narrow))) return rc;
        if ((rc = upload_any(plan->dtype, c.wave_tw, &d.wave_tw))) return rc;
        for (int q = 0; q < 4; ++q) if ((rc = upload_any(plan->dtype, c.tinymat[q], &d.tinymat[q]))) return rc;
        if ((rc = upload_any(plan->dtype, c.rader_bhat, &d.rader_bhat))) return rc;
        if ((rc = upload_any(plan->dtype, c.rader_twp, &d.rader_twp))) return rc;
        if ((rc = upload_any(plan->dtype, c.rader_twp2, &d.rader_twp2))) return rc;
        if ((rc = upload_any(plan->dtype, c.twp_rev, &d.twp_rev))) return rc;
        if ((rc = upload_any(plan->dtype, c.rader_ctw, &d.rader_ctw))) return rc;
        if (!c.rader_tab.empty()) {
            NDFFT_HIP(hipMalloc(&d.rader_tab, c.rader_tab.size() * sizeof(int32_t)));
            NDFFT_HIP(hipMemcpy(d.rader_tab, c.rader_tab.data(), c.rader_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    auto ins = plan->dev.emplace(dev, t);
    *out = &ins.first->second;
    return NDFFT_OK;
}

}  // namespace ndfft

using namespace ndfft;

// ---- environment switches (switches.h): parsed once -----------------------------------------------------------------
namespace ndfft {
namespace {
std::atomic<const Switches *> g_sw{nullptr};
std::mutex g_sw_mu;
const Switches *parse_switches() {
    Switches *S = new Switches();
    auto str = [](const char *name, bool &set, std::string &v) { const char *e = getenv(name); set = e != nullptr; v = e ? e : ""; };
    auto on = [](const char *name, bool def) { const char *e = getenv(name); return e ? e[0] != '0' : def; };       // "0" closes, anything else opens
    auto one = [](const char *name) { const char *e = getenv(name); return e && e[0] == '1'; };                      // only "1" enables
    // numeric switches are clamped to [lo, hi]; text that is not a number keeps the default
    auto num = [](const char *name, long def, long lo, long hi) {
        const char *e = getenv(name);
        if (!e) return def;
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end == e) return def;
        return std::min(hi, std::max(lo, v));
    };
    if (const char *e = getenv("NDFFT_JIT")) S->jit = e[0] == '0' ? 0 : !strcmp(e, "cached") ? 2 : 1;
    str("NDFFT_JIT_CACHE", S->jit_cache_set, S->jit_cache);
    str("NDFFT_JIT_PREBUILT", S->jit_prebuilt_set, S->jit_prebuilt);
    S->jit_verbose = getenv("NDFFT_JIT_VERBOSE") != nullptr;
    if (const char *e = getenv("NDFFT_JIT_DUMP_SRC")) S->jit_dump_src = e;
    if (const char *e = getenv("XDG_CACHE_HOME")) S->xdg_cache_home = e;
    if (const char *e = getenv("HOME")) S->home = e;
    S->wave = on("NDFFT_WAVE", true); S->tiny = on("NDFFT_TINY", true); S->plain = on("NDFFT_PLAIN", true); S->blue = on("NDFFT_BLUE", true);
    S->rader = on("NDFFT_RADER", true); S->colsplit = on("NDFFT_COLSPLIT", true); S->fourstep2 = on("NDFFT_FOURSTEP2", true);
    S->real_fourstep = (int)num("NDFFT_REAL_FOURSTEP", 1, 0, 2);
    if (const char *e = getenv("NDFFT_FS_DIRECT")) S->fs_direct = e[0] == '1' ? 1 : 0;
    S->narrow_dct = one("NDFFT_NARROW_DCT");
    S->rfs_c2r_tile = on("NDFFT_RFS_C2R_TILE", true);
    S->rfs_logn1 = (int)num("NDFFT_RFS_LOGN1", 0, 0, 11);
    S->cs_chunk_mb = (int)num("NDFFT_CS_CHUNK_MB", 144, 0, 1 << 20);
    S->stream_loads = (int)num("NDFFT_STREAM_LOADS", -1, -1, 1);
    if (const char *e = getenv("NDFFT_HOST_PIPE")) S->host_pipe = e[0] == '1' ? 1 : e[0] == '0' ? 0 : -1;
    S->host_reg_cache_mb = num("NDFFT_HOST_REG_CACHE_MB", 0, 0, 1L << 22);
    S->copy_threads = (int)num("NDFFT_COPY_THREADS", 0, 0, 64);
    S->shard_chunk_kb = num("NDFFT_SHARD_CHUNK_KB", 0, 1, 1L << 24);
    S->shard_force_remote = one("NDFFT_SHARD_FORCE_REMOTE");
    return S;
}
}  // namespace
const Switches &sw() {
    const Switches *p = g_sw.load(std::memory_order_acquire);
    if (__builtin_expect(p != nullptr, 1)) return *p;
    std::lock_guard<std::mutex> g(g_sw_mu);
    p = g_sw.load(std::memory_order_relaxed);
    if (!p) { p = parse_switches(); g_sw.store(p, std::memory_order_release); }
    return *p;
}
void reload_switches() {
    std::lock_guard<std::mutex> g(g_sw_mu);
    g_sw.store(parse_switches(), std::memory_order_release);     // (the previous struct is leaked on purpose: another thread may still hold a reference)
}
}  // namespace ndfft

extern "C" {

int ndfft_abi_version(void) { return NDFFT_ABI_VERSION; }
int ndfft_abi_minor(void) { return NDFFT_ABI_MINOR; }
int ndfft_reload_switches(void) { ndfft::reload_switches(); return NDFFT_OK; }
int ndfft_documented_switches(char *buf, size_t cap) {
    static const char *const names[] = {NDFFT_DOCUMENTED_SWITCHES};
    std::string s;
    for (const char *n : names) { s += n; s += '\n'; }
    if (buf && cap) { const size_t k = std::min(cap - 1, s.size()); memcpy(buf, s.data(), k); buf[k] = 0; }
    return (int)s.size();
}
const char *ndfft_last_error(void) { return last_err().c_str(); }
const char *ndfft_last_path(void) { return last_path(); }

int ndfft_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ndfft_set_device(int device) {
    clear_err();
    NDFFT_HIP(hipSetDevice(device));
    return NDFFT_OK;
}

int ndfft_plan_create(int kind, int dtype, size_t n, ndfft_plan **out_plan) {
    clear_err();
    if (!out_plan) return fail(NDFFT_ERR_INVALID_ARG, "out_plan is null");
    *out_plan = nullptr;
    if (kind < NDFFT_KIND_C2C || kind > NDFFT_KIND_DCT) return fail(NDFFT_ERR_INVALID_ARG, "bad kind");
    if (dtype != NDFFT_F32 && dtype != NDFFT_F64) return fail(NDFFT_ERR_INVALID_ARG, "bad dtype (T must be f32 or f64)");
    if (n > (size_t)(1 << 24)) return fail(NDFFT_ERR_UNSUPPORTED, "n > 2^24 is not supported yet");
    if (ndfft_device_count() <= 0)
        return fail(NDFFT_ERR_NO_DEVICE, "no HIP device visible: libndfft_mi355x has no CPU fallback");
    ndfft_plan *p = make_plan(kind, dtype, n);
    const DevTables *t;
    int rc = get_dev_tables(p, &t);   // eager upload to the current device
    if (rc) { delete p; return rc; }
    *out_plan = p;
    return NDFFT_OK;
}

int ndfft_plan_retain(ndfft_plan *plan) {
    if (!plan) return fail(NDFFT_ERR_INVALID_ARG, "plan is null");
    std::lock_guard<std::mutex> g(plan->mu);
    ++plan->refcount;
    return NDFFT_OK;
}

int ndfft_plan_destroy(ndfft_plan *plan) {
    if (!plan) return NDFFT_OK;
    {
        std::lock_guard<std::mutex> g(plan->mu);
        if (--plan->refcount > 0) return NDFFT_OK;
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : plan->dev) {
        (void)hipSetDevice(kv.first);
        for (int i = 0; i < CFG_COUNT; ++i) {
            DevConfig &d = kv.second.cfg[i];
            void *ptrs[] = {d.tw, d.twM, d.chirp, d.bhat, d.aux1, d.aux2, d.twp, d.twlo, d.twhi, d.twp_col, d.twp_col_w, d.twp_fs, d.twp_jcol, d.twp_narrow, d.cs_twlo, d.cs_twhi, d.rfs_twlo, d.rfs_twhi, d.rfs_c1, d.rfs_c2, d.wave_tw, d.tinymat[0], d.tinymat[1], d.tinymat[2], d.tinymat[3], d.rader_bhat, d.rader_twp, d.rader_twp2, d.rader_tab, d.twp_rev, d.rader_ctw};
            for (void *q : ptrs) if (q) (void)hipFree(q);
        }
    }
    (void)hipSetDevice(cur);
    for (int i = 0; i < CFG_COUNT; ++i) { ndfft_plan_destroy(plan->cfg[i].sub1); ndfft_plan_destroy(plan->cfg[i].sub2);
                                          ndfft_plan_destroy(plan->cfg[i].cs_sub1); ndfft_plan_destroy(plan->cfg[i].cs_sub2);
                                          ndfft_plan_destroy(plan->cfg[i].rfs_sub1); ndfft_plan_destroy(plan->cfg[i].rfs_sub2); }
    delete plan;
    return NDFFT_OK;
}

int ndfft_explain_plan(int kind, int dtype, size_t n, char *buf, size_t buflen) {
    clear_err();
    if (kind < NDFFT_KIND_C2C || kind > NDFFT_KIND_DCT) return -fail(NDFFT_ERR_INVALID_ARG, "bad kind");
    if (dtype != NDFFT_F32 && dtype != NDFFT_F64) return -fail(NDFFT_ERR_INVALID_ARG, "bad dtype");
    if (n > (size_t)(1 << 24)) return -fail(NDFFT_ERR_UNSUPPORTED, "n > 2^24 is not supported yet");
    ndfft_plan *p = make_plan(kind, dtype, n);      // host tables only: nothing is uploaded, no device is touched
    std::string out;
    auto radix = [](const std::vector<int> &r) { std::string t; for (size_t i = 0; i < r.size(); ++i) t += (i ? "." : "") + std::to_string(r[i]); return t.empty() ? std::string("-") : t; };
    static const char *slot_name[CFG_COUNT] = {"MAIN", "DCT1", "DCT4"};
    for (int i = 0; i < CFG_COUNT; ++i) {
        if (!p->has_cfg[i]) continue;
        const FftConfig &c = p->cfg[i];
        std::string l = std::string("slot=") + slot_name[i] + " F=" + std::to_string(c.F);
        if (c.unsupported) l += " route=unsupported";
        else if (c.rader) l += " route=rader p=" + std::to_string(c.radercfg.p) + " mc=" + std::to_string(c.radercfg.mc1) + "x" + std::to_string(c.radercfg.mc2) + " M=" + std::to_string(c.radercfg.fft.n) +
                               " tpl=" + std::to_string(c.radercfg.fft.tpl) + " e=" + std::to_string(c.radercfg.fft.e) + " radix=" + radix(c.radercfg.fft.radix) + " lanes=" + std::to_string(c.radercfg.fft.lpb) + (c.radercfg.sym ? " sym_rows=" + std::to_string(c.radercfg.rows()) + (c.radercfg.half() ? " half_conv=1" : "") : std::string());
        else if (c.pow2) l += " route=pow2";
        else if (c.jit) {
            l += " route=jit tpl=" + std::to_string(c.jitcfg.tpl) + " e=" + std::to_string(c.jitcfg.e) + " radix=" + radix(c.jitcfg.radix) + " lanes=" + std::to_string(c.jitcfg.row_lpb);
            if (i == CFG_MAIN && p->kind == NDFFT_KIND_C2C) { JitCfg cv = c.jitcfg; if (jit_c2c_row_vec(p->dtype, cv)) l += " rowvec_tpl=" + std::to_string(cv.tpl) + " rowvec_e=" + std::to_string(cv.e); }   // 16-byte-aligned f32 rows
            if (c.jit_col_alt) l += " col_tpl=" + std::to_string(c.jitcfg_col.tpl) + " col_e=" + std::to_string(c.jitcfg_col.e) + " col_radix=" + radix(c.jitcfg_col.radix);
            if (c.fs_jit) l += " fs_tpl=" + std::to_string(c.fs_jitcfg.tpl) + " fs_e=" + std::to_string(c.fs_jitcfg.e) + " fs_radix=" + radix(c.fs_jitcfg.radix);   // as a four-step factor (jit.hip: jit_fourstep_choose)
        }
        else if (c.big && !c.bigblue) l += " route=four_step F1=" + std::to_string(c.F1) + " F2=" + std::to_string(c.F2) +
                                           ((c.sub1 && c.sub1->cfg[CFG_MAIN].fs_jit) || (c.sub2 && c.sub2->cfg[CFG_MAIN].fs_jit) ? std::string(" jit_passes=") + (c.sub1->cfg[CFG_MAIN].fs_jit ? "1" : "") + (c.sub2->cfg[CFG_MAIN].fs_jit ? "2" : "") : std::string()) +
                                           (c.rfs ? " real_four_step=" + std::to_string(c.rfs_N1) + "x" + std::to_string(c.rfs_N2) + " ops=" + std::to_string(c.rfs_ops) : std::string());
        else if (c.F <= 1) l += " route=trivial";
        else if (!c.blue && !c.big) l += " route=lds radix=" + radix(c.radix);
        if (c.blue || c.bigblue) {
            l += std::string(c.rader ? " fallback=" : " route=") + (c.bigblue && !c.bluereg ? "blue_global" : c.bluereg ? "blue_reg" : "blue_lds") + " blueM=" + std::to_string(c.M);
            if (c.bluereg) l += " blue_tpl=" + std::to_string(c.jitcfg.tpl) + " blue_e=" + std::to_string(c.jitcfg.e) + " blue_radix=" + radix(c.jitcfg.radix);
        }
        out += l + "\n";
    }
    ndfft_plan_destroy(p);
    if (buf && buflen) { const size_t k = std::min(out.size(), buflen - 1); memcpy(buf, out.data(), k); buf[k] = '\0'; }
    return (int)out.size();
}

size_t ndfft_plan_n(const ndfft_plan *plan) { return plan ? plan->n : 0; }
int ndfft_plan_kind(const ndfft_plan *plan) { return plan ? plan->kind : -1; }
int ndfft_plan_dtype(const ndfft_plan *plan) { return plan ? plan->dtype : -1; }

size_t ndfft_plan_lane_len_in(const ndfft_plan *plan, int op) {
    if (!plan) return 0;
    return op == NDFFT_OP_C2R ? plan->n / 2 + 1 : plan->n;
}
size_t ndfft_plan_lane_len_out(const ndfft_plan *plan, int op) {
    if (!plan) return 0;
    return op == NDFFT_OP_R2C ? plan->n / 2 + 1 : plan->n;
}

int ndfft_dev_alloc(void **d_ptr, size_t bytes) {
    clear_err();
    if (!d_ptr) return fail(NDFFT_ERR_INVALID_ARG, "d_ptr is null");
    NDFFT_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return NDFFT_OK;
}
int ndfft_dev_free(void *d_ptr) {
    clear_err();
    if (d_ptr) NDFFT_HIP(hipFree(d_ptr));
    return NDFFT_OK;
}
int ndfft_dev_upload(void *d_dst, const void *h_src, size_t bytes) {
    clear_err();
    if (bytes) NDFFT_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return NDFFT_OK;
}
int ndfft_dev_download(void *h_dst, const void *d_src, size_t bytes) {
    clear_err();
    if (bytes) NDFFT_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return NDFFT_OK;
}
int ndfft_dev_sync(void *stream) {
    clear_err();
    NDFFT_HIP(hipStreamSynchronize((hipStream_t)stream));
    return NDFFT_OK;
}

}  // extern "C"
