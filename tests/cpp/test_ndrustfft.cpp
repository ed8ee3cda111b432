// tests/cpp/test_ndrustfft.cpp -- the reference's unit tests (src/lib.rs:903-1406) and examples
// (examples/fft2.rs, rfft2.rs, fft_norm.rs) re-stated against the C++ mirror include/ndrustfft.hpp,
// i.e. through the C ABI onto the MI355X.  Same fixtures, same tolerances (1e-3 / 1e-4).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

#include "ndrustfft.hpp"
#include "reference_vectors.h"

using namespace ndrustfft;
using C = Complex<double>;

static int g_fail = 0;
#define EXPECT(cond) do { if (!(cond)) { printf("    EXPECT failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); ++g_fail; } } while (0)

static void approx_eq(const std::vector<double> &r, const std::vector<double> &e, double dif = 1e-3) {   // lib.rs:852-864
    EXPECT(r.size() == e.size());
    for (size_t i = 0; i < r.size() && i < e.size(); ++i)
        if (std::fabs(r[i] - e[i]) > dif) { printf("    Large difference of values, got %g expected %g.\n", r[i], e[i]); ++g_fail; return; }
}
static void approx_eq_complex(const std::vector<C> &r, const std::vector<double> &ere, const std::vector<double> &eim, double dif = 1e-3) {
    EXPECT(r.size() == ere.size());
    for (size_t i = 0; i < r.size() && i < ere.size(); ++i)
        if (std::fabs(r[i].real() - ere[i]) > dif || std::fabs(r[i].imag() - eim[i]) > dif) {
            printf("    Large difference of values, got %g%+gi expected %g%+gi.\n", r[i].real(), r[i].imag(), ere[i], eim[i]); ++g_fail; return;
        }
}
static std::vector<C> cplx_of(const std::vector<double> &m) { std::vector<C> v; for (double x : m) v.emplace_back(x, x); return v; }
static std::vector<double> re_of(const std::vector<C> &v) { std::vector<double> r; for (auto &c : v) r.push_back(c.real()); return r; }
static std::vector<double> im_of(const std::vector<C> &v) { std::vector<double> r; for (auto &c : v) r.push_back(c.imag()); return r; }

static Array<double> test_matrix() { return Array<double>::from({6, 6}, refvec::test_matrix); }                    // lib.rs:880-889
static Array<C> test_matrix_complex() { return Array<C>::from({6, 6}, cplx_of(refvec::test_matrix)); }             // lib.rs:891-893
static Array<C> test_matrix_complex_f() { return Array<C>::from({6, 6}, cplx_of(refvec::test_matrix), true); }     // lib.rs:895-901

template <bool PAR> static void test_fft() {                                                                       // lib.rs:903-994
    auto v = test_matrix_complex(); auto v_copy = v;
    auto vhat = Array<C>::zeros({6, 6});
    FftHandler<double> handler(6);
    if (PAR) { ndfft_par(v, vhat, handler, 1); ndifft_par(vhat, v, handler, 1); }
    else { ndfft(v, vhat, handler, 1); ndifft(vhat, v, handler, 1); }
    approx_eq_complex(vhat.to_logical(), refvec::fft_axis1_re, refvec::fft_axis1_im);
    approx_eq_complex(v.to_logical(), re_of(v_copy.to_logical()), im_of(v_copy.to_logical()));
}
static void test_fft_f_layout() {                                                                                   // lib.rs:996-1040
    auto v = test_matrix_complex_f(); auto v_copy = v;
    auto vhat = Array<C>::zeros({6, 6});
    FftHandler<double> handler(6);
    ndfft(v, vhat, handler, 1); ndifft(vhat, v, handler, 1);
    approx_eq_complex(vhat.to_logical(), refvec::fft_axis1_re, refvec::fft_axis1_im);
    approx_eq_complex(v.to_logical(), re_of(v_copy.to_logical()), im_of(v_copy.to_logical()));
}
template <bool PAR> static void test_fft_r2c() {                                                                    // lib.rs:1042-1133
    auto v = test_matrix(); auto v_copy = v;
    auto vhat = Array<C>::zeros({6, 6 / 2 + 1});
    R2cFftHandler<double> handler(6);
    if (PAR) { ndfft_r2c_par(v, vhat, handler, 1); ndifft_r2c_par(vhat, v, handler, 1); }
    else { ndfft_r2c(v, vhat, handler, 1); ndifft_r2c(vhat, v, handler, 1); }
    approx_eq_complex(vhat.to_logical(), refvec::rfft_axis1_re, refvec::rfft_axis1_im);
    approx_eq(v.to_logical(), v_copy.to_logical());
}
static void test_ifft_c2r_first_last_element() {                                                                    // lib.rs:1135-1167
    const int n = 6;
    auto v = Array<double>::zeros({n}); auto vhat = Array<C>::zeros({n / 2 + 1});
    R2cFftHandler<double> rfft_handler(n);
    vhat(0) = C(1., 100.);
    ndifft_r2c(vhat, v, rfft_handler, 0);
    approx_eq(v.to_logical(), refvec::c2r_first_out);
    for (int i = 0; i < n / 2 + 1; ++i) vhat(i) = C(0., 0.);
    vhat(3) = C(1., 100.);
    ndifft_r2c(vhat, v, rfft_handler, 0);
    approx_eq(v.to_logical(), refvec::c2r_last_out);
}
template <bool PAR> static void test_fft_r2c_odd() {                                                                // lib.rs:1169-1202
    auto v = Array<double>::from({3, 3}, refvec::m3x3); auto v_copy = v;
    auto vhat = Array<C>::zeros({3, 3 / 2 + 1});
    R2cFftHandler<double> handler(3);
    if (PAR) { ndfft_r2c_par(v, vhat, handler, 1); ndifft_r2c_par(vhat, v, handler, 1); }
    else { ndfft_r2c(v, vhat, handler, 1); ndifft_r2c(vhat, v, handler, 1); }
    approx_eq(v.to_logical(), v_copy.to_logical());
}
template <int K, bool PAR> static void test_dct() {                                                                 // lib.rs:1204-1406
    auto v = test_matrix(); auto vhat = Array<double>::zeros({6, 6});
    DctHandler<double> handler(6);
    if (K == 1) { if (PAR) nddct1_par(v, vhat, handler, 1); else nddct1(v, vhat, handler, 1); }
    if (K == 2) { if (PAR) nddct2_par(v, vhat, handler, 1); else nddct2(v, vhat, handler, 1); }
    if (K == 3) { if (PAR) nddct3_par(v, vhat, handler, 1); else nddct3(v, vhat, handler, 1); }
    if (K == 4) { if (PAR) nddct4_par(v, vhat, handler, 1); else nddct4(v, vhat, handler, 1); }
    const std::vector<double> *sol[] = {&refvec::dct1_axis1, &refvec::dct2_axis1, &refvec::dct3_axis1, &refvec::dct4_axis1};
    approx_eq(vhat.to_logical(), *sol[K - 1]);
}

static void example_fft2() {                                                                                        // examples/fft2.rs
    auto v = Array<C>::from({3, 3}, cplx_of(refvec::m3x3));
    auto vhat = Array<C>::zeros({3, 3});
    FftHandler<double> handler_ax0(3), handler_ax1(3);
    { auto work = Array<C>::zeros({3, 3}); ndfft(v, work, handler_ax1, 1); ndfft(work, vhat, handler_ax0, 0); }
    approx_eq_complex(vhat.to_logical(), refvec::example_fft2_re, refvec::example_fft2_im, 1e-4);
    auto v_new = Array<C>::zeros({3, 3});
    { auto work = Array<C>::zeros({3, 3}); ndifft(vhat, work, handler_ax0, 0); ndifft(work, v_new, handler_ax1, 1); }
    approx_eq_complex(v_new.to_logical(), refvec::m3x3, refvec::m3x3, 1e-4);
}
static void example_rfft2() {                                                                                       // examples/rfft2.rs
    auto v = Array<double>::from({3, 3}, refvec::m3x3);
    auto vhat = Array<C>::zeros({3, 2});
    FftHandler<double> handler_ax0(3); R2cFftHandler<double> handler_ax1(3);
    { auto work = Array<C>::zeros({3, 2}); ndfft_r2c(v, work, handler_ax1, 1); ndfft(work, vhat, handler_ax0, 0); }
    approx_eq_complex(vhat.to_logical(), refvec::example_rfft2_re, refvec::example_rfft2_im, 1e-4);
    auto v_new = Array<double>::zeros({3, 3});
    { auto work = Array<C>::zeros({3, 2}); ndifft(vhat, work, handler_ax0, 0); ndifft_r2c(work, v_new, handler_ax1, 1); }
    approx_eq(v_new.to_logical(), refvec::m3x3, 1e-4);
}
static void my_norm(C *data, std::size_t len) { const double n = 2. / (double)len; for (std::size_t i = 0; i < len; ++i) data[i] *= n; }
static void example_fft_norm() {                                                                                    // examples/fft_norm.rs
    auto v = Array<C>::from({3}, cplx_of({1., 2., 3.}));
    auto vhat = Array<C>::zeros({3}); auto v2 = Array<C>::zeros({3});
    auto handler = FftHandler<double>(3).normalization(Normalization<C>::dflt());
    ndfft(v, vhat, handler, 0); ndifft(vhat, v2, handler, 0);
    approx_eq_complex(v2.to_logical(), {1., 2., 3.}, {1., 2., 3.}, 1e-12);
    handler = FftHandler<double>(3).normalization(Normalization<C>::none());
    ndfft(v, vhat, handler, 0); ndifft(vhat, v2, handler, 0);
    approx_eq_complex(v2.to_logical(), {3., 6., 9.}, {3., 6., 9.}, 1e-12);
    handler = FftHandler<double>(3).normalization(Normalization<C>::custom(my_norm));
    ndfft(v, vhat, handler, 0); ndifft(vhat, v2, handler, 0);
    approx_eq_complex(v2.to_logical(), {2., 4., 6.}, {2., 4., 6.}, 1e-12);
}
static void readme_r2c_6x4() {                                                                                      // lib.rs:38-50 (BASELINE configs[0])
    std::vector<double> d(24); for (int i = 0; i < 24; ++i) d[i] = i;
    auto data = Array<double>::from({6, 4}, d); auto vhat = Array<C>::zeros({6 / 2 + 1, 4});
    R2cFftHandler<double> handler(6);
    ndfft_r2c(data, vhat, handler, 0);
    approx_eq_complex(vhat.to_logical(), refvec::readme_r2c_6x4_re, refvec::readme_r2c_6x4_im, 1e-12);
}
static void device_resident_fft2() {   // examples/fft2.rs with the work array kept in HBM
    auto v = Array<C>::from({3, 3}, cplx_of(refvec::m3x3));
    FftHandler<double> handler_ax0(3), handler_ax1(3);
    auto dv = DeviceArray<C>::from_host(v);
    DeviceArray<C> work({3, 3}), dvhat({3, 3}), back({3, 3});
    ndfft(dv, work, handler_ax1, 1);
    ndfft(work, dvhat, handler_ax0, 0);
    approx_eq_complex(dvhat.to_host().to_logical(), refvec::example_fft2_re, refvec::example_fft2_im, 1e-4);
    ndifft(dvhat, work, handler_ax0, 0);
    ndifft(work, back, handler_ax1, 1);
    approx_eq_complex(back.to_host().to_logical(), refvec::m3x3, refvec::m3x3, 1e-4);
    // a larger 2-D real transform: rfft2 round trip on 64 x 96
    const int nx = 64, ny = 96;
    std::vector<double> d(nx * ny); for (int i = 0; i < nx * ny; ++i) d[i] = std::sin(0.37 * i) + 0.01 * (i % 17);
    auto x = Array<double>::from({nx, ny}, d);
    auto dx = DeviceArray<double>::from_host(x);
    DeviceArray<C> w1({nx, ny / 2 + 1}), w2({nx, ny / 2 + 1}), w3({nx, ny / 2 + 1});
    DeviceArray<double> dy({nx, ny});
    R2cFftHandler<double> hr(ny); FftHandler<double> hc(nx);
    ndfft_r2c(dx, w1, hr, 1); ndfft(w1, w2, hc, 0); ndifft(w2, w3, hc, 0); ndifft_r2c(w3, dy, hr, 1);
    approx_eq(dy.to_host().to_logical(), d, 1e-10);
}
static void my_norm_real(double *data, std::size_t len) { const double n = 3. / (double)len; for (std::size_t i = 0; i < len; ++i) data[i] *= n; }
static void device_resident_custom_norm() {   // examples/fft_norm.rs on DeviceArrays: Normalization::Custom runs on the host at the reference's points
    auto v = Array<C>::from({3}, cplx_of({1., 2., 3.}));
    auto dv = DeviceArray<C>::from_host(v);
    DeviceArray<C> dvhat({3}), dv2({3});
    auto handler = FftHandler<double>(3).normalization(Normalization<C>::custom(my_norm));
    ndfft(dv, dvhat, handler, 0); ndifft(dvhat, dv2, handler, 0);                       // AFTER, on the output lanes (lib.rs:326-330)
    approx_eq_complex(dv2.to_host().to_logical(), {2., 4., 6.}, {2., 4., 6.}, 1e-12);
    // BEFORE, on the input lanes of a DCT (lib.rs:692-696): same result as the host function, on a 2-D array along both axes
    std::vector<double> d(5 * 8); for (int i = 0; i < 40; ++i) d[i] = std::cos(0.3 * i) + 0.1 * i;
    auto x = Array<double>::from({5, 8}, d);
    for (std::size_t axis = 0; axis < 2; ++axis) {
        auto hd = DctHandler<double>(axis == 0 ? 5 : 8).normalization(Normalization<double>::custom(my_norm_real));
        auto yh = Array<double>::zeros({5, 8}); nddct2(x, yh, hd, axis);
        auto dx = DeviceArray<double>::from_host(x); DeviceArray<double> dy({5, 8});
        nddct2(dx, dy, hd, axis);
        approx_eq(dy.to_host().to_logical(), yh.to_logical(), 1e-12);
    }
}
static void par_over_devices() {   // `_par` = create_transform_par! (lib.rs:169-238): lanes handed to the workers -- here GPUs
    const int ndev = ndfft_device_count();
    std::vector<int> ids;
    if (ndev > 1) for (int d = 0; d < ndev; ++d) ids.push_back(d); else ids = {0, 0, 0};   // one GPU: three blocks on it
    set_par_devices(ids);
    const int rows = 37, n = 96;
    std::vector<C> in; for (int i = 0; i < rows * n; ++i) in.emplace_back(std::sin(0.11 * i), std::cos(0.07 * i));
    auto x = Array<C>::from({rows, n}, in); auto y1 = Array<C>::zeros({rows, n}); auto y2 = Array<C>::zeros({rows, n});
    FftHandler<double> h(n);
    ndfft(x, y1, h, 1);
    ndfft_par(x, y2, h, 1);
    EXPECT(std::string(ndfft_last_path()).rfind("sharded:", 0) == 0);
    auto a = y1.to_logical(), b = y2.to_logical();
    for (size_t i = 0; i < a.size(); ++i) EXPECT(a[i] == b[i]);
    // device-resident array: scatter / transform / gather without the host
    auto dx = DeviceArray<C>::from_host(x); DeviceArray<C> dy({rows, n});
    ndfft_par(dx, dy, h, 1);
    auto c = dy.to_host().to_logical();
    for (size_t i = 0; i < a.size(); ++i) EXPECT(a[i] == c[i]);
    // the reference's panic text comes out before any device starts
    auto bad = Array<C>::zeros({rows, n + 1});
    try { ndfft_par(bad, bad, h, 1); EXPECT(!"no panic"); }
    catch (const Panic &p) { EXPECT(std::string(p.what()) == "Size mismatch in fft, got 97 expected 96"); }
    set_par_devices({});
}
static void panics() {                                                                                              // lib.rs:340-347, 116, 120-121
    auto x = Array<C>::zeros({3, 5}); auto y = Array<C>::zeros({3, 5});
    try { ndfft(x, y, FftHandler<double>(6), 1); EXPECT(!"no panic"); }
    catch (const Panic &p) { EXPECT(std::string(p.what()) == "Size mismatch in fft, got 5 expected 6"); }
    try { ndfft(x, y, FftHandler<double>(5), 2); EXPECT(!"no panic"); }
    catch (const Panic &p) { EXPECT(p.status == NDFFT_ERR_AXIS); }
    auto y4 = Array<C>::zeros({4, 5});
    try { ndfft(x, y4, FftHandler<double>(5), 1); EXPECT(!"no panic"); }
    catch (const Panic &p) { EXPECT(p.status == NDFFT_ERR_SHAPE_MISMATCH); }
    auto xr = Array<double>::zeros({3, 5}); auto yr = Array<double>::zeros({3, 5});
    try { nddct1(xr, yr, DctHandler<double>(4), 1); EXPECT(!"no panic"); }
    catch (const Panic &p) { EXPECT(std::string(p.what()) == "Size mismatch in dct, got 5 expected 4"); }
}
static void f32_and_clone() {
    std::vector<Complex<float>> in; for (int i = 0; i < 8; ++i) in.emplace_back((float)i, (float)-i);
    auto x = Array<Complex<float>>::from({8}, in); auto y = Array<Complex<float>>::zeros({8}); auto z = Array<Complex<float>>::zeros({8});
    FftHandler<float> h(8); FftHandler<float> h2 = h;   // Clone
    ndfft(x, y, h, 0); ndifft(y, z, h2, 0);
    auto zl = z.to_logical();
    for (int i = 0; i < 8; ++i) EXPECT(std::abs(zl[i] - in[i]) < 1e-5f);
    EXPECT(std::abs(y.to_logical()[0] - Complex<float>(28.f, -28.f)) < 1e-4f);
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "--expect-no-device")) {
        try { FftHandler<double> h(8); printf("unexpected: a plan was created\n"); return 1; }
        catch (const Error &e) { printf("no device: %s\n", e.what()); return e.status == NDFFT_ERR_NO_DEVICE ? 0 : 1; }
    }
    struct T { const char *name; std::function<void()> fn; };
    const T tests[] = {
        {"test_fft", test_fft<false>}, {"test_fft_par", test_fft<true>}, {"test_fft_f_layout", test_fft_f_layout},
        {"test_fft_r2c", test_fft_r2c<false>}, {"test_fft_r2c_par", test_fft_r2c<true>},
        {"test_ifft_c2r_first_last_element", test_ifft_c2r_first_last_element},
        {"test_fft_r2c_odd", test_fft_r2c_odd<false>}, {"test_fft_r2c_odd_par", test_fft_r2c_odd<true>},
        {"test_dct1", test_dct<1, false>}, {"test_dct1_par", test_dct<1, true>}, {"test_dct2", test_dct<2, false>},
        {"test_dct2_par", test_dct<2, true>}, {"test_dct3", test_dct<3, false>}, {"test_dct3_par", test_dct<3, true>},
        {"test_dct4", test_dct<4, false>}, {"test_dct4_par", test_dct<4, true>},
        {"example_fft2", example_fft2}, {"example_rfft2", example_rfft2}, {"example_fft_norm", example_fft_norm},
        {"readme_r2c_6x4", readme_r2c_6x4}, {"panics", panics}, {"f32_and_clone", f32_and_clone},
        {"device_resident_fft2", device_resident_fft2}, {"device_resident_custom_norm", device_resident_custom_norm},
        {"par_over_devices", par_over_devices},
    };
    int bad = 0;
    for (const T &t : tests) {
        const int before = g_fail;
        try { t.fn(); } catch (const std::exception &e) { printf("    exception: %s\n", e.what()); ++g_fail; }
        printf("test %s ... %s\n", t.name, g_fail == before ? "ok" : "FAILED");
        bad += g_fail != before;
    }
    printf("test result: %s. %d passed; %d failed\n", bad ? "FAILED" : "ok", (int)(sizeof tests / sizeof tests[0]) - bad, bad);
    return bad ? 1 : 0;
}
