/*
 * ndfft_oracle.h -- CPU ORACLE for the ndrustfft axis-transform hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  The product path
 * (ndrustfft_amd/, include/ndfft_mi355x.h) never links, imports or calls it.
 *
 * What it restates (all citations are into /root/reference):
 *   - the lane iterator  create_transform!      src/lib.rs:100-167
 *                        create_transform_par!  src/lib.rs:169-238
 *   - FftHandler     (fft_lane / ifft_lane / norm_default / assert_size)   src/lib.rs:269-348
 *   - R2cFftHandler  (fft_r2c_lane / ifft_r2c_lane / norm_default)         src/lib.rs:451-541
 *   - DctHandler     (dct1..4_lane / norm_default / assert_size)           src/lib.rs:640-751
 *   - Normalization  (None / Default / Custom(fn))                         src/lib.rs:89-98
 *
 * The 1-D butterfly arithmetic of the reference lives in three third-party crates that are
 * NOT vendored under /root/reference (pinned by Cargo.lock): rustfft 6.1.0 (Cargo.lock:462),
 * realfft 3.2.0 (Cargo.lock:414), rustdct 0.7.0 (Cargo.lock:453).  Their *published
 * definitions* are restated here (unnormalised forward DFT with e^{-2 pi i jk/n}; realfft's
 * n -> n/2+1 R2C and its C2R inverse; rustdct's "half scale" DCT-I..IV sums) on top of a plain
 * mixed-radix Stockham FFT with Bluestein for large prime factors, computed in the lane's own
 * precision (f32 or f64) exactly as the reference does.
 *
 * PINNING.  The oracle is pinned against every golden vector the reference's own tests and
 * examples hold for this path (src/lib.rs:903-1406, examples/fft2.rs:30-46,
 * examples/rfft2.rs:36-40, examples/fft_norm.rs:20-32) -- those are numpy/scipy values
 * rounded to 3-5 decimals -- and, at 1e-12, against numpy/scipy-generated vectors
 * (tests/golden/make_golden.py) and the long-double O(n^2) definitions below.
 * Parity against rustfft's *bits* at 1e-10 is UNPINNED: no Rust toolchain exists in the build
 * container, so the reference itself cannot be run ("Oracle").
 */
#ifndef NDFFT_ORACLE_H
#define NDFFT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_F32 = 0, ORC_F64 = 1 };
enum { ORC_NORM_NONE = 0, ORC_NORM_DEFAULT = 1, ORC_NORM_CUSTOM = 2 };
enum { ORC_HANDLER_FFT = 0, ORC_HANDLER_R2C = 1, ORC_HANDLER_DCT = 2 };

/* The eight public transforms (src/lib.rs:350-421, 543-611, 753-844). */
enum {
    ORC_NDFFT = 0,      /* ndfft      : Complex<T> -> Complex<T>, FftHandler,    fft_lane      */
    ORC_NDIFFT = 1,     /* ndifft     : Complex<T> -> Complex<T>, FftHandler,    ifft_lane     */
    ORC_NDFFT_R2C = 2,  /* ndfft_r2c  : T          -> Complex<T>, R2cFftHandler, fft_r2c_lane  */
    ORC_NDIFFT_R2C = 3, /* ndifft_r2c : Complex<T> -> T,          R2cFftHandler, ifft_r2c_lane */
    ORC_NDDCT1 = 4,
    ORC_NDDCT2 = 5,
    ORC_NDDCT3 = 6,
    ORC_NDDCT4 = 7
};

enum {
    ORC_OK = 0,
    ORC_PANIC_SIZE = 1,  /* "Size mismatch in fft|dct, got {} expected {}"  (lib.rs:340,533,743) */
    ORC_PANIC_AXIS = 2,  /* output.shape()[axis] index panic              (lib.rs:116)        */
    ORC_PANIC_ZIP = 3,   /* ndarray Zip dimension mismatch                (lib.rs:120-121)    */
    ORC_BAD_ARG = 4
};

typedef struct orc_handler orc_handler;

/* Custom(fn(&mut [T])): `data` points at len elements of the handler's element type
 * (Complex<T> for the two FFT handlers, T for DctHandler) -- src/lib.rs:97. */
typedef void (*orc_custom_norm_fn)(void *data, size_t len);

/* FftHandler::new / R2cFftHandler::new / DctHandler::new (lib.rs:294, 477, 665); norm = Default. */
orc_handler *orc_handler_new(int handler_kind, int dtype, size_t n);
/* handler.normalization(norm) builder (lib.rs:308, 492, 683). */
void orc_handler_normalization(orc_handler *h, int norm_mode, orc_custom_norm_fn f);
void orc_handler_free(orc_handler *h);

/*
 * One nd* / nd*_par call.  Strides are in ELEMENTS, signed (ndarray semantics).
 * par != 0 selects create_transform_par! (OpenMP over lanes stands in for rayon's par_for_each).
 * On a restated panic returns the ORC_PANIC_* code and writes the panic message to err.
 */
int orc_nd(int func, int par, const void *in, void *out, int ndim,
           const int64_t *shape_in, const int64_t *strides_in,
           const int64_t *shape_out, const int64_t *strides_out,
           const orc_handler *h, size_t axis, char *err, size_t errlen);

/* Which of the three iterator strategies the last orc_nd call on this thread took: 1, 2 or 3. */
int orc_last_strategy(void);

/* ---- long-double O(n^2) definitions: the "truth" the oracle itself is checked against ---- */
/* forward (sign=-1) or backward (sign=+1) unnormalised DFT, interleaved re/im doubles */
void orc_truth_dft(const double *in, double *out, size_t n, int sign);
/* rustdct definitions (half-scale): type 1..4 */
void orc_truth_dct(int type, const double *in, double *out, size_t n);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
