/*
 * ndfft_mi355x.h -- C ABI of the MI355X-native axis-transform engine (libndfft_mi355x.so).
 *
 * This is the drop-in boundary for ndrustfft's ONE hot path: "apply a 1-D transform
 * (C2C / R2C / C2R / DCT-I..IV) to every lane of an n-d array along one axis".
 * The reference has no FFI (it is a single Rust file, /root/reference/src/lib.rs); the boundary is
 * placed at whole-call granularity: ONE call here replaces the whole body of one `nd*` /
 * `nd*_par` call -- the lane iterator macros (src/lib.rs:100-167, 169-238), the handler's
 * per-lane method (src/lib.rs:313-331, 497-523, 688-734) and the rustfft / realfft / rustdct
 * kernels underneath (call sites src/lib.rs:317, 325, 502, 522, 697, 709, 721, 733).
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns an ndfft_status and
 * never unwinds; ndfft_last_error() gives the message (for NDFFT_ERR_SIZE_MISMATCH it is the
 * reference's panic text, "Size mismatch in fft, got {} expected {}").
 *
 * The Rust-side binding a maintainer would add is shown in INTEGRATION.md (and rust/src/ffi.rs);
 * a C++ mirror of the reference API lives in include/ndrustfft.hpp.
 */
#ifndef NDFFT_MI355X_H
#define NDFFT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDFFT_ABI_VERSION 1
/* Minor revision: bumped when entry points are ADDED (existing ones keep their meaning).  1 = round-4 exports (ndfft_reload_switches,
 * ndfft_documented_switches, ndfft_set_input_hint ...); 2 = ndfft_abi_minor itself; 3 = ndfft_jit_prebuild (round 6). */
#define NDFFT_ABI_MINOR 3

typedef enum {
    NDFFT_OK = 0,
    NDFFT_ERR_INVALID_ARG = 1,    /* null pointer, bad enum, op does not belong to the plan's kind      */
    NDFFT_ERR_SIZE_MISMATCH = 2,  /* lane length != handler n / n/2+1: lib.rs:340-347, 533-540, 743-750 */
    NDFFT_ERR_SHAPE_MISMATCH = 3, /* a non-axis dimension differs between in and out (ndarray Zip panic, lib.rs:120-121) */
    NDFFT_ERR_AXIS = 4,           /* axis >= ndim: the index panic at lib.rs:116                         */
    NDFFT_ERR_UNSUPPORTED = 5,    /* prime factor beyond the global Bluestein (M > 2^21) or n > 2^24     */
    NDFFT_ERR_HIP = 6,            /* a HIP runtime call failed; message carries hipGetErrorString        */
    NDFFT_ERR_NO_DEVICE = 7,      /* no gfx950 device visible -- there is NO CPU fallback                */
    NDFFT_ERR_ALLOC = 8
} ndfft_status;

/* T: FftNum -- f32 and f64 only (lib.rs:111).  Complex<T> is {re, im} interleaved. */
typedef enum { NDFFT_F32 = 0, NDFFT_F64 = 1 } ndfft_dtype;

/* Which handler a plan stands for. */
typedef enum {
    NDFFT_KIND_C2C = 0, /* FftHandler<T>     lib.rs:269-348 */
    NDFFT_KIND_R2C = 1, /* R2cFftHandler<T>  lib.rs:451-541 */
    NDFFT_KIND_DCT = 2  /* DctHandler<T>     lib.rs:640-751 */
} ndfft_kind;

/* Which lane method to run; one per public nd* function (and its _par twin). */
typedef enum {
    NDFFT_OP_C2C_FWD = 0, /* ndfft / ndfft_par           -> FftHandler::fft_lane         lib.rs:313-318 */
    NDFFT_OP_C2C_INV = 1, /* ndifft / ndifft_par         -> FftHandler::ifft_lane        lib.rs:321-331 */
    NDFFT_OP_R2C = 2,     /* ndfft_r2c / ndfft_r2c_par   -> R2cFftHandler::fft_r2c_lane  lib.rs:497-503 */
    NDFFT_OP_C2R = 3,     /* ndifft_r2c / ndifft_r2c_par -> R2cFftHandler::ifft_r2c_lane lib.rs:506-523 */
    NDFFT_OP_DCT1 = 4,    /* nddct1 / nddct1_par         -> DctHandler::dct1_lane        lib.rs:688-698 */
    NDFFT_OP_DCT2 = 5,    /* nddct2                      -> dct2_lane                    lib.rs:700-710 */
    NDFFT_OP_DCT3 = 6,    /* nddct3                      -> dct3_lane                    lib.rs:712-722 */
    NDFFT_OP_DCT4 = 7     /* nddct4                      -> dct4_lane                    lib.rs:724-734 */
} ndfft_op;

/*
 * Normalization<T> (lib.rs:89-98).  None and Default are fused into the kernels at the point the
 * reference applies them: C2C inverse = after, on the output lane (x 1/n); C2R = before, on the
 * m-length complex lane (x 1/n, n = real length); DCT-I..IV = before, on the input lane (x 2).
 * Forward C2C and R2C ignore it (lib.rs:313-318, 497-503).  NDFFT_NORM_SCALE applies the caller's
 * scalar at that same point.  Normalization::Custom(fn) is a host function pointer and cannot
 * cross to the GPU: the language shim applies it on the host and calls with NONE.
 */
typedef enum { NDFFT_NORM_NONE = 0, NDFFT_NORM_DEFAULT = 1, NDFFT_NORM_SCALE = 2 } ndfft_norm;

typedef struct ndfft_plan ndfft_plan;

/* ---- library / device ------------------------------------------------------------------- */
int ndfft_abi_version(void);
int ndfft_abi_minor(void);   /* NDFFT_ABI_MINOR of the loaded library */
/* Message of the last failing call on THIS thread ("" if none). */
const char *ndfft_last_error(void);
/* Number of visible gfx950 devices (0 if none; never an error). */
int ndfft_device_count(void);
/* Select the device for this host thread (hipSetDevice). */
int ndfft_set_device(int device);

/* ---- plans: FftHandler::new / R2cFftHandler::new / DctHandler::new (lib.rs:294, 477, 665) --- */
/* Infallible for any n in the reference; here it fails only on bad enums / no device / HIP errors.
 * A plan is immutable after creation and may be used from many host threads at once (the
 * reference shares &handler across rayon workers, lib.rs:192-194). */
int ndfft_plan_create(int kind, int dtype, size_t n, ndfft_plan **out_plan);
/* #[derive(Clone)] on the handlers is an Arc bump (lib.rs:269, 451, 640): retain/release. */
int ndfft_plan_retain(ndfft_plan *plan);
int ndfft_plan_destroy(ndfft_plan *plan); /* release; frees at refcount 0 */
size_t ndfft_plan_n(const ndfft_plan *plan);
int ndfft_plan_kind(const ndfft_plan *plan);
int ndfft_plan_dtype(const ndfft_plan *plan);
/* Length of the lane on each side for `op` under this plan (n, or n/2+1 on the complex side of R2C/C2R). */
size_t ndfft_plan_lane_len_in(const ndfft_plan *plan, int op);
size_t ndfft_plan_lane_len_out(const ndfft_plan *plan, int op);

/* ---- execute: one nd* call ------------------------------------------------------------------
 * in / out     : base pointers of the two array views (element [0,0,...,0])
 * ndim         : D (same for both, as in the reference's signature lib.rs:105-110), 1..NDFFT_MAX_DIMS
 * shape_*      : extents; must agree except along `axis`
 * stride_*     : strides in ELEMENTS, signed (ndarray semantics; 0 = broadcast input is allowed)
 * axis         : transform axis
 * norm, scale  : see ndfft_norm; `scale` is read only for NDFFT_NORM_SCALE
 *
 * ndfft_exec        : host arrays.  Synchronous: `out` is fully written on return; nothing is
 *                     retained.  Staging is internal: calls whose input + output span at most 2 MiB run the
 *                     kernels straight from / into pinned, device-mapped bounce buffers (no DMA: two host copies
 *                     and one stream synchronisation -- 30-40 us for the reference's n = 128 / 129 bench
 *                     shapes; 264 x 264 c128 is 2.2 MiB and already takes the next path); larger ones are staged
 *                     through HBM with two synchronous copies; dense C-ordered calls are cut into a pipeline of row
 *                     chunks from 8 MiB when both arrays are pinned (ndfft_host_alloc) or cache-registered, and from
 *                     32 MiB (and at least 2^20 points) through pinned bounce buffers when they are pageable.
 * ndfft_exec_device : device-resident arrays (hipMalloc'd).  Asynchronous on `stream`
 *                     (a hipStream_t, NULL = default stream).  This is what the roofline numbers
 *                     are measured on and what multi-axis / multi-GPU callers chain.
 * in and out must not overlap (Rust's &/&mut guarantee).
 */
#define NDFFT_MAX_DIMS 16

int ndfft_exec(const ndfft_plan *plan, int op, const void *in, void *out, int ndim,
               const int64_t *shape_in, const int64_t *stride_in,
               const int64_t *shape_out, const int64_t *stride_out,
               int axis, int norm, double scale);

int ndfft_exec_device(const ndfft_plan *plan, int op, const void *d_in, void *d_out, int ndim,
                      const int64_t *shape_in, const int64_t *stride_in,
                      const int64_t *shape_out, const int64_t *stride_out,
                      int axis, int norm, double scale, void *stream);

/* ---- the same call spread over several GPUs of one node, from ONE host process ---------------------------
 * The `_par` twins of the reference hand the independent lanes to rayon's worker threads (create_transform_par!,
 * lib.rs:169-238; lanes never interact, lib.rs:187-194).  Here the workers are GPUs: the array is cut into
 * n_devices contiguous blocks along its outermost non-transform dimension and device_ids[g] transforms block g
 * with the ordinary single-device path -- no collective, no exchange.  Same argument checks and panic texts as
 * ndfft_exec.  device_ids may name a device more than once (its blocks then run one after the other).
 *
 * ndfft_exec_sharded        : host arrays; every device moves its block over its OWN PCIe link, concurrently.
 *                             Synchronous like ndfft_exec.
 * ndfft_exec_sharded_device : arrays resident on ONE device (the one that owns d_out): blocks for the other devices are
 *                             scattered and gathered host-less with hipMemcpyPeerAsync over xGMI.  `stream` is the stream
 *                             the input was produced on; it is synchronised first, and the call returns when the
 *                             whole output is complete.  (One xGMI link carries ~153 GB/s against ~6 TB/s of HBM: for a
 *                             single transform the two transfers dwarf the kernel -- DESIGN.md section 7 -- so this entry
 *                             point is for arrays that must end up back on one device; arrays that STAY sharded are
 *                             driven with ndfft_set_device + ndfft_exec_device from one host thread per device.) */
int ndfft_exec_sharded(const ndfft_plan *plan, int op, const void *in, void *out, int ndim,
                       const int64_t *shape_in, const int64_t *stride_in,
                       const int64_t *shape_out, const int64_t *stride_out,
                       int axis, int norm, double scale, int n_devices, const int *device_ids);
int ndfft_exec_sharded_device(const ndfft_plan *plan, int op, const void *d_in, void *d_out, int ndim,
                              const int64_t *shape_in, const int64_t *stride_in,
                              const int64_t *shape_out, const int64_t *stride_out,
                              int axis, int norm, double scale, int n_devices, const int *device_ids, void *stream);

/* ---- where does the input of the next calls come from?  (speed only; no reference counterpart) -------------------------------
 * The batched C2C row kernels (BASELINE configs[1] / [4]) read their input either with the default cache policy -- up to 15 %
 * faster when the array is resident in the MI355X's 256 MiB Infinity Cache -- or with streaming loads -- 6 % faster when it comes
 * from HBM.  AUTO (the default) decides from what this thread's own earlier calls imply: an array it read recently with the
 * default policy is resident; an array it WROTE (outputs are stored non-temporally) or has pushed out of the cache since is
 * not; an array it has never seen is assumed resident if it is <= 384 MiB.  A caller that knows better -- the input was just
 * produced by another kernel, or has not been touched for a long time -- says so.  Per host thread, sticky until changed. */
typedef enum { NDFFT_INPUT_AUTO = 0, NDFFT_INPUT_CACHED = 1, NDFFT_INPUT_COLD = 2 } ndfft_input_hint;
int ndfft_set_input_hint(int hint);
/* Diagnostic: the load policy the last ndfft_exec_device on this thread ran with -- 0 default policy, 1 streaming loads, -1 the kernel it
 * dispatched to has a fixed policy.  Used by the tests of the residency model. */
int ndfft_last_input_policy(void);

/* Name of the kernel path the last successful exec on this thread dispatched to
 * ("pow2_reg", "generic_row", "generic_col", "generic_strided", "transpose+row", ...). */
const char *ndfft_last_path(void);

/* Diagnostic (no reference counterpart; needs no GPU): writes a text description of the recipes a handler of (kind, dtype, n) would use -- one line per
 * inner-FFT slot: "slot=MAIN F=511 route=rader p=73 mc=7x1 M=72 tpl=9 e=9 radix=9.8 ..." -- into buf (NUL-terminated, truncated to buflen).
 * Returns the number of bytes the full text needs (excluding the NUL), or a negative status.  Used by tests/test_plan_recipes.py. */
int ndfft_explain_plan(int kind, int dtype, size_t n, char *buf, size_t buflen);

/* Environment switches (no reference counterpart).  The library reads its NDFFT_* environment switches ONCE, on first use, into one
 * struct (csrc/switches.h); nothing on the call path reads the environment.  INTEGRATION.md section 6 documents every switch.
 * ndfft_documented_switches writes their names, one per line, into buf (NUL-terminated, truncated to buflen) and returns the bytes the
 * full text needs.  ndfft_reload_switches re-reads the environment -- a TEST hook (the parity tests close a kernel route, reload, run a
 * case, restore): it must not run while another thread is inside ndfft_exec* / ndfft_plan_create, and plans keep the recipes they were
 * created with. */
int ndfft_documented_switches(char *buf, size_t buflen);
int ndfft_reload_switches(void);

/* Build step (no reference counterpart; needs NO GPU): the specialised kernels the reference's own lengths ask for are shipped as code objects beside the library
 * (<library directory>/jit_prebuilt, looked up read-only after the user's cache) so that a first nd* call on those lengths never waits for hiprtc.  They are BUILD
 * PRODUCTS: `manifest` (csrc/jit_prebuilt/manifest.txt, tracked) lists the kernels' few-line source texts; this call compiles entries first, first + stride, ... with
 * hiprtc against the kernel headers embedded in THIS library and writes them into out_dir under the names the library will look up (a hash of source + headers +
 * options: objects of other kernel text are never picked up).  Counts of entries compiled / already there / failed come back through the pointers (any may be NULL).
 * __graft_entry__.build() runs it; tools/prebuild_jit.py (on an MI355X) extends the manifest. */
int ndfft_jit_prebuild(const char *manifest, const char *out_dir, int first, int stride, int *built, int *present, int *failed);

/* ---- device memory helpers for shims that keep arrays resident between nd* calls ----------- */
int ndfft_dev_alloc(void **d_ptr, size_t bytes);
int ndfft_dev_free(void *d_ptr);
int ndfft_dev_upload(void *d_dst, const void *h_src, size_t bytes);   /* synchronous */
int ndfft_dev_download(void *h_dst, const void *d_src, size_t bytes); /* synchronous */
int ndfft_dev_sync(void *stream);                                     /* hipStreamSynchronize */

/* Pinned (page-locked) host memory.  ndfft_exec on arrays that BOTH live in memory from ndfft_host_alloc, are dense,
 * C-ordered in dimension 0 and transformed along another axis runs as a pipeline of row chunks -- upload of chunk c+1,
 * transform of chunk c and download of chunk c-1 overlap (PCIe is full duplex only for pinned memory: 2 x 256 MiB take
 * 5.6 ms instead of 9.7 ms).  Any other host memory takes the plain path.  No reference counterpart (ndarray
 * allocates pageable memory); a shim exposes it as an allocator for its array type. */
int ndfft_host_alloc(void **h_ptr, size_t bytes);
int ndfft_host_free(void *h_ptr);
/* Registration cache for ordinary (pageable) caller arrays -- what the reference's signature hands over, lib.rs:105-115.  OPT-IN:
 * ndfft_host_reg_cache(max_bytes) (or NDFFT_HOST_REG_CACHE_MB) switches it on with an LRU budget, 0 switches it off and drops every
 * registration.  When on, the SECOND ndfft_exec on the same array registers it with the driver (hipHostRegister, once) and later calls on
 * it run the pinned pipeline (2 x 256 MiB: ~6.2 ms instead of 8-10 ms through bounce buffers); one-shot arrays never pay.
 * Contract: before freeing (or reallocating) an array that has been through ndfft_exec, call ndfft_host_forget(ptr) -- the registration
 * covering ptr is dropped; NULL drops all.  A registration that outlives its array makes later copies from the reused addresses fail
 * ("invalid argument": this library then forgets the range and retries through the bounce buffers) or ABORT inside the HIP runtime
 * (measured on the MI355X, ROCm 7.2), in this library and in any other code of the process -- hence off by default, and only for
 * callers that own their arrays' lifetimes. */
int ndfft_host_reg_cache(size_t max_bytes);
int ndfft_host_forget(const void *h_ptr);

/* BLOCKS until every multi-device worker (ndfft_exec_sharded*) has drained its streams, then frees their chunk buffers and the CALLING
 * THREAD's device workspace on EVERY device it has used: the scratch arrays of the multi-pass paths (transpose route,
 * four-step, column four-step, global Bluestein) and the staging buffers of ndfft_exec.  They are otherwise
 * kept per thread and per stream for reuse (HIP-graph capture needs them stable).  Synchronises the device.
 * Also frees the chunk buffers of the multi-device workers behind ndfft_exec_sharded* (every worker, after its streams drained).
 * No reference counterpart: rustfft allocates its scratch inside every process() call (src/lib.rs:317). */
int ndfft_release_workspace(void);

#ifdef __cplusplus
}
#endif
#endif /* NDFFT_MI355X_H */
