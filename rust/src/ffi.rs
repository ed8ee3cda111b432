//! Raw bindings to `include/ndfft_mi355x.h` (ABI version 1).  One `ndfft_exec` call replaces the
//! whole body of one `nd*` call of the reference (lane iterator + handler lane method + rustfft /
//! realfft / rustdct kernel).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)]
pub struct ndfft_plan {
    _private: [u8; 0],
}

pub const NDFFT_OK: c_int = 0;
pub const NDFFT_ERR_SIZE_MISMATCH: c_int = 2;
pub const NDFFT_ERR_SHAPE_MISMATCH: c_int = 3;
pub const NDFFT_ERR_AXIS: c_int = 4;

pub const NDFFT_F32: c_int = 0;
pub const NDFFT_F64: c_int = 1;

pub const NDFFT_KIND_C2C: c_int = 0;
pub const NDFFT_KIND_R2C: c_int = 1;
pub const NDFFT_KIND_DCT: c_int = 2;

pub const NDFFT_OP_C2C_FWD: c_int = 0;
pub const NDFFT_OP_C2C_INV: c_int = 1;
pub const NDFFT_OP_R2C: c_int = 2;
pub const NDFFT_OP_C2R: c_int = 3;
pub const NDFFT_OP_DCT1: c_int = 4;
pub const NDFFT_OP_DCT2: c_int = 5;
pub const NDFFT_OP_DCT3: c_int = 6;
pub const NDFFT_OP_DCT4: c_int = 7;

pub const NDFFT_NORM_NONE: c_int = 0;
pub const NDFFT_NORM_DEFAULT: c_int = 1;

extern "C" {
    pub fn ndfft_last_error() -> *const c_char;
    /// NDFFT_ABI_VERSION / NDFFT_ABI_MINOR of the loaded library (minor grows when entry points are added)
    pub fn ndfft_abi_version() -> c_int;
    pub fn ndfft_abi_minor() -> c_int;
    pub fn ndfft_plan_create(kind: c_int, dtype: c_int, n: usize, out_plan: *mut *mut ndfft_plan) -> c_int;
    pub fn ndfft_plan_retain(plan: *mut ndfft_plan) -> c_int;
    pub fn ndfft_plan_destroy(plan: *mut ndfft_plan) -> c_int;
    pub fn ndfft_exec(
        plan: *const ndfft_plan, op: c_int, input: *const c_void, output: *mut c_void, ndim: c_int,
        shape_in: *const i64, stride_in: *const i64, shape_out: *const i64, stride_out: *const i64,
        axis: c_int, norm: c_int, scale: c_double,
    ) -> c_int;
    /// the same call spread over several GPUs from one process (contiguous lane blocks, no collective)
    pub fn ndfft_exec_sharded(
        plan: *const ndfft_plan, op: c_int, input: *const c_void, output: *mut c_void, ndim: c_int,
        shape_in: *const i64, stride_in: *const i64, shape_out: *const i64, stride_out: *const i64,
        axis: c_int, norm: c_int, scale: c_double, n_devices: c_int, device_ids: *const c_int,
    ) -> c_int;
    pub fn ndfft_exec_sharded_device(
        plan: *const ndfft_plan, op: c_int, d_input: *const c_void, d_output: *mut c_void, ndim: c_int,
        shape_in: *const i64, stride_in: *const i64, shape_out: *const i64, stride_out: *const i64,
        axis: c_int, norm: c_int, scale: c_double, n_devices: c_int, device_ids: *const c_int, stream: *mut c_void,
    ) -> c_int;
    pub fn ndfft_set_device(device: c_int) -> c_int;
    pub fn ndfft_device_count() -> c_int;
    /// frees the calling thread's device scratch / staging buffers (kept for reuse otherwise)
    pub fn ndfft_release_workspace() -> c_int;
    /// Diagnostic: text description of the recipes a handler of (kind, dtype, n) would use; needs no GPU.
    pub fn ndfft_explain_plan(kind: c_int, dtype: c_int, n: usize, buf: *mut c_char, buflen: usize) -> c_int;
    pub fn ndfft_documented_switches(buf: *mut c_char, buflen: usize) -> c_int;
    pub fn ndfft_reload_switches() -> c_int;
    /// Build step (no GPU needed): compiles the manifest of specialised kernels into `out_dir` (see include/ndfft_mi355x.h); a `build.rs` calls it once.
    pub fn ndfft_jit_prebuild(manifest: *const c_char, out_dir: *const c_char, first: c_int, stride: c_int, built: *mut c_int, present: *mut c_int, failed: *mut c_int) -> c_int;
    // device-resident arrays (no reference counterpart)
    pub fn ndfft_exec_device(
        plan: *const ndfft_plan, op: c_int, d_input: *const c_void, d_output: *mut c_void, ndim: c_int,
        shape_in: *const i64, stride_in: *const i64, shape_out: *const i64, stride_out: *const i64,
        axis: c_int, norm: c_int, scale: c_double, stream: *mut c_void,
    ) -> c_int;
    pub fn ndfft_dev_alloc(d_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn ndfft_dev_free(d_ptr: *mut c_void) -> c_int;
    pub fn ndfft_dev_upload(d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn ndfft_dev_download(h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    pub fn ndfft_dev_sync(stream: *mut c_void) -> c_int;
    /// pinned host memory: ndfft_exec on arrays that both live in it overlaps upload, transform and download
    pub fn ndfft_host_alloc(h_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn ndfft_host_free(h_ptr: *mut c_void) -> c_int;
    /// opt-in registration cache for the caller's own (pageable) arrays; `ndfft_host_forget` BEFORE such an array is freed
    pub fn ndfft_host_reg_cache(max_bytes: usize) -> c_int;
    pub fn ndfft_host_forget(h_ptr: *const c_void) -> c_int;
    /// 0 = auto (the library's Infinity-Cache model), 1 = the input is cache-resident, 2 = it comes from HBM; per host thread
    pub fn ndfft_set_input_hint(hint: c_int) -> c_int;
}
