//! ndrustfft on an AMD MI355X.
//!
//! Same public surface as ndrustfft 0.5.0 -- `ndfft`, `ndifft`, `ndfft_r2c`, `ndifft_r2c`,
//! `nddct1..4`, their `_par` twins, `FftHandler`, `R2cFftHandler`, `DctHandler`, `Normalization`,
//! and the re-exports `Complex`, `Zero`, `FftNum` -- but every call is ONE FFI call into `libndfft_mi355x`,
//! whose other side is hand-written HIP for gfx950.  Arrays may have any layout ndarray allows
//! (strides are passed through, signed, in elements).
//!
//! Differences a caller can observe:
//! * `T: FftNum + FloatConst` exactly as in the reference (src/lib.rs:111); `FftNum` is implemented for `f32` and
//!   `f64`, the two types rustfft implements it for, so code that is generic over the reference's bounds compiles.
//! * The `_par` twins spread one call over the GPUs named by [`set_par_devices`] (one GPU: same as the serial name).
//! * `Normalization::Custom(f)` runs `f` on the host, on a lane-major copy, at the point the
//!   reference applies it (after the inverse C2C transform; before the C2R and DCT transforms).
//! * Errors of the device runtime panic with the HIP error string.
pub use num_complex::Complex;
pub use num_traits::Zero;

use ndarray::{ArrayBase, Axis, Data, DataMut, Dimension};
use num_traits::FloatConst;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::sync::RwLock;

pub mod ffi;

mod sealed {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
}

/// The reference re-exports `rustfft::FftNum` (src/lib.rs:85) and bounds every function by `T: FftNum + FloatConst`
/// (src/lib.rs:111).  This is the same-named trait with the same supertraits rustfft 6.1 gives it, implemented for
/// the two element types rustfft implements it for; downstream code written `fn f<T: FftNum + FloatConst>` compiles
/// against this crate unchanged.  (Sealed: the device engine computes in f32 and f64.)
pub trait FftNum:
    Copy + num_traits::FromPrimitive + num_traits::Signed + Sync + Send + std::fmt::Debug + 'static + sealed::Sealed
{
    #[doc(hidden)]
    const DTYPE: c_int;
}
impl FftNum for f32 {
    const DTYPE: c_int = ffi::NDFFT_F32;
}
impl FftNum for f64 {
    const DTYPE: c_int = ffi::NDFFT_F64;
}

/// GPUs the `_par` functions spread one call over (`create_transform_par!`, src/lib.rs:169-238, hands the independent
/// lanes to rayon's workers; here the workers are GPUs).  Empty / one id: the current device only.
static PAR_DEVICES: RwLock<Vec<c_int>> = RwLock::new(Vec::new());
/// Selects the GPUs used by `ndfft_par` & co. (device ids as `rocm-smi` numbers them).
pub fn set_par_devices(ids: &[i32]) {
    *PAR_DEVICES.write().unwrap() = ids.iter().map(|&d| d as c_int).collect();
}

/// How the inverse / real / cosine transforms are scaled.
#[derive(Clone)]
pub enum Normalization<T> {
    /// Raw, unscaled transform.
    None,
    /// scipy-like: 1/n on the inverse FFTs, x2 on the DCTs.
    Default,
    /// A host function applied to every lane at the handler-specific point.
    Custom(fn(&mut [T])),
}

struct Plan(*mut ffi::ndfft_plan);
unsafe impl Send for Plan {}
unsafe impl Sync for Plan {} // plans are immutable after creation (see ndfft_mi355x.h)
impl Plan {
    fn new(kind: c_int, dtype: c_int, n: usize) -> Self {
        let mut p = std::ptr::null_mut();
        check(unsafe { ffi::ndfft_plan_create(kind, dtype, n, &mut p) });
        Plan(p)
    }
}
impl Clone for Plan {
    fn clone(&self) -> Self {
        check(unsafe { ffi::ndfft_plan_retain(self.0) });
        Plan(self.0)
    }
}
impl Drop for Plan {
    fn drop(&mut self) {
        unsafe { ffi::ndfft_plan_destroy(self.0) };
    }
}

fn check(status: c_int) {
    if status != ffi::NDFFT_OK {
        let msg = unsafe { CStr::from_ptr(ffi::ndfft_last_error()) }.to_string_lossy().into_owned();
        // size / axis / shape mismatches carry the reference's own panic text
        panic!("{}", msg);
    }
}

macro_rules! handler {
    ($(#[$m:meta])* $name:ident, $kind:expr, $norm_elem:ty) => {
        $(#[$m])*
        #[derive(Clone)]
        pub struct $name<T> {
            n: usize,
            plan: Plan,
            norm: Normalization<$norm_elem>,
        }
        impl<T: FftNum + FloatConst> $name<T> {
            /// Plans the transform of length `n` on the current device.
            #[must_use]
            pub fn new(n: usize) -> Self {
                Self { n, plan: Plan::new($kind, T::DTYPE, n), norm: Normalization::Default }
            }
            /// Builder: replaces the normalization.
            #[must_use]
            pub fn normalization(mut self, norm: Normalization<$norm_elem>) -> Self {
                self.norm = norm;
                self
            }
            /// Transform length.
            pub fn len(&self) -> usize { self.n }
        }
    };
}
handler!(/// Complex-to-complex handler.
    FftHandler, ffi::NDFFT_KIND_C2C, Complex<T>);
handler!(/// Real-to-complex / complex-to-real handler (`n` reals <-> `n/2+1` complex).
    R2cFftHandler, ffi::NDFFT_KIND_R2C, Complex<T>);
handler!(/// DCT-I..IV handler.
    DctHandler, ffi::NDFFT_KIND_DCT, T);

fn strides_i64<S, D: Dimension>(a: &ArrayBase<S, D>) -> (Vec<i64>, Vec<i64>)
where
    S: ndarray::RawData,
{
    (a.shape().iter().map(|&s| s as i64).collect(), a.strides().iter().map(|&s| s as i64).collect())
}

/// Applies a Custom normalization lane by lane (host side).
fn apply_custom<A: Clone, S: DataMut<Elem = A>, D: Dimension>(arr: &mut ArrayBase<S, D>, axis: usize, f: fn(&mut [A])) {
    for mut lane in arr.lanes_mut(Axis(axis)) {
        let mut tmp = lane.to_vec();
        f(&mut tmp);
        lane.assign(&ndarray::ArrayView1::from(&tmp));
    }
}

/// bad axis: the reference indexes `output.shape()[axis]` first (src/lib.rs:116) and panics with this text
fn check_axis(ndim: usize, axis: usize) {
    if axis >= ndim {
        panic!("index out of bounds: the len is {} but the index is {}", ndim, axis);
    }
}

/// one FFI call: serial -> ndfft_exec, parallel -> ndfft_exec_sharded over PAR_DEVICES (when more than one is set)
#[allow(clippy::too_many_arguments)]
fn exec(plan: &Plan, op: c_int, in_ptr: *const c_void, out_ptr: *mut c_void, shape_in: &[i64], stride_in: &[i64],
        shape_out: &[i64], stride_out: &[i64], axis: usize, mode: c_int, par: bool) {
    let devs = if par { PAR_DEVICES.read().unwrap().clone() } else { Vec::new() };
    check(unsafe {
        if devs.len() > 1 {
            ffi::ndfft_exec_sharded(plan.0, op, in_ptr, out_ptr, shape_in.len() as c_int, shape_in.as_ptr(), stride_in.as_ptr(),
                                    shape_out.as_ptr(), stride_out.as_ptr(), axis as c_int, mode, 0.0, devs.len() as c_int, devs.as_ptr())
        } else {
            ffi::ndfft_exec(plan.0, op, in_ptr, out_ptr, shape_in.len() as c_int, shape_in.as_ptr(), stride_in.as_ptr(),
                            shape_out.as_ptr(), stride_out.as_ptr(), axis as c_int, mode, 0.0)
        }
    });
}

// The three application points of `Normalization`: forward C2C and R2C ignore it (src/lib.rs:313-318,
// 497-503); the inverse C2C applies it AFTER, on the output lane (321-331); C2R and the DCTs apply it BEFORE, on a copy
// of the input lane (506-523, 688-734).  `Custom(f)` is a host function: it runs here, on the host, at that point.
macro_rules! transform_body {
    (ignored, $input:ident, $output:ident, $handler:ident, $axis:ident, $op:expr, $par:expr) => {{
        check_axis($output.ndim(), $axis);
        let (shape_in, stride_in) = strides_i64($input);
        let (shape_out, stride_out) = strides_i64($output);
        let mode = match $handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
        exec(&$handler.plan, $op, $input.as_ptr() as *const c_void, $output.as_mut_ptr() as *mut c_void, &shape_in, &stride_in,
             &shape_out, &stride_out, $axis, mode, $par);
    }};
    (after, $input:ident, $output:ident, $handler:ident, $axis:ident, $op:expr, $par:expr) => {{
        check_axis($output.ndim(), $axis);
        let (shape_in, stride_in) = strides_i64($input);
        let (shape_out, stride_out) = strides_i64($output);
        let mode = match $handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
        exec(&$handler.plan, $op, $input.as_ptr() as *const c_void, $output.as_mut_ptr() as *mut c_void, &shape_in, &stride_in,
             &shape_out, &stride_out, $axis, mode, $par);
        if let Normalization::Custom(f) = $handler.norm {
            apply_custom($output, $axis, f);
        }
    }};
    (before, $input:ident, $output:ident, $handler:ident, $axis:ident, $op:expr, $par:expr) => {{
        check_axis($output.ndim(), $axis);
        let (shape_out, stride_out) = strides_i64($output);
        if let Normalization::Custom(f) = $handler.norm {
            // acts on the input: run on an owned copy so that `input` stays untouched (it is a shared borrow)
            let mut staged = $input.to_owned();
            apply_custom(&mut staged, $axis, f);
            let (shape_in, stride_in) = strides_i64(&staged);
            exec(&$handler.plan, $op, staged.as_ptr() as *const c_void, $output.as_mut_ptr() as *mut c_void, &shape_in, &stride_in,
                 &shape_out, &stride_out, $axis, ffi::NDFFT_NORM_NONE, $par);
        } else {
            let (shape_in, stride_in) = strides_i64($input);
            let mode = match $handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
            exec(&$handler.plan, $op, $input.as_ptr() as *const c_void, $output.as_mut_ptr() as *mut c_void, &shape_in, &stride_in,
                 &shape_out, &stride_out, $axis, mode, $par);
        }
    }};
}

macro_rules! transform {
    ($(#[$m:meta])* $name:ident, $par:ident, $a:ty, $b:ty, $h:ident, $op:expr, $point:ident) => {
        $(#[$m])*
        pub fn $name<R, S, T, D>(input: &ArrayBase<R, D>, output: &mut ArrayBase<S, D>, handler: &$h<T>, axis: usize)
        where
            T: FftNum + FloatConst,
            R: Data<Elem = $a>,
            S: Data<Elem = $b> + DataMut,
            D: Dimension,
        {
            transform_body!($point, input, output, handler, axis, $op, false)
        }
        /// Parallel twin: one GPU processes every lane in parallel anyway; with [`set_par_devices`] the call is spread
        /// over several GPUs (contiguous blocks of the outermost non-transform dimension, no collective).
        #[cfg(feature = "parallel")]
        pub fn $par<R, S, T, D>(input: &ArrayBase<R, D>, output: &mut ArrayBase<S, D>, handler: &$h<T>, axis: usize)
        where
            T: FftNum + FloatConst,
            R: Data<Elem = $a>,
            S: Data<Elem = $b> + DataMut,
            D: Dimension,
        {
            transform_body!($point, input, output, handler, axis, $op, true)
        }
    };
}

transform!(/// Complex-to-complex forward FFT along `axis`.
    ndfft, ndfft_par, Complex<T>, Complex<T>, FftHandler, ffi::NDFFT_OP_C2C_FWD, ignored);
transform!(/// Complex-to-complex inverse FFT along `axis`.
    ndifft, ndifft_par, Complex<T>, Complex<T>, FftHandler, ffi::NDFFT_OP_C2C_INV, after);
transform!(/// Real-to-complex FFT along `axis`.
    ndfft_r2c, ndfft_r2c_par, T, Complex<T>, R2cFftHandler, ffi::NDFFT_OP_R2C, ignored);
transform!(/// Complex-to-real inverse FFT along `axis`.
    ndifft_r2c, ndifft_r2c_par, Complex<T>, T, R2cFftHandler, ffi::NDFFT_OP_C2R, before);
transform!(/// DCT-I along `axis`.
    nddct1, nddct1_par, T, T, DctHandler, ffi::NDFFT_OP_DCT1, before);
transform!(/// DCT-II along `axis`.
    nddct2, nddct2_par, T, T, DctHandler, ffi::NDFFT_OP_DCT2, before);
transform!(/// DCT-III along `axis`.
    nddct3, nddct3_par, T, T, DctHandler, ffi::NDFFT_OP_DCT3, before);
transform!(/// DCT-IV along `axis`.
    nddct4, nddct4_par, T, T, DctHandler, ffi::NDFFT_OP_DCT4, before);

/// Device-resident arrays: keep the `work` array of a multi-axis transform (examples/fft2.rs:23-27 in the
/// reference) in HBM between the axis passes instead of crossing PCIe twice per call.
/// No counterpart in ndrustfft; the functions mirror the host ones (same names, same handlers, same axis
/// semantics) on `DeviceArray`s.  `Normalization::Custom` is a host function: it costs one round trip of the array it acts on.
pub mod device {
    use super::*;

    /// A C-layout n-d array in device memory (owned).
    pub struct DeviceArray<A> {
        ptr: *mut c_void,
        shape: Vec<usize>,
        _elem: std::marker::PhantomData<A>,
    }
    unsafe impl<A: Send> Send for DeviceArray<A> {}

    impl<A: Copy> DeviceArray<A> {
        /// Uninitialised device array of `shape` (C layout).
        pub fn new(shape: &[usize]) -> Self {
            let len: usize = shape.iter().product();
            let mut p = std::ptr::null_mut();
            check(unsafe { ffi::ndfft_dev_alloc(&mut p, len.max(1) * std::mem::size_of::<A>()) });
            DeviceArray { ptr: p, shape: shape.to_vec(), _elem: std::marker::PhantomData }
        }
        /// Copies a host array (any layout) to the device in C layout.
        pub fn from_host<S: Data<Elem = A>, D: Dimension>(a: &ArrayBase<S, D>) -> Self {
            let d = Self::new(a.shape());
            let c = a.as_standard_layout();
            check(unsafe { ffi::ndfft_dev_upload(d.ptr, c.as_ptr() as *const c_void, c.len() * std::mem::size_of::<A>()) });
            d
        }
        /// Copies back into a C-layout host array of the same shape.
        pub fn to_host<S: DataMut<Elem = A>, D: Dimension>(&self, out: &mut ArrayBase<S, D>) {
            assert_eq!(out.shape(), &self.shape[..], "shape mismatch in to_host");
            assert!(out.is_standard_layout(), "to_host needs a C-layout destination");
            check(unsafe { ffi::ndfft_dev_sync(std::ptr::null_mut()) });
            check(unsafe { ffi::ndfft_dev_download(out.as_mut_ptr() as *mut c_void, self.ptr, out.len() * std::mem::size_of::<A>()) });
        }
        pub fn shape(&self) -> &[usize] { &self.shape }
        fn geom(&self) -> (Vec<i64>, Vec<i64>) {
            let shape: Vec<i64> = self.shape.iter().map(|&s| s as i64).collect();
            let mut stride = vec![1i64; shape.len()];
            for d in (0..shape.len().saturating_sub(1)).rev() { stride[d] = stride[d + 1] * shape[d + 1]; }
            (shape, stride)
        }
    }
    impl<A> Drop for DeviceArray<A> {
        fn drop(&mut self) { unsafe { ffi::ndfft_dev_free(self.ptr) }; }
    }

    impl<A: Copy + Zero> DeviceArray<A> {
        /// `Normalization::Custom(f)` is a host function: the array makes one round trip through host memory, `f` runs on every
        /// lane along `axis` there (the same `apply_custom` the host functions use), and the result is uploaded again.
        fn map_lanes_on_host(&self, axis: usize, f: fn(&mut [A])) -> DeviceArray<A> {
            let len: usize = self.shape.iter().product();
            let mut host = ndarray::ArrayD::<A>::from_elem(ndarray::IxDyn(&self.shape), A::zero());
            check(unsafe { ffi::ndfft_dev_sync(std::ptr::null_mut()) });
            check(unsafe { ffi::ndfft_dev_download(host.as_mut_ptr() as *mut c_void, self.ptr, len * std::mem::size_of::<A>()) });
            apply_custom(&mut host, axis, f);
            DeviceArray::from_host(&host)
        }
    }

    fn exec_device<T: FftNum + FloatConst, A, B>(plan: &Plan, op: c_int, input: &DeviceArray<A>, output: &mut DeviceArray<B>, axis: usize, mode: c_int)
    where A: Copy, B: Copy {
        let (si, sti) = input.geom();
        let (so, sto) = output.geom();
        check(unsafe {
            ffi::ndfft_exec_device(plan.0, op, input.ptr, output.ptr, si.len() as c_int, si.as_ptr(), sti.as_ptr(),
                                   so.as_ptr(), sto.as_ptr(), axis as c_int, mode, 0.0, std::ptr::null_mut())
        });
    }

    // the same three application points as the host functions (`transform_body!`): ignored / after / before
    macro_rules! device_transform {
        ($name:ident, $a:ty, $b:ty, $h:ident, $op:expr, ignored) => {
            /// Device-resident twin of the host function of the same name (asynchronous on the null stream).
            pub fn $name<T: FftNum + FloatConst>(input: &DeviceArray<$a>, output: &mut DeviceArray<$b>, handler: &$h<T>, axis: usize) {
                check_axis(output.shape.len(), axis);
                let mode = match handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
                exec_device::<T, $a, $b>(&handler.plan, $op, input, output, axis, mode);
            }
        };
        ($name:ident, $a:ty, $b:ty, $h:ident, $op:expr, after) => {
            /// Device-resident twin of the host function of the same name.  `Normalization::Custom` runs on the host, on the OUTPUT
            /// lanes after the transform (src/lib.rs:326-330): one extra round trip of the output array.
            pub fn $name<T: FftNum + FloatConst>(input: &DeviceArray<$a>, output: &mut DeviceArray<$b>, handler: &$h<T>, axis: usize) {
                check_axis(output.shape.len(), axis);
                let mode = match handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
                exec_device::<T, $a, $b>(&handler.plan, $op, input, output, axis, mode);
                if let Normalization::Custom(f) = handler.norm {
                    let fixed = output.map_lanes_on_host(axis, f);
                    *output = fixed;
                }
            }
        };
        ($name:ident, $a:ty, $b:ty, $h:ident, $op:expr, before) => {
            /// Device-resident twin of the host function of the same name.  `Normalization::Custom` runs on the host, on a copy of the
            /// INPUT lanes before the transform (src/lib.rs:511-515, 692-696): one extra round trip of the input array.
            pub fn $name<T: FftNum + FloatConst>(input: &DeviceArray<$a>, output: &mut DeviceArray<$b>, handler: &$h<T>, axis: usize) {
                check_axis(output.shape.len(), axis);
                if let Normalization::Custom(f) = handler.norm {
                    let staged = input.map_lanes_on_host(axis, f);
                    exec_device::<T, $a, $b>(&handler.plan, $op, &staged, output, axis, ffi::NDFFT_NORM_NONE);
                    check(unsafe { ffi::ndfft_dev_sync(std::ptr::null_mut()) });   // `staged` is freed on return
                } else {
                    let mode = match handler.norm { Normalization::Default => ffi::NDFFT_NORM_DEFAULT, _ => ffi::NDFFT_NORM_NONE };
                    exec_device::<T, $a, $b>(&handler.plan, $op, input, output, axis, mode);
                }
            }
        };
    }
    device_transform!(ndfft, Complex<T>, Complex<T>, FftHandler, ffi::NDFFT_OP_C2C_FWD, ignored);
    device_transform!(ndifft, Complex<T>, Complex<T>, FftHandler, ffi::NDFFT_OP_C2C_INV, after);
    device_transform!(ndfft_r2c, T, Complex<T>, R2cFftHandler, ffi::NDFFT_OP_R2C, ignored);
    device_transform!(ndifft_r2c, Complex<T>, T, R2cFftHandler, ffi::NDFFT_OP_C2R, before);
    device_transform!(nddct1, T, T, DctHandler, ffi::NDFFT_OP_DCT1, before);
    device_transform!(nddct2, T, T, DctHandler, ffi::NDFFT_OP_DCT2, before);
    device_transform!(nddct3, T, T, DctHandler, ffi::NDFFT_OP_DCT3, before);
    device_transform!(nddct4, T, T, DctHandler, ffi::NDFFT_OP_DCT4, before);
}
