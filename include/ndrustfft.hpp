// ndrustfft.hpp -- C++ host-side mirror of ndrustfft's public API over the C ABI of
// ndfft_mi355x.h.  (The reference's host language, Rust, has no toolchain in the build image; the
// Rust shim a maintainer would use is in rust/ and INTEGRATION.md.  This header keeps the same
// names, argument meaning and error behaviour so that tests/cpp reads like the reference's tests.)
//
//   reference (src/lib.rs)                       here
//   ------------------------------------------   ---------------------------------------------
//   Normalization<T> {None, Default, Custom(fn)}  ndrustfft::Normalization<T>            (89-98)
//   FftHandler<T>::new(n).normalization(..)       ndrustfft::FftHandler<T>(n).normalization(..) (269-311)
//   R2cFftHandler<T>, DctHandler<T>               same names                              (451-495, 640-686)
//   ndfft / ndifft / ndfft_r2c / ndifft_r2c /     same names, + _par twins                (350-421, 543-611,
//   nddct1..4 (input, output, handler, axis)                                               753-844)
//   panics                                        ndrustfft::Panic (what() = the panic text)
//   ndarray ArrayBase<S, D>                       ndrustfft::ArrayView<A> (ptr + shape + signed element
//                                                 strides) and the owning ndrustfft::Array<A>
//
// Header-only; link with -lndfft_mi355x.  The GPU does all the arithmetic; there is no CPU path.
#pragma once
#include <complex>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ndfft_mi355x.h"

namespace ndrustfft {

template <typename T> using Complex = std::complex<T>;   // {re, im}: layout-compatible with Complex<T>

struct Panic : std::runtime_error {
    int status;
    Panic(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};
struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

namespace detail {
inline void check(int st) {
    if (st == NDFFT_OK) return;
    std::string m = ndfft_last_error();
    if (st == NDFFT_ERR_SIZE_MISMATCH || st == NDFFT_ERR_SHAPE_MISMATCH || st == NDFFT_ERR_AXIS) throw Panic(st, m);
    throw Error(st, m);
}
template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = NDFFT_F32; };
template <> struct dtype_of<double> { static constexpr int value = NDFFT_F64; };
}  // namespace detail

// ---- Normalization<T> (lib.rs:89-98) -------------------------------------------------------------
template <typename T> struct Normalization {
    enum Kind { None, Default, Custom } kind = Default;
    void (*fn)(T *data, std::size_t len) = nullptr;   // Custom(fn(&mut [T])): a plain function pointer
    static Normalization none() { return {None, nullptr}; }
    static Normalization dflt() { return {Default, nullptr}; }
    static Normalization custom(void (*f)(T *, std::size_t)) { return {Custom, f}; }
};

// ---- a minimal strided view (what the nd* functions need from ndarray's ArrayBase) -------------------
template <typename A> struct ArrayView {
    A *ptr = nullptr;
    std::vector<std::int64_t> shape, strides;   // strides in ELEMENTS, signed
    std::size_t ndim() const { return shape.size(); }
    std::size_t len() const { std::size_t n = 1; for (auto s : shape) n *= (std::size_t)s; return n; }
    template <typename... I> A &operator()(I... idx) const {
        std::int64_t ix[] = {(std::int64_t)idx...}, off = 0;
        for (std::size_t d = 0; d < sizeof...(I); ++d) off += ix[d] * strides[d];
        return ptr[off];
    }
    operator ArrayView<const A>() const { return {ptr, shape, strides}; }
};

// owning array, C ("standard") or F layout -- Array2::zeros((nx, ny)) / .f()
template <typename A> class Array {
  public:
    Array() = default;
    static Array zeros(std::vector<std::int64_t> shape, bool f_layout = false) {
        Array a;
        a.v_.shape = shape;
        a.v_.strides.assign(shape.size(), 1);
        std::int64_t s = 1;
        if (!f_layout) for (std::size_t d = shape.size(); d-- > 0;) { a.v_.strides[d] = s; s *= shape[d]; }
        else for (std::size_t d = 0; d < shape.size(); ++d) { a.v_.strides[d] = s; s *= shape[d]; }
        a.data_.assign((std::size_t)s, A());
        a.v_.ptr = a.data_.data();
        return a;
    }
    // logical (row-major) fill order, like arr.iter_mut().zip(values)
    static Array from(std::vector<std::int64_t> shape, const std::vector<A> &values, bool f_layout = false) {
        Array a = zeros(shape, f_layout);
        a.assign_logical(values);
        return a;
    }
    void assign_logical(const std::vector<A> &values) {
        std::vector<std::int64_t> ix(v_.shape.size(), 0);
        for (std::size_t k = 0; k < values.size(); ++k) {
            std::int64_t off = 0;
            for (std::size_t d = 0; d < ix.size(); ++d) off += ix[d] * v_.strides[d];
            data_[(std::size_t)off] = values[k];
            for (std::size_t d = ix.size(); d-- > 0;) { if (++ix[d] < v_.shape[d]) break; ix[d] = 0; }
        }
    }
    std::vector<A> to_logical() const {
        std::vector<A> out(v_.len());
        std::vector<std::int64_t> ix(v_.shape.size(), 0);
        for (std::size_t k = 0; k < out.size(); ++k) {
            std::int64_t off = 0;
            for (std::size_t d = 0; d < ix.size(); ++d) off += ix[d] * v_.strides[d];
            out[k] = data_[(std::size_t)off];
            for (std::size_t d = ix.size(); d-- > 0;) { if (++ix[d] < v_.shape[d]) break; ix[d] = 0; }
        }
        return out;
    }
    Array(const Array &o) : data_(o.data_), v_(o.v_) { v_.ptr = data_.data(); }
    Array &operator=(const Array &o) { data_ = o.data_; v_ = o.v_; v_.ptr = data_.data(); return *this; }
    Array(Array &&o) noexcept : data_(std::move(o.data_)), v_(std::move(o.v_)) { v_.ptr = data_.data(); }
    Array &operator=(Array &&o) noexcept { data_ = std::move(o.data_); v_ = std::move(o.v_); v_.ptr = data_.data(); return *this; }
    ArrayView<A> view() { return v_; }
    ArrayView<const A> view() const { return {v_.ptr, v_.shape, v_.strides}; }
    const std::vector<std::int64_t> &shape() const { return v_.shape; }
    template <typename... I> A &operator()(I... idx) { return v_(idx...); }
    template <typename... I> const A &operator()(I... idx) const { return v_(idx...); }

  private:
    std::vector<A> data_;
    ArrayView<A> v_;
};

// ---- handlers --------------------------------------------------------------------------------------------
namespace detail {
class PlanRef {   // Arc<plan>: copy = retain (#[derive(Clone)], lib.rs:269, 451, 640)
  public:
    PlanRef(int kind, int dtype, std::size_t n) { check(ndfft_plan_create(kind, dtype, n, &p_)); }
    PlanRef(const PlanRef &o) : p_(o.p_) { check(ndfft_plan_retain(p_)); }
    PlanRef &operator=(const PlanRef &o) {
        if (this != &o) { ndfft_plan_destroy(p_); p_ = o.p_; check(ndfft_plan_retain(p_)); }
        return *this;
    }
    ~PlanRef() { ndfft_plan_destroy(p_); }
    ndfft_plan *get() const { return p_; }

  private:
    ndfft_plan *p_ = nullptr;
};
}  // namespace detail

template <typename T> class FftHandler {   // lib.rs:269-348
  public:
    explicit FftHandler(std::size_t n) : n_(n), plan_(NDFFT_KIND_C2C, detail::dtype_of<T>::value, n) {}
    FftHandler normalization(Normalization<Complex<T>> norm) && { norm_ = norm; return std::move(*this); }
    FftHandler normalization(Normalization<Complex<T>> norm) const & { FftHandler h(*this); h.norm_ = norm; return h; }
    std::size_t n() const { return n_; }
    const Normalization<Complex<T>> &norm() const { return norm_; }
    ndfft_plan *plan() const { return plan_.get(); }

  private:
    std::size_t n_;
    detail::PlanRef plan_;
    Normalization<Complex<T>> norm_;   // Default (lib.rs:302)
};

template <typename T> class R2cFftHandler {   // lib.rs:451-541
  public:
    explicit R2cFftHandler(std::size_t n) : n_(n), m_(n / 2 + 1), plan_(NDFFT_KIND_R2C, detail::dtype_of<T>::value, n) {}
    R2cFftHandler normalization(Normalization<Complex<T>> norm) && { norm_ = norm; return std::move(*this); }
    R2cFftHandler normalization(Normalization<Complex<T>> norm) const & { R2cFftHandler h(*this); h.norm_ = norm; return h; }
    std::size_t n() const { return n_; }
    std::size_t m() const { return m_; }
    const Normalization<Complex<T>> &norm() const { return norm_; }
    ndfft_plan *plan() const { return plan_.get(); }

  private:
    std::size_t n_, m_;
    detail::PlanRef plan_;
    Normalization<Complex<T>> norm_;
};

template <typename T> class DctHandler {   // lib.rs:640-751
  public:
    explicit DctHandler(std::size_t n) : n_(n), plan_(NDFFT_KIND_DCT, detail::dtype_of<T>::value, n) {}
    DctHandler normalization(Normalization<T> norm) && { norm_ = norm; return std::move(*this); }
    DctHandler normalization(Normalization<T> norm) const & { DctHandler h(*this); h.norm_ = norm; return h; }
    std::size_t n() const { return n_; }
    const Normalization<T> &norm() const { return norm_; }
    ndfft_plan *plan() const { return plan_.get(); }

  private:
    std::size_t n_;
    detail::PlanRef plan_;
    Normalization<T> norm_;
};

// ---- the transform body: one FFI call per nd* call --------------------------------------------------------
namespace detail {

// GPUs the _par functions spread one call over (create_transform_par!, lib.rs:169-238: the independent lanes go to
// rayon's workers; here the workers are GPUs).  Empty or one id: the current device only.
inline std::vector<int> &par_devices_ref() { static std::vector<int> ids; return ids; }

// one C-ABI call: serial names -> ndfft_exec, _par names -> ndfft_exec_sharded when several devices are selected
inline int exec_host(bool par, ndfft_plan *plan, int op, const void *in, void *out, int ndim, const std::int64_t *shape_in,
                     const std::int64_t *stride_in, const std::int64_t *shape_out, const std::int64_t *stride_out, int axis, int mode) {
    const std::vector<int> &ids = par_devices_ref();
    if (par && ids.size() > 1)
        return ndfft_exec_sharded(plan, op, in, out, ndim, shape_in, stride_in, shape_out, stride_out, axis, mode, 0.0, (int)ids.size(), ids.data());
    return ndfft_exec(plan, op, in, out, ndim, shape_in, stride_in, shape_out, stride_out, axis, mode, 0.0);
}

// apply a Custom normalisation to every lane along `axis` of a contiguous C-order copy
template <typename A> void for_each_lane(A *data, const std::vector<std::int64_t> &shape, std::size_t axis,
                                         void (*fn)(A *, std::size_t)) {
    std::size_t outer = 1, inner = 1, n = (std::size_t)shape[axis];
    for (std::size_t d = 0; d < axis; ++d) outer *= (std::size_t)shape[d];
    for (std::size_t d = axis + 1; d < shape.size(); ++d) inner *= (std::size_t)shape[d];
    std::vector<A> lane(n);
    for (std::size_t o = 0; o < outer; ++o)
        for (std::size_t i = 0; i < inner; ++i) {
            A *base = data + o * n * inner + i;
            for (std::size_t j = 0; j < n; ++j) lane[j] = base[j * inner];
            fn(lane.data(), n);
            for (std::size_t j = 0; j < n; ++j) base[j * inner] = lane[j];
        }
}

template <typename A> std::vector<A> gather_c_order(const ArrayView<const A> &v) {
    std::vector<A> out(v.len());
    std::vector<std::int64_t> ix(v.ndim(), 0);
    for (std::size_t k = 0; k < out.size(); ++k) {
        std::int64_t off = 0;
        for (std::size_t d = 0; d < ix.size(); ++d) off += ix[d] * v.strides[d];
        out[k] = v.ptr[off];
        for (std::size_t d = ix.size(); d-- > 0;) { if (++ix[d] < v.shape[d]) break; ix[d] = 0; }
    }
    return out;
}

template <typename A> std::vector<std::int64_t> c_strides(const std::vector<std::int64_t> &shape) {
    std::vector<std::int64_t> s(shape.size(), 1);
    std::int64_t acc = 1;
    for (std::size_t d = shape.size(); d-- > 0;) { s[d] = acc; acc *= shape[d]; }
    return s;
}

// NormT = element type the handler's Normalization acts on; pre = applied BEFORE the transform on the
// input lane (C2R, DCT: lib.rs:511-515, 692-696) or AFTER on the output lane (C2C inverse: 326-330)
template <typename In, typename Out, typename NormT>
void transform(int op, const ArrayView<const In> &input, ArrayView<Out> &output, ndfft_plan *plan,
               const Normalization<NormT> &norm, bool norm_applies, bool norm_is_pre, std::size_t axis, bool par = false) {
    if (input.ndim() != output.ndim()) throw Error(NDFFT_ERR_INVALID_ARG, "input and output must have the same dimensionality D");
    const int ndim = (int)input.ndim();
    int mode = NDFFT_NORM_DEFAULT;
    if (norm.kind == Normalization<NormT>::None) mode = NDFFT_NORM_NONE;
    const bool custom = norm.kind == Normalization<NormT>::Custom && norm_applies;
    if (norm.kind == Normalization<NormT>::Custom) mode = NDFFT_NORM_NONE;
    if (axis > 0x7fffffffu) throw Panic(NDFFT_ERR_AXIS, "index out of bounds");
    if (custom && norm_is_pre) {
        if constexpr (std::is_same<In, NormT>::value) {
            if (axis < input.ndim()) {
                std::vector<In> tmp = gather_c_order<In>(input);
                for_each_lane<In>(tmp.data(), input.shape, axis, norm.fn);
                auto cs = c_strides<In>(input.shape);
                check(exec_host(par, plan, op, tmp.data(), output.ptr, ndim, input.shape.data(), cs.data(), output.shape.data(),
                                output.strides.data(), (int)axis, mode));
                return;
            }
        }
    }
    check(exec_host(par, plan, op, input.ptr, output.ptr, ndim, input.shape.data(), input.strides.data(), output.shape.data(),
                    output.strides.data(), (int)axis, mode));
    if (custom && !norm_is_pre) {
        if constexpr (std::is_same<Out, NormT>::value) {
            std::vector<Out> tmp = gather_c_order<Out>(ArrayView<const Out>{output.ptr, output.shape, output.strides});
            for_each_lane<Out>(tmp.data(), output.shape, axis, norm.fn);
            // scatter back
            std::vector<std::int64_t> ix(output.ndim(), 0);
            for (std::size_t k = 0; k < tmp.size(); ++k) {
                std::int64_t off = 0;
                for (std::size_t d = 0; d < ix.size(); ++d) off += ix[d] * output.strides[d];
                output.ptr[off] = tmp[k];
                for (std::size_t d = ix.size(); d-- > 0;) { if (++ix[d] < output.shape[d]) break; ix[d] = 0; }
            }
        }
    }
}
}  // namespace detail

// ---- device-resident arrays: keep `work` arrays in HBM between axis passes ----------
// DeviceArray<A> owns a C-layout array in device memory.  The nd* overloads on DeviceArray go through
// ndfft_exec_device (asynchronous on the default stream); upload()/download() are the only PCIe traffic.
// Normalization::Custom is a host function: on this path it costs one round trip of the array it acts on.
template <typename A> class DeviceArray {
  public:
    DeviceArray() = default;
    explicit DeviceArray(std::vector<std::int64_t> shape) : shape_(std::move(shape)) {
        strides_.assign(shape_.size(), 1);
        std::int64_t s = 1;
        for (std::size_t d = shape_.size(); d-- > 0;) { strides_[d] = s; s *= shape_[d]; }
        len_ = (std::size_t)s;
        detail::check(ndfft_dev_alloc(&ptr_, len_ * sizeof(A)));
    }
    static DeviceArray from_host(const Array<A> &h) {
        DeviceArray d(h.shape());
        auto v = h.to_logical();
        detail::check(ndfft_dev_upload(d.ptr_, v.data(), v.size() * sizeof(A)));
        return d;
    }
    Array<A> to_host() const {
        std::vector<A> v(len_);
        detail::check(ndfft_dev_sync(nullptr));
        detail::check(ndfft_dev_download(v.data(), ptr_, len_ * sizeof(A)));
        return Array<A>::from(shape_, v);
    }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
    DeviceArray(DeviceArray &&o) noexcept { *this = std::move(o); }
    DeviceArray &operator=(DeviceArray &&o) noexcept {
        if (this != &o) { release(); ptr_ = o.ptr_; shape_ = std::move(o.shape_); strides_ = std::move(o.strides_); len_ = o.len_; o.ptr_ = nullptr; }
        return *this;
    }
    ~DeviceArray() { release(); }
    void *ptr() const { return ptr_; }
    const std::vector<std::int64_t> &shape() const { return shape_; }
    const std::vector<std::int64_t> &strides() const { return strides_; }

  private:
    void release() { if (ptr_) ndfft_dev_free(ptr_); ptr_ = nullptr; }
    void *ptr_ = nullptr;
    std::vector<std::int64_t> shape_, strides_;
    std::size_t len_ = 0;
};

namespace detail {
// Normalization::Custom is a host function: the array it acts on makes one round trip through host memory (the same
// for_each_lane the host functions use), at the reference's application point -- before the transform on the input lanes
// (C2R, DCT: lib.rs:511-515, 692-696) or after it on the output lanes (C2C inverse: lib.rs:326-330).
template <typename A> void custom_on_device(void *dptr, const std::vector<std::int64_t> &shape, std::size_t len, std::size_t axis, void (*fn)(A *, std::size_t),
                                            void *dst_dptr) {
    std::vector<A> host(len);
    check(ndfft_dev_sync(nullptr));
    check(ndfft_dev_download(host.data(), dptr, len * sizeof(A)));
    for_each_lane<A>(host.data(), shape, axis, fn);
    check(ndfft_dev_upload(dst_dptr, host.data(), len * sizeof(A)));
}
template <typename In, typename Out, typename NormT>
void transform_device(int op, const DeviceArray<In> &input, DeviceArray<Out> &output, ndfft_plan *plan,
                      const Normalization<NormT> &norm, bool norm_applies, bool norm_is_pre, std::size_t axis, bool par = false) {
    if (input.shape().size() != output.shape().size()) throw Error(NDFFT_ERR_INVALID_ARG, "input and output must have the same dimensionality D");
    const bool custom = norm.kind == Normalization<NormT>::Custom && norm_applies;
    const int mode = norm.kind == Normalization<NormT>::Default ? NDFFT_NORM_DEFAULT : NDFFT_NORM_NONE;
    if (axis > 0x7fffffffu) throw Panic(NDFFT_ERR_AXIS, "index out of bounds");
    const void *in_ptr = input.ptr();
    void *staged = nullptr;
    if (custom && norm_is_pre) {
        if constexpr (std::is_same<In, NormT>::value) {
            if (axis < input.shape().size()) {
                std::size_t len = 1;
                for (auto e : input.shape()) len *= (std::size_t)e;
                check(ndfft_dev_alloc(&staged, len * sizeof(In)));
                custom_on_device<In>(input.ptr(), input.shape(), len, axis, norm.fn, staged);
                in_ptr = staged;
            }
        }
    }
    const std::vector<int> &ids = par_devices_ref();
    int st;
    if (par && ids.size() > 1)   // scatter / transform / gather over xGMI, host-less
        st = ndfft_exec_sharded_device(plan, op, in_ptr, output.ptr(), (int)input.shape().size(), input.shape().data(), input.strides().data(),
                                       output.shape().data(), output.strides().data(), (int)axis, mode, 0.0, (int)ids.size(), ids.data(), nullptr);
    else
        st = ndfft_exec_device(plan, op, in_ptr, output.ptr(), (int)input.shape().size(), input.shape().data(), input.strides().data(),
                               output.shape().data(), output.strides().data(), (int)axis, mode, 0.0, nullptr);
    if (staged) { (void)ndfft_dev_sync(nullptr); (void)ndfft_dev_free(staged); }
    check(st);
    if (custom && !norm_is_pre) {
        if constexpr (std::is_same<Out, NormT>::value) {
            std::size_t len = 1;
            for (auto e : output.shape()) len *= (std::size_t)e;
            custom_on_device<Out>(output.ptr(), output.shape(), len, axis, norm.fn, output.ptr());
        }
    }
}
}  // namespace detail

#define NDRUSTFFT_DEFINE(NAME, IN, OUT, HANDLER, OP, NORMT, APPLIES, PRE)                                   \
    template <typename T>                                                                                    \
    void NAME(const ArrayView<const IN> &input, ArrayView<OUT> output, const HANDLER<T> &handler, std::size_t axis) { \
        detail::transform<IN, OUT, NORMT>(OP, input, output, handler.plan(), handler.norm(), APPLIES, PRE, axis);     \
    }                                                                                                        \
    template <typename T>                                                                                    \
    void NAME(const Array<IN> &input, Array<OUT> &output, const HANDLER<T> &handler, std::size_t axis) {    \
        NAME<T>(input.view(), output.view(), handler, axis);                                                 \
    }                                                                                                        \
    template <typename T>                                                                                    \
    void NAME##_par(const ArrayView<const IN> &input, ArrayView<OUT> output, const HANDLER<T> &handler, std::size_t axis) { \
        /* one GPU processes every lane in parallel anyway; set_par_devices() spreads the call over several */ \
        detail::transform<IN, OUT, NORMT>(OP, input, output, handler.plan(), handler.norm(), APPLIES, PRE, axis, true); \
    }                                                                                                        \
    template <typename T>                                                                                    \
    void NAME##_par(const Array<IN> &input, Array<OUT> &output, const HANDLER<T> &handler, std::size_t axis) { \
        NAME##_par<T>(input.view(), output.view(), handler, axis);                                           \
    }                                                                                                        \
    template <typename T>                                                                                    \
    void NAME(const DeviceArray<IN> &input, DeviceArray<OUT> &output, const HANDLER<T> &handler, std::size_t axis) { \
        detail::transform_device<IN, OUT, NORMT>(OP, input, output, handler.plan(), handler.norm(), APPLIES, PRE, axis);   \
    }                                                                                                        \
    template <typename T>                                                                                    \
    void NAME##_par(const DeviceArray<IN> &input, DeviceArray<OUT> &output, const HANDLER<T> &handler, std::size_t axis) { \
        detail::transform_device<IN, OUT, NORMT>(OP, input, output, handler.plan(), handler.norm(), APPLIES, PRE, axis, true); \
    }

NDRUSTFFT_DEFINE(ndfft, Complex<T>, Complex<T>, FftHandler, NDFFT_OP_C2C_FWD, Complex<T>, false, false)       // lib.rs:350-372
NDRUSTFFT_DEFINE(ndifft, Complex<T>, Complex<T>, FftHandler, NDFFT_OP_C2C_INV, Complex<T>, true, false)       // lib.rs:374-397
NDRUSTFFT_DEFINE(ndfft_r2c, T, Complex<T>, R2cFftHandler, NDFFT_OP_R2C, Complex<T>, false, false)             // lib.rs:543-564
NDRUSTFFT_DEFINE(ndifft_r2c, Complex<T>, T, R2cFftHandler, NDFFT_OP_C2R, Complex<T>, true, true)              // lib.rs:566-587
NDRUSTFFT_DEFINE(nddct1, T, T, DctHandler, NDFFT_OP_DCT1, T, true, true)                                      // lib.rs:753-775
NDRUSTFFT_DEFINE(nddct2, T, T, DctHandler, NDFFT_OP_DCT2, T, true, true)                                      // lib.rs:789-796
NDRUSTFFT_DEFINE(nddct3, T, T, DctHandler, NDFFT_OP_DCT3, T, true, true)                                      // lib.rs:808-815
NDRUSTFFT_DEFINE(nddct4, T, T, DctHandler, NDFFT_OP_DCT4, T, true, true)                                      // lib.rs:827-834
#undef NDRUSTFFT_DEFINE

/// Selects the GPUs the `_par` functions spread one call over (ids as ndfft_set_device numbers them).
inline void set_par_devices(std::vector<int> ids) { detail::par_devices_ref() = std::move(ids); }

/// Frees the calling thread's device scratch and staging buffers (the engine keeps them for reuse).
inline void release_workspace() { detail::check(ndfft_release_workspace()); }
// Opt-in: the library registers a caller-owned host array on its second use and DMAs straight from / to it afterwards.  Call
// host_forget(ptr) before freeing such an array (include/ndfft_mi355x.h: ndfft_host_reg_cache).
inline void host_reg_cache(size_t max_bytes) { detail::check(ndfft_host_reg_cache(max_bytes)); }
inline void host_forget(const void *ptr) { detail::check(ndfft_host_forget(ptr)); }
// Where the input of this thread's next device-resident calls comes from (speed only): NDFFT_INPUT_AUTO / _CACHED / _COLD
inline void set_input_hint(int hint) { detail::check(ndfft_set_input_hint(hint)); }

}  // namespace ndrustfft
