/*
 * ndfft_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ndfft_oracle.h for the rules).
 *
 * L2 restated here: the lane iterator macros create_transform! (src/lib.rs:100-167) and
 * create_transform_par! (src/lib.rs:169-238) with their three strategies kept separate:
 *   (i)   both arrays standard layout, axis == last       -> zip rows(), slices    (117-124)
 *   (ii)  both standard layout, other axis -> swap_axes, per lane to_vec / assign  (125-137)
 *   (iii) anything else -> lanes(Axis(axis)), 4-way contiguity branch              (138-164)
 * L1 / L0 live in oracle_lane.inc, instantiated for f32 and f64.
 */
#include "ndfft_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PIL 3.14159265358979323846264338327950288L
#define ORC_MAX_DIRECT_PRIME 13

struct orc_handler {
    int kind;     /* ORC_HANDLER_* */
    int dtype;    /* ORC_F32 / ORC_F64 */
    size_t n;     /* handler.n */
    size_t m;     /* R2cFftHandler.m = n/2 + 1 (lib.rs:483) */
    int norm;     /* Normalization, Default on construction (lib.rs:302, 486, 677) */
    orc_custom_norm_fn custom;
    void *plans;
};

#define REAL float
#define SFX(x) x##_f32
#define CPX cpx_f32
#include "oracle_lane.inc"
#undef REAL
#undef SFX
#undef CPX

#define REAL double
#define SFX(x) x##_f64
#define CPX cpx_f64
#include "oracle_lane.inc"
#undef REAL
#undef SFX
#undef CPX

static __thread int g_last_strategy = 0;
int orc_last_strategy(void) { return g_last_strategy; }

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

orc_handler *orc_handler_new(int kind, int dtype, size_t n) {
    orc_handler *h = (orc_handler *)calloc(1, sizeof(*h));
    h->kind = kind; h->dtype = dtype; h->n = n; h->m = n / 2 + 1;
    h->norm = ORC_NORM_DEFAULT;
    h->plans = dtype == ORC_F32 ? hplans_new_f32(kind, n) : hplans_new_f64(kind, n);
    return h;
}
void orc_handler_normalization(orc_handler *h, int norm_mode, orc_custom_norm_fn f) {
    h->norm = norm_mode; h->custom = f;
}
void orc_handler_free(orc_handler *h) {
    if (!h) return;
    if (h->dtype == ORC_F32) hplans_free_f32(h->plans); else hplans_free_f64(h->plans);
    free(h);
}

/* ndarray's is_standard_layout: C order, ignoring axes of length 1; empty arrays are standard */
static int is_standard_layout(int ndim, const int64_t *shape, const int64_t *strides) {
    for (int d = 0; d < ndim; ++d) if (shape[d] == 0) return 1;
    int64_t expect = 1;
    for (int d = ndim - 1; d >= 0; --d) {
        if (shape[d] != 1 && strides[d] != expect) return 0;
        expect *= shape[d];
    }
    return 1;
}

static void elem_sizes(int func, int dtype, size_t *ein, size_t *eout) {
    size_t r = dtype == ORC_F32 ? 4 : 8, c = 2 * r;
    switch (func) {
        case ORC_NDFFT: case ORC_NDIFFT: *ein = c; *eout = c; break;
        case ORC_NDFFT_R2C: *ein = r; *eout = c; break;
        case ORC_NDIFFT_R2C: *ein = c; *eout = r; break;
        default: *ein = r; *eout = r; break;
    }
}

static int call_lane(int func, const orc_handler *h, const void *x, size_t xl, void *y, size_t yl,
                     char *err, size_t errlen) {
    return h->dtype == ORC_F32 ? lane_f32(func, h, x, xl, y, yl, err, errlen)
                               : lane_f64(func, h, x, xl, y, yl, err, errlen);
}

/* x.to_vec(): gather a strided lane into a fresh contiguous Vec */
static void *to_vec(const char *x, size_t len, int64_t stride, size_t es) {
    char *v = (char *)malloc((len ? len : 1) * es);
    for (size_t i = 0; i < len; ++i) memcpy(v + i * es, x + (int64_t)i * stride * (int64_t)es, es);
    return v;
}
/* y.assign(&outvec): scatter */
static void assign(char *y, size_t len, int64_t stride, size_t es, const char *v) {
    for (size_t i = 0; i < len; ++i) memcpy(y + (int64_t)i * stride * (int64_t)es, v + i * es, es);
}

typedef struct {
    int func, strategy, par;
    const orc_handler *h;
    const char *in; char *out;
    size_t ein, eout;
    int nb;                          /* number of batch (non-axis) dims */
    int64_t bshape[32], bsi[32], bso[32];
    size_t nlanes, xlen, ylen;
    int64_t xs, ys;                  /* element stride along the axis, in / out */
} lane_job;

static int run_one_lane(const lane_job *J, size_t lane, void *shared_outvec, char *err, size_t errlen) {
    int64_t oi = 0, oo = 0; size_t t = lane;
    for (int d = J->nb - 1; d >= 0; --d) {
        int64_t i = (int64_t)(t % (size_t)J->bshape[d]); t /= (size_t)J->bshape[d];
        oi += i * J->bsi[d]; oo += i * J->bso[d];
    }
    const char *x = J->in + oi * (int64_t)J->ein;
    char *y = J->out + oo * (int64_t)J->eout;
    int rc;
    if (J->strategy == 1) {
        /* (i) lib.rs:122-124 / 192-194: handler.$p(x.as_slice().unwrap(), y.as_slice_mut().unwrap()) */
        return call_lane(J->func, J->h, x, J->xlen, y, J->ylen, err, errlen);
    }
    if (J->strategy == 2) {
        /* (ii) lib.rs:132-135 (serial: one reused outvec) / 202-206 (par: outvec per lane) */
        void *xv = to_vec(x, J->xlen, J->xs, J->ein);
        void *ov = shared_outvec ? shared_outvec : calloc(J->ylen ? J->ylen : 1, J->eout);
        rc = call_lane(J->func, J->h, xv, J->xlen, ov, J->ylen, err, errlen);
        if (!rc) assign(y, J->ylen, J->ys, J->eout, (const char *)ov);
        if (!shared_outvec) free(ov);
        free(xv);
        return rc;
    }
    /* (iii) lib.rs:141-163 / 212-234: 4-way branch on as_slice() of each lane */
    int x_contig = (J->xlen <= 1 || J->xs == 1), y_contig = (J->ylen <= 1 || J->ys == 1);
    size_t n = J->ylen;              /* `let n = output.shape()[axis]` (lib.rs:116) */
    if (x_contig && y_contig) return call_lane(J->func, J->h, x, J->xlen, y, J->ylen, err, errlen);
    if (x_contig) {
        void *ov = calloc(n ? n : 1, J->eout);
        rc = call_lane(J->func, J->h, x, J->xlen, ov, n, err, errlen);
        if (!rc) assign(y, J->ylen, J->ys, J->eout, (const char *)ov);
        free(ov); return rc;
    }
    void *xv = to_vec(x, J->xlen, J->xs, J->ein);
    if (y_contig) {
        rc = call_lane(J->func, J->h, xv, J->xlen, y, J->ylen, err, errlen);
    } else {
        void *ov = calloc(n ? n : 1, J->eout);
        rc = call_lane(J->func, J->h, xv, J->xlen, ov, n, err, errlen);
        if (!rc) assign(y, J->ylen, J->ys, J->eout, (const char *)ov);
        free(ov);
    }
    free(xv);
    return rc;
}

int orc_nd(int func, int par, const void *in, void *out, int ndim,
           const int64_t *shape_in, const int64_t *strides_in,
           const int64_t *shape_out, const int64_t *strides_out,
           const orc_handler *h, size_t axis, char *err, size_t errlen) {
    if (err && errlen) err[0] = 0;
    if (!h || ndim < 0 || ndim > 32 || func < ORC_NDFFT || func > ORC_NDDCT4) return ORC_BAD_ARG;
    {   /* the macro's `handler: &$h` type bound (lib.rs:108) */
        int want = func <= ORC_NDIFFT ? ORC_HANDLER_FFT : func <= ORC_NDIFFT_R2C ? ORC_HANDLER_R2C : ORC_HANDLER_DCT;
        if (h->kind != want) return ORC_BAD_ARG;
    }
    /* lib.rs:116  let n = output.shape()[axis];  -> index panic */
    if (axis >= (size_t)ndim) {
        if (err && errlen) snprintf(err, errlen, "index out of bounds: the len is %d but the index is %zu", ndim, axis);
        return ORC_PANIC_AXIS;
    }
    lane_job J; memset(&J, 0, sizeof(J));
    J.func = func; J.par = par; J.h = h; J.in = (const char *)in; J.out = (char *)out;
    elem_sizes(func, h->dtype, &J.ein, &J.eout);
    const int std_in = is_standard_layout(ndim, shape_in, strides_in);
    const int std_out = is_standard_layout(ndim, shape_out, strides_out);
    const int outer_axis = ndim - 1;
    if (std_in && std_out) J.strategy = ((int)axis == outer_axis) ? 1 : 2; else J.strategy = 3;
    g_last_strategy = J.strategy;
    /* Zip::from(input.rows()/lanes()).and(output...) requires equal producer shapes (lib.rs:120-121) */
    J.nlanes = 1;
    for (int d = 0; d < ndim; ++d) {
        if (d == (int)axis) continue;
        if (shape_in[d] != shape_out[d]) {
            if (err && errlen) snprintf(err, errlen, "ndarray: Zip dimension mismatch on axis %d (%lld vs %lld)", d,
                                        (long long)shape_in[d], (long long)shape_out[d]);
            return ORC_PANIC_ZIP;
        }
        J.bshape[J.nb] = shape_in[d]; J.bsi[J.nb] = strides_in[d]; J.bso[J.nb] = strides_out[d];
        J.nb++; J.nlanes *= (size_t)shape_in[d];
    }
    J.xlen = (size_t)shape_in[axis]; J.ylen = (size_t)shape_out[axis];
    J.xs = strides_in[axis]; J.ys = strides_out[axis];
    if (J.nlanes == 0) return ORC_OK;            /* empty producer: the closure never runs */

    int rc_all = 0;
    if (!par) {
        void *outvec = J.strategy == 2 ? calloc(J.ylen ? J.ylen : 1, J.eout) : NULL;   /* lib.rs:126 */
        for (size_t l = 0; l < J.nlanes && !rc_all; ++l) rc_all = run_one_lane(&J, l, outvec, err, errlen);
        free(outvec);
        return rc_all;
    }
    /* par_for_each (lib.rs:192, 202, 212): rayon's pool restated as an OpenMP loop over lanes */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (long long l = 0; l < (long long)J.nlanes; ++l) {
        char lerr[160]; lerr[0] = 0;
        int rc = 0, skip;
#ifdef _OPENMP
#pragma omp atomic read
#endif
        skip = rc_all;
        if (skip) continue;
        rc = run_one_lane(&J, (size_t)l, NULL, lerr, sizeof lerr);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical(orc_err)
#endif
            { if (!rc_all) { rc_all = rc; if (err && errlen) snprintf(err, errlen, "%s", lerr); } }
        }
    }
    return rc_all;
}

/* ---------------- long-double definitions ---------------- */
void orc_truth_dft(const double *in, double *out, size_t n, int sign) {
    for (size_t k = 0; k < n; ++k) {
        long double sr = 0, si = 0;
        for (size_t j = 0; j < n; ++j) {
            unsigned long long q = ((unsigned long long)j * k) % n;
            long double ang = 2.0L * ORC_PIL * (long double)q / (long double)n;
            long double c = cosl(ang), s = sign * sinl(ang);
            long double xr = in[2 * j], xi = in[2 * j + 1];
            sr += xr * c - xi * s; si += xr * s + xi * c;
        }
        out[2 * k] = (double)sr; out[2 * k + 1] = (double)si;
    }
}

void orc_truth_dct(int type, const double *in, double *out, size_t n) {
    for (size_t k = 0; k < n; ++k) {
        long double s = 0;
        if (type == 1) {
            if (n == 1) { s = in[0]; }
            else {
                s = 0.5L * in[0] + ((k & 1) ? -0.5L : 0.5L) * in[n - 1];
                for (size_t j = 1; j + 1 < n; ++j) {
                    unsigned long long q = ((unsigned long long)j * k) % (2ull * (n - 1));
                    s += (long double)in[j] * cosl(ORC_PIL * (long double)q / (long double)(n - 1));
                }
            }
        } else if (type == 2) {
            for (size_t j = 0; j < n; ++j) {
                unsigned long long q = ((unsigned long long)k * (2 * j + 1)) % (4ull * n);
                s += (long double)in[j] * cosl(ORC_PIL * (long double)q / (long double)(2 * n));
            }
        } else if (type == 3) {
            s = 0.5L * in[0];
            for (size_t j = 1; j < n; ++j) {
                unsigned long long q = ((unsigned long long)j * (2 * k + 1)) % (4ull * n);
                s += (long double)in[j] * cosl(ORC_PIL * (long double)q / (long double)(2 * n));
            }
        } else {
            for (size_t j = 0; j < n; ++j) {
                unsigned long long q = ((unsigned long long)(2 * j + 1) * (2 * k + 1)) % (8ull * n);
                s += (long double)in[j] * cosl(ORC_PIL * (long double)q / (long double)(4 * n));
            }
        }
        out[k] = (double)s;
    }
}
