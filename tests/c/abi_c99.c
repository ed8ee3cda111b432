/* The drop-in boundary is a C ABI: this file is compiled as strict C99 (tests/test_abi.py) to prove that include/ndfft_mi355x.h is a C header a
 * cgo / JNI / Rust-bindgen style binding can consume, and links every declared entry point.  With a GPU it also runs the README case of the
 * reference (6 x 4 f64, ndfft_r2c along axis 0, src/lib.rs:38-50); without one it checks the refusal (no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "ndfft_mi355x.h"

int main(void) {
    ndfft_plan *plan = NULL;
    double data[24];
    double out[2 * 16];
    const int64_t shape_in[2] = {6, 4}, stride_in[2] = {4, 1}, shape_out[2] = {4, 4}, stride_out[2] = {4, 1};
    int i, st;
    /* every entry point must at least link (address taken) */
    typedef void (*anyfn)(void);
    anyfn syms[] = {(anyfn)ndfft_abi_version, (anyfn)ndfft_abi_minor, (anyfn)ndfft_last_error, (anyfn)ndfft_device_count, (anyfn)ndfft_set_device, (anyfn)ndfft_plan_create,
                    (anyfn)ndfft_plan_retain, (anyfn)ndfft_plan_destroy, (anyfn)ndfft_plan_n, (anyfn)ndfft_plan_kind, (anyfn)ndfft_plan_dtype,
                    (anyfn)ndfft_plan_lane_len_in, (anyfn)ndfft_plan_lane_len_out, (anyfn)ndfft_exec, (anyfn)ndfft_exec_device, (anyfn)ndfft_exec_sharded,
                    (anyfn)ndfft_exec_sharded_device, (anyfn)ndfft_last_path, (anyfn)ndfft_explain_plan, (anyfn)ndfft_dev_alloc, (anyfn)ndfft_dev_free,
                    (anyfn)ndfft_dev_upload, (anyfn)ndfft_dev_download, (anyfn)ndfft_dev_sync, (anyfn)ndfft_release_workspace, (anyfn)ndfft_host_alloc,
                    (anyfn)ndfft_host_free, (anyfn)ndfft_set_input_hint, (anyfn)ndfft_last_input_policy, (anyfn)ndfft_host_reg_cache, (anyfn)ndfft_host_forget,
                    (anyfn)ndfft_documented_switches, (anyfn)ndfft_reload_switches, (anyfn)ndfft_jit_prebuild};
    for (i = 0; i < (int)(sizeof syms / sizeof syms[0]); ++i) if (!syms[i]) return 2;
    if (ndfft_abi_version() != 1 || ndfft_abi_minor() != NDFFT_ABI_MINOR) { printf("abi version\n"); return 1; }
    for (i = 0; i < 24; ++i) data[i] = (double)i;
    st = ndfft_plan_create(NDFFT_KIND_R2C, NDFFT_F64, 6, &plan);
    if (ndfft_device_count() == 0) {
        if (st != NDFFT_ERR_NO_DEVICE || plan != NULL || !strstr(ndfft_last_error(), "no CPU fallback")) { printf("expected a refusal, got %d: %s\n", st, ndfft_last_error()); return 1; }
        printf("c99 abi ok (no device: refused)\n");
        return 0;
    }
    if (st != NDFFT_OK) { printf("plan: %s\n", ndfft_last_error()); return 1; }
    st = ndfft_exec(plan, NDFFT_OP_R2C, data, out, 2, shape_in, stride_in, shape_out, stride_out, 0, NDFFT_NORM_DEFAULT, 0.0);
    if (st != NDFFT_OK) { printf("exec: %s\n", ndfft_last_error()); return 1; }
    /* expected rows: [60,66,72,78], [-12+20.7846i]x4, [-12+6.9282i]x4, [-12+0i]x4 */
    {
        const double re[4] = {60.0, -12.0, -12.0, -12.0}, im[4] = {0.0, 20.784609690826528, 6.928203230275509, 0.0};
        int r, c;
        for (r = 0; r < 4; ++r)
            for (c = 0; c < 4; ++c) {
                const double er = r == 0 ? 60.0 + 6.0 * c : re[r], ei = im[r];
                if (fabs(out[2 * (4 * r + c)] - er) > 1e-12 || fabs(out[2 * (4 * r + c) + 1] - ei) > 1e-12) { printf("mismatch at %d,%d\n", r, c); return 1; }
            }
    }
    /* the reference's panic text crosses the boundary as a status + message */
    st = ndfft_exec(plan, NDFFT_OP_R2C, data, out, 2, shape_out, stride_out, shape_out, stride_out, 0, NDFFT_NORM_DEFAULT, 0.0);
    if (st != NDFFT_ERR_SIZE_MISMATCH || !strstr(ndfft_last_error(), "Size mismatch in fft, got 4 expected 6")) { printf("panic text: %d %s\n", st, ndfft_last_error()); return 1; }
    ndfft_plan_destroy(plan);
    printf("c99 abi ok (README 6x4 case on the GPU)\n");
    return 0;
}
