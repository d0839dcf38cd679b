// Large-offset pre-alignment: LargeOffsetMatcher.match (large_offset.py:32-41: phase correlation) and shift_image (image.py:70-101).
#include "api_internal.hpp"

#include <cstring>
#include <vector>

extern "C" {

int km_phase_shift_dev(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double out_rc[2])
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_a, H, W, sa, "phase_shift")) || (rc = check_image(c, d_b, H, W, sb, "phase_shift"))) return rc;
    if (!out_rc || !km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d or null output", dtype);
    c->evs_used[c->ev_cur][ST_PHASE] = false;
    km_stage_timer t(c, ST_PHASE);
    return kp_phase_shift(c, d_a, d_b, dtype, H, W, sa, sb, out_rc);
}

int km_phase_shift(km_ctx *c, const void *a, const void *b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double out_rc[2])
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, a, H, W, sa, "phase_shift")) || (rc = check_image(c, b, H, W, sb, "phase_shift"))) return rc;
    const size_t es = km_dtype_size(dtype);
    if (!out_rc || !es) return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d or null output", dtype);
    void *d_a, *d_b;
    if ((rc = upload_image(c, WS_RAW_A, a, es, H, W, sa, &d_a)) || (rc = upload_image(c, WS_RAW_B, b, es, H, W, sb, &d_b))) return rc;
    return kp_phase_shift(c, d_a, d_b, dtype, H, W, W, W, out_rc);
}

int km_shift_image_dev(km_ctx *c, const void *d_img, int elem_size, int H, int W, ptrdiff_t stride, int y_off, int x_off, void *d_out)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_img, H, W, stride, "shift_image"))) return rc;
    if (!d_out) return km_fail(c, KM_E_ARG, "shift_image: null output");
    return kd_shift_image(c, d_img, elem_size, H, W, stride, y_off, x_off, d_out);
}

int km_shift_image(km_ctx *c, const void *img, int elem_size, int H, int W, ptrdiff_t stride, int y_off, int x_off, void *out)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, img, H, W, stride, "shift_image"))) return rc;
    if (!out || (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8)) return km_fail(c, KM_E_ARG, "shift_image: elem_size %d", elem_size);
    void *d_img;
    if ((rc = upload_image(c, WS_RAW_A, img, (size_t)elem_size, H, W, stride, &d_img))) return rc;
    void *d_out = km_ws(c, WS_RAW_B, (size_t)H * W * elem_size);
    if (!d_out) return KM_E_NOMEM;
    if ((rc = kd_shift_image(c, d_img, elem_size, H, W, W, y_off, x_off, d_out))) return rc;
    KM_D2H(c, out, d_out, (size_t)H * W * elem_size);
    KM_FLUSH(c);
    return KM_OK;
}

}  // extern "C"
