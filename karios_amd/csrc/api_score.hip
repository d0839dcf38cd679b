// Scoring entry points: ZNCCService.compute_zncc / _zncc2 (zncc_service.py:45-238), the mutual-information scores
// (mutual_info_service.py:73-130, zncc_service.py:240-287) and the DN-value filter of the key points (core.py:650-737).
#include "api_internal.hpp"

#include <cstring>
#include <vector>

extern "C" {

int km_zncc_batch_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
                      ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, double *d_out)
{
    int rc;
    if ((rc = begin_call(c, RESET_ZNCC)) || (rc = check_image(c, d_ref, Href, Wref, sref, "zncc")) || (rc = check_image(c, d_mon, Hmon, Wmon, smon, "zncc")))
        return rc;
    if (n < 0 || (n > 0 && (!d_x0 || !d_y0 || !d_dx || !d_dy || !d_out))) return km_fail(c, KM_E_ARG, "zncc: bad keypoint arrays");
    km_stage_timer t(c, ST_ZNCC);
    return kz_zncc(c, d_ref, d_mon, dtype, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_out);
}

int km_zncc_batch(km_ctx *c, const void *ref, const void *mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
                  ptrdiff_t smon, const float *x0, const float *y0, const float *dx, const float *dy, int n, double *out)
{
    int rc;
    if ((rc = begin_call(c, RESET_ZNCC)) || (rc = check_image(c, ref, Href, Wref, sref, "zncc")) || (rc = check_image(c, mon, Hmon, Wmon, smon, "zncc")))
        return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es) return km_fail(c, KM_E_ARG, "zncc: bad dtype %d", dtype);
    if (n < 0 || (n > 0 && (!x0 || !y0 || !dx || !dy || !out))) return km_fail(c, KM_E_ARG, "zncc: bad keypoint arrays");
    if (n == 0) return KM_OK;
    void *d_ref, *d_mon;
    if ((rc = upload_image(c, WS_RAW_A, ref, es, Href, Wref, sref, &d_ref)) || (rc = upload_image(c, WS_RAW_B, mon, es, Hmon, Wmon, smon, &d_mon)))
        return rc;
    float *kp = (float *)km_ws(c, WS_MISC0, (size_t)n * 4 * sizeof(float));
    double *d_out = (double *)km_ws(c, WS_MISC1, (size_t)n * sizeof(double));
    if (!kp || !d_out) return KM_E_NOMEM;
    const float *src[4] = {x0, y0, dx, dy};
    for (int i = 0; i < 4; i++) { const int rch = h2d_now(c, kp + (size_t)i * n, src[i], (size_t)n * sizeof(float)); if (rch) return rch; }
    {
        km_stage_timer t(c, ST_ZNCC);
        if ((rc = kz_zncc(c, d_ref, d_mon, dtype, Href, Wref, Hmon, Wmon, Wref, Wmon, kp, kp + n, kp + 2 * (size_t)n, kp + 3 * (size_t)n, n, d_out)))
            return rc;
    }
    KM_D2H(c, out, d_out, (size_t)n * sizeof(double));
    KM_FLUSH(c);
    return KM_OK;
}

int km_zncc_windows(km_ctx *c, const void *img1, const void *img2, int dtype1, int dtype2, int H1, int W1, int H2, int W2, ptrdiff_t stride1,
                    ptrdiff_t stride2, const int32_t *uv, int half_size, int count, double *out, uint8_t *out_outside)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, img1, H1, W1, stride1, "zncc_windows")) || (rc = check_image(c, img2, H2, W2, stride2, "zncc_windows")))
        return rc;
    const size_t e1 = km_any_dtype_size(dtype1), e2 = km_any_dtype_size(dtype2);
    if (!e1 || !e2) return km_fail(c, KM_E_ARG, "zncc_windows: bad dtypes %d / %d", dtype1, dtype2);
    if (half_size < 0) return km_fail(c, KM_E_ARG, "zncc_windows: window half-size must be non-negative");
    if (count < 0 || (count > 0 && (!uv || !out))) return km_fail(c, KM_E_ARG, "zncc_windows: bad window arrays");
    if (count == 0) return KM_OK;
    void *d1, *d2;
    if ((rc = upload_image(c, WS_RAW_A, img1, e1, H1, W1, stride1, &d1)) || (rc = upload_image(c, WS_RAW_B, img2, e2, H2, W2, stride2, &d2))) return rc;
    int *d_uv = (int *)km_ws(c, WS_MISC0, (size_t)count * 4 * sizeof(int));
    double *d_out = (double *)km_ws(c, WS_MISC1, (size_t)count * sizeof(double));
    uint8_t *d_fl = (uint8_t *)km_ws(c, WS_MISC2, (size_t)count);
    if (!d_uv || !d_out || !d_fl) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d_uv, uv, (size_t)count * 4 * sizeof(int)); if (rch) return rch; }
    if ((rc = kz_zncc_windows(c, d1, d2, dtype1, dtype2, H1, W1, H2, W2, W1, W2, d_uv, half_size, count, d_out, d_fl))) return rc;
    KM_D2H(c, out, d_out, (size_t)count * sizeof(double));
    if (out_outside) KM_D2H(c, out_outside, d_fl, (size_t)count);
    KM_FLUSH(c);
    return KM_OK;
}

int km_mi_batch_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
                    ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, double *d_st,
                    double *d_nmi)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_ref, Href, Wref, sref, "mi")) || (rc = check_image(c, d_mon, Hmon, Wmon, smon, "mi")))
        return rc;
    if (n < 0 || (n > 0 && (!d_x0 || !d_y0 || !d_dx || !d_dy || (!d_st && !d_nmi)))) return km_fail(c, KM_E_ARG, "mi: bad keypoint arrays");
    c->evs_used[c->ev_cur][ST_MI] = false;
    km_stage_timer t(c, ST_MI);
    return kmi_batch(c, d_ref, d_mon, dtype, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, nullptr, nullptr, 0.f, d_st, d_nmi);
}

int km_mi_batch(km_ctx *c, const void *ref, const void *mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
                ptrdiff_t smon, const float *x0, const float *y0, const float *dx, const float *dy, int n, double *out_st, double *out_nmi)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, ref, Href, Wref, sref, "mi")) || (rc = check_image(c, mon, Hmon, Wmon, smon, "mi")))
        return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es) return km_fail(c, KM_E_ARG, "mi: bad dtype %d", dtype);
    if (n < 0 || (n > 0 && (!x0 || !y0 || !dx || !dy || (!out_st && !out_nmi)))) return km_fail(c, KM_E_ARG, "mi: bad keypoint arrays");
    if (n == 0) return KM_OK;
    void *d_ref, *d_mon;
    if ((rc = upload_image(c, WS_RAW_A, ref, es, Href, Wref, sref, &d_ref)) || (rc = upload_image(c, WS_RAW_B, mon, es, Hmon, Wmon, smon, &d_mon)))
        return rc;
    float *kp = (float *)km_ws(c, WS_MISC0, (size_t)n * 4 * sizeof(float));
    double *d_out = (double *)km_ws(c, WS_MISC1, (size_t)n * 2 * sizeof(double));
    if (!kp || !d_out) return KM_E_NOMEM;
    const float *src[4] = {x0, y0, dx, dy};
    for (int i = 0; i < 4; i++) { const int rch = h2d_now(c, kp + (size_t)i * n, src[i], (size_t)n * sizeof(float)); if (rch) return rch; }
    if ((rc = kmi_batch(c, d_ref, d_mon, dtype, Href, Wref, Hmon, Wmon, Wref, Wmon, kp, kp + n, kp + 2 * (size_t)n, kp + 3 * (size_t)n, n, nullptr,
                        nullptr, 0.f, out_st ? d_out : nullptr, out_nmi ? d_out + n : nullptr)))
        return rc;
    if (out_st) KM_D2H(c, out_st, d_out, (size_t)n * sizeof(double));
    if (out_nmi) KM_D2H(c, out_nmi, d_out + n, (size_t)n * sizeof(double));
    KM_FLUSH(c);
    return KM_OK;
}

// KariosAPI._filter_by_dn_values (core.py:650-737) on resident images: x0 / y0 / no_values / keep are HOST arrays
// (n key points, n_no values), the images stay on the device.  keep[i] = 1 keep, 0 drop; a key point outside the image
// is an error (numpy's fancy indexing would raise or wrap).
int km_dn_keep_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon, const float *x0,
                   const float *y0, int n, const double *no_values, int n_no, const double *nodata_ref, const double *nodata_mon, uint8_t *keep)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_ref, H, W, sref, "dn_keep")) || (rc = check_image(c, d_mon, H, W, smon, "dn_keep"))) return rc;
    if (!km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "dn_keep: bad dtype %d", dtype);
    if (n < 0 || n_no < 0 || (n > 0 && (!x0 || !y0 || !keep)) || (n_no > 0 && !no_values)) return km_fail(c, KM_E_ARG, "dn_keep: bad arrays");
    if (n == 0) return KM_OK;
    float *d_xy = (float *)km_ws(c, WS_MISC0, (size_t)n * 2 * sizeof(float));
    double *d_nv = (double *)km_ws(c, WS_MISC1, (size_t)(n_no > 0 ? n_no : 1) * sizeof(double));
    uint8_t *d_keep = (uint8_t *)km_ws(c, WS_MISC2, (size_t)n);
    if (!d_xy || !d_nv || !d_keep) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d_xy, x0, (size_t)n * sizeof(float)); if (rch) return rch; }
    { const int rch = h2d_now(c, d_xy + n, y0, (size_t)n * sizeof(float)); if (rch) return rch; }
    if (n_no > 0) { const int rch = h2d_now(c, d_nv, no_values, (size_t)n_no * sizeof(double)); if (rch) return rch; }
    if ((rc = kf_dn_keep(c, d_ref, d_mon, dtype, H, W, sref, smon, d_xy, d_xy + n, n, d_nv, n_no, nodata_ref, nodata_mon, d_keep))) return rc;
    KM_D2H(c, keep, d_keep, (size_t)n);
    KM_FLUSH(c);
    for (int i = 0; i < n; i++)
        if (keep[i] > 1) return km_fail(c, KM_E_ARG, "dn_keep: key point %d (%g, %g) lies outside the %dx%d image", i, (double)x0[i], (double)y0[i], W, H);
    return KM_OK;
}

}  // extern "C"
