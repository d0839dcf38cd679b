// Fine-grained host mirrors: one entry point per third-party routine the reference reaches (cv2.Laplacian, goodFeaturesToTrack,
// calcOpticalFlowPyrLK, `_to_uint8`, the automatic mask), on caller-owned host buffers - what the parity tests compare operator by operator.
#include "api_internal.hpp"

#include <cstring>
#include <vector>

extern "C" {

// ------------------------------------------------------------------ fine-grained host mirrors
int km_to_uint8(km_ctx *c, const void *img, int dtype, int H, int W, ptrdiff_t stride, int invert, uint8_t *out, double *out_minmax)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, img, H, W, stride, "to_uint8"))) return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es || !out) return km_fail(c, KM_E_ARG, "to_uint8: bad dtype %d or null output", dtype);
    void *d_img;
    if ((rc = upload_image(c, WS_RAW_A, img, es, H, W, stride, &d_img))) return rc;
    km_scalars *sc = scalars(c);
    uint8_t *d_out = (uint8_t *)km_ws(c, WS_U8_A, (size_t)H * W);
    if (!sc || !d_out) return KM_E_NOMEM;
    if (dtype != KM_U8) { if ((rc = kd_minmax(c, d_img, dtype, H, W, W, sc->mm))) return rc; }
    else KM_HIP(c, hipMemsetAsync(sc->mm, 0, sizeof sc->mm, c->stream));
    if ((rc = kd_to_uint8(c, d_img, dtype, H, W, W, sc->mm, invert, d_out))) return rc;
    KM_D2H(c, out, d_out, (size_t)H * W);
    double mm[2];
    KM_D2H(c, mm, sc->mm, sizeof mm);
    KM_FLUSH(c);
    if (out_minmax) { out_minmax[0] = mm[0]; out_minmax[1] = mm[1]; }
    return KM_OK;
}

int km_auto_mask(km_ctx *c, const void *mon, const void *ref, int dtype, int H, int W, ptrdiff_t smon, ptrdiff_t sref,
                 const double *nodata_mon, const double *nodata_ref, uint8_t *mask, int64_t *valid)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, mon, H, W, smon, "auto_mask")) || (rc = check_image(c, ref, H, W, sref, "auto_mask")))
        return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es || !mask) return km_fail(c, KM_E_ARG, "auto_mask: bad dtype %d or null output", dtype);
    void *d_mon, *d_ref;
    if ((rc = upload_image(c, WS_RAW_A, ref, es, H, W, sref, &d_ref)) || (rc = upload_image(c, WS_RAW_B, mon, es, H, W, smon, &d_mon))) return rc;
    km_scalars *sc = scalars(c);
    uint8_t *d_mask = (uint8_t *)km_ws(c, WS_MASK, (size_t)H * W);
    if (!sc || !d_mask) return KM_E_NOMEM;
    if ((rc = kd_auto_mask(c, d_mon, d_ref, dtype, H, W, W, W, nodata_mon, nodata_ref, d_mask, &sc->valid))) return rc;
    unsigned long long v = 0;
    KM_D2H(c, mask, d_mask, (size_t)H * W);
    KM_D2H(c, &v, &sc->valid, sizeof v);
    KM_FLUSH(c);
    if (valid) *valid = (int64_t)v;
    return KM_OK;
}

int km_laplacian_u8(km_ctx *c, const uint8_t *src, int H, int W, int ksize, uint8_t *dst)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, src, H, W, W, "laplacian"))) return rc;
    if (!dst) return km_fail(c, KM_E_ARG, "laplacian: null output");
    void *d_src;
    if ((rc = upload_image(c, WS_RAW_A, src, 1, H, W, W, &d_src))) return rc;
    uint8_t *d_dst = (uint8_t *)km_ws(c, WS_U8_A, (size_t)H * W);
    if (!d_dst) return KM_E_NOMEM;
    if ((rc = kd_laplacian_u8(c, (const uint8_t *)d_src, H, W, ksize, d_dst))) return rc;
    KM_D2H(c, dst, d_dst, (size_t)H * W);
    KM_FLUSH(c);
    return KM_OK;
}

int km_min_eigen(km_ctx *c, const uint8_t *src, int H, int W, int block, float *eig)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, src, H, W, W, "min_eigen"))) return rc;
    if (!eig) return km_fail(c, KM_E_ARG, "min_eigen: null output");
    void *d_src;
    if ((rc = upload_image(c, WS_RAW_A, src, 1, H, W, W, &d_src))) return rc;
    km_scalars *sc = scalars(c);
    float *d_eig = (float *)km_ws(c, WS_EIG, (size_t)H * W * sizeof(float));
    if (!sc || !d_eig) return KM_E_NOMEM;
    if ((rc = kd_min_eigen(c, (const uint8_t *)d_src, nullptr, H, W, block, d_eig, &sc->max_eig_key))) return rc;
    KM_D2H(c, eig, d_eig, (size_t)H * W * sizeof(float));
    KM_FLUSH(c);
    return KM_OK;
}

int km_good_features(km_ctx *c, const uint8_t *img, const uint8_t *mask, int H, int W, int max_corners, double quality,
                     double min_distance, int block, float *out_xy, int cap, int *out_n)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_image(c, img, H, W, W, "good_features"))) return rc;
    if (!out_xy || !out_n || cap < 0) return km_fail(c, KM_E_ARG, "good_features: null output");
    if (!(quality > 0)) return km_fail(c, KM_E_ARG, "qualityLevel must be > 0");
    if (min_distance < 0) return km_fail(c, KM_E_ARG, "minDistance must be >= 0");
    if (max_corners > 0 && cap < max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    void *d_img, *d_mask = nullptr;
    if ((rc = upload_image(c, WS_RAW_A, img, 1, H, W, W, &d_img))) return rc;
    if (mask && (rc = upload_image(c, WS_MASK_IN, mask, 1, H, W, W, &d_mask))) return rc;
    km_scalars *sc = scalars(c);
    float *d_xy = (float *)km_ws(c, WS_PTS0, (size_t)(cap > 0 ? cap : 1) * 2 * sizeof(float));
    if (!sc || !d_xy) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    if ((rc = gftt_dev(c, (const uint8_t *)d_img, (const uint8_t *)d_mask, H, W, max_corners, quality, min_distance, block, d_xy, cap, sc)))
        return rc;
    if ((rc = read_stats(c, sc))) return rc;
    int n = c->stats.n_init;
    if (n > cap) return km_fail(c, KM_E_ARG, "good_features: %d corners exceed capacity %d", n, cap);
    if (n > 0) {
        KM_D2H(c, out_xy, d_xy, (size_t)n * 2 * sizeof(float));
        KM_FLUSH(c);
    }
    *out_n = n;
    return KM_OK;
}

int km_pyrdown_u8(km_ctx *c, const uint8_t *src, int H, int W, uint8_t *dst)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, src, H, W, W, "pyrdown"))) return rc;
    if (!dst) return km_fail(c, KM_E_ARG, "pyrdown: null output");
    void *d_src;
    if ((rc = upload_image(c, WS_RAW_A, src, 1, H, W, W, &d_src))) return rc;
    const size_t on = (size_t)((H + 1) / 2) * ((W + 1) / 2);
    uint8_t *d_dst = (uint8_t *)km_ws(c, WS_U8_A, on);
    if (!d_dst) return KM_E_NOMEM;
    if ((rc = kd_pyrdown_u8(c, (const uint8_t *)d_src, H, W, d_dst))) return rc;
    KM_D2H(c, dst, d_dst, on);
    KM_FLUSH(c);
    return KM_OK;
}

int km_pyrlk(km_ctx *c, const uint8_t *prev, const uint8_t *next, int H, int W, const float *pts, int n, int win, int max_level,
             int max_count, double epsilon, float *out_pts)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, prev, H, W, W, "pyrlk")) || (rc = check_image(c, next, H, W, W, "pyrlk"))) return rc;
    if (n < 0 || (n > 0 && (!pts || !out_pts))) return km_fail(c, KM_E_ARG, "pyrlk: bad points");
    if (win <= 2 || max_level < 0) return km_fail(c, KM_E_ARG, "pyrlk: winSize %d / maxLevel %d", win, max_level);
    if (n == 0) return KM_OK;
    void *d_prev, *d_next;
    if ((rc = upload_image(c, WS_U8_A, prev, 1, H, W, W, &d_prev)) || (rc = upload_image(c, WS_U8_B, next, 1, H, W, W, &d_next))) return rc;
    float *d_in = (float *)km_ws(c, WS_PTS0, (size_t)n * 2 * sizeof(float));
    float *d_out = (float *)km_ws(c, WS_PTS1, (size_t)n * 2 * sizeof(float));
    if (!d_in || !d_out) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d_in, pts, (size_t)n * 2 * sizeof(float)); if (rch) return rch; }
    km_pyr A, B;
    if ((rc = build_pyramid_pair(c, (const uint8_t *)d_prev, (const uint8_t *)d_next, H, W, win, max_level, &A, &B))) return rc;
    if ((rc = kl_track(c, A, B, d_in, nullptr, n, win, max_count, epsilon, false, d_out, nullptr))) return rc;
    KM_D2H(c, out_pts, d_out, (size_t)n * 2 * sizeof(float));
    KM_FLUSH(c);
    return KM_OK;
}

// test hook: the oscillation predicate of the LK kernels (k_lk.hip: lk_oscillates) on n host quadruples (ddx, pdx, ddy, pdy)
int km_lk_oscillation_probe(km_ctx *c, const float *quads, int n, uint8_t *out)
{
    int rc;
    if ((rc = begin_call(c))) return rc;
    if (n < 0 || (n > 0 && (!quads || !out))) return km_fail(c, KM_E_ARG, "lk_oscillation_probe: bad arguments");
    if (n == 0) return KM_OK;
    float *d_q = (float *)km_ws(c, WS_MISC0, (size_t)n * 4 * sizeof(float));
    uint8_t *d_o = (uint8_t *)km_ws(c, WS_MISC2, (size_t)n);
    if (!d_q || !d_o) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d_q, quads, (size_t)n * 4 * sizeof(float)); if (rch) return rch; }
    if ((rc = kl_oscillation_probe(c, d_q, n, d_o))) return rc;
    KM_D2H(c, out, d_o, (size_t)n);
    KM_FLUSH(c);
    return KM_OK;
}

}  // extern "C"
