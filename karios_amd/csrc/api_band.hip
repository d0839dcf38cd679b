#include "api_internal.hpp"

#include <cstring>
#include <vector>

// ================================================================== SURVEY 8(f)-3: ONE tile matched by several GPUs, exactly
// The reference's default configuration is a single 10980^2 tile (tile_size 20000): its uint8 stretch uses the tile's global
// min / max, goodFeaturesToTrack its global maximum eigenvalue and ONE ranked greedy selection with one maxCorners cut
// (klt.py:42-49, 120).  Row bands reproduce that across ranks: every rank holds the rows of its band plus a halo, the entry
// points below are the per-rank steps between which karios_amd.parallel.match_tile_banded exchanges two scalars (min / max,
// maximum eigenvalue key), the ranks' strongest candidate keys and, at the end, the tracks.  Everything a rank computes for
// a row of its OWN band is bit-identical to the single-GPU result: the halo keeps the artificial band edges out of reach of
// every stencil, and the trackers work in IMAGE coordinates on a virtual row origin (km_pyr::oy).
extern "C" {

int km_minmax_dev(km_ctx *c, const void *d_img, int dtype, int H, int W, ptrdiff_t stride, double out_minmax[2])
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_img, H, W, stride, "minmax"))) return rc;
    if (!km_dtype_size(dtype) || !out_minmax) return km_fail(c, KM_E_ARG, "minmax: bad dtype %d or null output", dtype);
    km_scalars *sc = scalars(c);
    if (!sc) return KM_E_NOMEM;
    if (dtype == KM_U8) { out_minmax[0] = out_minmax[1] = 0.0; return KM_OK; }   // (_to_uint8 passes uint8 through, klt.py:44-45)
    if ((rc = kd_minmax(c, d_img, dtype, H, W, stride, sc->mm))) return rc;
    KM_D2H(c, out_minmax, sc->mm, 2 * sizeof(double));
    KM_FLUSH(c);
    return KM_OK;
}

// stretch with the GIVEN (global) min / max, Laplacians, mask restricted to the band's own rows [own_y0, own_y1)
int km_band_prefilter_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                          const double minmax[4], const double *nodata_ref, const double *nodata_mon, int ksize_ref, int ksize_mon, int invert_mon,
                          int own_y0, int own_y1, const uint8_t *d_user_mask, uint8_t *d_lap_ref, uint8_t *d_lap_mon, uint8_t *d_mask,
                          int64_t *valid_owned)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_ref, H, W, sref, "band_prefilter")) || (rc = check_image(c, d_mon, H, W, smon, "band_prefilter")))
        return rc;
    if (!km_dtype_size(dtype) || !minmax || !d_lap_ref || !d_lap_mon || !d_mask || !valid_owned) return km_fail(c, KM_E_ARG, "band_prefilter: bad arguments");
    if (own_y0 < 0 || own_y1 > H || own_y0 > own_y1) return km_fail(c, KM_E_ARG, "band_prefilter: own rows [%d, %d) outside 0..%d", own_y0, own_y1, H);
    km_scalars *sc = scalars(c);
    if (!sc) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    { const int rch = h2d_now(c, sc->mm, minmax, 4 * sizeof(double)); if (rch) return rch; }
    if ((rc = kd_stretch_laplacian_pair(c, d_ref, d_mon, dtype, H, W, sref, smon, sc->mm, ksize_ref, ksize_mon, invert_mon, nodata_ref, nodata_mon,
                                        d_lap_ref, d_lap_mon, d_user_mask ? nullptr : d_mask, &sc->valid)))
        return rc;
    if (d_user_mask) KM_HIP(c, hipMemcpyAsync(d_mask, d_user_mask, (size_t)H * W, hipMemcpyDeviceToDevice, c->stream));
    // corners are only sought in the band's own rows; masked pixels still shape their neighbours' eigenvalues (App. A.2 step 5)
    if (own_y0 > 0) KM_HIP(c, hipMemsetAsync(d_mask, 0, (size_t)own_y0 * W, c->stream));
    if (own_y1 < H) KM_HIP(c, hipMemsetAsync(d_mask + (size_t)own_y1 * W, 0, (size_t)(H - own_y1) * W, c->stream));
    KM_HIP(c, hipMemsetAsync(&sc->valid, 0, sizeof sc->valid, c->stream));
    if ((rc = kd_count_nonzero(c, d_mask, (size_t)H * W, &sc->valid))) return rc;
    unsigned long long v = 0;
    KM_D2H(c, &v, &sc->valid, sizeof v);
    KM_FLUSH(c);
    *valid_owned = (int64_t)v;
    return KM_OK;
}

// minimum-eigenvalue pass of the band: candidate keys stay in the context, *local_max_key = ordered-uint key of the band's maximum
// over its own rows (0: no valid pixel).  The keys carry raster indices of the BAND image.
int km_band_eigen_dev(km_ctx *c, const uint8_t *d_lap_ref, const uint8_t *d_mask, int H, int W, int block, double quality, unsigned *local_max_key)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, d_lap_ref, H, W, W, "band_eigen"))) return rc;
    if (!d_mask || !local_max_key || !(quality > 0) || block < 1) return km_fail(c, KM_E_ARG, "band_eigen: bad arguments");
    km_scalars *sc = scalars(c);
    if (!sc) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(&sc->max_eig_key, 0, sizeof(km_scalars) - offsetof(km_scalars, max_eig_key), c->stream));
    c->band_capk = (size_t)H * W / 4 + 4096 * KM_NSHARD;       // generous: the running threshold of a band is only its own maximum's
    unsigned long long *keys = (unsigned long long *)km_ws(c, WS_KEYS0, c->band_capk * sizeof(unsigned long long));
    if (!keys) return KM_E_NOMEM;
    rc = c->fused_eig ? k2_eig_candidates(c, d_lap_ref, d_mask, H, W, block, quality, sc, keys, c->band_capk, false) : KM_E_UNSUPPORTED;
    c->band_fused = rc == KM_OK;
    if (rc == KM_E_UNSUPPORTED) {
        float *eig = (float *)km_ws(c, WS_EIG, (size_t)H * W * sizeof(float));
        if (!eig) return KM_E_NOMEM;
        if ((rc = kd_min_eigen(c, d_lap_ref, d_mask, H, W, block, eig, &sc->max_eig_key))) return rc;
    } else if (rc) return rc;
    KM_D2H(c, local_max_key, &sc->max_eig_key, sizeof(unsigned));
    KM_FLUSH(c);
    return KM_OK;
}

// with the GLOBAL maximum: the band's candidates above the exact threshold, strongest first as far as `cap` reaches.
// *n_total = candidates of the band, *n_out = keys delivered (all of them, or at least k_target: the strongest value bins).
int km_band_keys_dev(km_ctx *c, const uint8_t *d_mask, int H, int W, double quality, unsigned global_max_key, size_t k_target,
                     unsigned long long *out_keys, size_t cap, size_t *n_out, size_t *n_total)
{
    int rc;
    if ((rc = begin_call(c))) return rc;
    if (!out_keys || !n_out || !n_total || !d_mask) return km_fail(c, KM_E_ARG, "band_keys: null argument");
    km_scalars *sc = scalars(c);
    unsigned long long *keys = (unsigned long long *)c->ws[WS_KEYS0].p;
    if (!sc || !keys || !c->band_capk) return km_fail(c, KM_E_ARG, "band_keys: km_band_eigen_dev has not run on this context");
    { const int rch = h2d_now(c, &sc->max_eig_key, &global_max_key, sizeof(unsigned)); if (rch) return rch; }
    if (!c->band_fused) {
        const float *eig = (const float *)c->ws[WS_EIG].p;
        if ((rc = kd_candidates(c, eig, d_mask, H, W, quality, sc, keys, c->band_capk, false))) return rc;
    }
    unsigned long long *kept = nullptr;
    size_t nkept = 0, ntotal = 0;
    km_scalars hs;
    if ((rc = ks_topk_prefilter(c, keys, c->band_capk, k_target, sc, quality, &kept, &nkept, &ntotal, &hs, false))) return rc;
    if (c->band_fused && hs.pad0 != 0u) return km_fail(c, KM_E_UNSUPPORTED, "band_keys: plateau image overflowed the fused kernel's stage");
    if ((size_t)hs.n_cand > c->band_capk) return km_fail(c, KM_E_UNSUPPORTED, "band_keys: candidate buffer overflow (%u keys)", hs.n_cand);
    if (nkept > cap) return km_fail(c, KM_E_ARG, "band_keys: %zu keys exceed the capacity %zu", nkept, cap);
    if (nkept) KM_D2H(c, out_keys, kept, nkept * sizeof(unsigned long long));
    KM_FLUSH(c);
    *n_out = nkept; *n_total = ntotal;
    return KM_OK;
}

// the ordering primitives of the exact paths on host buffers (parity tests; callers that need the library's key order)
int km_sort_pairs_u64(km_ctx *c, unsigned long long *keys, unsigned *vals, size_t n, int descending)
{
    int rc;
    if ((rc = begin_call(c))) return rc;
    if (n && !keys) return km_fail(c, KM_E_ARG, "sort_pairs: bad arguments");
    if (n == 0) return KM_OK;
    unsigned long long *dk = (unsigned long long *)km_ws(c, WS_KEYS0, 2 * n * sizeof(unsigned long long));
    unsigned *dv = vals ? (unsigned *)km_ws(c, WS_MISC1, 2 * n * sizeof(unsigned)) : nullptr;
    if (!dk || (vals && !dv)) return KM_E_NOMEM;
    { const int rch = h2d_now(c, dk, keys, n * sizeof(unsigned long long)); if (rch) return rch; }
    if (vals) { const int rch = h2d_now(c, dv, vals, n * sizeof(unsigned)); if (rch) return rch; }
    if ((rc = km_sort_u64(c, dk, dk + n, dv, dv ? dv + n : nullptr, n, descending != 0))) return rc;
    KM_D2H(c, keys, dk, n * sizeof(unsigned long long));
    if (vals) KM_D2H(c, vals, dv, n * sizeof(unsigned));
    KM_FLUSH(c);
    return KM_OK;
}

int km_exclusive_scan_u32(km_ctx *c, const unsigned *in, unsigned *out, size_t n, int count_ones)
{
    int rc;
    if ((rc = begin_call(c))) return rc;
    if (n && (!in || !out)) return km_fail(c, KM_E_ARG, "exclusive_scan: bad arguments");
    if (n == 0) return KM_OK;
    unsigned *d = (unsigned *)km_ws(c, WS_MISC1, 2 * n * sizeof(unsigned));
    if (!d) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d, in, n * sizeof(unsigned)); if (rch) return rch; }
    if ((rc = km_exclusive_scan(c, d, d + n, n, count_ones ? KM_SCAN_IS_ONE : KM_SCAN_PLAIN, WS_SORT_TMP))) return rc;
    KM_D2H(c, out, d + n, n * sizeof(unsigned));
    KM_FLUSH(c);
    return KM_OK;
}

// goodFeaturesToTrack steps 6-8 on a GIVEN list of candidate keys (value bits << 32 | raster index of the H x W image), any order
int km_select_keys(km_ctx *c, const unsigned long long *keys, size_t n, int H, int W, int max_corners, double min_distance, float *out_xy, int cap,
                   int *out_n)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT))) return rc;
    if ((n && !keys) || !out_xy || !out_n || cap <= 0 || H <= 0 || W <= 0) return km_fail(c, KM_E_ARG, "select_keys: bad arguments");
    if (max_corners > 0 && cap < max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, max_corners);
    km_scalars *sc = scalars(c);
    unsigned long long *d_keys = (unsigned long long *)km_ws(c, WS_KEYS0, (n + 16) * sizeof(unsigned long long));
    float *d_xy = (float *)km_ws(c, WS_PTS0, (size_t)cap * 2 * sizeof(float));
    if (!sc || !d_keys || !d_xy) return KM_E_NOMEM;
    c->band_capk = 0;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    *out_n = 0;
    if (n == 0) return KM_OK;
    { const int rch = h2d_now(c, d_keys, keys, n * sizeof(unsigned long long)); if (rch) return rch; }
    unsigned long long *sorted = nullptr;
    if ((rc = ks_sort_keys_desc(c, d_keys, n, &sorted))) return rc;
    if ((rc = ks_select(c, sorted, n, H, W, max_corners, min_distance, d_xy, cap, sc, nullptr, true))) return rc;
    if ((rc = read_stats(c, sc))) return rc;
    const int m = c->stats.n_init;
    if (m > cap) return km_fail(c, KM_E_ARG, "select_keys: %d corners exceed capacity %d", m, cap);
    if (m > 0) {
        KM_D2H(c, out_xy, d_xy, (size_t)m * 2 * sizeof(float));
        KM_FLUSH(c);
    }
    *out_n = m;
    return KM_OK;
}

// LK forward + backward of `n` key points given in IMAGE coordinates on a band: rows [oy, oy + H) of an H_image-row image pair are
// resident (Laplacians).  *left_band = 1 when some window needed rows outside the band (its tracks are then undefined).
int km_band_track_dev(km_ctx *c, const uint8_t *d_lap_ref, const uint8_t *d_lap_mon, int H, int W, int oy, int H_image, const km_klt_params *prm,
                      const float *p0, int n, float *p1, float *p0r, int *left_band)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, d_lap_ref, H, W, W, "band_track")) ||
        (rc = check_image(c, d_lap_mon, H, W, W, "band_track")))
        return rc;
    if (n < 0 || (n > 0 && (!p0 || !p1 || !p0r)) || !left_band) return km_fail(c, KM_E_ARG, "band_track: bad point arrays");
    if (oy < 0 || oy + H > H_image) return km_fail(c, KM_E_ARG, "band_track: rows [%d, %d) outside the %d-row image", oy, oy + H, H_image);
    *left_band = 0;
    if (n == 0) return KM_OK;
    // pyramid depth is decided by the IMAGE size (buildOpticalFlowPyramid stops when the next level would be <= winSize)
    int levels = 0, hf[5] = {H_image}, wf[5] = {W};
    for (int l = 0; l < (prm->max_level > 4 ? 4 : prm->max_level); l++) {
        const int nw = (wf[l] + 1) / 2, nh = (hf[l] + 1) / 2;
        if (nw <= prm->win_size || nh <= prm->win_size) break;
        hf[l + 1] = nh; wf[l + 1] = nw; levels = l + 1;
    }
    if (oy % (1 << levels)) return km_fail(c, KM_E_ARG, "band_track: band origin %d must be a multiple of %d", oy, 1 << levels);
    km_pyr A, B;
    A.levels = B.levels = levels;
    size_t total = 0;
    int hr[5] = {H};
    for (int l = 1; l <= levels; l++) { hr[l] = (hr[l - 1] + 1) / 2; total += ((size_t)wf[l] * hr[l] + 255) & ~(size_t)255; }
    uint8_t *sa = total ? (uint8_t *)km_ws(c, WS_PYR_A, total) : nullptr, *sb = total ? (uint8_t *)km_ws(c, WS_PYR_B, total) : nullptr;
    if (total && (!sa || !sb)) return KM_E_NOMEM;
    const uint8_t *ra[5] = {d_lap_ref}, *rb[5] = {d_lap_mon};
    size_t off = 0;
    {
        km_stage_timer t(c, ST_PYRAMID);
        for (int l = 1; l <= levels; l++) {
            if ((rc = kd_pyrdown_u8_pair(c, ra[l - 1], rb[l - 1], hr[l - 1], wf[l - 1], sa + off, sb + off))) return rc;
            ra[l] = sa + off; rb[l] = sb + off;
            off += ((size_t)wf[l] * hr[l] + 255) & ~(size_t)255;
        }
    }
    int trim[5] = {0};   // rows of level l next to an artificial band edge that the 5-tap pyrDown computed from mirrored (not the image's) rows:
    for (int l = 1; l <= levels; l++) trim[l] = (trim[l - 1] + 2 + 1) / 2;   // output row r reads rows 2r-2 .. 2r+2: trim_l = ceil((trim_{l-1} + 2) / 2) = 1, 2, 2, 2
    for (int l = 0; l <= levels; l++) {
        // ... they are declared absent
        const int o = oy >> l, trim_top = oy > 0 ? trim[l] : 0, trim_bot = oy + H < H_image ? trim[l] : 0;
        A.H[l] = B.H[l] = hf[l]; A.W[l] = B.W[l] = wf[l];
        A.oy[l] = B.oy[l] = o + trim_top;
        A.Hres[l] = B.Hres[l] = hr[l] - trim_top - trim_bot;
        if (A.Hres[l] <= 0) return km_fail(c, KM_E_ARG, "band_track: band of %d rows too thin at pyramid level %d", H, l);
        A.img[l] = ra[l] - (ptrdiff_t)o * wf[l];            // virtual row 0 of the level
        B.img[l] = rb[l] - (ptrdiff_t)o * wf[l];
    }
    const size_t pb = (size_t)n * 2 * sizeof(float);
    float *d_p0 = (float *)km_ws(c, WS_PTS0, pb), *d_p1 = (float *)km_ws(c, WS_PTS1, pb), *d_p0r = (float *)km_ws(c, WS_PTS2, pb);
    km_scalars *sc = scalars(c);
    if (!d_p0 || !d_p1 || !d_p0r || !sc) return KM_E_NOMEM;
    { const int rch = h2d_now(c, d_p0, p0, pb); if (rch) return rch; }
    int *d_flag = &sc->n_batches;
    KM_HIP(c, hipMemsetAsync(d_flag, 0, sizeof(int), c->stream));
    {
        km_stage_timer t(c, ST_LK);
        if ((rc = kl_track(c, A, B, d_p0, nullptr, n, prm->win_size, prm->max_count, prm->epsilon, true, d_p1, d_p0r, d_flag))) return rc;
    }
    KM_D2H(c, p1, d_p1, pb);
    KM_D2H(c, p0r, d_p0r, pb);
    KM_D2H(c, left_band, d_flag, sizeof(int));
    KM_FLUSH(c);
    return KM_OK;
}

}  // extern "C"
