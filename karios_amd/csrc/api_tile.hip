// Tile entry points of libkarios_hip.so: the device-resident KLT tile pipeline (KLT._match_tile, klt.py:236-349: stretch -> Laplacians -> mask
// -> goodFeaturesToTrack -> LK forward / backward), the frame forms (FB test, score, (x0, y0) order and the score columns in the same
// device call), the asynchronous submission and the batched kernel-size search.  Batched units: api_units.hip.
#include "api_internal.hpp"
#include <time.h>
#include <sys/prctl.h>

#include <cstring>

// pyramid of one image into caller-provided storage (levels >= 1 packed from `store`); returns the bytes used
int build_pyramid_single(km_ctx *c, const uint8_t *d_img, int H, int W, int win, int max_level, uint8_t *store, km_pyr *P, size_t *used)
{
    P->img[0] = d_img; P->H[0] = H; P->W[0] = W; P->levels = 0;
    if (max_level > 4) max_level = 4;
    size_t off = 0;
    int w = W, h = H;
    for (int l = 1; l <= max_level; l++) {
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win || nh <= win) break;
        if (store) {
            int rc = kd_pyrdown_u8(c, P->img[l - 1], h, w, store + off);
            if (rc) return rc;
            P->img[l] = store + off;
        }
        P->H[l] = nh; P->W[l] = nw; P->levels = l;
        off += ((size_t)nw * nh + 255) & ~(size_t)255;
        w = nw; h = nh;
    }
    if (used) *used = off;
    return KM_OK;
}

// both pyramids of a pair, one launch per level
int build_pyramid_pair(km_ctx *c, const uint8_t *d_a, const uint8_t *d_b, int H, int W, int win, int max_level, km_pyr *A, km_pyr *B)
{
    A->img[0] = d_a; B->img[0] = d_b;
    A->H[0] = B->H[0] = H; A->W[0] = B->W[0] = W; A->levels = B->levels = 0;
    if (max_level > 4) max_level = 4;
    size_t total = 0;
    int w = W, h = H, nl = 0;
    int hs[5], wsz[5];
    for (int l = 0; l < max_level; l++) {
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win || nh <= win) break;
        nl = l + 1; hs[nl] = nh; wsz[nl] = nw;
        total += ((size_t)nw * nh + 255) & ~(size_t)255;
        w = nw; h = nh;
    }
    if (nl == 0) return KM_OK;
    uint8_t *ba = (uint8_t *)km_ws(c, WS_PYR_A, total), *bb = (uint8_t *)km_ws(c, WS_PYR_B, total);
    if (!ba || !bb) return KM_E_NOMEM;
    size_t off = 0;
    for (int l = 1; l <= nl; l++) {
        int rc = kd_pyrdown_u8_pair(c, A->img[l - 1], B->img[l - 1], A->H[l - 1], A->W[l - 1], ba + off, bb + off);
        if (rc) return rc;
        A->img[l] = ba + off; B->img[l] = bb + off;
        A->H[l] = B->H[l] = hs[l]; A->W[l] = B->W[l] = wsz[l];
        off += ((size_t)wsz[l] * hs[l] + 255) & ~(size_t)255;
    }
    A->levels = B->levels = nl;
    return KM_OK;
}

// goodFeaturesToTrack on a dense device u8 image.  Leaves the corner count in scalars->n_corners
// (device) and the corner list in d_xy.  One host sync (candidate count).
int gftt_dev(km_ctx *c, const uint8_t *d_img, const uint8_t *d_mask, int H, int W, int max_corners, double quality,
                    double min_distance, int block, float *d_xy, int cap, km_scalars *sc)
{
    int rc;
    // Strongest-first shortcut: rank and select on the top slice only (a rank prefix, so a sufficient slice gives the
    // exact result); fall back to the complete list when that slice cannot supply maxCorners corners.
    size_t k_target = (max_corners > 0 && min_distance >= 1) ? (size_t)max_corners * (c->opt_topk_factor > 0 ? c->opt_topk_factor : 8) : 0;
    size_t capk = (size_t)H * W / 8 + 4096 * KM_NSHARD;
    if (c->opt_key_cap > 0) capk = (size_t)c->opt_key_cap * KM_NSHARD;   // test knob: tiny shards, so that the regrow path runs
    unsigned long long *kept = nullptr;
    size_t nkept = 0, ntotal = 0;
    km_scalars hs;
    bool fused_overflow = false;   // a row group overflowed the fused kernel's candidate stage (plateau image): use the two-kernel path
    for (int attempt = 0; attempt < 4; attempt++) {
        unsigned long long *keys = (unsigned long long *)km_ws(c, WS_KEYS0, capk * sizeof(unsigned long long));
        if (!keys) return KM_E_NOMEM;
        // K3 + K4 fused (2 pixels per lane, no eig map: k_eig2.hip) when it covers the case, else eig map + candidate kernel.
        // km_set_option("fused_eig", 0) selects the two-kernel path.
        bool fused = false;
        if (c->fused_eig && !fused_overflow) {
            km_stage_timer t(c, ST_EIGEN);
            rc = k2_eig_candidates(c, d_img, d_mask, H, W, block, quality, sc, keys, capk, attempt > 0);
            if (rc == KM_OK) fused = true;
            else if (rc != KM_E_UNSUPPORTED) return rc;
        }
        if (!fused) {
            float *eig = (float *)km_ws(c, WS_EIG, (size_t)H * W * sizeof(float));
            if (!eig) return KM_E_NOMEM;
            {
                km_stage_timer t(c, ST_EIGEN);
                if ((rc = kd_min_eigen(c, d_img, d_mask, H, W, block, eig, &sc->max_eig_key))) return rc;
            }
            {
                km_stage_timer t(c, ST_CANDIDATES);
                if ((rc = kd_candidates(c, eig, d_mask, H, W, quality, sc, keys, capk, attempt > 0))) return rc;
            }
        }
        {
            km_stage_timer t(c, ST_SORT);
            if ((rc = ks_topk_prefilter(c, keys, capk, k_target, sc, quality, &kept, &nkept, &ntotal, &hs, attempt > 0))) return rc;
        }
        c->stats.valid_pixels = (int64_t)hs.valid;
        c->stats.max_eig = hs.max_eig;
        c->stats.min_ref = hs.mm[0]; c->stats.max_ref = hs.mm[1]; c->stats.min_mon = hs.mm[2]; c->stats.max_mon = hs.mm[3];
        if (fused && hs.pad0 != 0u) {   // candidates were dropped: repeat with the eig-map + candidate kernels
            fused_overflow = true;
            c->stats.path_flags |= KM_PATH_STAGE_FALLBACK;
            KM_HIP(c, hipMemsetAsync(&sc->run_max_key, 0, (2 + KM_NSHARD) * sizeof(unsigned), c->stream));
            continue;
        }
        if ((size_t)hs.n_cand <= capk) break;
        capk = (size_t)hs.n_cand + hs.n_cand / 4 + 4096 * KM_NSHARD;   // a shard overflowed: grow the key buffer and redo
        c->stats.path_flags |= KM_PATH_KEY_REGROW;
        if (attempt == 3) return km_fail(c, KM_E_INTERNAL, "candidate buffer kept overflowing");
    }
    c->stats.n_candidates = (int64_t)ntotal;
    c->stats.emitted_ratio = ntotal ? (float)((double)hs.n_cand / (double)ntotal) : 0.f;
    unsigned long long *keys = (unsigned long long *)c->ws[WS_KEYS0].p;
    for (int pass = 0; pass < 2; pass++) {
        unsigned long long *sorted = kept;
        if (nkept > 0) {
            km_stage_timer t(c, ST_SORT);
            if ((rc = ks_sort_keys_desc(c, kept, nkept, &sorted))) return rc;
        }
        int found = -1;
        {
            km_stage_timer t(c, ST_SELECT);
            if ((rc = ks_select(c, sorted, nkept, H, W, max_corners, min_distance, d_xy, cap, sc, nkept < ntotal ? &found : nullptr, pass == 0))) return rc;
        }
        if (nkept >= ntotal || found >= max_corners) break;
        // the top slice did not contain maxCorners mutually distant corners: repeat on every candidate
        k_target = 0;
        c->stats.path_flags |= KM_PATH_SECOND_PASS;
        km_scalars hs2;
        if ((rc = ks_topk_prefilter(c, keys, capk, 0, sc, quality, &kept, &nkept, &ntotal, &hs2, true))) return rc;
    }
    return KM_OK;
}

int read_stats(km_ctx *c, km_scalars *sc)
{
    km_scalars h;
    KM_D2H(c, &h, sc, sizeof h);
    KM_FLUSH(c);
    c->stats.n_init = h.n_corners;
    c->stats.n_select_batches = h.n_batches;
    c->stats.max_eig = h.max_eig;
    c->stats.min_ref = h.mm[0]; c->stats.max_ref = h.mm[1]; c->stats.min_mon = h.mm[2]; c->stats.max_mon = h.mm[3];
    if (c->spec_used) {   // the speculative corner path read nothing back on the way: its diagnostics arrive here
        c->stats.valid_pixels = (int64_t)h.valid;
        c->stats.n_candidates = (int64_t)h.cut[3];
        c->spec_flags = h.flags;
    }
    c->stats.tie_rows = (int32_t)h.tie_rows;
    if (h.n_cand == 0xffffffffu) return km_fail(c, KM_E_INTERNAL, "corner grid cell overflow");
    return KM_OK;
}

// klt_tracker numeric core on dense device u8 images (klt.py:103-142)
int klt_track_dev(km_ctx *c, const uint8_t *d_ref_lap, const uint8_t *d_mon_lap, const uint8_t *d_mask, int H, int W,
                         const km_klt_params *prm, const float *d_p0_in, int n_p0, float *d_p0, float *d_p1, float *d_p0r, int cap,
                         km_scalars *sc)
{
    int rc;
    if (d_p0_in) {
        if (n_p0 > cap) return km_fail(c, KM_E_ARG, "p0 count %d exceeds capacity %d", n_p0, cap);
        if (n_p0 > 0 && d_p0_in != d_p0)
            KM_HIP(c, hipMemcpyAsync(d_p0, d_p0_in, (size_t)n_p0 * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        { const int rch = h2d_now(c, &sc->n_corners, &n_p0, sizeof(int)); if (rch) return rch; }   // (n_p0 is a stack variable: staged)
    }
    km_pyr A, B;
    // Speculative corner path (k_select2.hip): no host synchronisation, fixed capacities, flags instead of retries.  The
    // caller reads sc->flags with the tile's result and repeats a flagged tile with c->spec_allowed = false.
    bool spec = !d_p0_in && c->spec_allowed && c->opt_speculative && c->fused_eig && prm->max_corners > 0 && prm->min_distance >= 1 &&
                !c->opt_key_cap && !c->opt_stage_cap && !c->opt_topk_factor && !c->opt_select_first;
    if (!spec && (rc = kd_run_valid_sum(c))) return rc;  // (... and the valid-pixel sum)
    if (spec) {
        const size_t capk = (size_t)H * W / 8 + 4096 * KM_NSHARD;
        unsigned long long *keys = (unsigned long long *)km_ws(c, WS_KEYS0, capk * sizeof(unsigned long long));
        if (!keys) return KM_E_NOMEM;
        // The pyramids depend on the Laplacians only: they run on a second stream, joined before LK.  Forked BEFORE the fused
        // eigenvalue pass ("aux_early", default): that kernel is bound by instruction issue at 3 waves per SIMD and leaves the
        // memory system idle, and the ranking / selection chain behind it (small latency-bound kernels) then has the GPU to itself;
        // forked behind it (round 2) the pyramids stretched the chain's one-workgroup kernels from 8 to 36 us.
        bool forked = false;
        auto fork_pyramids = [&]() -> int {
            if (!c->aux_stream) {
                // lowest priority: when a kernel of the main stream and a pyramid kernel become ready together (both wait for the
                // Laplacians), the main stream's takes the compute units first and the pyramids fill what it leaves
                int prio_lo = 0, prio_hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
                KM_HIP(c, hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, c->opt_aux_priority ? prio_lo : 0));
                KM_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                KM_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
            KM_HIP(c, hipEventRecord(c->ev_fork, c->stream));
            KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
            hipStream_t main_stream = c->stream;
            c->stream = c->aux_stream;
            int r = kd_run_valid_sum(c);                     // (the Laplacian pass's deferred valid-pixel sum: off the main stream)
            if (r == KM_OK) {
                km_stage_timer t(c, ST_PYRAMID);
                r = build_pyramid_pair(c, d_ref_lap, d_mon_lap, H, W, prm->win_size, prm->max_level, &A, &B);
            }
            if (r == KM_OK && hipEventRecord(c->ev_join, c->aux_stream) != hipSuccess) r = km_fail(c, KM_E_HIP, "hipEventRecord(join)");
            c->stream = main_stream;
            forked = r == KM_OK;
            return r;
        };
        {
            // (the stage's start event sits in front of the fork: recorded between the fork and the kernel it would let the pyramid
            // kernels take the compute units first, and the bracketed kernel would measure 0.48 instead of 0.33 ms)
            km_stage_timer t(c, ST_EIGEN);
            if (c->opt_aux_pyramid && c->opt_aux_early && (rc = fork_pyramids())) return rc;
            c->eig_defer_max = true; c->eig_partial = nullptr; c->eig_npartial = 0;   // the ranking's first launch reduces the per-wave maxima itself
            rc = k2_eig_candidates(c, d_ref_lap, d_mask, H, W, prm->block_size, prm->quality_level, sc, keys, capk, false);
            c->eig_defer_max = false;
        }
        if (rc == KM_E_UNSUPPORTED) {
            spec = false;
            if (forked) { KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0)); forked = false; }   // (the general path builds its own pyramids in the same buffers)
        } else if (rc) return rc;
        else {
            if (c->opt_aux_pyramid && !forked && (rc = fork_pyramids())) return rc;
            {
                km_stage_timer t(c, ST_SORT);
                rc = kf_rank(c, keys, capk, H, W, prm->max_corners, prm->quality_level, prm->min_distance, sc);
            }
            if (rc == KM_OK) {
                km_stage_timer t(c, ST_SELECT);
                rc = kf_select(c, H, W, prm->max_corners, prm->min_distance, d_p0, cap, sc);
            }
            if (forked) KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));   // (whatever happens next reuses the pyramid buffers)
            if (rc == KM_E_UNSUPPORTED) {   // (grid too large for the fixed-slot cells: nothing irreversible was enqueued)
                spec = false;
                KM_HIP(c, hipMemsetAsync(&sc->max_eig_key, 0, sizeof(km_scalars) - offsetof(km_scalars, max_eig_key), c->stream));
            } else if (rc) return rc;
        }
        if (spec) {
            c->spec_used = true;
            if (!forked) {
                km_stage_timer t(c, ST_PYRAMID);
                if ((rc = build_pyramid_pair(c, d_ref_lap, d_mon_lap, H, W, prm->win_size, prm->max_level, &A, &B))) return rc;
            }
        }
    }
    if ((rc = kd_run_valid_sum(c))) return rc;           // (a path that never forked the second stream)
    if (spec) {
        // corners, their count and the pyramids are enqueued
    } else if (d_p0_in) {
        km_stage_timer t(c, ST_PYRAMID);
        if ((rc = build_pyramid_pair(c, d_ref_lap, d_mon_lap, H, W, prm->win_size, prm->max_level, &A, &B))) return rc;
    } else {
        // the pyramids do not depend on the corners: they are queued as deferred jobs and fill the GPU during the two
        // host read-backs of the corner selection (km_wait_readback); whatever is left runs right after it
        c->deferred.clear();
        size_t pyr_bytes = 0;
        build_pyramid_single(c, d_ref_lap, H, W, prm->win_size, prm->max_level, nullptr, &A, &pyr_bytes);   // sizes only
        uint8_t *store_a = pyr_bytes ? (uint8_t *)km_ws(c, WS_PYR_A, pyr_bytes) : nullptr, *store_b = pyr_bytes ? (uint8_t *)km_ws(c, WS_PYR_B, pyr_bytes) : nullptr;
        if (pyr_bytes && (!store_a || !store_b)) return KM_E_NOMEM;
        B = A; B.img[0] = d_mon_lap;
        if (pyr_bytes) {                       // one job per image: one for each of the two read-backs
            c->deferred.push_back([=, &A]() -> int {
                km_stage_timer t(c, ST_PYRAMID);
                return build_pyramid_single(c, d_ref_lap, H, W, prm->win_size, prm->max_level, store_a, &A, nullptr);
            });
            c->deferred.push_back([=, &B]() -> int { return build_pyramid_single(c, d_mon_lap, H, W, prm->win_size, prm->max_level, store_b, &B, nullptr); });
        }
        rc = gftt_dev(c, d_ref_lap, d_mask, H, W, prm->max_corners, prm->quality_level, prm->min_distance, prm->block_size, d_p0, cap, sc);
        const int rc2 = rc ? (c->deferred.clear(), rc) : km_run_deferred(c);
        if (rc2) return rc2;
    }
    const int n_max = d_p0_in ? n_p0 : (prm->max_corners > 0 && prm->max_corners < cap ? prm->max_corners : cap);
    {
        km_stage_timer t(c, ST_LK);
        if (spec && !c->lk_start_valid) {
            // (the next unit's early min / max starts here: beside LK - in front of the selection sweeps, the ranking or behind LK it
            // measured slower, CHANGELOG.md round 4)
            if (!c->ev_lk_start) KM_HIP(c, hipEventCreateWithFlags(&c->ev_lk_start, hipEventDisableTiming));
            KM_HIP(c, hipEventRecord(c->ev_lk_start, c->stream));
            c->lk_start_valid = true;
        }
        if ((rc = kl_track(c, A, B, d_p0, &sc->n_corners, n_max, prm->win_size, prm->max_count, prm->epsilon, true, d_p1, d_p0r)))
            return rc;
    }
    return KM_OK;
}

int check_params(km_ctx *c, const km_klt_params *p)
{
    if (!p) return km_fail(c, KM_E_ARG, "null params");
    if (p->block_size < 1) return km_fail(c, KM_E_ARG, "blockSize %d < 1", p->block_size);
    if (p->win_size <= 2) return km_fail(c, KM_E_ARG, "winSize %d must be > 2", p->win_size);
    if (p->max_level < 0) return km_fail(c, KM_E_ARG, "maxLevel %d < 0", p->max_level);
    if (!(p->quality_level > 0)) return km_fail(c, KM_E_ARG, "qualityLevel must be > 0");
    if (p->min_distance < 0) return km_fail(c, KM_E_ARG, "minDistance must be >= 0");
    return KM_OK;
}

int klt_tile_dev_impl(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                             const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm,
                             float *d_p0, float *d_p1, float *d_p0r, int cap, km_scalars *sc, bool *no_valid)
{
    int rc;
    const size_t n = (size_t)H * W;
    uint8_t *lap_ref = (uint8_t *)km_ws(c, WS_U8_A, n), *lap_mon = (uint8_t *)km_ws(c, WS_U8_B, n);
    if (!lap_ref || !lap_mon) return KM_E_NOMEM;
    uint8_t *mask_auto = nullptr;
    if (!d_mask) { mask_auto = (uint8_t *)km_ws(c, WS_MASK, n); if (!mask_auto) return KM_E_NOMEM; }
    else if (smask != W) {
        // box of a larger resident mask: the kernels index masks densely, so pack the box first (1 B/px copy)
        if (smask < W) return km_fail(c, KM_E_ARG, "mask stride %td < width %d", smask, W);
        uint8_t *dense = (uint8_t *)km_ws(c, WS_MASK, n);
        if (!dense) return KM_E_NOMEM;
        KM_HIP(c, hipMemcpy2DAsync(dense, (size_t)W, d_mask, (size_t)smask, (size_t)W, (size_t)H, hipMemcpyDeviceToDevice, c->stream));
        d_mask = dense;
    }
    const double *mm = sc->mm;
    if (dtype != KM_U8 && c->mm_early_allowed && c->opt_mm_early && c->lk_start_prev && c->aux_stream) {
        // Early min / max: K1 of THIS unit does not queue behind the tail of the previous one (LK, FB test, ZNCC - instruction-bound
        // kernels of short-lived waves that leave HBM idle) but starts on the second stream the moment the previous unit's LK launch
        // starts, and streams the two rasters beside it.  The previous tile call of this context recorded ev_lk_start; if the GPU is
        // already past it, the kernel simply runs at once.  Result and partials live in slots of their own (the scalar block is
        // zeroed on the main stream at the start of every call, WS_PARTIAL belongs to the kernels of the unit still running).
        double *mm_early = (double *)km_ws(c, WS_MM_EARLY, 4 * sizeof(double));
        if (!mm_early) return KM_E_NOMEM;
        if (!c->ev_mm) KM_HIP(c, hipEventCreateWithFlags(&c->ev_mm, hipEventDisableTiming));
        KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_lk_start, 0));
        hipStream_t main_stream = c->stream;
        c->stream = c->aux_stream;
        {
            km_stage_timer t(c, ST_MINMAX);
            rc = kd_minmax_pair_ws(c, d_ref, d_mon, dtype, H, W, sref, smon, mm_early, WS_MM_PARTIAL);
        }
        if (rc == KM_OK && hipEventRecord(c->ev_mm, c->aux_stream) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipEventRecord(min/max)");
        c->stream = main_stream;
        if (rc) return rc;
        KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_mm, 0));
        mm = mm_early;
        c->stats.path_flags |= KM_PATH_MM_EARLY;
    } else if (dtype != KM_U8) {
        km_stage_timer t(c, ST_MINMAX);
        if ((rc = kd_minmax_pair(c, d_ref, d_mon, dtype, H, W, sref, smon, &sc->mm[0]))) return rc;
    }   // (u8 input: mm stays 0 from the scalar block the entry point zeroed)
    {
        km_stage_timer t(c, ST_LAPLACIAN);
        if (d_mask) { if ((rc = kd_count_nonzero(c, d_mask, n, &sc->valid))) return rc; }
        // on the sync-free path with the pyramids on a second stream the valid-pixel sum goes there too (klt_track_dev: same condition)
        c->defer_valid_sum = c->opt_defer_valid && c->spec_allowed && c->opt_speculative && c->fused_eig && c->opt_aux_pyramid && prm->max_corners > 0 &&
                             prm->min_distance >= 1 && !c->opt_key_cap && !c->opt_stage_cap && !c->opt_topk_factor && !c->opt_select_first;
        rc = kd_stretch_laplacian_pair(c, d_ref, d_mon, dtype, H, W, sref, smon, mm, prm->ksize_ref, prm->ksize_mon,
                                       prm->invert_mon, nodata_ref, nodata_mon, lap_ref, lap_mon, mask_auto, &sc->valid);
        c->defer_valid_sum = false;
        if (rc) return rc;
    }
    // "No valid pixels" (klt.py:276-279) needs no early exit: an all-zero mask gives max-eig 0, no candidate, no corner.
    // The count itself reaches the host with the candidate count (gftt_dev), i.e. without an extra synchronisation.
    *no_valid = false;
    const int rc2 = klt_track_dev(c, lap_ref, lap_mon, d_mask ? d_mask : mask_auto, H, W, prm, nullptr, 0, d_p0, d_p1, d_p0r, cap, sc);
    *no_valid = c->stats.valid_pixels == 0;
    return rc2;
}

extern "C" {

extern "C++" int fetch_tracks(km_ctx *c, km_scalars *sc, const float *d_p0, const float *d_p1, const float *d_p0r, float *p0, float *p1,
                        float *p0r, int cap, int *out_n)
{
    int rc;
    if ((rc = read_stats(c, sc))) return rc;
    int n = c->stats.n_init;
    if (n > cap) return km_fail(c, KM_E_ARG, "%d corners exceed capacity %d", n, cap);
    if (n > 0) {
        const size_t b = (size_t)n * 2 * sizeof(float);
        KM_D2H(c, p0, d_p0, b);
        KM_D2H(c, p1, d_p1, b);
        KM_D2H(c, p0r, d_p0r, b);
        KM_FLUSH(c);
    }
    *out_n = n;
    return KM_OK;
}

int km_klt_track(km_ctx *c, const uint8_t *ref_lap, const uint8_t *mon_lap, const uint8_t *mask, int H, int W, const km_klt_params *prm,
                 const float *p0_in, int n_p0, float *p0, float *p1, float *p0r, int cap, int *out_n)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, ref_lap, H, W, W, "klt_track")) ||
        (rc = check_image(c, mon_lap, H, W, W, "klt_track")))
        return rc;
    if (!p0 || !p1 || !p0r || !out_n || cap <= 0) return km_fail(c, KM_E_ARG, "klt_track: null output");
    if (!p0_in && prm->max_corners > 0 && cap < prm->max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, prm->max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    void *d_ref, *d_mon, *d_mask = nullptr;
    if ((rc = upload_image(c, WS_U8_A, ref_lap, 1, H, W, W, &d_ref)) || (rc = upload_image(c, WS_U8_B, mon_lap, 1, H, W, W, &d_mon))) return rc;
    if (mask && (rc = upload_image(c, WS_MASK_IN, mask, 1, H, W, W, &d_mask))) return rc;
    km_scalars *sc = scalars(c);
    const size_t pb = (size_t)cap * 2 * sizeof(float);
    float *d_p0 = (float *)km_ws(c, WS_PTS0, pb), *d_p1 = (float *)km_ws(c, WS_PTS1, pb), *d_p0r = (float *)km_ws(c, WS_PTS2, pb);
    if (!sc || !d_p0 || !d_p1 || !d_p0r) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    const float *d_p0_in = nullptr;
    if (p0_in) {
        if (n_p0 < 0 || n_p0 > cap) return km_fail(c, KM_E_ARG, "klt_track: p0 count %d (capacity %d)", n_p0, cap);
        if (n_p0 > 0) { const int rch = h2d_now(c, d_p0, p0_in, (size_t)n_p0 * 2 * sizeof(float)); if (rch) return rch; }
        d_p0_in = d_p0;
    }
    if ((rc = klt_track_dev(c, (const uint8_t *)d_ref, (const uint8_t *)d_mon, (const uint8_t *)d_mask, H, W, prm, d_p0_in, n_p0, d_p0, d_p1,
                            d_p0r, cap, sc)))
        return rc;
    return fetch_tracks(c, sc, d_p0, d_p1, d_p0r, p0, p1, p0r, cap, out_n);
}

int km_klt_tile(km_ctx *c, const void *ref, const void *mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon, const uint8_t *mask,
                const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm, float *p0, float *p1, float *p0r, int cap,
                int *out_n)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, ref, H, W, sref, "klt_tile")) ||
        (rc = check_image(c, mon, H, W, smon, "klt_tile")))
        return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es) return km_fail(c, KM_E_ARG, "klt_tile: bad dtype %d", dtype);
    if (!p0 || !p1 || !p0r || !out_n || cap <= 0) return km_fail(c, KM_E_ARG, "klt_tile: null output");
    if (prm->max_corners > 0 && cap < prm->max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, prm->max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    void *d_ref, *d_mon, *d_mask = nullptr;
    if ((rc = upload_image(c, WS_RAW_A, ref, es, H, W, sref, &d_ref)) || (rc = upload_image(c, WS_RAW_B, mon, es, H, W, smon, &d_mon))) return rc;
    if (mask && (rc = upload_image(c, WS_MASK_IN, mask, 1, H, W, W, &d_mask))) return rc;
    km_scalars *sc = scalars(c);
    const size_t pb = (size_t)cap * 2 * sizeof(float);
    float *d_p0 = (float *)km_ws(c, WS_PTS0, pb), *d_p1 = (float *)km_ws(c, WS_PTS1, pb), *d_p0r = (float *)km_ws(c, WS_PTS2, pb);
    if (!sc || !d_p0 || !d_p1 || !d_p0r) return KM_E_NOMEM;
    for (int attempt = 0; attempt < 2; attempt++) {
        KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
        bool no_valid = false;
        c->spec_allowed = attempt == 0; c->spec_used = false; c->spec_flags = 0;
        rc = klt_tile_dev_impl(c, d_ref, d_mon, dtype, H, W, W, W, (const uint8_t *)d_mask, W, nodata_ref, nodata_mon, prm, d_p0, d_p1, d_p0r, cap, sc,
                               &no_valid);
        c->spec_allowed = false;
        if (rc) return rc;
        if ((rc = fetch_tracks(c, sc, d_p0, d_p1, d_p0r, p0, p1, p0r, cap, out_n))) return rc;
        (void)verify_upload(c, "end of km_klt_tile (ref)", WS_RAW_A, ref, es, H, W, sref, d_ref);
        (void)verify_upload(c, "end of km_klt_tile (mon)", WS_RAW_B, mon, es, H, W, smon, d_mon);
        if (!(c->spec_used && c->spec_flags)) break;       // flagged speculative run: once more through the exact path
        memset(&c->stats, 0, sizeof c->stats);
        c->stats.path_flags |= KM_PATH_SPEC_RETRY;
    }
    return KM_OK;
}

// KLT._match_tile pre-filter on host buffers: uint8 stretch + Laplacian of both images and the automatic mask in the
// fused kernel the tile path uses (klt.py:268-273, 407-436).  out_mask may be null (then no mask is derived).
int km_tile_prefilter(km_ctx *c, const void *ref, const void *mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                      const double *nodata_ref, const double *nodata_mon, int ksize_ref, int ksize_mon, int invert_mon, uint8_t *out_lap_ref,
                      uint8_t *out_lap_mon, uint8_t *out_mask, int64_t *out_valid)
{
    int rc;
    if ((rc = begin_call(c)) || (rc = check_image(c, ref, H, W, sref, "tile_prefilter")) || (rc = check_image(c, mon, H, W, smon, "tile_prefilter")))
        return rc;
    const size_t es = km_dtype_size(dtype);
    if (!es) return km_fail(c, KM_E_ARG, "tile_prefilter: bad dtype %d", dtype);
    if (!out_lap_ref || !out_lap_mon) return km_fail(c, KM_E_ARG, "tile_prefilter: null output");
    void *d_ref, *d_mon;
    if ((rc = upload_image(c, WS_RAW_A, ref, es, H, W, sref, &d_ref)) || (rc = upload_image(c, WS_RAW_B, mon, es, H, W, smon, &d_mon))) return rc;
    const size_t n = (size_t)H * W;
    km_scalars *sc = scalars(c);
    uint8_t *lap_ref = (uint8_t *)km_ws(c, WS_U8_A, n), *lap_mon = (uint8_t *)km_ws(c, WS_U8_B, n);
    uint8_t *d_mask = out_mask ? (uint8_t *)km_ws(c, WS_MASK, n) : nullptr;
    if (!sc || !lap_ref || !lap_mon || (out_mask && !d_mask)) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    if (dtype != KM_U8) {
        if ((rc = kd_minmax(c, d_ref, dtype, H, W, W, &sc->mm[0])) || (rc = kd_minmax(c, d_mon, dtype, H, W, W, &sc->mm[2]))) return rc;
    }
    if ((rc = kd_stretch_laplacian_pair(c, d_ref, d_mon, dtype, H, W, W, W, sc->mm, ksize_ref, ksize_mon, invert_mon, nodata_ref, nodata_mon,
                                        lap_ref, lap_mon, d_mask, &sc->valid)))
        return rc;
    unsigned long long valid = 0;
    KM_D2H(c, out_lap_ref, lap_ref, n);
    KM_D2H(c, out_lap_mon, lap_mon, n);
    if (out_mask) {
        KM_D2H(c, out_mask, d_mask, n);
        KM_D2H(c, &valid, &sc->valid, sizeof valid);
    }
    KM_FLUSH(c);
    if (out_valid) *out_valid = out_mask ? (int64_t)valid : -1;
    return KM_OK;
}

int km_klt_tile_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                    const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm, float *d_p0,
                    float *d_p1, float *d_p0r, int cap, int *d_n)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, d_ref, H, W, sref, "klt_tile_dev")) ||
        (rc = check_image(c, d_mon, H, W, smon, "klt_tile_dev")))
        return rc;
    if (!km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "klt_tile_dev: bad dtype %d", dtype);
    if (!d_p0 || !d_p1 || !d_p0r || !d_n || cap <= 0) return km_fail(c, KM_E_ARG, "klt_tile_dev: null output");
    if (prm->max_corners > 0 && cap < prm->max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, prm->max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    km_scalars *sc = scalars(c);
    if (!sc) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    bool no_valid = false;
    if ((rc = klt_tile_dev_impl(c, d_ref, d_mon, dtype, H, W, sref, smon, d_mask, smask, nodata_ref, nodata_mon, prm, d_p0, d_p1, d_p0r, cap, sc,
                                &no_valid)))
        return rc;
    KM_HIP(c, hipMemcpyAsync(d_n, &sc->n_corners, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    return KM_OK;
}

// The previous submitted frame's block may still be on its way to the host (on the d2h stream): WS_FRAME may be rewritten once it
// has left - a device-side wait that never stalls in practice (the copy takes 13 us, the next frame is written ~1 ms later).
extern "C++" int frame_block_free(km_ctx *c)
{
    if (c->frame_copy) {
        KM_HIP(c, hipStreamWaitEvent(c->stream, c->frame_copy, 0));
        c->frame_copy = nullptr;
    }
    return KM_OK;
}

static int tile_frame_impl(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                           const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm, float x_off,
                           float y_off, const void *d_ref_full, const void *d_mon_full, int Hf, int Wf, ptrdiff_t sref_f, ptrdiff_t smon_f,
                           bool with_zncc, double zncc_threshold, void *host_out, int cap, km_frame_slot *slot = nullptr)
{
    // slot != nullptr: km_klt_tile_frame_submit - the block goes to the slot's pinned buffer and the call returns without
    // waiting for the tail of the pipeline (LK, FB test, ZNCC, copy), which then overlaps the caller's next submission
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, d_ref, H, W, sref, "klt_tile_frame_dev")) ||
        (rc = check_image(c, d_mon, H, W, smon, "klt_tile_frame_dev")))
        return rc;
    if (with_zncc && ((rc = check_image(c, d_ref_full, Hf, Wf, sref_f, "klt_tile_frame_zncc_dev")) ||
                      (rc = check_image(c, d_mon_full, Hf, Wf, smon_f, "klt_tile_frame_zncc_dev"))))
        return rc;
    if (!km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "klt_tile_frame_dev: bad dtype %d", dtype);
    // the frame's (x0, y0) ordering buckets the rows by tile column (k_frame.hip: x0 - x_off < 65536); wider tiles are refused, not mis-ordered
    if (W > 65535) return km_fail(c, KM_E_ARG, "klt_tile_frame_dev: tile of %d columns (the device-side frame ordering holds at most 65535)", W);
    if ((!host_out && !slot) || cap <= 0) return km_fail(c, KM_E_ARG, "klt_tile_frame_dev: null output");
    if (prm->max_corners > 0 && cap < prm->max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, prm->max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    c->evs_used[c->ev_cur][ST_ZNCC] = false; c->evs_used[c->ev_cur][ST_MI] = false;
    km_scalars *sc = scalars(c);
    const size_t pb = (size_t)cap * 2 * sizeof(float);
    // block: header | x0 | y0 | dx | dy | score | index bits (float32) | zncc [| mutual_info_score | mi_score] (float64)
    const bool with_mi = with_zncc && c->opt_frame_mi;
    const size_t fb = 16 + (size_t)cap * 6 * sizeof(float), ob = fb + (with_zncc ? (size_t)cap * sizeof(double) : 0) + (with_mi ? (size_t)cap * 2 * sizeof(double) : 0);
    float *d_p0 = (float *)km_ws(c, WS_PTS0, pb), *d_p1 = (float *)km_ws(c, WS_PTS1, pb), *d_p0r = (float *)km_ws(c, WS_PTS2, pb);
    char *d_out = (char *)km_ws(c, WS_FRAME, ob);
    if (!sc || !d_p0 || !d_p1 || !d_p0r || !d_out) return KM_E_NOMEM;
    for (int attempt = 0;; attempt++) {
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    bool no_valid = false;
    // corners without a host synchronisation where the case allows it; header word 2 of the frame block carries the flags of
    // that speculative path: the synchronous variants repeat a flagged tile right here, a submitted one is repeated by the
    // caller that waits for it (karios_amd.resident)
    c->spec_allowed = attempt == 0; c->spec_used = false; c->spec_flags = 0;
    c->mm_early_allowed = slot != nullptr;     // (the synchronous forms report min / max in their statistics: scalar block)
    rc = klt_tile_dev_impl(c, d_ref, d_mon, dtype, H, W, sref, smon, d_mask, smask, nodata_ref, nodata_mon, prm, d_p0, d_p1, d_p0r, cap, sc, &no_valid);
    c->spec_allowed = false; c->mm_early_allowed = false;
    if (rc) return rc;
    const int n_max = prm->max_corners > 0 && prm->max_corners < cap ? prm->max_corners : cap;
    if ((rc = frame_block_free(c))) return rc;
    {
        km_stage_timer t(c, ST_FRAME);
        if ((rc = kf_frame(c, d_p0, d_p1, d_p0r, &sc->n_corners, n_max, cap, 0.1f, x_off, y_off, d_out, c->spec_used ? sc : nullptr, W))) return rc;
    }
    if (with_zncc) {
        km_stage_timer t(c, ST_ZNCC);
        const float *f = (const float *)(d_out + 16);
        if ((rc = kz_zncc_filtered(c, d_ref_full, d_mon_full, dtype, Hf, Wf, Hf, Wf, sref_f, smon_f, f, f + cap, f + 2 * (size_t)cap,
                                   f + 3 * (size_t)cap, n_max, (const int *)d_out, f + 4 * (size_t)cap, (float)zncc_threshold,
                                   (double *)(d_out + fb))))
            return rc;
    }
    if (with_mi) {
        // the other two scores of _handle_klt_results (core.py:894-907) for the same rows, behind ZNCC in the same call: the chips of
        // a key point (57 x 57, around the 43 x 43 ZNCC window) are still in the XCD's L2
        km_stage_timer t(c, ST_MI);
        const float *f = (const float *)(d_out + 16);
        double *st = (double *)(d_out + fb) + cap;
        if ((rc = kmi_batch(c, d_ref_full, d_mon_full, dtype, Hf, Wf, Hf, Wf, sref_f, smon_f, f, f + cap, f + 2 * (size_t)cap, f + 3 * (size_t)cap, n_max,
                            (const int *)d_out, f + 4 * (size_t)cap, (float)zncc_threshold, st, st + cap)))
            return rc;
    }
    if (c->frame_sink && c->frame_sink_cap < ob)
        return km_fail(c, KM_E_ARG, "frame sink of %zu bytes is smaller than the %zu-byte frame block", c->frame_sink_cap, ob);
    if (c->frame_sink && !slot) KM_HIP(c, hipMemcpyAsync(c->frame_sink, d_out, ob, hipMemcpyDeviceToDevice, c->stream));
    if (slot) {
        if (slot->cap < ob) {
            if (slot->host) KM_HIP(c, hipHostFree(slot->host));
            slot->host = nullptr; slot->cap = 0;
            KM_HIP(c, hipHostMalloc(&slot->host, ob + ob / 8, hipHostMallocDefault));
            slot->cap = ob + ob / 8;
        }
        if (!slot->done) KM_HIP(c, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
        // the block leaves on a stream of its own: 13 us of DMA that the next submission's first kernels need not wait for
        // (the next frame is written into WS_FRAME ~1 ms later, behind a wait for this copy: see frame_block_free)
        if (!c->d2h_stream) {
            KM_HIP(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
            KM_HIP(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
        }
        KM_HIP(c, hipEventRecord(c->ev_tail, c->stream));
        KM_HIP(c, hipStreamWaitEvent(c->d2h_stream, c->ev_tail, 0));
        slot->sunk_valid = false;
        if (c->frame_sink) {
            // the device-side copy of the block (km_set_frame_sink) leaves there too: on the compute stream it cost the next unit 11 us.
            // km_stream_wait_frame: a stream of the caller (the one an RCCL collective is issued on) can wait for exactly this copy
            KM_HIP(c, hipMemcpyAsync(c->frame_sink, d_out, ob, hipMemcpyDeviceToDevice, c->d2h_stream));
            if (!slot->sunk) KM_HIP(c, hipEventCreateWithFlags(&slot->sunk, hipEventDisableTiming));
            KM_HIP(c, hipEventRecord(slot->sunk, c->d2h_stream));
            slot->sunk_valid = true;
        }
        KM_HIP(c, hipMemcpyAsync(slot->host, d_out, ob, hipMemcpyDeviceToHost, c->d2h_stream));
        KM_HIP(c, hipEventRecord(slot->done, c->d2h_stream));
        c->frame_copy = slot->done;
        slot->bytes = ob;
        return KM_OK;
    }
    km_scalars *land = c->spec_used ? (km_scalars *)km_pinned_rb(c, sizeof(km_scalars)) : nullptr;
    if (land) KM_HIP(c, hipMemcpyAsync(land, sc, sizeof *land, hipMemcpyDeviceToHost, c->stream));   // diagnostics of the sync-free corner path
    KM_D2H(c, host_out, d_out, ob);
    KM_FLUSH(c);
    c->stats.n_init = ((const int *)host_out)[1];
    if (land) {
        c->stats.valid_pixels = (int64_t)land->valid;
        c->stats.n_candidates = (int64_t)land->cut[3];
        c->stats.tie_rows = (int32_t)land->tie_rows;
        c->stats.max_eig = land->max_eig;
        c->stats.min_ref = land->mm[0]; c->stats.max_ref = land->mm[1]; c->stats.min_mon = land->mm[2]; c->stats.max_mon = land->mm[3];
        if (land->flags) {                                   // did not fit the fixed capacities: the exact path decides
            memset(&c->stats, 0, sizeof c->stats);
            c->stats.path_flags |= KM_PATH_SPEC_RETRY;
            continue;
        }
    }
    return KM_OK;
    }
}

// KLT._match_tile_auto_ksize (klt.py:465-545) on resident data: every Laplacian, pyramid and corner list is built ONCE
// and stays on the device; the nk*nk tracker runs reuse them.  Best pair = highest inlier ratio, first wins ties, in
// itertools.product order (mon outer, ref inner).
int km_klt_auto_ksize_frame_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                                const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm,
                                const int *ksizes, int nk, float x_off, float y_off, void *host_out, int cap, double *out_ratios, int *out_best)
{
    int rc;
    if ((rc = begin_call(c, RESET_KLT)) || (rc = check_params(c, prm)) || (rc = check_image(c, d_ref, H, W, sref, "klt_auto_ksize")) ||
        (rc = check_image(c, d_mon, H, W, smon, "klt_auto_ksize")))
        return rc;
    if (!km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "klt_auto_ksize: bad dtype %d", dtype);
    if (W > 65535) return km_fail(c, KM_E_ARG, "klt_auto_ksize: tile of %d columns (the device-side frame ordering holds at most 65535)", W);
    if (!ksizes || nk < 1 || nk > 8 || !host_out || !out_ratios || !out_best || cap <= 0) return km_fail(c, KM_E_ARG, "klt_auto_ksize: bad arguments");
    if (prm->max_corners > 0 && cap < prm->max_corners) return km_fail(c, KM_E_ARG, "capacity %d < maxCorners %d", cap, prm->max_corners);
    memset(&c->stats, 0, sizeof c->stats);
    const size_t n = (size_t)H * W, na = (n + 255) & ~(size_t)255;
    // scalar blocks: [0] the call's own (min / max, valid pixels, the exact corner path), [1 + k] the corner detection of reference kernel k
    const size_t sc_stride = (sizeof(km_scalars) + 255) & ~(size_t)255;
    char *sc_base = (char *)km_ws(c, WS_SCALARS, sc_stride * (size_t)(nk + 1));
    km_scalars *sc = (km_scalars *)sc_base;
    if (!sc) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc_base, 0, sc_stride * (size_t)(nk + 1), c->stream));
    // ---- mask: the user's (packed to the box) - or the automatic one, which the FIRST Laplacian pass below derives from the raw rasters
    const uint8_t *mask = d_mask;
    uint8_t *mask_auto = nullptr;
    if (!d_mask) {
        mask_auto = (uint8_t *)km_ws(c, WS_MASK, n);
        if (!mask_auto) return KM_E_NOMEM;
        mask = mask_auto;
    } else {
        if (smask < W) return km_fail(c, KM_E_ARG, "mask stride %td < width %d", smask, W);
        if (smask != W) {
            uint8_t *dense = (uint8_t *)km_ws(c, WS_MASK, n);
            if (!dense) return KM_E_NOMEM;
            KM_HIP(c, hipMemcpy2DAsync(dense, (size_t)W, d_mask, (size_t)smask, (size_t)W, (size_t)H, hipMemcpyDeviceToDevice, c->stream));
            mask = dense;
        }
        if ((rc = kd_count_nonzero(c, mask, n, &sc->valid))) return rc;
    }
    if (dtype != KM_U8) {
        km_stage_timer t(c, ST_MINMAX);
        if ((rc = kd_minmax_pair(c, d_ref, d_mon, dtype, H, W, sref, smon, &sc->mm[0]))) return rc;
    }
    // ---- arena: 2*nk Laplacians, 2*nk pyramids, nk corner lists, nk*nk track pairs, counters
    km_pyr probe;
    size_t pyr_bytes = 0;
    build_pyramid_single(c, nullptr, H, W, prm->win_size, prm->max_level, nullptr, &probe, &pyr_bytes);
    const size_t pts = ((size_t)cap * 2 * sizeof(float) + 255) & ~(size_t)255;
    const size_t total = (size_t)2 * nk * (na + pyr_bytes) + (size_t)nk * pts + (size_t)2 * nk * nk * pts + 4096;
    uint8_t *arena = (uint8_t *)km_ws(c, WS_AUTO, total);
    if (!arena) return KM_E_NOMEM;
    uint8_t *lap_ref = arena, *lap_mon = lap_ref + (size_t)nk * na, *pyr_store = lap_mon + (size_t)nk * na;
    uint8_t *p0_store = pyr_store + (size_t)2 * nk * pyr_bytes, *trk_store = p0_store + (size_t)nk * pts;
    int *d_counts = (int *)(trk_store + (size_t)2 * nk * nk * pts);      // [nk] corners per ref kernel, [nk*nk] kept tracks
    KM_HIP(c, hipMemsetAsync(d_counts, 0, (size_t)(nk + nk * nk) * sizeof(int), c->stream));
    km_pyr PR[8], PM[8];
    {
        // the uint8 stretch (klt.py:42-49; [+ 255 - x, klt.py:419]) rides in every Laplacian pass, as in the tile pipeline: both images
        // of a kernel size in ONE launch from the raw rasters - the marching kernel for k <= 7; no uint8 copies of the rasters exist
        km_stage_timer t(c, ST_LAPLACIAN);
        for (int k = 0; k < nk; k++) {
            const bool first = k == 0 && mask_auto != nullptr;
            if ((rc = kd_stretch_laplacian_pair(c, d_ref, d_mon, dtype, H, W, sref, smon, sc->mm, ksizes[k], ksizes[k], prm->invert_mon, nodata_ref, nodata_mon,
                                                lap_ref + (size_t)k * na, lap_mon + (size_t)k * na, first ? mask_auto : nullptr, first ? &sc->valid : nullptr)))
                return rc;
        }
    }
    const int n_lim = prm->max_corners > 0 && prm->max_corners < cap ? prm->max_corners : cap;
    // ---- ONE pipeline for the whole search where the batched forms cover the case (round 6; two-level pyramids, maxCorners > 0, the
    // synchronisation-free corner path): the 2 nk pyramids in one launch, the nk corner detections as one batch of units (fused
    // eigenvalue pass + selection chains side by side, nothing read back in between), the nk^2 tracker runs as ONE LK launch and one
    // count launch.  Two host synchronisations per search (corner counts + flags; inlier counts) instead of nk + 2, 4 + 2 nk launches of
    // dense kernels instead of 7 nk + 2 nk^2.  Anything the batched forms refuse goes through the loops below, run by run.
    int n_p0[8];
    const int *d_np0[8];                                             // device word holding the corner count of reference kernel k
    bool corners_done = false, tracks_done = false;
    const bool batchable = prm->max_level == 1 && probe.levels == 1 && nk <= KM_UNITS_MAX && nk * nk <= KM_LK_JOBS_MAX && prm->max_corners > 0 &&
                           prm->min_distance >= 1 && c->opt_speculative && c->fused_eig && !c->opt_key_cap && !c->opt_stage_cap && !c->opt_topk_factor &&
                           !c->opt_select_first && c->opt_eig3 && c->opt_lk2 && W >= 512 && H >= 2 * prm->block_size + 8;
    km_units U;
    if (batchable) {
        U.n = nk; U.dtype = KM_U8; U.capk = n / 8 + 4096 * KM_NSHARD;
        unsigned long long *keys = (unsigned long long *)km_ws(c, WS_KEYS0, U.capk * sizeof(unsigned long long) * (size_t)nk);
        if (!keys) return KM_E_NOMEM;
        for (int k = 0; k < nk; k++) {
            U.H[k] = H; U.W[k] = W; U.x_off[k] = 0.f; U.y_off[k] = 0.f;
            U.lap_ref[k] = lap_ref + (size_t)k * na; U.lap_mon[k] = lap_mon + (size_t)k * na; U.mask[k] = const_cast<uint8_t *>(mask);
            U.sc[k] = (km_scalars *)(sc_base + sc_stride * (size_t)(k + 1));
            U.keys[k] = keys + U.capk * (size_t)k;
            U.p0[k] = (float *)(p0_store + (size_t)k * pts);
            U.eig_partial[k] = nullptr; U.eig_npartial[k] = 0;
            km_pyr &A = U.A[k], &B = U.B[k];
            A.img[0] = U.lap_ref[k]; B.img[0] = U.lap_mon[k];
            A.H[0] = B.H[0] = H; A.W[0] = B.W[0] = W;
            A.img[1] = pyr_store + (size_t)(2 * k) * pyr_bytes; B.img[1] = pyr_store + (size_t)(2 * k + 1) * pyr_bytes;
            A.H[1] = B.H[1] = (H + 1) / 2; A.W[1] = B.W[1] = (W + 1) / 2;
            A.levels = B.levels = 1;
            d_np0[k] = &U.sc[k]->n_corners;
        }
        {
            km_stage_timer t(c, ST_PYRAMID);
            if ((rc = kd_pyrdown_units(c, U, 1))) return rc;
        }
        for (int k = 0; k < nk; k++) { PR[k] = U.A[k]; PM[k] = U.B[k]; }
        {
            km_stage_timer t(c, ST_EIGEN);
            rc = k3_eig_candidates_units(c, U, prm->block_size, prm->quality_level);
        }
        if (rc == KM_OK) {
            km_stage_timer t(c, ST_SELECT);
            rc = kf_rank_select_units(c, U, prm->max_corners, prm->quality_level, prm->min_distance, cap);
        }
        if (rc != KM_OK && rc != KM_E_UNSUPPORTED) return rc;
        if (rc == KM_OK) {
            unsigned flags[8];
            for (int k = 0; k < nk; k++) { KM_D2H(c, &n_p0[k], &U.sc[k]->n_corners, sizeof(int)); KM_D2H(c, &flags[k], &U.sc[k]->flags, sizeof(unsigned)); }
            KM_FLUSH(c);
            for (int k = 0; k < nk; k++) {
                if (!flags[k]) continue;
                // the unit did not fit the fixed capacities of the synchronisation-free corner path: its corners through the exact one
                c->stats.path_flags |= KM_PATH_SPEC_RETRY;
                KM_HIP(c, hipMemsetAsync(&sc->max_eig_key, 0, sizeof(km_scalars) - offsetof(km_scalars, max_eig_key), c->stream));
                if ((rc = gftt_dev(c, lap_ref + (size_t)k * na, mask, H, W, prm->max_corners, prm->quality_level, prm->min_distance, prm->block_size, U.p0[k], cap, sc)))
                    return rc;
                KM_HIP(c, hipMemcpyAsync(&U.sc[k]->n_corners, &sc->n_corners, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
                KM_D2H(c, &n_p0[k], &sc->n_corners, sizeof(int));
                KM_FLUSH(c);
            }
            corners_done = true;
        }
    }
    if (!corners_done) {
        {
            km_stage_timer t(c, ST_PYRAMID);
            for (int k = 0; k < nk; k++)
                if ((rc = build_pyramid_single(c, lap_ref + (size_t)k * na, H, W, prm->win_size, prm->max_level, pyr_store + (size_t)(2 * k) * pyr_bytes, &PR[k], nullptr)) ||
                    (rc = build_pyramid_single(c, lap_mon + (size_t)k * na, H, W, prm->win_size, prm->max_level, pyr_store + (size_t)(2 * k + 1) * pyr_bytes, &PM[k], nullptr)))
                    return rc;
        }
        // ---- corners of every reference Laplacian (klt.py:494), one after the other
        for (int k = 0; k < nk; k++) {
            float *p0 = (float *)(p0_store + (size_t)k * pts);
            // gftt_dev starts from a clean scalar block; the min/max and the valid-pixel count gathered above stay
            KM_HIP(c, hipMemsetAsync(&sc->max_eig_key, 0, sizeof(km_scalars) - offsetof(km_scalars, max_eig_key), c->stream));
            if ((rc = gftt_dev(c, lap_ref + (size_t)k * na, mask, H, W, prm->max_corners, prm->quality_level, prm->min_distance, prm->block_size, p0, cap, sc)))
                return rc;
            KM_HIP(c, hipMemcpyAsync(&d_counts[k], &sc->n_corners, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
            KM_D2H(c, &n_p0[k], &sc->n_corners, sizeof(int));
            d_np0[k] = &d_counts[k];
        }
        KM_FLUSH(c);
    }
    // ---- nk*nk tracker runs (mon kernel outer, ref kernel inner)
    if (corners_done) {
        km_lk_job jobs[KM_LK_JOBS_MAX];
        km_count_jobs CJ;
        for (int im = 0; im < nk; im++)
            for (int ir = 0; ir < nk; ir++) {
                const int combo = im * nk + ir;
                km_lk_job &j = jobs[combo];
                j.A = PR[ir]; j.B = PM[im]; j.pts_in = (const float *)(p0_store + (size_t)ir * pts); j.d_n = d_np0[ir];
                j.p1 = (float *)(trk_store + (size_t)(2 * combo) * pts); j.p0r = (float *)(trk_store + (size_t)(2 * combo + 1) * pts);
                CJ.p0[combo] = j.pts_in; CJ.p0r[combo] = j.p0r; CJ.d_n[combo] = j.d_n;      // (a reference kernel without corners: 0 points, count 0)
            }
        {
            km_stage_timer t(c, ST_LK);
            rc = kl_jobs_launch(c, jobs, nk * nk, n_lim, prm->win_size, prm->max_count, prm->epsilon);
            if (rc == KM_OK) rc = kf_count_kept_jobs(c, CJ, nk * nk, n_lim, 0.1f, &d_counts[nk]);
        }
        if (rc != KM_OK && rc != KM_E_UNSUPPORTED) return rc;
        tracks_done = rc == KM_OK;
    }
    if (!tracks_done) {
        km_stage_timer t(c, ST_LK);
        for (int im = 0; im < nk; im++)
            for (int ir = 0; ir < nk; ir++) {
                if (n_p0[ir] <= 0) continue;
                const int combo = im * nk + ir;
                float *p0 = (float *)(p0_store + (size_t)ir * pts);
                float *p1 = (float *)(trk_store + (size_t)(2 * combo) * pts), *p0r = (float *)(trk_store + (size_t)(2 * combo + 1) * pts);
                if ((rc = kl_track(c, PR[ir], PM[im], p0, d_np0[ir], n_lim, prm->win_size, prm->max_count, prm->epsilon, true, p1, p0r)) ||
                    (rc = kf_count_kept(c, p0, p0r, d_np0[ir], n_lim, 0.1f, &d_counts[nk + combo])))
                    return rc;
            }
    }
    int kept[64];
    KM_D2H(c, kept, d_counts + nk, (size_t)nk * nk * sizeof(int));
    unsigned long long valid = 0;
    KM_D2H(c, &valid, &sc->valid, sizeof valid);
    KM_FLUSH(c);
    c->stats.valid_pixels = (int64_t)valid;
    double best_ratio = -1.0;
    int best = -1;
    for (int im = 0; im < nk; im++)
        for (int ir = 0; ir < nk; ir++) {
            const int combo = im * nk + ir;
            if (n_p0[ir] <= 0) { out_ratios[combo] = 0.0; continue; }          // klt_tracker returned None: score 0, never the best
            const double ratio = (double)kept[combo] / (double)n_p0[ir];
            out_ratios[combo] = ratio;
            if (ratio > best_ratio) { best_ratio = ratio; best = combo; }
        }
    const size_t fb = 16 + (size_t)cap * 6 * sizeof(float);
    char *d_out = (char *)km_ws(c, WS_FRAME, fb);
    if (!d_out) return KM_E_NOMEM;
    if ((rc = frame_block_free(c))) return rc;
    out_best[0] = out_best[1] = -1;
    if (best < 0) {
        memset(host_out, 0, 16);
        return KM_OK;
    }
    const int bm = best / nk, br = best % nk;
    out_best[0] = ksizes[bm]; out_best[1] = ksizes[br];
    {
        km_stage_timer t(c, ST_FRAME);
        if ((rc = kf_frame(c, (const float *)(p0_store + (size_t)br * pts), (const float *)(trk_store + (size_t)(2 * best) * pts),
                           (const float *)(trk_store + (size_t)(2 * best + 1) * pts), d_np0[br], n_lim, cap, 0.1f, x_off, y_off, d_out, nullptr, W)))
            return rc;
    }
    KM_D2H(c, host_out, d_out, fb);
    KM_FLUSH(c);
    c->stats.n_init = ((const int *)host_out)[1];
    return KM_OK;
}

int km_klt_tile_frame_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                          const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm,
                          float x_off, float y_off, void *host_out, int cap)
{
    return tile_frame_impl(c, d_ref, d_mon, dtype, H, W, sref, smon, d_mask, smask, nodata_ref, nodata_mon, prm, x_off, y_off, nullptr, nullptr, 0, 0,
                           0, 0, false, 0.0, host_out, cap);
}

int km_klt_tile_frame_zncc_dev(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                               const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon,
                               const km_klt_params *prm, float x_off, float y_off, const void *d_ref_full, const void *d_mon_full, int Hf,
                               int Wf, ptrdiff_t sref_f, ptrdiff_t smon_f, double zncc_threshold, void *host_out, int cap)
{
    return tile_frame_impl(c, d_ref, d_mon, dtype, H, W, sref, smon, d_mask, smask, nodata_ref, nodata_mon, prm, x_off, y_off, d_ref_full, d_mon_full,
                           Hf, Wf, sref_f, smon_f, true, zncc_threshold, host_out, cap);
}

// Asynchronous form of km_klt_tile_frame[_zncc]_dev for a stream of tiles / band pairs: returns as soon as the last
// kernel and the copy of the frame block are ENQUEUED (the corner selection still synchronises inside), so the caller's
// next submission queues its dense stages right behind this frame's tail and the GPU never idles between frames.
// km_frame_wait (any thread) blocks until frame `ticket` is complete and hands out its block in pinned host memory, valid
// until KM_FRAME_SLOTS further submissions.  d_ref_full == NULL: no ZNCC column.
int km_klt_tile_frame_submit(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
                             const uint8_t *d_mask, ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon,
                             const km_klt_params *prm, float x_off, float y_off, const void *d_ref_full, const void *d_mon_full, int Hf,
                             int Wf, ptrdiff_t sref_f, ptrdiff_t smon_f, double zncc_threshold, int cap, int *ticket)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    if (!ticket) return km_fail(c, KM_E_ARG, "klt_tile_frame_submit: null ticket");
    const int k = c->fslot_next;
    km_frame_slot *slot = &c->fslot[k];
    if (slot->pending.load(std::memory_order_acquire)) {   // never waited for: its block is about to be overwritten
        KM_HIP(c, hipEventSynchronize(slot->done));
        slot->pending.store(0, std::memory_order_release);
    }
    c->ev_cur = 1 + k;
    const int rc = tile_frame_impl(c, d_ref, d_mon, dtype, H, W, sref, smon, d_mask, smask, nodata_ref, nodata_mon, prm, x_off, y_off, d_ref_full,
                                   d_mon_full, Hf, Wf, sref_f, smon_f, d_ref_full != nullptr, zncc_threshold, nullptr, cap, slot);
    c->ev_cur = 0;
    if (rc) return rc;
    slot->pending.store(1, std::memory_order_release);
    c->fslot_next = (k + 1) % KM_FRAME_SLOTS;
    *ticket = k;
    return KM_OK;
}

// Touches only the slot (no context state, no error string): safe from another thread while the context is submitting.
int km_frame_wait(km_ctx *c, int ticket, const void **block, size_t *bytes)
{
    if (!c || ticket < 0 || ticket >= KM_FRAME_SLOTS || !block) return KM_E_ARG;
    km_frame_slot *slot = &c->fslot[ticket];
    if (!slot->done || !slot->pending.load(std::memory_order_acquire)) return KM_E_ARG;
    // The waiting thread has nothing else to do: poll and SLEEP (hipEventSynchronize spins - also on an event created with
    // hipEventBlockingSync - and kept one CPU per rank at 100 %: 0.72 of every 0.91-ms step; eight ranks want those CPUs for RCCL's proxies)
    // (the kernel rounds a sleep up by the thread's timer slack, 50 us by default: 1 us for the threads that wait here)
    // (the slack is this thread's for the duration of the wait only: the caller may be an application thread - ADVICE r5)
    const int slack_before = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    struct slack_restore { int v; ~slack_restore() { if (v > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)v, 0UL, 0UL, 0UL); } } slack_guard{slack_before};
    // a pipelined batched submission enqueues its tail (LK .. copy-out) with the NEXT submission or km_frame_flush: wait for that first.
    // Nobody doing either for 100 ms is a caller that forgot to flush - the tail is then enqueued from here (under the context's
    // enqueue lock; every other entry point of the submitting thread flushes before it touches the context)
    for (int spins = 0; slot->deferred.load(std::memory_order_acquire); spins++) {
        if (spins >= 5000) { const int rf = km_units_flush(c, false); if (rf) return rf; break; }
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
    for (;;) {
        const hipError_t e = hipEventQuery(slot->done);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return KM_E_HIP;
        struct timespec ts = {0, 20000};                  // 20 us: a hundredth of a batched submission
        nanosleep(&ts, nullptr);
    }
    slot->pending.store(0, std::memory_order_release);
    *block = slot->host;
    if (bytes) *bytes = slot->bytes;
    return KM_OK;
}

// Device-side hand-over of a submitted frame's block to a stream of the CALLER: `hip_stream` (a hipStream_t, e.g. the stream an RCCL
// all-gather of the frame sink is issued on) waits - on the device, the host does not block - until the block of frame `ticket` has
// been written to the frame sink that was set when the frame was submitted.
int km_stream_wait_frame(km_ctx *c, int ticket, void *hip_stream)
{
    if (!c || ticket < 0 || ticket >= KM_FRAME_SLOTS) return km_fail(c, KM_E_ARG, "km_stream_wait_frame: bad ticket %d", ticket);
    km_frame_slot *slot = &c->fslot[ticket];
    if (!slot->sunk || !slot->sunk_valid) return km_fail(c, KM_E_ARG, "km_stream_wait_frame: frame %d was submitted without a frame sink", ticket);
    KM_HIP(c, hipStreamWaitEvent((hipStream_t)hip_stream, slot->sunk, 0));
    return KM_OK;
}

// Stage spans of frame `ticket` (after km_frame_wait; profiling enabled), same order as km_get_stage_ms.
int km_frame_stage_ms(km_ctx *c, int ticket, float *out, int cap, int *n)
{
    if (!c || ticket < 0 || ticket >= KM_FRAME_SLOTS || !out) return KM_E_ARG;
    const int m = cap < ST_COUNT ? cap : ST_COUNT;
    for (int i = 0; i < m; i++) {
        out[i] = 0.f;
        float ms = 0.f;
        if (c->ev_ready && c->evs_used[1 + ticket][i] &&
            hipEventElapsedTime(&ms, c->evs[1 + ticket][i][0], c->evs[1 + ticket][i][1]) == hipSuccess)
            out[i] = ms;
    }
    if (n) *n = m;
    return KM_OK;
}

}  // extern "C"
