// Helpers shared by the entry-point translation units (api*.hip).
#pragma once
#include "common.hpp"

// stage timers are cleared per pipeline: the KLT entry points own [ST_MINMAX, ST_LK], ZNCC owns ST_ZNCC
enum { RESET_NONE = 0, RESET_KLT = 1, RESET_ZNCC = 2 };
int begin_call(km_ctx *c, int reset = RESET_NONE);   // start of every entry point: device, pending uploads, stale jobs, stage timers
int check_image(km_ctx *c, const void *p, int H, int W, ptrdiff_t stride, const char *what);
int check_params(km_ctx *c, const km_klt_params *p);
int frame_block_free(km_ctx *c);                     // WS_FRAME may be rewritten once the previous submitted frame's block has left

// results for the caller: DMA into the context's page-locked landing arena, then (KM_FLUSH) complete the stream and copy out
#define KM_D2H(c, dst, src, bytes)                                           \
    do {                                                                     \
        const int rq_ = km_d2h_queue((c), (dst), (src), (bytes));            \
        if (rq_) return rq_;                                                 \
    } while (0)
#define KM_FLUSH(c)                                                          \
    do {                                                                     \
        const int rq_ = km_d2h_flush(c);                                     \
        if (rq_) return rq_;                                                 \
    } while (0)


// ---- api.hip: caller memory <-> device (through the page-locked ring / landing arena, staging.hip)
int h2d_now(km_ctx *c, void *dst, const void *src, size_t bytes);
int verify_upload(km_ctx *c, const char *when, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, const void *d);
int upload_image(km_ctx *c, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, void **dptr);
km_scalars *scalars(km_ctx *c);
// ---- api_tile.hip: the stages of a tile on dense device images
int build_pyramid_single(km_ctx *c, const uint8_t *d_img, int H, int W, int win, int max_level, uint8_t *store, km_pyr *P, size_t *used);
int build_pyramid_pair(km_ctx *c, const uint8_t *d_a, const uint8_t *d_b, int H, int W, int win, int max_level, km_pyr *A, km_pyr *B);
int gftt_dev(km_ctx *c, const uint8_t *d_img, const uint8_t *d_mask, int H, int W, int max_corners, double quality, double min_distance, int block,
             float *d_xy, int cap, km_scalars *sc);
int read_stats(km_ctx *c, km_scalars *sc);
int klt_track_dev(km_ctx *c, const uint8_t *d_ref_lap, const uint8_t *d_mon_lap, const uint8_t *d_mask, int H, int W, const km_klt_params *prm,
                  const float *d_p0_in, int n_p0, float *d_p0, float *d_p1, float *d_p0r, int cap, km_scalars *sc);
int klt_tile_dev_impl(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon, const uint8_t *d_mask,
                      ptrdiff_t smask, const double *nodata_ref, const double *nodata_mon, const km_klt_params *prm, float *d_p0, float *d_p1, float *d_p0r,
                      int cap, km_scalars *sc, bool *no_valid);
int fetch_tracks(km_ctx *c, km_scalars *sc, const float *d_p0, const float *d_p1, const float *d_p0r, float *p0, float *p1, float *p0r, int cap,
                 int *out_n);

// api_units.hip: Laplacian kernel sizes the batched stretch + Laplacian pass covers (k_dense.hip kd_stretch_laplacian_units)
static inline bool km_units_ksize_supported(int k) { return k >= 1 && k <= 11 && (k & 1); }
