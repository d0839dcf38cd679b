/*
 * karios_hip.h -- C ABI of libkarios_hip.so, the MI355X (gfx950) implementation of
 * the KARIOS image-matching hot path.
 *
 * The reference (telespazio-tim/karios) has no FFI of its own: the seam is the
 * Python module surface of `karios.matcher`.  Each entry
 * point below replaces the third-party C++ the reference reaches through cv2 /
 * skimage / numpy at the cited call site (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a KARIOS maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - every function returns an int status: 0 = OK, <0 = error (KM_E_*);
 *     km_last_error(ctx) gives the message.  Nothing throws across the boundary.
 *   - "host" entry points take caller-owned host buffers (numpy) and do the
 *     H2D/D2H copies themselves; "_dev" entry points take device pointers
 *     (hipMalloc'ed or a torch CUDA tensor's data_ptr) and run entirely on the
 *     context's stream.  No pointer is retained after return.
 *   - images are row-major, `stride` is in ELEMENTS between rows.
 *   - a context owns one HIP stream plus a grow-only device workspace; it is not
 *     thread-safe: use one context per calling thread (the reference calls
 *     klt_tracker from a ThreadPoolExecutor, klt.py:526-527).
 */
#ifndef KARIOS_HIP_H
#define KARIOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct km_ctx km_ctx;

/* pixel types accepted for raw images (KM_F64 / KM_I32 / KM_U32: km_zncc_windows only) */
enum { KM_U8 = 0, KM_U16 = 1, KM_I16 = 2, KM_F32 = 3, KM_F64 = 4, KM_I32 = 5, KM_U32 = 6 };

/* status codes */
enum {
    KM_OK = 0,
    KM_E_ARG = -1,      /* malformed argument (cv2.error equivalent) */
    KM_E_HIP = -2,      /* HIP runtime failure */
    KM_E_NOMEM = -3,
    KM_E_UNSUPPORTED = -4,
    KM_E_NO_DEVICE = -5,
    KM_E_INTERNAL = -6
};

/* KLTConfiguration fields (core/configuration.py:36-50) + the fixed LK criteria of
 * klt.py:128-132 (maxLevel 1, COUNT|EPS 30 / 0.03) + back_threshold klt.py:143. */
typedef struct km_klt_params {
    int32_t max_corners;      /* maxCorners */
    int32_t block_size;       /* blocksize */
    int32_t win_size;         /* matching_winsize */
    int32_t max_level;        /* 1 */
    int32_t max_count;        /* 30 */
    int32_t ksize_mon;        /* laplacian_kernel_size (mon) */
    int32_t ksize_ref;        /* laplacian_kernel_size (ref) */
    int32_t invert_mon;       /* laplacian_invert_polarity */
    double quality_level;     /* qualityLevel */
    double min_distance;      /* minDistance */
    double epsilon;           /* 0.03 */
} km_klt_params;

/* diagnostics of the last km_klt_* / km_good_features call on a context */
typedef struct km_klt_stats {
    int64_t valid_pixels;     /* mask > 0 count (klt.py:276) */
    int64_t n_candidates;     /* local maxima above threshold */
    int32_t n_init;           /* corners selected (Ninit, klt.py:150) */
    int32_t n_select_batches; /* greedy-selection batches executed */
    double min_ref, max_ref, min_mon, max_mon; /* _to_uint8 stretch bounds */
    float max_eig;            /* maxVal of minMaxLoc */
    float emitted_ratio;      /* candidate keys emitted by the fused eig kernel / exact candidate count */
    int32_t path_flags;       /* KM_PATH_* bits: which retry paths of the corner detector the call went through */
    int32_t tie_rows;         /* (wavefront, row) steps of the fused 8-px eigenvalue pass in which a lane held more than two candidates
                                 of its 8 pixels (ties / plateaus): the per-pixel emission path; diagnostics of the blocking tile entry points */
} km_klt_stats;
#define KM_PATH_KEY_REGROW 1      /* a key-buffer shard overflowed: buffer regrown, detection repeated */
#define KM_PATH_STAGE_FALLBACK 2  /* the fused kernel's key stage overflowed: eig map + candidate kernel instead */
#define KM_PATH_SECOND_PASS 4     /* the top-K slice held too few mutually distant corners: selection on all candidates */
#define KM_PATH_PREFIX_GROWN 8    /* the selection's first ranked prefix was enlarged */
#define KM_PATH_SPEC_RETRY 16     /* the speculative (no host synchronisation) corner path flagged the tile: repeated exactly */
#define KM_PATH_MM_EARLY 32      /* a submitted unit's min / max ran on the second stream beside the previous unit's LK ("mm_early") */

/* ---- context ------------------------------------------------------------ */
int km_version(void);
int km_ctx_create(int device, km_ctx **out);
int km_ctx_destroy(km_ctx *ctx);
const char *km_last_error(km_ctx *ctx);   /* ctx may be NULL: last global error */
int km_ctx_sync(km_ctx *ctx);
/* enable (1) / disable (0) hipEvent stage timing; read back after a call */
int km_set_profiling(km_ctx *ctx, int enable);
/* Options (no counterpart in the reference).  Results NEVER depend on them: every option selects between forms the parity suite
 * holds bit-identical, or shrinks a capacity so that a retry path runs (tests/test_gpu_forced_paths.py, tests/test_gpu_parity.py).
 * The RELEASE library accepts exactly the names below and returns KM_E_ARG for anything else (tests/test_host_logic.py checks that
 * the development names are rejected); the development build (make DEV=1, km_is_dev_build) adds A/B switches of settled choices.
 *  forms
 *   "fused_eig"    1 (default): GFTT's minimum-eigenvalue + candidate detection fused in one pass (no eig map); 0: eig map + candidate
 *                  scan.  Initial value from the environment variable KARIOS_HIP_FUSED_EIG
 *   "eig3"         1 (default): the fused pass runs 8 pixels per lane (k_eig3.hip) on images >= 512 columns wide; 0: always the
 *                  2-pixels-per-lane kernel (k_eig2.hip).  Initial value from KARIOS_HIP_EIG3
 *   "lk2"          1 (default): LK on four resident patches per key point (two-level pyramids); 0: the first form (one patch per
 *                  direction and level), which also serves deeper pyramids and row bands
 *   "speculative"  1 (default): the tile entry points detect corners without a host synchronisation and without sorts (fixed
 *                  capacities, k_select2.hip); a tile that does not fit is flagged (frame header word 2) and repeated through the exact
 *                  path - inside the call for the blocking entry points, by the caller of km_frame_wait for submitted frames
 *                  (PendingFrame.result()).  0: always the exact path (two scalar read-backs per tile, k_select.hip + k_sort.hip).
 *                  Initial value from KARIOS_HIP_SPECULATIVE
 *   "aux_pyramid"  1 (default): on the synchronisation-free path the two pyramids are built on a second stream beside the fused
 *                  eigenvalue pass and joined before LK; 0: on the library's stream.  Initial value from KARIOS_HIP_AUX_PYRAMID
 *   "mm_early"     1 (default): a unit submitted (km_klt_tile_frame_submit / km_klt_units_frame_submit) directly behind another one
 *                  starts its min / max on the second stream as soon as the previous unit's LK launch starts (KM_PATH_MM_EARLY);
 *                  0: on the library's stream, behind the previous unit's tail
 *   "units_pipeline" 1: consecutive km_klt_units_frame_submit calls on this context form a SOFTWARE PIPELINE (csrc/api_units.hip): two
 *                  workspace sets alternate, the dense stages of neighbouring submissions interleave on the library's stream and the
 *                  latency-bound chains of one (corner selection; frame stage + scores) run on a second stream beside the dense kernels
 *                  of the other.  The tail of a submission (LK, frame stage, scores, copy-out) is then enqueued by the NEXT submission,
 *                  by km_frame_flush, by any other entry point or by km_ctx_sync; km_frame_wait on such a frame waits for one of them
 *                  (and enqueues the tail itself after 100 ms).  Frames are bit-identical.  0 (default): every submission is complete
 *                  in stream order when the call returns.  karios_amd.stream.FrameStream switches it on for the contexts it drives
 *   "frame_mi"     1: frame blocks that carry the ZNCC column also carry `mutual_info_score` (MutualInfoService,
 *                  mutual_info_service.py:73-130) and `mi_score` (ZNCCService.compute_mi, zncc_service.py:240-287) of the same rows -
 *                  the whole scoring of KariosAPI._handle_klt_results (api/core.py:894-907) in the tile call: two more float64 columns
 *                  of `cap` entries behind the zncc column (NaN where score < threshold or the chip leaves the image); 0 (default)
 *   "phase_fp64"   1: km_phase_shift* always evaluates in double precision (k_fft64.hip), the reference's precision; 0 (default):
 *                  hand-written float32 FFT where the image sides factor into {2,3,5,7,61}, double precision only when the
 *                  float32 correlation peak is not at least 1 % above every other sample
 *   "fft61"        1 (default): rows of length 61 M of the float32 transform through the wave-local form; 0: the generic row kernel
 *   "fft_herm"     1 (default): the inverse float32 transform works on the Hermitian half of the cross-power spectrum; 0: full plane
 *   "f64_pair"     1 (default): the inverse along the rows of the double-precision transform packs two image rows into one complex
 *                  transform (the correlation surface of two real images is real); 0: one row per transform
 *   "f64_half"     1 (default): ... and its inverse column levels only run the columns kx <= W / 2; 0: all columns
 *   "f64_plain"    1: the double-precision transform packs the images in a pass of its own and finds the arg-max in two passes behind
 *                  the last level - the form that serves sides with a Bluestein dimension - on every shape (default 0: both ends ride in
 *                  the level kernels where the shape allows)
 *  test knobs that shrink internal capacities so that the corner detector's retry paths run on every call (0 = default)
 *   "key_cap"      candidate keys per shard of the first attempt          -> key-buffer overflow + regrow
 *   "stage_cap"    usable slots of the fused kernel's per-wave key stage  -> stage overflow + two-kernel repeat
 *   "topk_factor"  top-K pre-filter keeps factor * maxCorners keys (8)    -> second selection pass on all candidates
 *   "select_first" first ranked prefix of the selection sweeps (3 * maxCorners) -> prefix growth
 *   "stash_cap"    kept keys a workgroup of the scatter launch stashes in LDS -> its second read of the keys
 *   "spec_flag"    KM_FLAG bits the synchronisation-free path raises artificially -> the flag-and-repeat logic
 *   "defer"        0: pyramid jobs of the exact path run after the selection's read-back waits instead of under them
 *  profiling
 *   "profile_stage" with km_set_profiling(1): time only stage i of km_stage_name (every timed span records two events on the
 *                  library stream and the kernels either side no longer overlap: ~6 us per span); -1 (default): every stage
 *   "profile_every" N: only every N-th tile call records its stage events
 *   "roctx"        1: every stage's host-side enqueue span becomes a roctx range (rocprofv3 --marker-trace; resolved at run time)
 * Environment variables read by the release library: KARIOS_HIP_FUSED_EIG, KARIOS_HIP_EIG3, KARIOS_HIP_AUX_PYRAMID,
 * KARIOS_HIP_SPECULATIVE (initial option values, above) and KARIOS_HIP_UPLOAD_CHECKSUM (km_upload_check_stats). */
int km_set_option(km_ctx *ctx, const char *name, int value);
/* 1: development build (make -C karios_amd/csrc DEV=1): km_set_option additionally accepts A/B switches of settled choices and the
 * library reads KARIOS_HIP_* tuning variables; 0: release build (the default; what __graft_entry__.build() produces) */
int km_is_dev_build(void);
/* development build only (KM_E_UNSUPPORTED otherwise): counters of the "eig3_count" option after a blocking tile call */
int km_dev_counters(km_ctx *ctx, unsigned long long out[2]);
/* stage times (ms) of the last pipeline call; names via km_stage_name(i) */
int km_get_stage_ms(km_ctx *ctx, float *out, int cap, int *n);
const char *km_stage_name(int i);
int km_get_klt_stats(km_ctx *ctx, km_klt_stats *out);

/* ---- device memory helpers (bench / multi-GPU plumbing) ----------------- */
int km_dev_alloc(km_ctx *ctx, size_t bytes, void **dptr);
int km_dev_free(km_ctx *ctx, void *dptr);
int km_h2d(km_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int km_d2h(km_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Page-locked host memory for image buffers that GDAL (or numpy) fills and the GPU fetches: an upload from such a buffer
 * runs at full PCIe rate and asynchronously (a pageable source is staged chunk by chunk and blocks the caller). */
int km_host_alloc(km_ctx *ctx, size_t bytes, void **hptr);
int km_host_free(km_ctx *ctx /* may be NULL: the block outlived its context */, void *hptr);
/* Asynchronous strided upload on the context's COPY stream: `rows` rows of `width_bytes` bytes, pitches in bytes.  Returns
 * at once when `src_host` is page-locked; a pageable source is packed into the context's page-locked ring chunk by chunk
 * (csrc/staging.hip) and has been read completely when the call returns - no runtime copy ever touches pageable memory; every later call on
 * the context that launches kernels waits (on the device, not
 * on the host) for the uploads queued so far, so pair / tile i+1 can travel while pair / tile i computes.  The caller keeps
 * `src_host` alive and unmodified, and `dst_dev` unused by earlier work, until km_upload_wait or that later call returns. */
int km_upload_async(km_ctx *ctx, void *dst_dev, size_t dst_pitch, const void *src_host, size_t src_pitch,
                    size_t width_bytes, size_t rows);
int km_upload_wait(km_ctx *ctx);   /* host-side wait for the uploads queued so far */
/* Diagnosis of klt.py:252-253 ("a tile is matched as read"): with KARIOS_HIP_UPLOAD_CHECKSUM=1 in the environment a row-checksum
 * kernel runs on the library stream right behind every host-buffer upload of the blocking entry points and is compared with the
 * host's checksum of the source rows when the call completes (mismatches are reported on stderr, see csrc/staging.hip).
 * *armed = uploads checked so far on this context, *missed = uploads whose first consumer saw rows that differ from the source. */
int km_upload_check_stats(km_ctx *ctx, int64_t *armed, int64_t *missed);
/* Finer ordering for pipelines: km_upload_mark returns a ticket for "the uploads queued so far" and takes them out of the
 * automatic wait above; km_upload_join makes the compute stream wait (on the device) for that ticket only.  Upload pair i+1,
 * mark, launch the kernels of pair i, join, launch the kernels of pair i+1: the copy hides under pair i's compute. */
int km_upload_mark(km_ctx *ctx, int *ticket);
int km_upload_join(km_ctx *ctx, int ticket);
/* The buffers passed as full images to the ZNCC / MI entry points (km_klt_tile_frame_zncc_dev, km_klt_tile_frame_submit,
 * km_zncc_batch[_dev], km_mi_batch[_dev]) hold only a WINDOW of the H_image x W_image image, starting at image pixel (ox, oy) -
 * a multi-GPU rank keeps just its tile plus a margin resident.  Key-point coordinates stay image coordinates (the float32
 * sum x0 + dx that the reference rounds, zncc_service.py:195-196, depends on their magnitude), the reference's bounds rule
 * is applied with the image size, pixels are fetched relative to (ox, oy).  A chip inside the image but outside the window
 * scores the NaN with bit pattern KM_NAN_OUTSIDE_WINDOW (the caller's margin was too small).  H_image = W_image = 0: off. */
#define KM_NAN_OUTSIDE_WINDOW 0x7ff80000dead0000ull
int km_set_image_window(km_ctx *ctx, int ox, int oy, int H_image, int W_image);
/* Which evaluation the last km_phase_shift* call used: *path = 1 float32 hand-written FFT (k_fft.hip), 2 double precision
 * (hand-written too, k_fft64.hip); *margin = (largest - second largest) / largest sample of |cross-correlation| seen by the
 * float32 path. */
int km_phase_info(km_ctx *ctx, int *path, double *margin);
/* Test hook, host only (no device, no context): how the double-precision transform behind km_phase_shift* (large_offset.py:39)
 * decomposes a side of n pixels - levels[2 i] = length of level i, levels[2 i + 1] = 0 (radices 2/3/5/7 in LDS) or 1 (one prime,
 * 11 .. 127); *bluestein = length of the chirp-z convolution when n has a larger prime factor (then no levels), else 0;
 * negpos[pos] (n ints, may be NULL) = position at which the forward levels leave the negated frequency of position pos. */
int km_phase_plan(int n, int along_columns, int *levels, int cap_levels, int *n_levels, int *bluestein, int *negpos);
/* Optional device-side copy of every frame block the km_klt_tile_frame_* entry points produce (same layout), e.g. a slice of
 * the send buffer of an RCCL all-gather: the block then never bounces through host memory.  NULL switches it off. */
int km_set_frame_sink(km_ctx *ctx, void *d_dst, size_t capacity_bytes);
/* ... for batched submissions (km_klt_units_frame_submit): unit k's block goes to d_dst + k * pitch_bytes (pitch 0: the block size),
 * e.g. the rows of an all-gather's send buffer that carry a unit id in front of every block */
int km_set_frame_sink_pitch(km_ctx *ctx, void *d_dst, size_t capacity_bytes, size_t pitch_bytes);
/* Hand-over of a SUBMITTED frame (km_klt_tile_frame_submit with a frame sink set) to a stream of the caller without the host:
 * `hip_stream` (a hipStream_t - e.g. the stream the RCCL all-gather of the per-tile blocks, klt.py:220-253 / SURVEY 8e, is issued
 * on) waits on the device until the block of frame `ticket` has reached the sink. */
int km_stream_wait_frame(km_ctx *ctx, int ticket, void *hip_stream);

/* ---- batched units ------------------------------------------------------- */
/* One work unit of a batched submission: a tile of `KLT.match` (klt.py:220-253) - a box of a resident pair - with the rasters its
 * ZNCC / MI chips are cut from (the whole pair, or the window of it a rank holds: win_* as km_set_image_window, win_H = 0: none). */
typedef struct km_unit {
    const void *d_ref, *d_mon;              /* first pixel of the box in the reference / monitored raster (device) */
    ptrdiff_t sref, smon;                   /* row strides in elements */
    const void *d_ref_full, *d_mon_full;    /* rasters of the score columns (NULL: bare frames; all units alike) */
    ptrdiff_t sref_f, smon_f;
    int32_t H, W;                           /* box size */
    int32_t Hf, Wf;                         /* size of the full rasters */
    float x_off, y_off;                     /* origin added to the key points (klt.py:341-342) */
    int32_t win_ox, win_oy, win_H, win_W;
    const uint8_t *d_mask;                  /* user mask of the box (klt.py:258-266; != 0: valid), NULL: the automatic mask (klt.py:268-273); all units alike */
    ptrdiff_t smask;                        /* its row stride in bytes */
} km_unit;
#define KM_UNITS_PER_SUBMISSION 16
/* km_klt_tile_frame_submit for n <= KM_UNITS_PER_SUBMISSION independent units of one pixel type and one parameter set in ONE device
 * pipeline (csrc/api_units.hip): every dense kernel, the corner-selection chain, LK, the frame stage and the scores are launched once
 * for all units.  The frame blocks (layout of km_klt_tile_frame_zncc_dev, unit order) are bit-identical to the unit-by-unit calls';
 * km_frame_wait(ticket) hands out n consecutive blocks, a frame sink receives them at its pitch.  Header word 2 of a block != 0: that
 * unit did not fit the fixed capacities of the synchronisation-free corner path - repeat it alone through km_klt_tile_frame_zncc_dev
 * with "speculative" 0.  Returns KM_E_UNSUPPORTED (no error text) when the batch form does not cover the case (maxCorners 0,
 * minDistance < 1, a unit narrower than 512 columns or without a level-1 pyramid, float32 score columns,
 * a shrunken test capacity): submit the units one by one then.  Either every unit carries a user mask (km_unit.d_mask) or none. */
int km_klt_units_frame_submit(km_ctx *ctx, const km_unit *units, int n_units, int dtype, const double *nodata_ref, const double *nodata_mon,
                              const km_klt_params *prm, double zncc_threshold, int cap, int *ticket);
/* "units_pipeline": enqueue what the last km_klt_units_frame_submit deferred (its LK, frame stage, scores and copy-out) - if that
 * submission is frame `ticket` (ticket < 0: whichever it is).  Call it on the submitting thread when no further submission follows
 * before the frame is waited for; a no-op otherwise. */
int km_frame_flush(km_ctx *ctx, int ticket);

/* ---- fine-grained mirrors (host buffers) -------------------------------- */
/* _to_uint8 (matcher/klt.py:42-49) [+ 255-x, klt.py:419]; out_minmax[2] nullable */
int km_to_uint8(km_ctx *ctx, const void *img, int dtype, int H, int W, ptrdiff_t stride,
                int invert, uint8_t *out, double *out_minmax);
/* automatic validity mask (klt.py:268-273); nodata pointers nullable */
int km_auto_mask(km_ctx *ctx, const void *mon, const void *ref, int dtype, int H, int W,
                 ptrdiff_t stride_mon, ptrdiff_t stride_ref, const double *nodata_mon,
                 const double *nodata_ref, uint8_t *mask, int64_t *valid);
/* Test hook: the oscillation stop of the LK kernels' iteration (cv2.calcOpticalFlowPyrLK, klt.py:134-140; OpenCV compares the
 * float32 |delta + prevDelta| with the DOUBLE literal 0.01) evaluated on the device for n host quadruples (ddx, pdx, ddy, pdy). */
int km_lk_oscillation_probe(km_ctx *ctx, const float *quads, int n, uint8_t *out);
/* cv2.Laplacian(u8, CV_8U, ksize) (klt.py:359-360, 427-434, 480-483) */
int km_laplacian_u8(km_ctx *ctx, const uint8_t *src, int H, int W, int ksize, uint8_t *dst);
/* cornerMinEigenVal inside cv2.goodFeaturesToTrack (klt.py:120) */
int km_min_eigen(km_ctx *ctx, const uint8_t *src, int H, int W, int block_size, float *eig);
/* cv2.goodFeaturesToTrack(img, mask=, maxCorners, qualityLevel, minDistance, blockSize)
 * (klt.py:120, 494).  out_xy: 2*cap floats (x,y interleaved); *out_n = 0 <=> None */
int km_good_features(km_ctx *ctx, const uint8_t *img, const uint8_t *mask, int H, int W,
                     int max_corners, double quality_level, double min_distance,
                     int block_size, float *out_xy, int cap, int *out_n);
/* cv::pyrDown u8 (pyramid level of calcOpticalFlowPyrLK) */
int km_pyrdown_u8(km_ctx *ctx, const uint8_t *src, int H, int W, uint8_t *dst);
/* cv2.calcOpticalFlowPyrLK(prev, next, pts, None, winSize=(w,w), maxLevel, criteria)
 * (klt.py:134-140); status/err are not produced (discarded at klt.py:142-153) */
int km_pyrlk(km_ctx *ctx, const uint8_t *prev, const uint8_t *next, int H, int W,
             const float *pts, int n, int win_size, int max_level, int max_count,
             double epsilon, float *out_pts);
/* klt_tracker body up to the forward-backward distance (klt.py:103-142):
 * GFTT on ref (unless p0 given) + LK ref->mon + LK mon->ref.
 * Outputs, each 2*cap floats: p0, p1, p0r; *out_n = Ninit (0 <=> "No features"). */
int km_klt_track(km_ctx *ctx, const uint8_t *ref_lap, const uint8_t *mon_lap,
                 const uint8_t *mask, int H, int W, const km_klt_params *prm,
                 const float *p0_in, int n_p0, float *p0, float *p1, float *p0r, int cap,
                 int *out_n);
/* KLT._match_tile numeric core for one box (klt.py:252-301, 407-436): raw images ->
 * uint8 stretch -> Laplacians -> auto mask (if mask NULL) -> klt_track.
 * *out_n = 0 with stats.valid_pixels == 0 <=> "No valid pixels" (klt.py:276-279). */
int km_klt_tile(km_ctx *ctx, const void *ref, const void *mon, int dtype, int H, int W,
                ptrdiff_t stride_ref, ptrdiff_t stride_mon, const uint8_t *mask,
                const double *nodata_ref, const double *nodata_mon,
                const km_klt_params *prm, float *p0, float *p1, float *p0r, int cap,
                int *out_n);
/* Pre-filter of KLT._match_tile on one tile (klt.py:268-273 automatic mask, :407-436 `_to_uint8` + inversion +
 * cv2.Laplacian of both images) exactly as the tile entry points run it: ONE fused kernel for both images.
 * out_mask may be NULL (no mask derived, *out_valid = -1), else *out_valid = count of valid pixels. */
int km_tile_prefilter(km_ctx *ctx, const void *ref, const void *mon, int dtype, int H, int W,
                      ptrdiff_t stride_ref, ptrdiff_t stride_mon, const double *nodata_ref,
                      const double *nodata_mon, int ksize_ref, int ksize_mon, int invert_mon,
                      uint8_t *out_lap_ref, uint8_t *out_lap_mon, uint8_t *out_mask,
                      int64_t *out_valid);
/* ZNCCService.compute_zncc per keypoint (matcher/zncc_service.py:186-238, _zncc2 :45-126):
 * out[k] = NaN where the reference returns NaN */
int km_zncc_batch(km_ctx *ctx, const void *ref, const void *mon, int dtype, int Href,
                  int Wref, int Hmon, int Wmon, ptrdiff_t stride_ref, ptrdiff_t stride_mon,
                  const float *x0, const float *y0, const float *dx, const float *dy, int n,
                  double *out);
/* _zncc2(img1, img2, u1, v1, u2, v2, n) for `count` window pairs of any half-size n >= 0 (matcher/zncc_service.py:45-126;
 * the reference's known-answer tests use 3x3 and 5x5 windows, tests/test_zncc_service.py:107-125) and any two pixel types:
 * uv = u1[count] | v1[count] | u2[count] | v2[count] (rows, columns of the window centres).  out[k] = NaN for a window
 * without variance; out_outside[k] = 1 (and NaN) where a window leaves its image - the reference raises IndexError there. */
int km_zncc_windows(km_ctx *ctx, const void *img1, const void *img2, int dtype1, int dtype2, int H1, int W1, int H2,
                    int W2, ptrdiff_t stride1, ptrdiff_t stride2, const int32_t *uv, int half_size, int count,
                    double *out, uint8_t *out_outside);
/* Mutual-information scores per keypoint on the 57x57 chips (next to ZNCC in _handle_klt_results, api/core.py:894-907):
 * out_studholme[k] = MutualInfoService._mutual_info  (matcher/mutual_info_service.py:32-63, column mutual_info_score)
 * out_nmi[k]       = ZNCCService._mutual_information (matcher/zncc_service.py:129-151,      column mi_score)
 * either output may be NULL; NaN where the reference returns NaN */
int km_mi_batch(km_ctx *ctx, const void *ref, const void *mon, int dtype, int Href, int Wref,
                int Hmon, int Wmon, ptrdiff_t stride_ref, ptrdiff_t stride_mon, const float *x0,
                const float *y0, const float *dx, const float *dy, int n, double *out_studholme,
                double *out_nmi);
/* skimage.registration.phase_cross_correlation(reference_image, moving_image)[0]
 * (matcher/large_offset.py:39): out_rc = [row, col] */
int km_phase_shift(km_ctx *ctx, const void *reference_image, const void *moving_image,
                   int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                   double out_rc[2]);
/* shift_image (core/image.py:70-101); elem_size in bytes (1, 2, 4 or 8) */
int km_shift_image(km_ctx *ctx, const void *img, int elem_size, int H, int W,
                   ptrdiff_t stride, int y_off, int x_off, void *out);

/* ---- device-resident pipeline (inputs already in HBM) -------------------- */
/* same as km_klt_tile with d_ref/d_mon/d_mask device pointers (the mask, if any, has its own row stride:
 * a box of a full-resolution resident mask); outputs are device
 * pointers too (each 2*cap floats) plus a device int for the count.  Asynchronous on
 * the context stream; call km_ctx_sync before reading results. */
int km_klt_tile_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H,
                    int W, ptrdiff_t stride_ref, ptrdiff_t stride_mon,
                    const uint8_t *d_mask, ptrdiff_t stride_mask, const double *nodata_ref,
                    const double *nodata_mon, const km_klt_params *prm, float *d_p0,
                    float *d_p1, float *d_p0r, int cap, int *d_n);
/* KLT._match_tile end to end on resident data (klt.py:236-349): km_klt_tile_dev, then the forward-backward
 * test / score of klt_tracker (klt.py:142-155) and the (x0, y0) ordering (klt.py:341-348) on the device.
 * host_out receives, in ONE device-to-host copy, 4 int32 {n_rows, n_init, 0, 0} followed by 6*cap float32:
 * x0 | y0 | dx | dy | score | index (int32 bit pattern: the row's label after pandas' in-place sort).
 * The optional 3-sigma outlier filter (klt.py:161-163) is not applied here. */
int km_klt_tile_frame_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H, int W,
                          ptrdiff_t stride_ref, ptrdiff_t stride_mon, const uint8_t *d_mask,
                          ptrdiff_t stride_mask, const double *nodata_ref, const double *nodata_mon,
                          const km_klt_params *prm, float x_off, float y_off, void *host_out, int cap);
/* Same, plus the ZNCC column of _handle_klt_results (core.py:876-893) for the rows with score >= zncc_threshold,
 * computed on the FULL-resolution resident images (key points carry full-image coordinates through x_off / y_off).
 * host_out: the layout above followed by cap float64 (NaN where not scored). */
int km_klt_tile_frame_zncc_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H, int W,
                               ptrdiff_t stride_ref, ptrdiff_t stride_mon, const uint8_t *d_mask,
                               ptrdiff_t stride_mask, const double *nodata_ref, const double *nodata_mon,
                               const km_klt_params *prm, float x_off, float y_off, const void *d_ref_full, const void *d_mon_full,
                               int H_full, int W_full, ptrdiff_t stride_ref_full, ptrdiff_t stride_mon_full,
                               double zncc_threshold, void *host_out, int cap);
/* Stream-of-tiles form of the two calls above (a KLT.match loop over tiles, klt.py:220-234, or the per-band loop of
 * KariosAPI, core.py:845-871): km_klt_tile_frame_submit returns once the last kernel and the copy of the frame block
 * are ENQUEUED; the next submission then queues its dense stages right behind this frame's tail.  d_ref_full == NULL:
 * no ZNCC column.  km_frame_wait blocks until frame `ticket` is complete and returns its block (layout above) in pinned
 * host memory owned by the context, valid until KM_FRAME_SLOTS (3) further submissions; it touches nothing but that
 * frame's slot, so another thread may call it while this context is already submitting the next frame.
 * km_frame_stage_ms: the frame's stage spans (km_set_profiling), order of km_stage_name. */
int km_klt_tile_frame_submit(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H, int W,
                             ptrdiff_t stride_ref, ptrdiff_t stride_mon, const uint8_t *d_mask,
                             ptrdiff_t stride_mask, const double *nodata_ref, const double *nodata_mon,
                             const km_klt_params *prm, float x_off, float y_off, const void *d_ref_full,
                             const void *d_mon_full, int H_full, int W_full, ptrdiff_t stride_ref_full,
                             ptrdiff_t stride_mon_full, double zncc_threshold, int cap, int *ticket);
int km_frame_wait(km_ctx *ctx, int ticket, const void **block, size_t *bytes);
int km_frame_stage_ms(km_ctx *ctx, int ticket, float *out, int cap, int *n);
int km_zncc_batch_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype,
                      int Href, int Wref, int Hmon, int Wmon, ptrdiff_t stride_ref,
                      ptrdiff_t stride_mon, const float *d_x0, const float *d_y0,
                      const float *d_dx, const float *d_dy, int n, double *d_out);
int km_mi_batch_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref,
                    int Hmon, int Wmon, ptrdiff_t stride_ref, ptrdiff_t stride_mon, const float *d_x0,
                    const float *d_y0, const float *d_dx, const float *d_dy, int n, double *d_out_studholme,
                    double *d_out_nmi);
/* KLT._match_tile_auto_ksize (klt.py:465-545) on resident data: the nk Laplacians of each image, their
 * pyramids and the nk corner lists (goodFeaturesToTrack of every reference Laplacian) are built once and stay on the
 * device; the nk*nk tracker runs (mon kernel outer, ref kernel inner = itertools.product order) reuse them.
 * out_ratios[im*nk + ir] = inlier ratio len(points)/Ninit of (ksizes[im], ksizes[ir]) (0 where the reference's
 * klt_tracker returns None); the best pair is the first maximum; out_best = {mon_ksize, ref_ksize} or {-1,-1}.
 * host_out: frame block of the best pair (layout of km_klt_tile_frame_dev).  prm->ksize_* are ignored, prm->invert_mon
 * applies (255 - uint8(mon) before the Laplacians, klt.py:419); outlier filtering is not part of this entry point. */
int km_klt_auto_ksize_frame_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H,
                                int W, ptrdiff_t stride_ref, ptrdiff_t stride_mon,
                                const uint8_t *d_mask, ptrdiff_t stride_mask,
                                const double *nodata_ref, const double *nodata_mon,
                                const km_klt_params *prm, const int *ksizes, int nk, float x_off,
                                float y_off, void *host_out, int cap, double *out_ratios,
                                int *out_best);
/* KariosAPI._filter_by_dn_values (api/core.py:650-737) on resident images: key point i (x = int(x0[i]), y = int(y0[i]))
 * is dropped (keep[i] = 0) when the reference OR the monitored pixel equals one of `no_values`, or when a pixel equals
 * its own image's no-data value (nodata_* nullable).  x0 / y0 / no_values / keep are host arrays; a key point outside
 * the image is an error. */
int km_dn_keep_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H, int W,
                   ptrdiff_t stride_ref, ptrdiff_t stride_mon, const float *x0, const float *y0,
                   int n, const double *no_values, int n_no, const double *nodata_ref,
                   const double *nodata_mon, uint8_t *keep);
int km_phase_shift_dev(km_ctx *ctx, const void *d_reference_image,
                       const void *d_moving_image, int dtype, int H, int W,
                       ptrdiff_t stride_a, ptrdiff_t stride_b, double out_rc[2]);
int km_shift_image_dev(km_ctx *ctx, const void *d_img, int elem_size, int H, int W,
                       ptrdiff_t stride, int y_off, int x_off, void *d_out);

/* ---- ONE tile matched exactly by several GPUs (row bands) ---------------------------------------------
 * The reference's default is a single tile (tile_size 20000 > 10980): one global min / max for the uint8 stretch, one global
 * maximum eigenvalue, ONE ranked greedy selection and one maxCorners cut (klt.py:42-49, 120).  A rank holds the rows of its band
 * plus a halo; karios_amd.parallel.match_tile_banded runs these steps and exchanges, between them, min / max and the maximum
 * eigenvalue key (all-reduce), the ranks' strongest candidate keys (all-gather) and finally the tracks.  Row origins must be
 * even (pyramid alignment). */
int km_minmax_dev(km_ctx *ctx, const void *d_img, int dtype, int H, int W, ptrdiff_t stride, double out_minmax[2]);
/* stretch with the GIVEN minmax = {min_ref, max_ref, min_mon, max_mon}, Laplacians, automatic (or user) mask cleared outside the
 * band's own rows [own_y0, own_y1); *valid_owned = valid pixels in those rows */
int km_band_prefilter_dev(km_ctx *ctx, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t stride_ref,
                          ptrdiff_t stride_mon, const double minmax[4], const double *nodata_ref, const double *nodata_mon,
                          int ksize_ref, int ksize_mon, int invert_mon, int own_y0, int own_y1, const uint8_t *d_user_mask,
                          uint8_t *d_lap_ref, uint8_t *d_lap_mon, uint8_t *d_mask, int64_t *valid_owned);
/* cornerMinEigenVal + candidate detection of the band; *local_max_key = ordered key of its maximum over the mask (0: none) */
int km_band_eigen_dev(km_ctx *ctx, const uint8_t *d_lap_ref, const uint8_t *d_mask, int H, int W, int block_size, double quality_level,
                      unsigned *local_max_key);
/* the band's candidate keys (value bits << 32 | raster index IN THE BAND IMAGE) above quality_level * global maximum: all of
 * them (k_target = 0) or the strongest value bins holding at least k_target */
int km_band_keys_dev(km_ctx *ctx, const uint8_t *d_mask, int H, int W, double quality_level, unsigned global_max_key, size_t k_target,
                     unsigned long long *out_keys, size_t cap, size_t *n_out, size_t *n_total);
/* goodFeaturesToTrack steps 6-8 (rank, greedy minDistance selection, maxCorners) on candidate keys of an H x W image, any order */
int km_select_keys(km_ctx *ctx, const unsigned long long *keys, size_t n, int H, int W, int max_corners, double min_distance,
                   float *out_xy, int cap, int *out_n);
/* The ordering primitives behind the exact paths (k_sort.hip: rank of every candidate in cv::goodFeaturesToTrack's order, row order
 * of frames beyond 32 768 rows), on host buffers.  km_sort_pairs_u64: stable sort of n 64-bit keys in place, ascending or
 * descending; vals (NULL or n 32-bit words) travel with their keys.  km_exclusive_scan_u32: out[i] = sum of in[j], j < i;
 * count_ones != 0: sum of (in[j] == 1) instead. */
int km_sort_pairs_u64(km_ctx *ctx, unsigned long long *keys, unsigned *vals, size_t n, int descending);
int km_exclusive_scan_u32(km_ctx *ctx, const unsigned *in, unsigned *out, size_t n, int count_ones);
/* LK forward + backward of n points in IMAGE coordinates; rows [oy, oy + H) of the H_image-row Laplacian pair are resident.
 * *left_band = 1: a window needed rows outside the band (halo too small for this displacement) */
int km_band_track_dev(km_ctx *ctx, const uint8_t *d_lap_ref, const uint8_t *d_lap_mon, int H, int W, int oy, int H_image,
                      const km_klt_params *prm, const float *p0, int n, float *p1, float *p0r, int *left_band);

#ifdef __cplusplus
}
#endif
#endif /* KARIOS_HIP_H */
