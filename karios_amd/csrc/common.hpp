// Internal declarations shared by the HIP translation units of libkarios_hip.so.
// gfx950 only: 64-lane wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <functional>
#include <string>
#include <vector>

#include "../../include/karios_hip.h"

#define KM_WAVE 64
// candidate keys are appended through KM_NSHARD independent counters (one region of the key buffer each): a single
// counter caps the whole kernel at the ~90 atomics/us one address sustains
#define KM_NSHARD 16
#define KM_TK_NB 2048   // bins of the top-K pre-filter histogram
#define KM_UNITS_MAX 16 // work units one batched submission holds (km_klt_units_frame_submit)

// OpenCV borderInterpolate(p, len, BORDER_REFLECT_101)
__host__ __device__ __forceinline__ int km_reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

enum km_stage {
    ST_MINMAX = 0,
    ST_LAPLACIAN,
    ST_EIGEN,
    ST_CANDIDATES,
    ST_SORT,
    ST_SELECT,
    ST_PYRAMID,
    ST_LK,
    ST_ZNCC,
    ST_FRAME,
    ST_MI,
    ST_PHASE,
    ST_COUNT
};

struct km_buf {
    void *p = nullptr;
    size_t cap = 0;
};

// Grow-only device workspace slots.
enum km_slot {
    WS_RAW_A = 0,   // host-API staging of raw image A (ref)
    WS_RAW_B,       // host-API staging of raw image B (mon)
    WS_MASK_IN,     // host-API staging of a user mask
    WS_U8_A,        // uint8 / Laplacian of ref
    WS_U8_B,        // uint8 / Laplacian of mon
    WS_MASK,        // auto mask
    WS_EIG,         // f32 min-eigenvalue map
    WS_PYR_A,       // pyramid levels >= 1 of image A
    WS_PYR_B,
    WS_KEYS0,       // candidate keys
    WS_KEYS1,       // sort double buffer
    WS_MM_PARTIAL,  // partial minima / maxima of the NEXT unit's rasters (early min/max on the second stream, beside LK of the current unit)
    WS_MM_EARLY,    // ... and their result {min_ref, max_ref, min_mon, max_mon}
    WS_SORT_TMP,    // k_sort.hip: digit tables of the radix sort / tile sums of the scan
    WS_GRID,        // accepted-point grid of the greedy selection
    WS_PTS0,        // p0
    WS_PTS1,        // p1
    WS_PTS2,        // p0r
    WS_SCALARS,     // min/max doubles, counters, max-eig
    WS_PARTIAL,     // reduction partials
    WS_MISC0,
    WS_MISC1,
    WS_MISC2,
    WS_MISC3,
    WS_FFT_A,
    WS_FFT_B,
    WS_FFT_WORK,
    WS_FRAME,
    WS_AUTO,        // batched auto-ksize search: all Laplacians, pyramids, tracks
    WS_FFT_TW0,     // twiddle tables of the float32 FFT (row length of the first / second dimension), kept between calls
    WS_FFT_TW1,
    WS_FFT_TOP2,    // per-row (largest, second-largest) |cc| of the last inverse pass
    WS_MI_TABLE,    // c ln c, c = 0 .. 57^2 (k_mi.hip)
    WS_LAP_VALID,   // k_dense.hip: per-wave counts of valid pixels of the Laplacian pass (summed later, maybe on the second stream)
    WS_FRAME_CNT,   // k_frame.hip: per-workgroup counts of the compaction (a slot of its own: the frame stage of unit k may run beside the Laplacian / eigenvalue kernels of unit k + 1, which own WS_PARTIAL)
    WS_F64_TWX,     // k_fft64.hip: exp(-2 pi i j / W), exp(-2 pi i j / H) ...
    WS_F64_TWY,
    WS_F64_NEGX,    //   ... position of the negated frequency along x / y ...
    WS_F64_NEGY,
    WS_F64_BLUEX,   //   ... and the tables of a Bluestein dimension
    WS_F64_BLUEY,
    WS_F64_MASK,    //   ... which columns / column tiles the inverse needs on the Hermitian half plane
    WS_UNITS_LK,    // k_lk.hip: the per-unit argument table of a batched LK launch
    WS_UNITS_MM,    // api_units.hip: early min / max results of a batch ({min_ref, max_ref, min_mon, max_mon} per unit)
    WS_COUNT
};

// Device-side scalar block (lives in WS_SCALARS).
struct km_scalars {
    double mm[4];             // min_ref, max_ref, min_mon, max_mon
    unsigned long long valid; // mask>0 count
    unsigned int max_eig_key; // ordered-uint encoding of max eig over mask
    unsigned int n_cand;      // candidates emitted (may exceed capacity)
    int n_corners;            // corners selected
    int n_batches;
    float thr;                // maxVal * qualityLevel as f32
    float max_eig;
    unsigned long long argmax_key; // phase correlation arg-max
    unsigned int run_max_key;      // running max-eig key of the fused eig+candidate kernel
    unsigned int pad0;
    unsigned int shard_cnt[KM_NSHARD];  // keys appended per shard (may exceed the shard capacity: overflow)
    unsigned int cut[4];           // top-K pre-filter: D, kept, compaction cursor, exact candidate count
    unsigned int und[8];           // undecided counters of the selection sweeps (one per launch slot)
    unsigned int hist[KM_TK_NB];   // top-K pre-filter histogram (zeroed with the block at the start of a call)
    unsigned int run_max_shard[64]; // sharded running max-eig keys of the 2-px fused kernel (same-address device-scope traffic serialises)
    unsigned int flags;            // KM_FLAG_*: the speculative (no host sync) corner path could not complete, repeat through the exact path
    unsigned int tickets[3];       // "last workgroup finishes the job" counters of the merged launches of k_select2.hip (zeroed with the block)
    unsigned int bin_off[KM_TK_NB]; // k_select2.hip: first slot of every value bin in the kept list
    unsigned int bin_cur[KM_TK_NB]; //                fill cursors of the bins
    unsigned int tie_rows;         // k_eig3.hip: (wavefront, row) steps that took the per-pixel emission path (a lane held two candidates)
    unsigned int pad1;
    unsigned long long skip_total; // development build, "eig3_count": (wavefront, row, pixel slot) triples of the fused 8-px eigenvalue pass ...
    unsigned long long skip_hit;   //   ... of which every lane satisfies min(S_xx, S_yy) scale^2 <= the running lower bound of the threshold
};
#define KM_FLAG_SHARD_OVERFLOW 1u   // a key-buffer shard overflowed
#define KM_FLAG_STAGE_OVERFLOW 2u   // the fused kernel's per-wave key stage overflowed (plateau image)
#define KM_FLAG_KEPT_OVERFLOW 4u    // more keys above the cut than the fixed-capacity kept list holds
#define KM_FLAG_BIN_TOO_LARGE 8u    // a value bin does not fit one workgroup's LDS sort
#define KM_FLAG_CELL_OVERFLOW 16u   // more candidates in one grid cell than its fixed slots
#define KM_FLAG_NOT_CONVERGED 32u   // undecided candidates left after the fixed number of sweeps
#define KM_FLAG_SLICE_SHORT 64u     // fewer than maxCorners corners in the top slice while weaker candidates exist

// km_set_image_window: the buffers handed to the ZNCC / MI kernels as "full images" hold only a window of the real image.
// Key-point coordinates stay IMAGE coordinates (the float32 sum x0 + dx the reference rounds depends on their magnitude), the
// bounds rule is the image's, the pixels are fetched relative to (ox, oy).  H == 0: no window.
struct km_window {
    int ox = 0, oy = 0, H = 0, W = 0;
};

// One frame in flight of km_klt_tile_frame_submit: pinned host block, completion event.
#define KM_LANE_EVENTS 8      // events of one lane of the batched units' software pipeline (api_units.hip)
#define KM_FRAME_SLOTS 3
struct km_frame_slot {
    void *host = nullptr;
    size_t cap = 0, bytes = 0;
    hipEvent_t done = nullptr;
    hipEvent_t sunk = nullptr;      // the frame's block is in the frame sink (km_set_frame_sink): recorded on the compute stream behind the copy
    bool sunk_valid = false;
    std::atomic<int> pending{0};
    std::atomic<int> deferred{0};   // pipelined batched submission: the tail (LK .. copy-out) of this frame has not been enqueued yet
};

// staging.hip: page-locked ring between caller memory and the device (no runtime copy ever reads or writes pageable memory)
#define KM_RING_SLOTS 4
struct km_ring_slot {
    void *buf = nullptr;
    hipEvent_t done = nullptr;      // the DMA that reads the slot
    bool busy = false;
};
struct km_stage_ring {
    km_ring_slot slot[KM_RING_SLOTS];
    size_t chunk = 0;
    int next = 0;
};
struct km_land_job { void *dst; const void *pinned; size_t bytes; };
struct km_chk_job { const char *what; const void *host; size_t elem; int H, W; ptrdiff_t stride; const void *d; size_t off; };

struct km_ctx {
    int device = 0;
    int n_cu = 256;            // compute units of the device (wave slots = n_cu * 4 SIMDs * waves per SIMD)
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // km_upload_async: uploads of the next pair / tile under the compute of the current one
    hipEvent_t ev_copy = nullptr;
    hipStream_t aux_stream = nullptr;    // sync-free tile path: the pyramids (they depend on the Laplacians only) run here next to the
    hipStream_t d2h_stream = nullptr;    // km_klt_tile_frame_submit: the finished frame block travels to the host here
    // ---- software-pipelined batched submissions ("units_pipeline", api_units.hip): two LANES (workspace sets) alternate; the dense
    // kernels of both lanes interleave on `stream`, the latency-bound chains (corner selection; frame stage + scores) run on
    // `chain_stream` beside the other lane's dense kernels; a submission's tail (LK, frame stage, scores, copy-out) is enqueued by
    // the NEXT submission (or km_frame_flush)
    hipStream_t chain_stream = nullptr;
    hipEvent_t ev_lane[2][KM_LANE_EVENTS] = {};
    bool lane_f_recorded[2] = {false, false};   // EV_F_DONE of the lane has been recorded: its next submission waits for it
    bool in_units_submit = false;
    bool opt_units_pipeline = false;
    int lane = 0;                        // workspace set km_ws hands out (0: `ws`, 1: `ws_b`)
    struct km_units_tail *utail = nullptr;   // the deferred tail (armed: enqueue pending)
    std::mutex *enqueue_mu = nullptr;    // submit / flush / the fallback flush of km_frame_wait
    hipEvent_t ev_tail = nullptr, frame_copy = nullptr;   // frame_copy: completion of the last block copy (WS_FRAME must not be rewritten before)
    hipEvent_t ev_lk_start = nullptr, ev_mm = nullptr;   // early min/max: the next unit's K1 starts on aux_stream when this unit's LK launch starts
    bool lk_start_valid = false, lk_start_prev = false;   //   ... ev_lk_start was recorded by the tile call that directly preceded this one
    bool mm_early_allowed = false;       //   ... the running entry point reads no min / max statistics back (km_klt_tile_frame_submit)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // chain of small corner-selection kernels on `stream`, joined before LK
    bool copy_pending = false;           // uploads queued since the compute stream last waited for the whole copy stream
    std::vector<hipEvent_t> upload_marks;   // km_upload_mark tickets: events on the copy stream, nullptr = ticket consumed
    std::vector<hipEvent_t> free_marks;
    km_window window;                    // km_set_image_window
    size_t band_capk = 0;                // km_band_eigen_dev -> km_band_keys_dev: capacity of the candidate keys left in WS_KEYS0
    bool band_fused = false;             //   ... emitted by the fused kernel (else: eig map in WS_EIG, candidates still to be scanned)
    void *frame_sink = nullptr;          // km_set_frame_sink: device-side copy of every frame block
    size_t frame_sink_cap = 0;
    size_t frame_sink_pitch = 0;         // km_set_frame_sink_pitch: distance between the blocks of a batched submission (0: block size)
    km_buf ws[WS_COUNT];
    km_buf ws_b[WS_COUNT];               // lane 1 of the pipelined batched submissions
    std::vector<void *> retired;         // workspace buffers replaced by larger ones (km_ws): freed at the next km_ctx_sync / destroy
    size_t retired_mark = 0;             // ... the first `retired_mark` of them were retired before the running entry point began (begin_call)
    km_stage_ring ring;                  // host -> device staging (staging.hip)
    void *land = nullptr;                // device -> host landing arena (page-locked), km_d2h_queue / km_d2h_flush
    size_t land_cap = 0, land_used = 0;
    hipEvent_t land_ev[2] = {nullptr, nullptr};
    std::vector<km_land_job> land_jobs;
    void *chk_dev = nullptr, *chk_host = nullptr;   // KARIOS_HIP_UPLOAD_CHECKSUM: row checksums seen by a kernel right behind each upload
    size_t chk_used = 0;
    std::vector<km_chk_job> chk_jobs;
    long long chk_armed_total = 0, chk_miss_total = 0;
    std::string err;
    // Independent device work queued by the caller to fill the GPU while the host waits for a small read-back (the two
    // synchronisations of the corner selection): km_wait_readback runs ONE deferred job between the copy and the wait.
    std::vector<std::function<int()>> deferred;
    hipEvent_t ev_readback = nullptr;
    bool profiling = false;
    int fused_eig = 1;         // km_set_option("fused_eig"): 1 = K3+K4 fused (k_eig2.hip, no eig map), 0 = eig map + candidate scan
    // test knobs (km_set_option, 0 = default): shrunken capacities that force the corner detector's retry paths
    int opt_key_cap = 0;       // "key_cap": candidate keys per shard of the first attempt (forces the regrow path)
    int opt_stage_cap = 0;     // "stage_cap": usable slots of the fused kernel's per-wave LDS stage (forces the two-kernel fallback)
    int opt_topk_factor = 0;   // "topk_factor": the top-K pre-filter keeps factor * maxCorners keys (default 8; 1 forces the second selection pass)
    int opt_select_first = 0;  // "select_first": first prefix of the selection sweeps = value candidates (default 3 * maxCorners; small values force prefix growth)
    bool opt_speculative = true;   // "speculative" 1 (default): corners through the synchronisation-free, sort-free path (k_select2.hip) where a tile entry point can repeat a flagged tile; 0: always the exact path (k_select.hip)
    bool opt_aux_pyramid = true;   // "aux_pyramid": sync-free tile path builds the pyramids on a second stream, next to the corner selection
    bool opt_aux_early = true;     // "aux_early": ... forked before the fused eigenvalue pass (0: behind it, next to the selection chain only)
    bool opt_aux_priority = false; // "aux_priority" 1: the second stream gets the lowest priority.  Off: it bought nothing for one context (the start event in
                                   // front of the fork already lets the main stream's kernel go first) and with several contexts on one GPU it made the
                                   // selection sweeps of one context wait on the others (tiles flagged as unconverged and repeated)
    bool opt_eig3 = true;          // "eig3": fused eig + candidate pass with 8 pixels per lane where the image is >= 512 wide (0: always the 2-px kernel)
    int opt_profile_every = 1;     // "profile_every": with profiling on, only every N-th tile call records stage events
    unsigned profile_tick = 0;
    bool profile_skip = false;
    int opt_profile_stage = -1;    // "profile_stage": with profiling on, time only this stage (-1: every stage; each timed span costs two events = two pipeline drains)
    int opt_stash_cap = 0;         // "stash_cap": kept keys a workgroup of the scatter launch stashes in LDS (small values force its second read of the keys)
    int opt_spec_flag = 0;         // "spec_flag": KM_FLAG_* bits raised artificially by the speculative path (tests of the repeat logic)
    const unsigned *eig_partial = nullptr;   // speculative path: per-wave maxima the fused eigenvalue pass left for the ranking's first launch
    unsigned eig_npartial = 0;               //   (the one-workgroup reduction launch in between is skipped), consumed by kf_rank
    bool eig_defer_max = false;
    bool spec_used = false;        // the running call went through the speculative corner path
    unsigned spec_flags = 0;       // sc->flags of the speculative run, once read back
    bool spec_allowed = false;     // set by the entry points that check sc->flags with their result (and cleared for the repeat)
    bool opt_roctx = false;        // "roctx": roctx ranges around the stages
    bool opt_fft_cross_fused = true;   // "fft_cross": the cross-power step fused into the first inverse pass's row load (61 M rows)
    bool opt_fft_herm = true;      // "fft_herm" 1 (default): the inverse transform of the float32 phase correlation works on the Hermitian half plane (rows of length 61 M on both sides)
    bool opt_fft61 = true;         // "fft61" 1 (default): rows of length 61 M through the wave-local form (k_fft.hip, second form); 0: the Stockham kernel
    bool opt_phase_fp64 = false;   // "phase_fp64" 1: phase correlation always in double precision, k_fft64.hip (the reference's precision)
    bool opt_lk2 = true;           // "lk2" 1 (default): LK on four resident patches per key point (two-level pyramids); 0: the first form
    bool opt_mm_early = true;      // "mm_early" 0: min / max of a submitted unit on the main stream behind the previous unit's tail (round-2 order)
    bool opt_frame_mi = false;     // "frame_mi" 1: frames scored by the tile entry points (ZNCC of the rows with score >= threshold) also carry the two mutual-information scores of those rows (core.py:894-907): two more float64 columns behind zncc
    bool opt_no_defer = false; // "defer" 0: the deferred pyramid jobs run after the read-back waits instead of under them
    // stage-timer events: set 0 serves the synchronous calls, sets 1..KM_FRAME_SLOTS the frames in flight of
    // km_klt_tile_frame_submit (a frame's spans are read after ITS completion, while the next one is already recording)
    hipEvent_t evs[KM_FRAME_SLOTS + 1][ST_COUNT][2];
    bool evs_used[KM_FRAME_SLOTS + 1][ST_COUNT];
    int ev_cur = 0;
    bool ev_ready = false;
    km_frame_slot fslot[KM_FRAME_SLOTS];
    int fslot_next = 0;
    // pinned landing zone of the small mid-pipeline read-backs: a copy into pageable memory blocks the host until the
    // stream has drained, so nothing could be queued behind it (the deferred jobs of km_wait_readback arrived too late)
    void *pinned_rb = nullptr;
    size_t pinned_rb_cap = 0;
    km_klt_stats stats;
    int phase_path = 0;            // last km_phase_shift*: 1 = float32 hand-written FFT, 2 = double precision (k_fft64.hip)
    double phase_margin = 0.0;     // (max - second largest) / max of |cc| seen by the float32 path
    bool defer_valid_sum = false;  // set around the Laplacian pass of a unit on the sync-free path: the valid-pixel sum becomes a job for the second stream
    bool valid_job_pending = false;
    const unsigned *valid_job_partial = nullptr;
    unsigned valid_job_n = 0;
    unsigned long long *valid_job_out = nullptr;
    bool opt_eig3_count = false;   // (KM_DEV) "eig3_count": the fused 8-px eigenvalue pass counts the (wave, row, pixel slot) triples a bound could skip
    bool opt_defer_valid = true;   // "defer_valid" 0: the sum stays behind the Laplacian pass on the main stream
    int f64_h = 0, f64_w = 0;      // shape whose tables sit in WS_F64_TW* / WS_F64_NEG* (k_fft64.hip)
    int opt_f64_prime_t = 0;       // "f64_prime_t": cap on the transforms per tile of the prime level kernel (0: as many as fit, <= 64)
    int opt_f64_smooth_t = 0;      // "f64_smooth_t": the same for the smooth level kernel (default 8)
    bool opt_f64_pair = true;      // "f64_pair" 1 (default): the inverse along the rows of the float64 transform packs two image rows into one complex transform
    bool opt_f64_half = true;      // "f64_half" 1 (default): ... and the inverse column levels only run the columns kx <= W / 2 (the rest follows by symmetry)
    bool opt_f64_plain = false;    // "f64_plain" 1: pack pass in front, two arg-max passes behind (instead of fusing both ends into the level kernels)
    int fft_tw_n[2] = {0, 0};      // row lengths whose twiddle tables sit in WS_FFT_TW0 / WS_FFT_TW1
    bool mi_table_ready = false;   // WS_MI_TABLE holds its table
    int fft_tw_m[2] = {-1, -1};    //   ... and the 61 M plan they were laid out for (0: Stockham table only)
};

int km_fail(km_ctx *ctx, int code, const char *fmt, ...);
// KARIOS_HIP_* TUNING / DEBUGGING variables (row counts of the marching kernels, grid sizes, workspace poisoning ...) only exist in the
// development build (make DEV=1 -> -DKM_DEV); the release library reads the variables include/karios_hip.h documents and nothing else
#ifdef KM_DEV
static inline const char *km_dev_env(const char *name) { return getenv(name); }
#else
static inline const char *km_dev_env(const char *) { return nullptr; }
#endif
// km_set_option("roctx", 1): every stage's host-side enqueue span becomes a roctx range (rocprofv3 --marker-trace); the library
// is looked up at run time (librocprofiler-sdk-roctx.so / libroctx64.so), nothing is linked
void km_roctx_push(int stage);
void km_roctx_pop();
void *km_pinned_rb(km_ctx *c, size_t bytes);   // >= bytes of pinned host memory owned by the context (nullptr on failure)

// MI355X dispatches workgroup w of a grid to XCD w % 8, and every XCD has its own L2.  The marching kernels therefore
// launch km_xcd_grid(ntiles) workgroups and let workgroup w process tile (w % 8) * ceil(ntiles / 8) + w / 8: each XCD
// then owns one contiguous range of tiles (neighbouring column strips and row blocks share their halo in ONE L2 instead of
// being fetched from HBM once per XCD).  Returns false for the padding workgroups.
#define KM_XCDS 8u
static inline unsigned km_xcd_grid(unsigned ntiles) { return ((ntiles + KM_XCDS - 1) / KM_XCDS) * KM_XCDS; }
#ifdef __HIPCC__
// One pixel of a chip through a buffer descriptor: `rs` = the chip's origin (wave-uniform), voff = the lane's byte offset inside a row,
// soff = the row's byte offset (scalar): `buffer_load_ushort v, v_off, s[rs], s_row offen` - no vector instruction goes into addresses.
template <typename T>
__device__ __forceinline__ unsigned km_chip_px(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    if constexpr (sizeof(T) == 1) return (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rs, voff, soff, 0);
    else return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
}
__device__ __forceinline__ bool km_xcd_tile(unsigned ntiles, unsigned &tile)
{
    const unsigned per = (ntiles + KM_XCDS - 1) / KM_XCDS, w = blockIdx.x;
    tile = (w % KM_XCDS) * per + w / KM_XCDS;
    return tile < ntiles;
}
#endif

// Rows per work item of a marching kernel (one wavefront walks `rows` output rows of one column strip, re-reading
// `halo` rows): the value in [lo, hi] that minimises  ceil(items / wave_slots) * (rows + halo), i.e. whole rounds of
// resident waves times the work of one item - a grid that needs 2.02 rounds costs three.
static inline int km_pick_rows(int H, int nstrips, int halo, long wave_slots, int lo, int hi)
{
    int best = lo;
    double best_cost = 1e300;
    for (int r = lo; r <= hi; r++) {
        const long items = (long)nstrips * ((H + r - 1) / r);
        const long rounds = (items + wave_slots - 1) / wave_slots;
        // a thinly filled last round still lasts about half an item (fewer waves per SIMD, each one faster)
        const double last = (double)(items - (rounds - 1) * wave_slots) / (double)wave_slots;
        const double cost = ((double)(rounds - 1) + 0.5 + 0.5 * last) * (double)(r + halo);
        if (cost < best_cost) { best_cost = cost; best = r; }
    }
    return best;
}
void *km_ws(km_ctx *ctx, int slot, size_t bytes);  // nullptr on failure (error set); the slot of the context's current lane
void *km_ws_peek(km_ctx *ctx, int slot);           // the slot's current buffer (no growth)
int km_units_flush(km_ctx *c, bool join = true);   // api_units.hip: enqueue the deferred tail of a pipelined batched submission (no-op without one)
void km_units_free(km_ctx *c);                     // ... its host-side state (km_ctx_destroy)
// staging.hip.  Host -> device: returns when `src` has been read completely; the DMAs (from the ring) are ordered on stream s.
int km_h2d_staged(km_ctx *c, hipStream_t s, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows);
static inline int km_h2d_small(km_ctx *c, void *dst, const void *src, size_t bytes) { return km_h2d_staged(c, c->stream, dst, bytes, src, bytes, bytes, 1); }
// Device -> host on c->stream: queue any number of results, then flush (completes the stream, copies them out of the landing arena)
int km_d2h_queue(km_ctx *c, void *dst, const void *d_src, size_t bytes);
int km_d2h_flush(km_ctx *c);
void km_ring_destroy(km_ctx *c);
bool km_upload_check_enabled();
int km_upload_check_arm(km_ctx *c, const char *what, const void *host, size_t elem, int H, int W, ptrdiff_t stride, const void *d);
int km_upload_check_verify(km_ctx *c);
void km_upload_check_drop(km_ctx *c);
// Call right after queuing device-to-host copies the host needs NOW: records an event, queues one deferred job (if any)
// behind it, then waits for the event only - the GPU keeps working on the job while the host continues.
int km_wait_readback(km_ctx *ctx);
int km_run_deferred(km_ctx *ctx);   // runs every job still pending (call before results that depend on them are used)

#define KM_HIP(ctx, call)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return km_fail((ctx), KM_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,    \
                           hipGetErrorString(e_));                                          \
    } while (0)

#define KM_LAUNCH_CHECK(ctx) KM_HIP(ctx, hipGetLastError())

struct km_stage_timer {
    km_ctx *c;
    int s;
    bool marked = false;
    km_stage_timer(km_ctx *ctx, int stage) : c(ctx), s(stage)
    {
        if (c->opt_roctx) { km_roctx_push(stage); marked = true; }
        if ((c->opt_profile_stage >= 0 && stage != c->opt_profile_stage) || c->profile_skip) { s = -1; return; }
        if (c->profiling && c->ev_ready && !c->evs_used[c->ev_cur][s]) {
            (void)hipEventRecord(c->evs[c->ev_cur][s][0], c->stream);
        } else if (c->profiling && c->ev_ready) {
            s = -1;  // stage already timed in this call (only the first span is kept)
        }
    }
    ~km_stage_timer()
    {
        if (marked) km_roctx_pop();
        if (s >= 0 && c->profiling && c->ev_ready) {
            (void)hipEventRecord(c->evs[c->ev_cur][s][1], c->stream);
            c->evs_used[c->ev_cur][s] = true;
        }
    }
};

static inline size_t km_dtype_size(int dtype)
{
    switch (dtype) {
    case KM_U8: return 1;
    case KM_U16: return 2;
    case KM_I16: return 2;
    case KM_F32: return 4;
    default: return 0;   // KM_F64 / KM_I32 / KM_U32 are read by km_zncc_windows only (km_any_dtype_size)
    }
}

static inline size_t km_any_dtype_size(int dtype)
{
    return dtype == KM_F64 ? 8 : (dtype == KM_I32 || dtype == KM_U32) ? 4 : km_dtype_size(dtype);
}

// per-unit arguments of the batched scoring kernels (k_zncc.hip, k_mi.hip): by value in the kernel arguments
struct km_score_unit {
    const void *ref, *mon;                // rasters the chips are cut from
    const float *x0, *y0, *dx, *dy, *score;
    const int *d_n;
    double *out, *out2;                   // ZNCC: out; MI: out = mutual_info_score, out2 = mi_score
    ptrdiff_t sref, smon;
    int Href, Wref, Hmon, Wmon;
    km_window win;
};
struct km_score_units {
    km_score_unit u[KM_UNITS_MAX];
};

// ---- launchers implemented in the kernel translation units (all asynchronous) ----
// k_dense.hip
int kd_minmax(km_ctx *c, const void *d_img, int dtype, int H, int W, ptrdiff_t stride,
              double *d_mm /* [2] */);
int kd_minmax_pair_ws(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *d_mm, int ws_slot);
int kd_minmax_pair(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb,
                   double *d_mm /* [4]: min_a, max_a, min_b, max_b */);
int kd_to_uint8(km_ctx *c, const void *d_img, int dtype, int H, int W, ptrdiff_t stride,
                const double *d_mm, int invert, uint8_t *d_out);
int kd_auto_mask(km_ctx *c, const void *d_mon, const void *d_ref, int dtype, int H, int W,
                 ptrdiff_t stride_mon, ptrdiff_t stride_ref, const double *nodata_mon,
                 const double *nodata_ref, uint8_t *d_mask, unsigned long long *d_valid);
int kd_count_nonzero(km_ctx *c, const uint8_t *d_mask, size_t n, unsigned long long *d_valid);
int kd_run_valid_sum(km_ctx *c);
int kd_laplacian_u8(km_ctx *c, const uint8_t *d_src, int H, int W, int ksize, uint8_t *d_dst);
// fused: raw ref+mon -> uint8 stretch -> Laplacians (+ auto mask when d_mask_out != null)
int kd_stretch_laplacian_pair(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H,
                              int W, ptrdiff_t stride_ref, ptrdiff_t stride_mon,
                              const double *d_mm /* [4] */, int ksize_ref, int ksize_mon,
                              int invert_mon, const double *nodata_ref, const double *nodata_mon,
                              uint8_t *d_lap_ref, uint8_t *d_lap_mon, uint8_t *d_mask_out,
                              unsigned long long *d_valid);
int kd_min_eigen(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block,
                 float *d_eig, unsigned int *d_max_key);
// k_frame.hip: DN-value filter of the key points (core.py:650-737)
// count of tracks passing the forward-backward test (max |p0 - p0r| < thr) among the first min(*d_n, n_max) points
int kf_count_kept(km_ctx *c, const float *d_p0, const float *d_p0r, const int *d_n, int n_max, float back_thr, int *d_count);
int kf_dn_keep(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon, const float *d_x0,
               const float *d_y0, int n, const double *d_no_values, int n_no, const double *ref_nd, const double *mon_nd, uint8_t *d_keep);
// k_eig2.hip: minimum-eigenvalue map + masked maximum, 2 pixels per lane (KM_E_UNSUPPORTED when the case is not covered)
int k2_min_eigen(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, float *d_eig, unsigned *d_max_key);
int k2_eig_candidates(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, double quality, km_scalars *sc,
                      unsigned long long *d_keys, size_t cap, bool rezero);
// k_eig3.hip
int k3_eig_candidates(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, double quality, km_scalars *sc,
                      unsigned long long *d_keys, size_t cap);
int kd_candidates(km_ctx *c, const float *d_eig, const uint8_t *d_mask, int H, int W,
                  double quality, km_scalars *d_sc, unsigned long long *d_keys, size_t cap, bool rezero);
int kd_pyrdown_u8(km_ctx *c, const uint8_t *d_src, int H, int W, uint8_t *d_dst);
int kd_pyrdown_u8_pair(km_ctx *c, const uint8_t *d_src_a, const uint8_t *d_src_b, int H, int W, uint8_t *d_dst_a, uint8_t *d_dst_b);
int kd_shift_image(km_ctx *c, const void *d_img, int elem_size, int H, int W, ptrdiff_t stride,
                   int y_off, int x_off, void *d_out);
// k_select.hip
int ks_sort_keys_desc(km_ctx *c, unsigned long long *d_keys, size_t n, unsigned long long **d_sorted);
// k_sort.hip: the ordering primitives of the exact fallback paths (hand-written; no library)
enum { KM_SCAN_PLAIN = 0, KM_SCAN_IS_ONE = 1 };
int km_sort_u64(km_ctx *c, unsigned long long *keys_a, unsigned long long *keys_b, unsigned *vals_a, unsigned *vals_b, size_t n, bool descending);
int km_exclusive_scan(km_ctx *c, const unsigned *in, unsigned *out, size_t n, int mode, int tmp_slot);
int ks_select(km_ctx *c, const unsigned long long *d_sorted, size_t n, int H, int W,
              int max_corners, double min_distance, float *d_xy, int cap, km_scalars *d_sc, int *n_found, bool fresh_scalars);
int ks_topk_prefilter(km_ctx *c, const unsigned long long *d_keys, size_t cap_keys, size_t k_target, km_scalars *d_sc, double quality,
                      unsigned long long **d_kept, size_t *n_kept, size_t *n_total, km_scalars *hs, bool rezero);
// k_select2.hip: ranking + selection without host synchronisation or library sorts (flags instead of retries)
size_t kf_kept_capacity(int max_corners);
int kf_rank(km_ctx *c, const unsigned long long *d_keys, size_t cap_keys, int H, int W, int max_corners, double quality, double min_distance,
            km_scalars *sc);
int kf_select(km_ctx *c, int H, int W, int max_corners, double min_distance, float *d_xy, int cap, km_scalars *sc);
// k_lk.hip
struct km_pyr {
    const uint8_t *img[5];
    int H[5], W[5];
    int levels;  // highest level index actually built (<= max_level)
    // Row band of a larger image (single-tile multi-GPU mode): only rows [oy, oy + Hres) of every level are resident, H stays
    // the height of the WHOLE level and img points at the (virtual) row 0, i.e. resident pointer - oy * W.  Hres == 0: all rows.
    int oy[5] = {0, 0, 0, 0, 0}, Hres[5] = {0, 0, 0, 0, 0};
};
int kl_track(km_ctx *c, const km_pyr &A, const km_pyr &B, const float *d_pts_in, const int *d_n,
             int n_max, int win, int max_count, double epsilon, bool backward_too, float *d_p1,
             float *d_p0r, int *d_left_band = nullptr);
int kl_oscillation_probe(km_ctx *c, const float *d_q, int n, uint8_t *d_out);   // test hook of the LK kernels' oscillation predicate
// independent forward + backward tracker runs in one launch (k_lk.hip kl_jobs_launch; the kernel-size search of klt.py:465-545)
#define KM_LK_JOBS_MAX 64
struct km_lk_job {
    km_pyr A, B;                  // pyramids of the image the points live in / the image they are tracked into
    const float *pts_in;
    const int *d_n;
    float *p1, *p0r;
};
int kl_jobs_launch(km_ctx *c, const km_lk_job *jobs, int n_jobs, int n_max, int win, int max_count, double epsilon);
// forward-backward inlier counts of n_jobs tracker runs in one launch: counts[j] += tracks of job j with max |p0 - p0r| < thr
struct km_count_jobs {
    const float *p0[KM_LK_JOBS_MAX], *p0r[KM_LK_JOBS_MAX];
    const int *d_n[KM_LK_JOBS_MAX];
};
int kf_count_kept_jobs(km_ctx *c, const km_count_jobs &J, int n_jobs, int n_max, float back_thr, int *d_counts);
// ---- batched units (km_klt_units_frame_submit, api_units.hip): U independent work units - tiles of one or several pairs, klt.py:220-253 -
// through ONE set of launches: the dense kernels take the unit as part of their linear work-item space (items tall enough to amortise
// their halo, one launch's worth of waves instead of U thin ones), the corner-selection chain runs its U latency chains side by side
// (unit = blockIdx.z), LK / frame / ZNCC / MI take the unit as blockIdx.y.  The per-unit tables travel BY VALUE in the kernel
// arguments (<= 4 KB), so nothing has to be uploaded or kept alive.
struct km_units {
    int n = 0;                                   // units of the batch (<= KM_UNITS_MAX)
    int dtype = 0;
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX];
    const void *ref[KM_UNITS_MAX], *mon[KM_UNITS_MAX];      // the unit's box in the two rasters
    ptrdiff_t sref[KM_UNITS_MAX], smon[KM_UNITS_MAX];
    float x_off[KM_UNITS_MAX], y_off[KM_UNITS_MAX];
    // per-unit workspace (slices of the context's slots)
    km_scalars *sc[KM_UNITS_MAX];
    const double *mm[KM_UNITS_MAX];              // {min_ref, max_ref, min_mon, max_mon} the stretch reads (sc->mm, or the early slot)
    uint8_t *lap_ref[KM_UNITS_MAX], *lap_mon[KM_UNITS_MAX], *mask[KM_UNITS_MAX];
    const uint8_t *user_mask[KM_UNITS_MAX];      // user mask of the box (nullptr: the automatic mask is derived, klt.py:268-273) ...
    ptrdiff_t user_smask[KM_UNITS_MAX];          // ... and its row stride; it is packed into `mask` (dense rows) by the Laplacian pass
    bool has_user_mask = false;
    unsigned long long *keys[KM_UNITS_MAX];      // candidate keys (KM_NSHARD regions of capk / KM_NSHARD slots)
    size_t capk = 0;                             // ... the same capacity for every unit (sized for the largest)
    const unsigned *eig_partial[KM_UNITS_MAX];   // per-wave maxima the fused eigenvalue pass leaves for the ranking's first launch
    unsigned eig_npartial[KM_UNITS_MAX];
    km_pyr A[KM_UNITS_MAX], B[KM_UNITS_MAX];     // pyramids (ref, mon Laplacians)
    float *p0[KM_UNITS_MAX], *p1[KM_UNITS_MAX], *p0r[KM_UNITS_MAX];
    char *frame[KM_UNITS_MAX];                   // frame block (header | 6 cap float32 | score columns)
    // scoring (rasters the chips are cut from; window of km_set_image_window per unit)
    const void *ref_full[KM_UNITS_MAX], *mon_full[KM_UNITS_MAX];
    int Hf[KM_UNITS_MAX], Wf[KM_UNITS_MAX];
    ptrdiff_t sref_f[KM_UNITS_MAX], smon_f[KM_UNITS_MAX];
    km_window win[KM_UNITS_MAX];
};

// deferred per-unit sums of the valid-pixel counts of the batched Laplacian pass
struct km_valid_units {
    int n = 0;
    const unsigned *partial[KM_UNITS_MAX];
    unsigned n_partial[KM_UNITS_MAX];
    unsigned long long *out[KM_UNITS_MAX];
};
// batched launchers (KM_E_UNSUPPORTED without a message: the batch form does not cover the case - the caller submits the units one by one)
int kd_minmax_units(km_ctx *c, const km_units &U, double *const *d_out, int ws_slot);
int kd_stretch_laplacian_units(km_ctx *c, const km_units &U, int ksize_ref, int ksize_mon, int invert_mon, const double *nodata_ref,
                               const double *nodata_mon, km_valid_units *job);
int kd_valid_sum_units(km_ctx *c, const km_valid_units &J);
int k3_eig_candidates_units(km_ctx *c, km_units &U, int block, double quality);
int kd_pyrdown_units(km_ctx *c, const km_units &U, int level);
int kf_rank_select_units(km_ctx *c, const km_units &U, int max_corners, double quality, double min_distance, int cap);
int kl_units_prepare(km_ctx *c, const km_units &U, int n_max, int win, int max_count, double epsilon);
int kl_units_launch(km_ctx *c, int n_units, int n_max, int win);
int kf_frame_units(km_ctx *c, const km_units &U, int n_max, int cap, float back_thr);
int kz_zncc_units(km_ctx *c, const km_score_units &A, int n_units, int dtype, int n, float score_thr);
int kmi_units(km_ctx *c, const km_score_units &A, int n_units, int dtype, int n, float score_thr);
// k_frame.hip
int kf_frame(km_ctx *c, const float *d_p0, const float *d_p1, const float *d_p0r, const int *d_n, int n_max, int cap, float back_thr,
             float x_off, float y_off, void *d_out, const km_scalars *d_sc_header = nullptr, int width = 0);   // d_sc_header: header words 2 / 3 = flags, candidate count; width: corner columns < width (0: unknown)
int kf_row_checksum(km_ctx *c, const void *d_img, size_t row_bytes, int rows, unsigned long long *d_out);   // diagnosis of host-buffer uploads (staging.hip)
// k_zncc.hip
int kz_zncc(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref,
            int Hmon, int Wmon, ptrdiff_t stride_ref, ptrdiff_t stride_mon, const float *d_x0,
            const float *d_y0, const float *d_dx, const float *d_dy, int n, double *d_out);
int kz_zncc_filtered(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon,
                     ptrdiff_t stride_ref, ptrdiff_t stride_mon, const float *d_x0, const float *d_y0, const float *d_dx,
                     const float *d_dy, int n, const int *d_n, const float *d_score, float score_thr, double *d_out);   // honours c->window
int kz_zncc_windows(km_ctx *c, const void *d_img1, const void *d_img2, int dt1, int dt2, int H1, int W1, int H2, int W2, ptrdiff_t s1, ptrdiff_t s2,
                    const int *d_uv, int half, int count, double *d_out, uint8_t *d_flags);
// k_mi.hip
int kmi_batch(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t stride_ref,
              ptrdiff_t stride_mon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, const int *d_n,
              const float *d_score, float score_thr, double *d_studholme, double *d_nmi);
// k_phase.hip
int kp_phase_shift(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W,
                   ptrdiff_t stride_a, ptrdiff_t stride_b, double out_rc[2]);
void kp_destroy(km_ctx *c);
// k_fft64.hip: the double-precision evaluation (any side length)
int kp_phase_shift_f64(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                       double out_rc[2]);
// k_fft.hip: hand-written float32 phase correlation (sides with prime factors in {2,3,5,7,61}, <= 12288)
bool kp_fast_supported(int H, int W);
int kp_phase_shift_fast(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                        double out_rc[2], double *margin);
