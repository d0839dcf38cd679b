// km_klt_units_frame_submit: U independent work units - the tiles of `KLT.match` (reference karios/matcher/klt.py:220-253: no halo, per-tile
// uint8 stretch, per-tile quality threshold and maxCorners), of one pair or of several bands - through ONE device pipeline.
//
// Why.  Submitted one by one, every 30-Mpx unit pays the fixed costs of 20 000 corners serially: a corner-selection chain of nine
// latency-bound launches (0.12 ms at 4 % VALU occupancy), an LK launch with its fill + drain, three small frame launches, and dense
// kernels whose items are too short to amortise their halo (a 5490^2 unit fills two thirds of ONE round of resident waves).  Here
//   * min / max, stretch + Laplacians + mask, the fused eigenvalue pass and the pyramids take the unit as part of their linear item
//     space: one launch each, items as tall as a single 10980^2 tile's;
//   * the selection chain runs its U latency chains side by side (unit = blockIdx.z): nine launches for all units;
//   * ONE LK launch tracks every unit's corners, one frame / ZNCC / MI launch each scores them;
//   * the U frame blocks leave in one copy (host slot) / one strided copy (frame sink: the send buffer of the all-gather).
// Results are the unit-by-unit results bit for bit (tests/test_gpu_units.py): every kernel runs exactly the single-unit item code on
// the unit's own rasters, scalar block, key buffer and grid.  A unit the fixed capacities of the synchronisation-free corner path do
// not fit is flagged in its block's header (word 2) and repeated exactly by the caller, as for km_klt_tile_frame_submit.
#include "api_internal.hpp"

#include <cstring>

static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- software pipeline over consecutive submissions ("units_pipeline" 1; karios_amd.stream.FrameStream switches it on)
// A submission is three instruction-bound dense stages - L (stretch + Laplacians + mask), E (fused eigenvalue pass), K (LK) - and two
// latency-bound chains - C (corner selection: nine launches, between E and K) and F (FB test / frame order / ZNCC / MI, behind K).  In
// stream order the chains leave the GPU idle: 0.57 ms of a 3.5-ms submission of four 10980^2 pairs (profiles/timeline_r06_a.txt).  With
// the pipeline on, two LANES (workspace sets) alternate and the dense stages of consecutive submissions interleave on the main stream,
//     ... o.E | n.L | o.K | n.E | n'.L | n.K | n'.E ...            (o = previous submission, n = this one, n' = the next)
// while the chains run on a second stream beside the other lane's dense kernels: o.C beside n.L, o.F beside n.E.  The tail of a
// submission (K, F, copy-out) is therefore enqueued by the NEXT submission - or by km_frame_flush / any other entry point / km_ctx_sync.
// The chains' workgroups are 256 threads (k_select2.hip KF_T): they take the slots single retiring dense workgroups leave.
// Same kernels on the same data in the same per-submission order: frames are bit-identical to the unpipelined form.
enum { EV_MM = 0, EV_FORK, EV_JOIN, EV_E_DONE, EV_C_DONE, EV_K_DONE, EV_F_DONE, EV_SETUP };
static_assert(EV_SETUP < KM_LANE_EVENTS, "lane events");

struct km_units_tail {
    bool armed = false;
    int lane = 0, slot = 0;
    km_units U;
    km_klt_params prm;
    int n = 0, n_max = 0, cap = 0, dtype = 0;
    bool with_zncc = false, with_mi = false, piped = false, profile_skip = false;
    double zncc_threshold = 0.0;
    size_t fb = 0, ob = 0, ob_al = 0;
    char *d_out = nullptr;
    void *sink = nullptr;                 // the frame sink as it was set at submission time
    size_t sink_cap = 0, sink_pitch = 0;
};

// K, F and the copy-out of a submission.  piped: K on the main stream behind the lane's chain, F on the chain stream behind K.
static int units_tail(km_ctx *c, km_units_tail &T)
{
    int rc;
    km_units &U = T.U;
    const int n = T.n;
    km_frame_slot *slot = &c->fslot[T.slot];
    hipStream_t const main_stream = c->stream;
    const int lane_before = c->lane, ev_before = c->ev_cur;
    struct restore_t { km_ctx *c; hipStream_t s; int lane, ev; bool skip; ~restore_t() { c->stream = s; c->lane = lane; c->ev_cur = ev; c->profile_skip = skip; } }
        restore{c, main_stream, lane_before, ev_before, c->profile_skip};
    c->lane = T.lane;
    c->ev_cur = 1 + T.slot;
    c->profile_skip = T.profile_skip;             // (the stage events of THIS submission: sampled with its own tick)
    hipEvent_t *ev = c->ev_lane[T.lane];
    KM_HIP(c, hipStreamWaitEvent(main_stream, ev[EV_JOIN], 0));
    if (T.piped) KM_HIP(c, hipStreamWaitEvent(main_stream, ev[EV_C_DONE], 0));
    // ---- K7: LK forward + backward of every unit's corners in one launch
    {
        km_stage_timer t(c, ST_LK);
        if (!T.piped) {          // (unpipelined: the next submission's early min / max starts here)
            if (!c->ev_lk_start) KM_HIP(c, hipEventCreateWithFlags(&c->ev_lk_start, hipEventDisableTiming));
            KM_HIP(c, hipEventRecord(c->ev_lk_start, c->stream));
            c->lk_start_valid = true;
        }
        if ((rc = kl_units_launch(c, n, T.n_max, T.prm.win_size))) return rc;
    }
    if (T.piped) {
        KM_HIP(c, hipEventRecord(ev[EV_K_DONE], main_stream));
        KM_HIP(c, hipStreamWaitEvent(c->chain_stream, ev[EV_K_DONE], 0));
        c->stream = c->chain_stream;
    }
    // ---- K8: FB test, score, (x0, y0) order; K9 / K12: scores of the confident rows
    if ((rc = frame_block_free(c))) return rc;
    {
        km_stage_timer t(c, ST_FRAME);
        if ((rc = kf_frame_units(c, U, T.n_max, T.cap, 0.1f))) return rc;
    }
    if (T.with_zncc) {
        km_score_units S;
        for (int u = 0; u < n; u++) {
            km_score_unit &s = S.u[u];
            const float *f = (const float *)(U.frame[u] + 16);
            s.ref = U.ref_full[u]; s.mon = U.mon_full[u]; s.Href = s.Hmon = U.Hf[u]; s.Wref = s.Wmon = U.Wf[u]; s.sref = U.sref_f[u]; s.smon = U.smon_f[u];
            s.x0 = f; s.y0 = f + T.cap; s.dx = f + 2 * (size_t)T.cap; s.dy = f + 3 * (size_t)T.cap; s.score = f + 4 * (size_t)T.cap;
            s.d_n = (const int *)U.frame[u];
            s.out = (double *)(U.frame[u] + T.fb); s.out2 = nullptr;
            s.win = U.win[u];
        }
        {
            km_stage_timer t(c, ST_ZNCC);
            if ((rc = kz_zncc_units(c, S, n, T.dtype, T.n_max, (float)T.zncc_threshold))) return rc;
        }
        if (T.with_mi) {
            for (int u = 0; u < n; u++) { S.u[u].out = (double *)(U.frame[u] + T.fb) + T.cap; S.u[u].out2 = S.u[u].out + T.cap; }
            km_stage_timer t(c, ST_MI);
            if ((rc = kmi_units(c, S, n, T.dtype, T.n_max, (float)T.zncc_threshold))) return rc;
        }
    }
    // ---- the blocks leave: one strided device copy into the frame sink, one copy into the slot's page-locked buffer.  Unpipelined: on the
    // block-copy stream (the next submission's dense kernels queue behind this tail on the main stream).  Pipelined: in the chain stream's
    // own order - one stream fewer: the runtime maps streams onto FOUR hardware queues, and whenever the copy stream shared one with the
    // main stream its device-side wait for this chain held the next eigenvalue pass up (1 ms per 16-unit step, profiles/timeline_r06_c4.txt)
    const size_t ob = T.ob, ob_al = T.ob_al;
    if (!slot->done) KM_HIP(c, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
    hipStream_t out_stream = c->stream;
    if (!T.piped) {
        if (!c->d2h_stream) {
            KM_HIP(c, hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking));
            KM_HIP(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
        }
        KM_HIP(c, hipEventRecord(c->ev_tail, c->stream));
        KM_HIP(c, hipStreamWaitEvent(c->d2h_stream, c->ev_tail, 0));
        out_stream = c->d2h_stream;
    }
    slot->sunk_valid = false;
    if (T.sink) {
        const size_t pitch = T.sink_pitch ? T.sink_pitch : ob;
        KM_HIP(c, hipMemcpy2DAsync(T.sink, pitch, T.d_out, ob_al, ob, (size_t)n, hipMemcpyDeviceToDevice, out_stream));
        if (!slot->sunk) KM_HIP(c, hipEventCreateWithFlags(&slot->sunk, hipEventDisableTiming));
        KM_HIP(c, hipEventRecord(slot->sunk, out_stream));
        slot->sunk_valid = true;
    }
    KM_HIP(c, hipMemcpy2DAsync(slot->host, ob, T.d_out, ob_al, ob, (size_t)n, hipMemcpyDeviceToHost, out_stream));
    KM_HIP(c, hipEventRecord(slot->done, out_stream));
    if (T.piped) {      // the lane's next submission may rewrite its scalars / points / frame buffers once this chain (copies included) is through
        KM_HIP(c, hipEventRecord(c->ev_lane[T.lane][EV_F_DONE], c->chain_stream));
        c->lane_f_recorded[T.lane] = true;
    }
    c->frame_copy = T.piped ? nullptr : slot->done;     // (pipelined: every lane owns its frame buffer and waits for EV_F_DONE)
    slot->bytes = ob * n;
    slot->deferred.store(0, std::memory_order_release);
    return KM_OK;
}

// the deferred tail, if any (submitting thread, or the fallback of km_frame_wait: under c->enqueue_mu)
static int units_flush_locked(km_ctx *c)
{
    if (!c->utail || !c->utail->armed) return KM_OK;
    c->utail->armed = false;
    const int rc = units_tail(c, *c->utail);
    if (rc) c->fslot[c->utail->slot].deferred.store(0, std::memory_order_release);   // (a waiter must not spin for ever: its wait fails on the event)
    return rc;
}

extern "C++" void km_units_free(km_ctx *c)
{
    delete c->utail; c->utail = nullptr;
    delete c->enqueue_mu; c->enqueue_mu = nullptr;
}

// join = true (any entry point other than the next batched submission, km_ctx_sync, the option going off): the library's streams also
// wait for both lanes' chains - whatever follows on them may reuse lane 0's workspace in plain stream order
extern "C++" int km_units_flush(km_ctx *c, bool join)
{
    if (!c || !c->utail || !c->enqueue_mu) return KM_OK;
    std::lock_guard<std::mutex> lk(*c->enqueue_mu);
    if (c->utail->armed) KM_HIP(c, hipSetDevice(c->device));      // (the fallback of km_frame_wait runs on a thread of the caller's)
    const int rc = units_flush_locked(c);
    if (rc || !join) return rc;
    for (int l = 0; l < 2; l++)
        if (c->lane_f_recorded[l]) {
            KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_lane[l][EV_F_DONE], 0));
            if (c->aux_stream) KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_lane[l][EV_F_DONE], 0));
            c->lane_f_recorded[l] = false;
        }
    return KM_OK;
}

extern "C" {

// Enqueue whatever a pipelined batched submission deferred (include/karios_hip.h).  The submitting thread calls it when no further
// submission follows (FrameStream.drain, PendingBatch.wait on the submitting thread); every other entry point and km_ctx_sync do it
// themselves.
int km_frame_flush(km_ctx *c, int ticket)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    if (ticket >= 0 && (ticket >= KM_FRAME_SLOTS || !c->fslot[ticket].deferred.load(std::memory_order_acquire))) return KM_OK;   // not the deferred one
    return km_units_flush(c, false);
}

int km_klt_units_frame_submit(km_ctx *c, const km_unit *units, int n, int dtype, const double *nodata_ref, const double *nodata_mon,
                              const km_klt_params *prm, double zncc_threshold, int cap, int *ticket)
{
    int rc;
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    if (!ticket || !units) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: null argument");
    if (n < 1 || n > KM_UNITS_MAX) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: %d units (1 .. %d per submission)", n, KM_UNITS_MAX);
    if ((rc = check_params(c, prm))) return rc;
    if (!km_dtype_size(dtype)) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: bad dtype %d", dtype);
    if (cap <= 0 || prm->max_corners <= 0 || cap < prm->max_corners) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: capacity %d / maxCorners %d", cap, prm->max_corners);
    // the batch form IS the synchronisation-free corner path: whatever that path does not cover goes unit by unit (no message).  Every
    // coverage check sits HERE, in front of the first launch: a refused batch has queued nothing
    if (!(prm->min_distance >= 1) || !c->opt_speculative || !c->fused_eig || c->opt_key_cap || c->opt_stage_cap || c->opt_topk_factor || c->opt_select_first ||
        prm->max_level != 1 || cap > 32768)
        return KM_E_UNSUPPORTED;
    if (prm->block_size < 1 || prm->block_size > 15 || (prm->block_size & 1) == 0 || !c->opt_eig3 || !c->opt_lk2 || prm->win_size <= 2 || prm->win_size > 40)
        return KM_E_UNSUPPORTED;
    if (!km_units_ksize_supported(prm->ksize_ref) || !km_units_ksize_supported(prm->ksize_mon)) return KM_E_UNSUPPORTED;
    const bool with_zncc = units[0].d_ref_full != nullptr;
    const bool user_mask = units[0].d_mask != nullptr;
    const bool with_mi = with_zncc && c->opt_frame_mi;
    if (with_zncc && dtype == KM_F32) return KM_E_UNSUPPORTED;
    for (int u = 0; u < n; u++) {
        const km_unit &q = units[u];
        if ((rc = check_image(c, q.d_ref, q.H, q.W, q.sref, "klt_units_frame_submit")) || (rc = check_image(c, q.d_mon, q.H, q.W, q.smon, "klt_units_frame_submit")))
            return rc;
        if ((q.d_ref_full != nullptr) != with_zncc) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: either every unit carries full rasters or none");
        if ((q.d_mask != nullptr) != user_mask) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: either every unit carries a user mask or none");
        if (user_mask && q.smask < q.W) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: mask stride %td < width %d", q.smask, q.W);
        if (with_zncc && ((rc = check_image(c, q.d_ref_full, q.Hf, q.Wf, q.sref_f, "klt_units_frame_submit")) ||
                          (rc = check_image(c, q.d_mon_full, q.Hf, q.Wf, q.smon_f, "klt_units_frame_submit"))))
            return rc;
        if (q.W > 65535) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: unit of %d columns (the device-side frame ordering holds at most 65535)", q.W);
        if (q.W < 512 || q.H < 2 * prm->block_size + 8 || (q.W + 1) / 2 <= prm->win_size || (q.H + 1) / 2 <= prm->win_size) return KM_E_UNSUPPORTED;
        if ((prm->ksize_ref == 11 || prm->ksize_mon == 11) && q.H < 16) return KM_E_UNSUPPORTED;      // (the marching kernel's radius-5 form: kd_stretch_laplacian_units)
    }
    if (!c->enqueue_mu) c->enqueue_mu = new std::mutex;
    if (!c->utail) c->utail = new km_units_tail;
    std::lock_guard<std::mutex> enqueue_lock(*c->enqueue_mu);
    km_units_tail &old_tail = *c->utail;
    const bool piped = c->opt_units_pipeline && dtype != KM_U8;
    if (!piped) {      // (a pipelined submission was the previous one: its tail first, and this one behind both lanes' chains)
        if ((rc = units_flush_locked(c))) return rc;
        for (int l = 0; l < 2; l++)
            if (c->lane_f_recorded[l]) {
                KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_lane[l][EV_F_DONE], 0));
                if (c->aux_stream) KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_lane[l][EV_F_DONE], 0));
                c->lane_f_recorded[l] = false;
            }
    }
    const int k = c->fslot_next;
    km_frame_slot *slot = &c->fslot[k];
    if (slot->pending.load(std::memory_order_acquire)) {   // never waited for: its block is about to be overwritten
        if (slot->deferred.load(std::memory_order_acquire)) return km_fail(c, KM_E_ARG, "klt_units_frame_submit: frame slot %d still holds a deferred submission", k);
        KM_HIP(c, hipEventSynchronize(slot->done));
        slot->pending.store(0, std::memory_order_release);
    }
    const int lane = piped ? (old_tail.armed ? 1 - old_tail.lane : 0) : 0;
    hipStream_t const main_stream = c->stream;
    struct guard_t { km_ctx *c; hipStream_t s; ~guard_t() { c->ev_cur = 0; c->stream = s; c->lane = 0; c->in_units_submit = false; } } guard{c, main_stream};
    c->ev_cur = 1 + k;
    c->in_units_submit = true;
    if ((rc = begin_call(c, RESET_KLT))) return rc;
    memset(&c->stats, 0, sizeof c->stats);
    c->evs_used[c->ev_cur][ST_ZNCC] = false; c->evs_used[c->ev_cur][ST_MI] = false;
    c->lane = lane;
    if (!c->aux_stream) {
        KM_HIP(c, hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, 0));
        KM_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        KM_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    for (int i = 0; i < KM_LANE_EVENTS; i++)
        if (!c->ev_lane[lane][i]) KM_HIP(c, hipEventCreateWithFlags(&c->ev_lane[lane][i], hipEventDisableTiming));
    if (piped && !c->chain_stream) KM_HIP(c, hipStreamCreateWithFlags(&c->chain_stream, hipStreamNonBlocking));
    hipEvent_t *ev = c->ev_lane[lane];

    // ---- layout: unit u's slices of the lane's workspace slots
    km_units U;
    U.n = n; U.dtype = dtype; U.has_user_mask = user_mask;
    size_t px_total = 0, pyr_total = 0, max_px = 0;
    size_t px_off[KM_UNITS_MAX], pyr_off[KM_UNITS_MAX];
    for (int u = 0; u < n; u++) {
        const km_unit &q = units[u];
        U.H[u] = q.H; U.W[u] = q.W; U.ref[u] = q.d_ref; U.mon[u] = q.d_mon; U.sref[u] = q.sref; U.smon[u] = q.smon; U.x_off[u] = q.x_off; U.y_off[u] = q.y_off;
        U.ref_full[u] = q.d_ref_full; U.mon_full[u] = q.d_mon_full; U.Hf[u] = q.Hf; U.Wf[u] = q.Wf; U.sref_f[u] = q.sref_f; U.smon_f[u] = q.smon_f;
        U.win[u].ox = q.win_ox; U.win[u].oy = q.win_oy; U.win[u].H = q.win_H; U.win[u].W = q.win_W;
        U.user_mask[u] = q.d_mask; U.user_smask[u] = q.smask;
        const size_t px = (size_t)q.H * q.W;
        px_off[u] = px_total; px_total += up256(px);
        max_px = px > max_px ? px : max_px;
        pyr_off[u] = pyr_total; pyr_total += up256((size_t)((q.H + 1) / 2) * ((q.W + 1) / 2));
    }
    U.capk = max_px / 8 + 4096 * KM_NSHARD;
    const size_t sc_stride = up256(sizeof(km_scalars)), pb = up256((size_t)cap * 2 * sizeof(float));
    const size_t fb = 16 + (size_t)cap * 6 * sizeof(float), ob = fb + (with_zncc ? (size_t)cap * sizeof(double) : 0) + (with_mi ? (size_t)cap * 2 * sizeof(double) : 0);
    const size_t ob_al = up256(ob);
    uint8_t *lap_ref = (uint8_t *)km_ws(c, WS_U8_A, px_total), *lap_mon = (uint8_t *)km_ws(c, WS_U8_B, px_total), *mask = (uint8_t *)km_ws(c, WS_MASK, px_total);
    unsigned long long *keys = (unsigned long long *)km_ws(c, WS_KEYS0, U.capk * sizeof(unsigned long long) * n);
    char *sc = (char *)km_ws(c, WS_SCALARS, sc_stride * n);
    char *p0 = (char *)km_ws(c, WS_PTS0, pb * n), *p1 = (char *)km_ws(c, WS_PTS1, pb * n), *p0r = (char *)km_ws(c, WS_PTS2, pb * n);
    uint8_t *pyr_a = (uint8_t *)km_ws(c, WS_PYR_A, pyr_total), *pyr_b = (uint8_t *)km_ws(c, WS_PYR_B, pyr_total);
    char *d_out = (char *)km_ws(c, WS_FRAME, ob_al * n);
    if (!lap_ref || !lap_mon || !mask || !keys || !sc || !p0 || !p1 || !p0r || !pyr_a || !pyr_b || !d_out) return KM_E_NOMEM;
    for (int u = 0; u < n; u++) {
        U.lap_ref[u] = lap_ref + px_off[u]; U.lap_mon[u] = lap_mon + px_off[u]; U.mask[u] = mask + px_off[u];
        U.keys[u] = keys + U.capk * u;
        U.sc[u] = (km_scalars *)(sc + sc_stride * u);
        U.mm[u] = U.sc[u]->mm;
        U.p0[u] = (float *)(p0 + pb * u); U.p1[u] = (float *)(p1 + pb * u); U.p0r[u] = (float *)(p0r + pb * u);
        U.frame[u] = d_out + ob_al * u;
        U.eig_partial[u] = nullptr; U.eig_npartial[u] = 0;
        km_pyr &A = U.A[u], &B = U.B[u];
        A.img[0] = U.lap_ref[u]; B.img[0] = U.lap_mon[u];
        A.H[0] = B.H[0] = U.H[u]; A.W[0] = B.W[0] = U.W[u];
        A.img[1] = pyr_a + pyr_off[u]; B.img[1] = pyr_b + pyr_off[u];
        A.H[1] = B.H[1] = (U.H[u] + 1) / 2; A.W[1] = B.W[1] = (U.W[u] + 1) / 2;
        A.levels = B.levels = 1;
    }
    const size_t pitch = c->frame_sink_pitch ? c->frame_sink_pitch : ob;
    if (c->frame_sink && (pitch < ob || c->frame_sink_cap < pitch * (size_t)(n - 1) + ob))
        return km_fail(c, KM_E_ARG, "frame sink of %zu bytes (pitch %zu) is smaller than %d frame blocks of %zu bytes", c->frame_sink_cap, pitch, n, ob);
    if (slot->cap < ob * n) {
        if (slot->host) KM_HIP(c, hipHostFree(slot->host));
        slot->host = nullptr; slot->cap = 0;
        KM_HIP(c, hipHostMalloc(&slot->host, ob * n + ob / 8, hipHostMallocDefault));
        slot->cap = ob * n + ob / 8;
    }
    // The lane's previous submission must have left its frame stage before its scalars, points and frame buffers are reset - the scalar
    // blocks' reset and the LK table's small copy ("setup").  Unpipelined: here, on the main stream.  Pipelined: on the second stream
    // BEHIND the min / max (below): the min / max touches none of those buffers, and with the wait in front of it the Laplacians of this
    // submission - which need the min / max only - stood behind the other lane's frame stage, which runs beside the previous
    // eigenvalue pass and outlasts it (16 units of 5490^2: 0.8 ms of an idle main stream per step, tools/investigations/stage_order.py).  The main
    // stream waits for the setup in front of the eigenvalue pass, the first kernel of this submission that touches the scalars.
    const int n_max = prm->max_corners < cap ? prm->max_corners : cap;
    const bool setup_on_aux = piped && dtype != KM_U8;
    auto setup = [&]() -> int {
        int r = hipMemsetAsync(sc, 0, sc_stride * n, c->stream) == hipSuccess ? KM_OK : km_fail(c, KM_E_HIP, "hipMemsetAsync(scalars)");
        if (r == KM_OK) r = kl_units_prepare(c, U, n_max, prm->win_size, prm->max_count, prm->epsilon);
        return r;
    };
    if (!setup_on_aux) {
        if (c->lane_f_recorded[lane]) {
            KM_HIP(c, hipStreamWaitEvent(c->stream, ev[EV_F_DONE], 0));
            KM_HIP(c, hipStreamWaitEvent(c->aux_stream, ev[EV_F_DONE], 0));
            c->lane_f_recorded[lane] = false;
        }
        if ((rc = setup())) return rc;
    }

    // ---- K1: min / max of every raster, on the second stream.  Pipelined: at once - beside whatever dense kernel the older submissions
    // are in (HBM-bound work under instruction-bound kernels).  Unpipelined, directly behind another submission: beside its LK.
    if (dtype != KM_U8) {
        double *out[KM_UNITS_MAX];
        const bool early = piped || (c->opt_mm_early && c->lk_start_prev);
        if (early) {
            double *slot_mm = (double *)km_ws(c, WS_UNITS_MM, (size_t)4 * KM_UNITS_MAX * sizeof(double));
            if (!slot_mm) return KM_E_NOMEM;
            for (int u = 0; u < n; u++) { out[u] = slot_mm + 4 * u; U.mm[u] = out[u]; }
            if (!piped) KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_lk_start, 0));
            c->stream = c->aux_stream;
            {
                km_stage_timer t(c, ST_MINMAX);
                rc = kd_minmax_units(c, U, out, WS_MM_PARTIAL);
            }
            if (rc == KM_OK && hipEventRecord(ev[EV_MM], c->aux_stream) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipEventRecord(min/max)");
            if (rc == KM_OK && setup_on_aux) {      // the setup, behind the min / max and behind the lane's previous frame stage
                if (c->lane_f_recorded[lane]) {
                    if (hipStreamWaitEvent(c->aux_stream, ev[EV_F_DONE], 0) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipStreamWaitEvent(lane)");
                    c->lane_f_recorded[lane] = false;
                }
                if (rc == KM_OK) rc = setup();
                if (rc == KM_OK && hipEventRecord(ev[EV_SETUP], c->aux_stream) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipEventRecord(setup)");
            }
            c->stream = main_stream;
            if (rc) return rc;
            KM_HIP(c, hipStreamWaitEvent(c->stream, ev[EV_MM], 0));
            c->stats.path_flags |= KM_PATH_MM_EARLY;
        } else {
            for (int u = 0; u < n; u++) out[u] = U.sc[u]->mm;
            km_stage_timer t(c, ST_MINMAX);
            if ((rc = kd_minmax_units(c, U, out, WS_PARTIAL))) return rc;
        }
    }
    // ---- K2: stretch + Laplacians + automatic mask
    km_valid_units vjob;
    {
        km_stage_timer t(c, ST_LAPLACIAN);
        if ((rc = kd_stretch_laplacian_units(c, U, prm->ksize_ref, prm->ksize_mon, prm->invert_mon, nodata_ref, nodata_mon, &vjob))) return rc;
    }
    // ---- the valid-pixel sums and the pyramids (they depend on the Laplacians only) on the second stream
    {
        KM_HIP(c, hipEventRecord(ev[EV_FORK], c->stream));
        KM_HIP(c, hipStreamWaitEvent(c->aux_stream, ev[EV_FORK], 0));
        c->stream = c->aux_stream;
        rc = kd_valid_sum_units(c, vjob);
        if (rc == KM_OK) {
            km_stage_timer tp(c, ST_PYRAMID);
            rc = kd_pyrdown_units(c, U, 1);
        }
        if (rc == KM_OK && hipEventRecord(ev[EV_JOIN], c->aux_stream) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipEventRecord(join)");
        c->stream = main_stream;
        if (rc) return rc;
    }
    // ---- pipelined: the PREVIOUS submission's LK goes here, between this one's Laplacians and its eigenvalue pass (its frame stage and
    // scores follow on the chain stream, beside this submission's eigenvalue pass)
    if (piped && (rc = units_flush_locked(c))) { (void)hipStreamWaitEvent(c->stream, ev[EV_JOIN], 0); return rc; }
    c->lane = lane; c->ev_cur = 1 + k;
    if (setup_on_aux) KM_HIP(c, hipStreamWaitEvent(c->stream, ev[EV_SETUP], 0));     // (scalars reset, the lane's previous frame stage through)
    // ---- K3 + K4 fused
    {
        km_stage_timer t(c, ST_EIGEN);
        rc = k3_eig_candidates_units(c, U, prm->block_size, prm->quality_level);
    }
    if (rc) { (void)hipStreamWaitEvent(c->stream, ev[EV_JOIN], 0); return rc; }
    // ---- K5: ranking + greedy selection, every unit's chain side by side (pipelined: on the chain stream, beside the next submission's Laplacians)
    if (piped) {
        KM_HIP(c, hipEventRecord(ev[EV_E_DONE], main_stream));
        KM_HIP(c, hipStreamWaitEvent(c->chain_stream, ev[EV_E_DONE], 0));
        c->stream = c->chain_stream;
    }
    {
        km_stage_timer t(c, ST_SELECT);
        rc = kf_rank_select_units(c, U, prm->max_corners, prm->quality_level, prm->min_distance, cap);
    }
    if (piped) {
        if (rc == KM_OK && hipEventRecord(ev[EV_C_DONE], c->chain_stream) != hipSuccess) rc = km_fail(c, KM_E_HIP, "hipEventRecord(chain)");
        c->stream = main_stream;
    }
    if (rc) { (void)hipStreamWaitEvent(c->stream, ev[EV_JOIN], 0); return rc; }
    c->spec_used = true;
    // ---- the tail: LK, frame stage, scores, copy-out - now, or with the next submission
    km_units_tail &T = *c->utail;
    T.lane = lane; T.slot = k; T.U = U; T.prm = *prm; T.n = n; T.n_max = n_max; T.cap = cap; T.dtype = dtype;
    T.with_zncc = with_zncc; T.with_mi = with_mi; T.piped = piped; T.zncc_threshold = zncc_threshold; T.profile_skip = c->profile_skip;
    T.fb = fb; T.ob = ob; T.ob_al = ob_al; T.d_out = d_out;
    T.sink = c->frame_sink; T.sink_cap = c->frame_sink_cap; T.sink_pitch = c->frame_sink_pitch;
    slot->sunk_valid = false;
    if (piped) {
        T.armed = true;
        if (!slot->done) KM_HIP(c, hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
        slot->deferred.store(1, std::memory_order_release);
    } else if ((rc = units_tail(c, T)))
        return rc;
    slot->pending.store(1, std::memory_order_release);
    c->fslot_next = (k + 1) % KM_FRAME_SLOTS;
    *ticket = k;
    return KM_OK;
}

}  // extern "C"
