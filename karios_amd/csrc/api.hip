// C ABI of libkarios_hip.so (include/karios_hip.h): context, workspace, host-buffer wrappers and
// the device-resident KLT tile pipeline.  Nothing here computes on the CPU: every entry point
// ends in HIP kernels on the context stream; there is no fallback path.
#include "api_internal.hpp"
#include "fft64_plan.hpp"
#include <vector>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

static thread_local std::string g_last_error;

int km_fail(km_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}

void *km_ws_peek(km_ctx *c, int slot) { return c->lane ? c->ws_b[slot].p : c->ws[slot].p; }

void *km_ws(km_ctx *c, int slot, size_t bytes)
{
    km_buf &b = c->lane ? c->ws_b[slot] : c->ws[slot];
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return b.p;
    if (b.p) {
        // The library's streams may still use the old buffer: it is RETIRED, not freed - work already queued keeps a valid buffer,
        // new work gets the new one, and nothing synchronises (round 3 did hipDeviceSynchronize + hipFree here: a device-wide stall
        // under the feet of every other context on the GPU).  Retired buffers are released when the context is next synchronised
        // by its owner (km_ctx_sync) or destroyed.  Slots only grow, and by 6 % head-room at least: a handful of regrows per context.
        c->retired.push_back(b.p);
        b.p = nullptr; b.cap = 0;
    }
    size_t want = bytes + bytes / 16 + 256;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        km_fail(c, KM_E_NOMEM, "hipMalloc(%zu) for workspace slot %d: %s", want, slot, hipGetErrorString(e));
        return nullptr;
    }
    b.p = p; b.cap = want;
    return p;
}

void *km_pinned_rb(km_ctx *c, size_t bytes)
{
    if (c->pinned_rb_cap >= bytes) return c->pinned_rb;
    if (c->pinned_rb) { (void)hipStreamSynchronize(c->stream); (void)hipHostFree(c->pinned_rb); c->pinned_rb = nullptr; c->pinned_rb_cap = 0; }
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes + 256, hipHostMallocDefault);
    if (e != hipSuccess) { km_fail(c, KM_E_NOMEM, "hipHostMalloc(%zu): %s", bytes + 256, hipGetErrorString(e)); return nullptr; }
    c->pinned_rb = p; c->pinned_rb_cap = bytes + 256;
    return p;
}

int km_wait_readback(km_ctx *c)
{
    if (!c->ev_readback) KM_HIP(c, hipEventCreateWithFlags(&c->ev_readback, hipEventDisableTiming));
    static const bool no_defer = km_dev_env("KARIOS_HIP_NO_DEFER") != nullptr;   // A/B switch: wait first, run the jobs afterwards
    KM_HIP(c, hipEventRecord(c->ev_readback, c->stream));
    if (!c->deferred.empty() && !no_defer && !c->opt_no_defer) {
        std::function<int()> job = std::move(c->deferred.front());
        c->deferred.erase(c->deferred.begin());
        const int rc = job();
        if (rc) return rc;
    }
    KM_HIP(c, hipEventSynchronize(c->ev_readback));
    return KM_OK;
}

int km_run_deferred(km_ctx *c)
{
    while (!c->deferred.empty()) {
        std::function<int()> job = std::move(c->deferred.front());
        c->deferred.erase(c->deferred.begin());
        const int rc = job();
        if (rc) return rc;
    }
    return KM_OK;
}

// uploads queued on the copy stream (km_upload_async): whatever the compute stream does next starts behind them
static int join_uploads(km_ctx *c)
{
    if (c->copy_pending) {
        KM_HIP(c, hipEventRecord(c->ev_copy, c->copy_stream));
        KM_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
        if (c->aux_stream) KM_HIP(c, hipStreamWaitEvent(c->aux_stream, c->ev_copy, 0));   // (the early min / max reads the rasters there)
        c->copy_pending = false;
    }
    return KM_OK;
}

extern "C" {

int km_version(void) { return 101; }

const char *km_last_error(km_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int km_ctx_create(int device, km_ctx **out)
{
    if (!out) return km_fail(nullptr, KM_E_ARG, "km_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return km_fail(nullptr, KM_E_NO_DEVICE, "no HIP device available (%s); libkarios_hip has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return km_fail(nullptr, KM_E_ARG, "device %d out of range [0,%d)", device, ndev);
    km_ctx *c = new (std::nothrow) km_ctx();
    if (!c) return km_fail(nullptr, KM_E_NOMEM, "out of host memory");
    c->device = device;
    memset(&c->stats, 0, sizeof c->stats);
    memset(c->evs_used, 0, sizeof c->evs_used);
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        int rc = km_fail(nullptr, KM_E_HIP, "stream creation on device %d: %s", device, hipGetErrorString(e));
        delete c;
        return rc;
    }
    if (const char *e = getenv("KARIOS_HIP_FUSED_EIG")) c->fused_eig = atoi(e) != 0;
    if (const char *e = getenv("KARIOS_HIP_EIG3")) c->opt_eig3 = atoi(e) != 0;
    if (const char *e = getenv("KARIOS_HIP_AUX_PYRAMID")) c->opt_aux_pyramid = atoi(e) != 0;
    if (const char *e = getenv("KARIOS_HIP_SPECULATIVE")) c->opt_speculative = atoi(e) != 0;   // A/B switch for the sync-free corner path
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    *out = c;
    return KM_OK;
}

int km_ctx_destroy(km_ctx *c)
{
    if (!c) return KM_OK;
    (void)hipSetDevice(c->device);
    (void)km_units_flush(c);
    (void)hipStreamSynchronize(c->stream);
    if (c->chain_stream) (void)hipStreamSynchronize(c->chain_stream);
    kp_destroy(c);
    if (c->ev_readback) (void)hipEventDestroy(c->ev_readback);
    if (c->pinned_rb) (void)hipHostFree(c->pinned_rb);
    km_ring_destroy(c);
    for (km_frame_slot &f : c->fslot) {
        if (f.host) (void)hipHostFree(f.host);
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.sunk) (void)hipEventDestroy(f.sunk);
    }
    for (int i = 0; i < WS_COUNT; i++)
        if (c->ws[i].p) (void)hipFree(c->ws[i].p);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->d2h_stream) (void)hipStreamSynchronize(c->d2h_stream);
    for (void *p : c->retired) (void)hipFree(p);
    c->retired.clear();
    if (c->ev_ready)
        for (int k = 0; k <= KM_FRAME_SLOTS; k++)
            for (int i = 0; i < ST_COUNT; i++) { (void)hipEventDestroy(c->evs[k][i][0]); (void)hipEventDestroy(c->evs[k][i][1]); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_copy) (void)hipEventDestroy(c->ev_copy);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->d2h_stream) { (void)hipStreamSynchronize(c->d2h_stream); (void)hipStreamDestroy(c->d2h_stream); }
    if (c->chain_stream) { (void)hipStreamSynchronize(c->chain_stream); (void)hipStreamDestroy(c->chain_stream); }
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < KM_LANE_EVENTS; i++)
            if (c->ev_lane[l][i]) (void)hipEventDestroy(c->ev_lane[l][i]);
    for (int i = 0; i < WS_COUNT; i++)
        if (c->ws_b[i].p) (void)hipFree(c->ws_b[i].p);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->ev_lk_start) (void)hipEventDestroy(c->ev_lk_start);
    if (c->ev_mm) (void)hipEventDestroy(c->ev_mm);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (hipEvent_t e : c->upload_marks) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->free_marks) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    km_units_free(c);
    delete c;
    return KM_OK;
}

int km_ctx_sync(km_ctx *c)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    { const int rf = km_units_flush(c); if (rf) return rf; }              // (the deferred tail of a pipelined batched submission)
    KM_HIP(c, hipStreamSynchronize(c->stream));
    if (c->chain_stream) KM_HIP(c, hipStreamSynchronize(c->chain_stream));
    if (c->d2h_stream) KM_HIP(c, hipStreamSynchronize(c->d2h_stream));   // frame blocks of submitted tiles
    if (!c->retired.empty()) {
        // workspace buffers replaced by larger ones: nothing of this context uses them any more once its streams are idle
        if (c->aux_stream) KM_HIP(c, hipStreamSynchronize(c->aux_stream));
        if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
        for (void *p : c->retired) (void)hipFree(p);
        c->retired.clear();
        c->retired_mark = 0;
    }
    return KM_OK;
}

int km_set_option(km_ctx *c, const char *name, int value)
{
    if (!c || !name) return km_fail(c, KM_E_ARG, "km_set_option: null argument");
    // ---- the names include/karios_hip.h documents.  No option changes a result: each selects between forms that the parity suite holds
    // bit-identical (tests/test_gpu_forced_paths.py, tests/test_gpu_parity.py), or shrinks a capacity so that a retry path runs
    if (strcmp(name, "fused_eig") == 0) { c->fused_eig = value != 0; return KM_OK; }
    if (strcmp(name, "eig3") == 0) { c->opt_eig3 = value != 0; return KM_OK; }
    if (strcmp(name, "lk2") == 0) { c->opt_lk2 = value != 0; return KM_OK; }
    if (strcmp(name, "speculative") == 0) { c->opt_speculative = value != 0; return KM_OK; }
    if (strcmp(name, "aux_pyramid") == 0) { c->opt_aux_pyramid = value != 0; return KM_OK; }
    if (strcmp(name, "mm_early") == 0) { c->opt_mm_early = value != 0; return KM_OK; }
    if (strcmp(name, "frame_mi") == 0) { c->opt_frame_mi = value != 0; return KM_OK; }
    if (strcmp(name, "units_pipeline") == 0) {
        if (!value) { const int rf = km_units_flush(c); if (rf) return rf; }
        c->opt_units_pipeline = value != 0;
        return KM_OK;
    }
    if (strcmp(name, "key_cap") == 0) { c->opt_key_cap = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "stage_cap") == 0) { c->opt_stage_cap = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "topk_factor") == 0) { c->opt_topk_factor = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "select_first") == 0) { c->opt_select_first = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "stash_cap") == 0) { c->opt_stash_cap = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "spec_flag") == 0) { c->opt_spec_flag = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "defer") == 0) { c->opt_no_defer = value == 0; return KM_OK; }
    if (strcmp(name, "phase_fp64") == 0) { c->opt_phase_fp64 = value != 0; return KM_OK; }
    if (strcmp(name, "fft61") == 0) { c->opt_fft61 = value != 0; return KM_OK; }
    if (strcmp(name, "fft_herm") == 0) { c->opt_fft_herm = value != 0; return KM_OK; }
    if (strcmp(name, "f64_half") == 0) { c->opt_f64_half = value != 0; return KM_OK; }
    if (strcmp(name, "f64_pair") == 0) { c->opt_f64_pair = value != 0; return KM_OK; }
    if (strcmp(name, "f64_plain") == 0) { c->opt_f64_plain = value != 0; return KM_OK; }
    if (strcmp(name, "roctx") == 0) { c->opt_roctx = value != 0; return KM_OK; }
    if (strcmp(name, "profile_every") == 0) { c->opt_profile_every = value < 1 ? 1 : value; return KM_OK; }
    if (strcmp(name, "profile_stage") == 0) { c->opt_profile_stage = (value >= 0 && value < ST_COUNT) ? value : -1; return KM_OK; }
#ifdef KM_DEV
    // ---- development build only (make DEV=1): A/B switches of measured-and-settled choices, tuning of the double-precision transform
    if (strcmp(name, "fft_cross") == 0) { c->opt_fft_cross_fused = value != 0; return KM_OK; }
    if (strcmp(name, "f64_prime_t") == 0) { c->opt_f64_prime_t = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "f64_smooth_t") == 0) { c->opt_f64_smooth_t = value < 0 ? 0 : value; return KM_OK; }
    if (strcmp(name, "defer_valid") == 0) { c->opt_defer_valid = value != 0; return KM_OK; }
    if (strcmp(name, "aux_early") == 0) { c->opt_aux_early = value != 0; return KM_OK; }
    if (strcmp(name, "aux_priority") == 0) { c->opt_aux_priority = value != 0; return KM_OK; }   // (before the first tile: the stream is created once)
    if (strcmp(name, "eig3_count") == 0) { c->opt_eig3_count = value != 0; return KM_OK; }
#endif
    return km_fail(c, KM_E_ARG, "km_set_option: unknown option '%s'", name);
}

// development build: the counters of "eig3_count" after a BLOCKING tile call (out = {triples, triples every lane could skip}); release:
// KM_E_UNSUPPORTED
int km_dev_counters(km_ctx *c, unsigned long long out[2])
{
#ifdef KM_DEV
    if (!c || !out) return km_fail(c, KM_E_ARG, "km_dev_counters: null argument");
    km_scalars *sc = (km_scalars *)c->ws[WS_SCALARS].p;
    if (!sc) return km_fail(c, KM_E_ARG, "km_dev_counters: no tile call yet");
    unsigned long long h[2];
    { const int rq = km_d2h_queue(c, h, &sc->skip_total, sizeof h); if (rq) return rq; }
    { const int rq = km_d2h_flush(c); if (rq) return rq; }
    out[0] = h[0]; out[1] = h[1];
    return KM_OK;
#else
    (void)out;
    return km_fail(c, KM_E_UNSUPPORTED, "km_dev_counters: development build only");
#endif
}

// 1: the library was built with -DKM_DEV (development options and KARIOS_HIP_* tuning variables are live); 0: release build
int km_is_dev_build(void)
{
#ifdef KM_DEV
    return 1;
#else
    return 0;
#endif
}

int km_set_profiling(km_ctx *c, int enable)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    if (enable && !c->ev_ready) {
        for (int k = 0; k <= KM_FRAME_SLOTS; k++)
            for (int i = 0; i < ST_COUNT; i++) { KM_HIP(c, hipEventCreate(&c->evs[k][i][0])); KM_HIP(c, hipEventCreate(&c->evs[k][i][1])); }
        c->ev_ready = true;
    }
    c->profiling = enable != 0;
    return KM_OK;
}

static const char *const kStageNames[ST_COUNT] = {"minmax", "stretch_laplacian_mask", "min_eigen", "candidates", "sort",
                                                  "select", "pyramid", "lk_fwd_bwd", "zncc", "fb_frame", "mutual_info", "phase_correlation"};

const char *km_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

// ---- roctx ranges (tracing): resolved lazily with dlopen, a no-op when no roctx library is present
extern "C++" {
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static bool g_roctx_tried = false;
static void roctx_resolve()
{
    if (g_roctx_tried) return;
    g_roctx_tried = true;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
        void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_roctx_push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) return;
        g_roctx_push = nullptr; g_roctx_pop = nullptr;
    }
}
void km_roctx_push(int stage)
{
    roctx_resolve();
    if (g_roctx_push) {
        char name[64];
        snprintf(name, sizeof name, "karios:%s", km_stage_name(stage));
        g_roctx_push(name);
    }
}
void km_roctx_pop()
{
    if (g_roctx_pop) g_roctx_pop();
}
}  // extern "C++"

int km_get_stage_ms(km_ctx *c, float *out, int cap, int *n)
{
    if (!c || !out) return km_fail(c, KM_E_ARG, "km_get_stage_ms: null argument");
    KM_HIP(c, hipStreamSynchronize(c->stream));
    int m = cap < ST_COUNT ? cap : ST_COUNT;
    for (int i = 0; i < m; i++) {
        out[i] = 0.f;
        if (c->ev_ready && c->evs_used[0][i]) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->evs[0][i][0], c->evs[0][i][1]) == hipSuccess) out[i] = ms;
        }
    }
    if (n) *n = m;
    return KM_OK;
}

int km_get_klt_stats(km_ctx *c, km_klt_stats *out)
{
    if (!c || !out) return km_fail(c, KM_E_ARG, "km_get_klt_stats: null argument");
    *out = c->stats;
    return KM_OK;
}

int km_dev_alloc(km_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr) return km_fail(c, KM_E_ARG, "km_dev_alloc: null argument");
    KM_HIP(c, hipSetDevice(c->device));
    KM_HIP(c, hipMalloc(dptr, bytes ? bytes : 16));
    return KM_OK;
}
int km_dev_free(km_ctx *c, void *dptr)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    KM_HIP(c, hipStreamSynchronize(c->stream));
    if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
    KM_HIP(c, hipFree(dptr));
    return KM_OK;
}
int km_h2d(km_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    { const int rcs = km_h2d_staged(c, c->stream, dst, bytes, src, bytes, bytes, 1); if (rcs) return rcs; }
    KM_FLUSH(c);
    return KM_OK;
}
int km_d2h(km_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    KM_D2H(c, dst, src, bytes);
    KM_FLUSH(c);
    return KM_OK;
}

int km_host_alloc(km_ctx *c, size_t bytes, void **hptr)
{
    if (!c || !hptr) return km_fail(c, KM_E_ARG, "km_host_alloc: null argument");
    KM_HIP(c, hipSetDevice(c->device));
    KM_HIP(c, hipHostMalloc(hptr, bytes ? bytes : 16, hipHostMallocDefault));
    return KM_OK;
}
int km_host_free(km_ctx *c, void *hptr)
{
    // The block may outlive the context it was allocated through (a numpy array over it is released by the garbage
    // collector): with a null context hipHostFree alone decides - it waits for the copies that still read the block.
    if (!hptr) return KM_OK;
    if (!c) return hipHostFree(hptr) == hipSuccess ? KM_OK : KM_E_HIP;
    if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
    KM_HIP(c, hipHostFree(hptr));
    return KM_OK;
}
int km_upload_async(km_ctx *c, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows)
{
    if (!c || !dst || !src) return km_fail(c, KM_E_ARG, "km_upload_async: null argument");
    if (dst_pitch < width_bytes || src_pitch < width_bytes) return km_fail(c, KM_E_ARG, "km_upload_async: pitch below the row width");
    KM_HIP(c, hipSetDevice(c->device));
    if (!c->copy_stream) {
        KM_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        KM_HIP(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    }
    if (rows == 0 || width_bytes == 0) return KM_OK;
    // page-locked source (km_host_alloc / hipHostMalloc / hipHostRegister): DMA'd in place, truly asynchronous.  Pageable source:
    // packed into the context's page-locked ring chunk by chunk (staging.hip) - the call returns when the source has been read,
    // the DMAs stay ordered on the copy stream; no runtime copy ever reads pageable memory
    { const int rcs = km_h2d_staged(c, c->copy_stream, dst, dst_pitch, src, src_pitch, width_bytes, rows); if (rcs) return rcs; }
    c->copy_pending = true;
    return KM_OK;
}
// Tickets let a caller tie a LATER piece of compute to exactly the uploads it needs: mark after queuing the uploads of pair
// i+1, join before the first kernel of pair i+1 - the kernels of pair i, launched in between, do not wait for them.
int km_upload_mark(km_ctx *c, int *ticket)
{
    if (!c || !ticket) return km_fail(c, KM_E_ARG, "km_upload_mark: null argument");
    *ticket = -1;
    if (!c->copy_stream) return KM_OK;                      // nothing was ever uploaded asynchronously: ticket -1 = already complete
    hipEvent_t ev = nullptr;
    if (!c->free_marks.empty()) { ev = c->free_marks.back(); c->free_marks.pop_back(); }
    else KM_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    KM_HIP(c, hipEventRecord(ev, c->copy_stream));
    size_t slot = 0;
    while (slot < c->upload_marks.size() && c->upload_marks[slot]) slot++;
    if (slot == c->upload_marks.size()) c->upload_marks.push_back(nullptr);
    c->upload_marks[slot] = ev;
    c->copy_pending = false;                                // the caller took charge of the ordering
    *ticket = (int)slot;
    return KM_OK;
}
int km_upload_join(km_ctx *c, int ticket)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    if (ticket < 0) return KM_OK;
    if ((size_t)ticket >= c->upload_marks.size() || !c->upload_marks[ticket]) return km_fail(c, KM_E_ARG, "km_upload_join: unknown ticket %d", ticket);
    hipEvent_t ev = c->upload_marks[ticket];
    KM_HIP(c, hipStreamWaitEvent(c->stream, ev, 0));        // device-side wait: the host does not block
    if (c->aux_stream) KM_HIP(c, hipStreamWaitEvent(c->aux_stream, ev, 0));
    c->upload_marks[ticket] = nullptr;
    c->free_marks.push_back(ev);
    return KM_OK;
}
int km_upload_wait(km_ctx *c)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
    return KM_OK;
}
int km_upload_check_stats(km_ctx *c, int64_t *armed, int64_t *missed)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    if (armed) *armed = (int64_t)c->chk_armed_total;
    if (missed) *missed = (int64_t)c->chk_miss_total;
    return KM_OK;
}
int km_set_image_window(km_ctx *c, int ox, int oy, int H_image, int W_image)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    if (H_image < 0 || W_image < 0 || ox < 0 || oy < 0 || (H_image > 0) != (W_image > 0)) return km_fail(c, KM_E_ARG, "km_set_image_window: bad window");
    c->window.ox = ox; c->window.oy = oy; c->window.H = H_image; c->window.W = W_image;
    return KM_OK;
}
int km_phase_info(km_ctx *c, int *path, double *margin)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    if (path) *path = c->phase_path;
    if (margin) *margin = c->phase_margin;
    return KM_OK;
}
int km_phase_plan(int n, int along_columns, int *levels, int cap_levels, int *n_levels, int *bluestein, int *negpos)
{
    if (n < 1 || cap_levels < 0 || (cap_levels > 0 && !levels)) return KM_E_ARG;
    f64plan::dimplan P;
    if (!f64plan::plan_dim(n, along_columns ? 256 : F64_SMOOTH_MAX, &P)) return KM_E_UNSUPPORTED;
    if ((int)P.lv.size() > cap_levels) return KM_E_ARG;
    for (size_t i = 0; i < P.lv.size(); i++) { levels[2 * i] = P.lv[i].n; levels[2 * i + 1] = P.lv[i].kind; }
    if (n_levels) *n_levels = (int)P.lv.size();
    if (bluestein) *bluestein = P.blue ? P.L : 0;
    if (negpos) {
        std::vector<int> ng;
        f64plan::host_negpos(P, ng);
        for (int i = 0; i < n; i++) negpos[i] = ng[(size_t)i];
    }
    return KM_OK;
}
int km_set_frame_sink(km_ctx *c, void *d_dst, size_t capacity_bytes)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    c->frame_sink = d_dst;
    c->frame_sink_cap = d_dst ? capacity_bytes : 0;
    c->frame_sink_pitch = 0;
    return KM_OK;
}
int km_set_frame_sink_pitch(km_ctx *c, void *d_dst, size_t capacity_bytes, size_t pitch_bytes)
{
    if (!c) return km_fail(c, KM_E_ARG, "null context");
    c->frame_sink = d_dst;
    c->frame_sink_cap = d_dst ? capacity_bytes : 0;
    c->frame_sink_pitch = d_dst ? pitch_bytes : 0;
    return KM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ helpers
// (api_internal.hpp: RESET_* - stage timers are cleared per pipeline: the KLT entry points own [ST_MINMAX, ST_LK], ZNCC owns ST_ZNCC)
int begin_call(km_ctx *c, int reset)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    KM_HIP(c, hipSetDevice(c->device));
    // any entry point other than the next batched submission: the deferred tail of a pipelined submission goes first, in stream order
    if (!c->in_units_submit) { const int rf = km_units_flush(c); if (rf) return rf; }
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    km_upload_check_drop(c);   // (checks armed by a call that failed half-way: their sources may be gone)
    c->land_jobs.clear(); c->land_used = 0;   // (... and results it queued for a caller buffer that may be gone too: km_d2h_queue without its flush)
    c->spec_used = false; c->spec_flags = 0;
    c->retired_mark = c->retired.size();        // (km_d2h_flush inside this call frees only what was retired before it)
    c->valid_job_pending = false;      // (a call that failed between the Laplacian pass and the fork)
    // early min / max (klt_tile_dev_impl): only a tile call that DIRECTLY follows a tile call may start its K1 beside the previous
    // unit's LK - any call in between may have produced the rasters on the main stream (km_shift_image_dev ...)
    c->lk_start_prev = c->lk_start_valid; c->lk_start_valid = false;
    // debugging aid: KARIOS_HIP_POISON_WS=<byte> fills every workspace buffer at the start of a tile call, so a kernel that reads
    // workspace it (or its predecessors in the call) never wrote shows up as a parity failure instead of a once-in-a-while one
    static const char *const poison = km_dev_env("KARIOS_HIP_POISON_WS");
    if (poison && reset == RESET_KLT)
        for (int i = 0; i < WS_COUNT; i++)
            if (c->ws[i].p && i != WS_AUTO && i != WS_MM_EARLY && i != WS_MM_PARTIAL && i != WS_UNITS_MM) KM_HIP(c, hipMemsetAsync(c->ws[i].p, atoi(poison) & 0xff, c->ws[i].cap, c->stream));
    if (reset == RESET_KLT) {
        for (int i = ST_MINMAX; i <= ST_LK; i++) c->evs_used[c->ev_cur][i] = false;
        c->evs_used[c->ev_cur][ST_FRAME] = false;
        // "profile_every" N: only every N-th tile call records its stage events (a sample of the calls: an event record is a point
        // where consecutive kernels may not overlap, and on some boxes bracketing one stage of EVERY call costs 3 - 5 % of a step)
        c->profile_tick++;
        c->profile_skip = c->opt_profile_every > 1 && (c->profile_tick % (unsigned)c->opt_profile_every) != 0;
    }
    else if (reset == RESET_ZNCC)
        c->evs_used[c->ev_cur][ST_ZNCC] = false;
    return KM_OK;
}

// Host -> device copy of caller memory on the library stream through the page-locked ring: `src` has been read completely on
// return, the DMA is ordered on the stream like any kernel (staging.hip).
int h2d_now(km_ctx *c, void *dst, const void *src, size_t bytes)
{
    return km_h2d_staged(c, c->stream, dst, bytes, src, bytes, bytes, 1);
}

// diagnosis (KARIOS_HIP_VERIFY_UPLOAD): read a device image back on the library stream and compare it with its host source
int verify_upload(km_ctx *c, const char *when, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, const void *d)
{
    static const bool verify = km_dev_env("KARIOS_HIP_VERIFY_UPLOAD") != nullptr;
    if (!verify) return KM_OK;
    const size_t row = (size_t)W * elem;
    std::vector<char> back((size_t)H * row);
    KM_D2H(c, back.data(), d, back.size());
    KM_HIP(c, hipStreamSynchronize(c->stream));
    { const int rq = km_d2h_flush(c); if (rq) return rq; }
    int bad = 0, first = -1, last = -1;
    for (int y = 0; y < H; y++)
        if (memcmp(back.data() + (size_t)y * row, (const char *)host + (size_t)y * stride * elem, row) != 0) { bad++; if (first < 0) first = y; last = y; }
    if (bad) fprintf(stderr, "KARIOS_HIP_VERIFY_UPLOAD: %s: slot %d, %d x %d x %zu B: %d rows differ from the host source (first %d, last %d)\n", when, slot, H, W, elem,
                     bad, first, last);
    return KM_OK;
}

int upload_image(km_ctx *c, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, void **dptr)
{
    void *d = km_ws(c, slot, (size_t)H * W * elem);
    if (!d) return KM_E_NOMEM;
    // Caller memory is pageable: the rows travel through the library's own page-locked ring (staging.hip) - the runtime never copies
    // from pageable memory (round 3 saw two stale-input mismatches behind hipMemcpy2DAsync(pageable rows): CHANGELOG.md, round 4)
    { const int rcs = km_h2d_staged(c, c->stream, d, (size_t)W * elem, host, (size_t)stride * elem, (size_t)W * elem, (size_t)H); if (rcs) return rcs; }
    { const int rca = km_upload_check_arm(c, slot == WS_RAW_A ? "ref / image A" : slot == WS_RAW_B ? "mon / image B" : slot == WS_MASK_IN ? "mask" : "u8 image", host, elem, H, W, stride, d); if (rca) return rca; }
    *dptr = d;
    return verify_upload(c, "after upload", slot, host, elem, H, W, stride, d);
}

int check_image(km_ctx *c, const void *p, int H, int W, ptrdiff_t stride, const char *what)
{
    if (!p) return km_fail(c, KM_E_ARG, "%s: null image", what);
    if (H <= 0 || W <= 0) return km_fail(c, KM_E_ARG, "%s: empty image %dx%d", what, H, W);
    if (stride < W) return km_fail(c, KM_E_ARG, "%s: stride %td < width %d", what, stride, W);
    return KM_OK;
}

km_scalars *scalars(km_ctx *c) { return (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars)); }

