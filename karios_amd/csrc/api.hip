// C ABI of libkarios_hip.so (include/karios_hip.h): context, workspace, host-buffer wrappers and
// the device-resident KLT tile pipeline.  Nothing here computes on the CPU: every entry point
// ends in HIP kernels on the context stream; there is no fallback path.
#include "common.hpp"
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

void *km_ws(km_ctx *c, int slot, size_t bytes)
{
    km_buf &b = c->ws[slot];
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
    (void)hipStreamSynchronize(c->stream);
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
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->ev_lk_start) (void)hipEventDestroy(c->ev_lk_start);
    if (c->ev_mm) (void)hipEventDestroy(c->ev_mm);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (hipEvent_t e : c->upload_marks) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->free_marks) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return KM_OK;
}

int km_ctx_sync(km_ctx *c)
{
    if (!c) return km_fail(nullptr, KM_E_ARG, "null context");
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    KM_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d2h_stream) KM_HIP(c, hipStreamSynchronize(c->d2h_stream));   // frame blocks of submitted tiles
    if (!c->retired.empty()) {
        // workspace buffers replaced by larger ones: nothing of this context uses them any more once its streams are idle
        if (c->aux_stream) KM_HIP(c, hipStreamSynchronize(c->aux_stream));
        if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
        for (void *p : c->retired) (void)hipFree(p);
        c->retired.clear();
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
    { const int rcj = join_uploads(c); if (rcj) return rcj; }
    km_upload_check_drop(c);   // (checks armed by a call that failed half-way: their sources may be gone)
    c->land_jobs.clear(); c->land_used = 0;   // (... and results it queued for a caller buffer that may be gone too: km_d2h_queue without its flush)
    c->spec_used = false; c->spec_flags = 0;
    c->valid_job_pending = false;      // (a call that failed between the Laplacian pass and the fork)
    // early min / max (klt_tile_dev_impl): only a tile call that DIRECTLY follows a tile call may start its K1 beside the previous
    // unit's LK - any call in between may have produced the rasters on the main stream (km_shift_image_dev ...)
    c->lk_start_prev = c->lk_start_valid; c->lk_start_valid = false;
    // debugging aid: KARIOS_HIP_POISON_WS=<byte> fills every workspace buffer at the start of a tile call, so a kernel that reads
    // workspace it (or its predecessors in the call) never wrote shows up as a parity failure instead of a once-in-a-while one
    static const char *const poison = km_dev_env("KARIOS_HIP_POISON_WS");
    if (poison && reset == RESET_KLT)
        for (int i = 0; i < WS_COUNT; i++)
            if (c->ws[i].p && i != WS_AUTO && i != WS_MM_EARLY && i != WS_MM_PARTIAL) KM_HIP(c, hipMemsetAsync(c->ws[i].p, atoi(poison) & 0xff, c->ws[i].cap, c->stream));
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
static int h2d_now(km_ctx *c, void *dst, const void *src, size_t bytes)
{
    return km_h2d_staged(c, c->stream, dst, bytes, src, bytes, bytes, 1);
}

// diagnosis (KARIOS_HIP_VERIFY_UPLOAD): read a device image back on the library stream and compare it with its host source
static int verify_upload(km_ctx *c, const char *when, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, const void *d)
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

static int upload_image(km_ctx *c, int slot, const void *host, size_t elem, int H, int W, ptrdiff_t stride, void **dptr)
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

static km_scalars *scalars(km_ctx *c) { return (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars)); }

// pyramid of one image into caller-provided storage (levels >= 1 packed from `store`); returns the bytes used
static int build_pyramid_single(km_ctx *c, const uint8_t *d_img, int H, int W, int win, int max_level, uint8_t *store, km_pyr *P, size_t *used)
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
static int build_pyramid_pair(km_ctx *c, const uint8_t *d_a, const uint8_t *d_b, int H, int W, int win, int max_level, km_pyr *A, km_pyr *B)
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
static int gftt_dev(km_ctx *c, const uint8_t *d_img, const uint8_t *d_mask, int H, int W, int max_corners, double quality,
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

static int read_stats(km_ctx *c, km_scalars *sc)
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
static int klt_track_dev(km_ctx *c, const uint8_t *d_ref_lap, const uint8_t *d_mon_lap, const uint8_t *d_mask, int H, int W,
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

static int klt_tile_dev_impl(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon,
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

static int fetch_tracks(km_ctx *c, km_scalars *sc, const float *d_p0, const float *d_p1, const float *d_p0r, float *p0, float *p1,
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
        if ((rc = kf_frame(c, d_p0, d_p1, d_p0r, &sc->n_corners, n_max, cap, 0.1f, x_off, y_off, d_out, c->spec_used ? sc : nullptr))) return rc;
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
    km_scalars *sc = scalars(c);
    uint8_t *u8_ref = (uint8_t *)km_ws(c, WS_U8_A, n), *u8_mon = (uint8_t *)km_ws(c, WS_U8_B, n);
    if (!sc || !u8_ref || !u8_mon) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(sc, 0, sizeof *sc, c->stream));
    // ---- mask (user mask packed to the box, or the automatic one) and uint8 stretch
    const uint8_t *mask = d_mask;
    if (!d_mask) {
        uint8_t *m = (uint8_t *)km_ws(c, WS_MASK, n);
        if (!m) return KM_E_NOMEM;
        if ((rc = kd_auto_mask(c, d_mon, d_ref, dtype, H, W, smon, sref, nodata_mon, nodata_ref, m, &sc->valid))) return rc;
        mask = m;
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
        if ((rc = kd_minmax(c, d_ref, dtype, H, W, sref, &sc->mm[0])) || (rc = kd_minmax(c, d_mon, dtype, H, W, smon, &sc->mm[2]))) return rc;
    }
    if ((rc = kd_to_uint8(c, d_ref, dtype, H, W, sref, &sc->mm[0], 0, u8_ref)) || (rc = kd_to_uint8(c, d_mon, dtype, H, W, smon, &sc->mm[2], prm->invert_mon, u8_mon)))
        return rc;
    // ---- arena: 2*nk Laplacians, 2*nk pyramids, nk corner lists, nk*nk track pairs, counters
    km_pyr probe;
    size_t pyr_bytes = 0;
    build_pyramid_single(c, u8_ref, H, W, prm->win_size, prm->max_level, nullptr, &probe, &pyr_bytes);
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
        km_stage_timer t(c, ST_LAPLACIAN);
        for (int k = 0; k < nk; k++)
            if ((rc = kd_laplacian_u8(c, u8_ref, H, W, ksizes[k], lap_ref + (size_t)k * na)) ||
                (rc = kd_laplacian_u8(c, u8_mon, H, W, ksizes[k], lap_mon + (size_t)k * na)))
                return rc;
    }
    {
        km_stage_timer t(c, ST_PYRAMID);
        for (int k = 0; k < nk; k++)
            if ((rc = build_pyramid_single(c, lap_ref + (size_t)k * na, H, W, prm->win_size, prm->max_level, pyr_store + (size_t)(2 * k) * pyr_bytes, &PR[k], nullptr)) ||
                (rc = build_pyramid_single(c, lap_mon + (size_t)k * na, H, W, prm->win_size, prm->max_level, pyr_store + (size_t)(2 * k + 1) * pyr_bytes, &PM[k], nullptr)))
                return rc;
    }
    // ---- corners of every reference Laplacian (klt.py:494)
    int n_p0[8];
    for (int k = 0; k < nk; k++) {
        float *p0 = (float *)(p0_store + (size_t)k * pts);
        // gftt_dev starts from a clean scalar block; the min/max and the valid-pixel count gathered above stay
        KM_HIP(c, hipMemsetAsync(&sc->max_eig_key, 0, sizeof(km_scalars) - offsetof(km_scalars, max_eig_key), c->stream));
        if ((rc = gftt_dev(c, lap_ref + (size_t)k * na, mask, H, W, prm->max_corners, prm->quality_level, prm->min_distance, prm->block_size, p0, cap, sc)))
            return rc;
        KM_HIP(c, hipMemcpyAsync(&d_counts[k], &sc->n_corners, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        KM_D2H(c, &n_p0[k], &sc->n_corners, sizeof(int));
    }
    KM_FLUSH(c);
    // ---- nk*nk tracker runs (mon kernel outer, ref kernel inner), all queued before one synchronisation
    const int n_lim = prm->max_corners > 0 && prm->max_corners < cap ? prm->max_corners : cap;
    {
        km_stage_timer t(c, ST_LK);
        for (int im = 0; im < nk; im++)
            for (int ir = 0; ir < nk; ir++) {
                if (n_p0[ir] <= 0) continue;
                const int combo = im * nk + ir;
                float *p0 = (float *)(p0_store + (size_t)ir * pts);
                float *p1 = (float *)(trk_store + (size_t)(2 * combo) * pts), *p0r = (float *)(trk_store + (size_t)(2 * combo + 1) * pts);
                if ((rc = kl_track(c, PR[ir], PM[im], p0, &d_counts[ir], n_lim, prm->win_size, prm->max_count, prm->epsilon, true, p1, p0r)) ||
                    (rc = kf_count_kept(c, p0, p0r, &d_counts[ir], n_lim, 0.1f, &d_counts[nk + combo])))
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
                           (const float *)(trk_store + (size_t)(2 * best + 1) * pts), &d_counts[br], n_lim, cap, 0.1f, x_off, y_off, d_out)))
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
    if (hipEventSynchronize(slot->done) != hipSuccess) return KM_E_HIP;
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
