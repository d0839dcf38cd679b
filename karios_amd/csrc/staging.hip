// Transfers between CALLER memory and the device.  The boundary hands over numpy arrays (klt.py:252-253: every tile is read
// into fresh host arrays and must be matched as read), i.e. pageable memory.  The library never gives such memory to an
// asynchronous runtime copy: it moves through the context's own page-locked ring -
//
//   host -> device   the caller's rows are packed into a ring slot by the host (a small pool of copy threads for large
//                    chunks) while the DMA of the previous slot runs; the DMA source is always the ring.  When the call
//                    returns the caller's buffer has been read completely, the DMAs are ordinary stream-ordered copies.
//   device -> host   results land in a page-locked arena (DMA), the stream is completed, the host copies them out.
//
// A source that is already page-locked (km_host_alloc, hipHostMalloc, hipHostRegister) is DMA'd directly.
//
// Why: twice in 36 700 tile cases of a six-process soak the kernels behind `hipMemcpy2DAsync(pageable rows)` saw partly stale
// destination rows (profiles/r03_fuzz_parity.md).  KARIOS_HIP_UPLOAD_CHECKSUM=1 arms the diagnosis of that event: a row
// checksum kernel is enqueued right behind every host-buffer upload ON THE SAME STREAM, its result is compared with the host's
// checksum of the source rows when the call completes, and a row that differs is checksummed again after the stream has
// drained - "wrong at kernel time, right after the wait" is a copy / ordering fault below the library, "still wrong" is ours.
#include "common.hpp"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

// ------------------------------------------------------------------ host copy pool
namespace {

struct copy_pool {
    std::mutex run;                     // one parallel copy at a time (contexts of several threads share the pool)
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)> *job = nullptr;
    int nparts = 0, next = 0, pending = 0;
    unsigned long long gen = 0;

    void worker()
    {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_work.wait(lk, [&] { return gen != seen && job; });
            seen = gen;
            while (job && next < nparts) {
                const int p = next++;
                const std::function<void(int)> *j = job;
                lk.unlock();
                (*j)(p);
                lk.lock();
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void start(int n)
    {
        for (int i = 0; i < n; i++) {
            th.emplace_back([this] { worker(); });
            th.back().detach();          // (the pool lives as long as the process; nothing to join at exit)
        }
    }
    void parallel(int parts, const std::function<void(int)> &f)
    {
        std::lock_guard<std::mutex> one(run);
        std::unique_lock<std::mutex> lk(m);
        job = &f; nparts = parts; next = 0; pending = parts; gen++;
        cv_work.notify_all();
        while (next < nparts) {          // the caller works too
            const int p = next++;
            lk.unlock();
            f(p);
            lk.lock();
            --pending;
        }
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
};

copy_pool *g_pool = nullptr;
int g_pool_threads = -1;
std::once_flag g_pool_once;

int pool_threads()
{
    std::call_once(g_pool_once, [] {
        int n = 3;                                               // + the calling thread: 45.6 GB/s on the GPU box (8 / 12 / 16 threads: 34 / 33 / 32 GB/s under its 16-CPU quota)
        if (const char *e = km_dev_env("KARIOS_HIP_COPY_THREADS")) n = atoi(e) - 1;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && n > (int)hw - 1) n = (int)hw - 1;
        if (n < 0) n = 0;
        if (n > 15) n = 15;
        g_pool_threads = n;
        if (n > 0) { g_pool = new copy_pool(); g_pool->start(n); }
    });
    return g_pool_threads;
}

// rows [0, rows) of width wb: (dst, dpitch) <- (src, spitch), by one thread or by the pool
void copy_rows(char *dst, size_t dpitch, const char *src, size_t spitch, size_t wb, size_t rows)
{
    const size_t total = wb * rows;
    auto range = [=](size_t b0, size_t b1) {                     // dense byte range [b0, b1) of the wb x rows block
        while (b0 < b1) {
            const size_t r = b0 / wb, x = b0 % wb;
            size_t n = wb - x;
            if (n > b1 - b0) n = b1 - b0;
            if (x == 0 && dpitch == wb && spitch == wb) n = b1 - b0;   // contiguous on both sides: one memcpy
            memcpy(dst + r * dpitch + x, src + r * spitch + x, n);
            b0 += n;
        }
    };
    const int extra = total >= ((size_t)1 << 20) ? pool_threads() : 0;
    if (extra <= 0) { range(0, total); return; }
    const int parts = extra + 1;
    const size_t per = ((total + parts - 1) / parts + 63) & ~(size_t)63;
    std::function<void(int)> f = [&](int p) {
        const size_t b0 = (size_t)p * per, b1 = b0 + per < total ? b0 + per : total;
        if (b0 < b1) range(b0, b1);
    };
    g_pool->parallel(parts, f);
}

bool is_page_locked(const void *p)
{
    hipPointerAttribute_t at;
    const bool pinned = hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return pinned;
}

}  // namespace

// ------------------------------------------------------------------ the ring
static int ring_ready(km_ctx *c)
{
    km_stage_ring &r = c->ring;
    if (r.slot[0].buf) return KM_OK;
    size_t chunk = (size_t)8 << 20;     // 4 x 8 MB: 50 GB/s from pageable memory on the GPU box (4 MB: 45.6, 1 - 2 MB: 30; page-locked source: 57.5)
    if (const char *e = km_dev_env("KARIOS_HIP_RING_CHUNK_KB")) { const long v = atol(e); if (v >= 64 && v <= (1 << 18)) chunk = (size_t)v << 10; }
    for (int i = 0; i < KM_RING_SLOTS; i++) {
        KM_HIP(c, hipHostMalloc(&r.slot[i].buf, chunk, hipHostMallocDefault));
        KM_HIP(c, hipEventCreateWithFlags(&r.slot[i].done, hipEventDisableTiming));
        r.slot[i].busy = false;
    }
    r.chunk = chunk;
    return KM_OK;
}

void km_ring_destroy(km_ctx *c)
{
    for (int i = 0; i < KM_RING_SLOTS; i++) {
        if (c->ring.slot[i].done) { (void)hipEventSynchronize(c->ring.slot[i].done); (void)hipEventDestroy(c->ring.slot[i].done); }
        if (c->ring.slot[i].buf) (void)hipHostFree(c->ring.slot[i].buf);
        c->ring.slot[i] = km_ring_slot();
    }
    for (int i = 0; i < 2; i++) { if (c->land_ev[i]) (void)hipEventDestroy(c->land_ev[i]); c->land_ev[i] = nullptr; }
    if (c->land) (void)hipHostFree(c->land);
    c->land = nullptr; c->land_cap = c->land_used = 0;
    c->land_jobs.clear();
    if (c->chk_dev) (void)hipFree(c->chk_dev);
    if (c->chk_host) (void)hipHostFree(c->chk_host);
    c->chk_dev = nullptr; c->chk_host = nullptr;
    c->chk_jobs.clear();
}

int km_h2d_staged(km_ctx *c, hipStream_t s, void *dst, size_t dpitch, const void *src, size_t spitch, size_t wb, size_t rows)
{
    if (wb == 0 || rows == 0) return KM_OK;
    if (dpitch < wb || spitch < wb) return km_fail(c, KM_E_ARG, "staged upload: pitch below the row width");
    if (is_page_locked(src)) {                                   // already a legal DMA source: no staging
        if (dpitch == wb && spitch == wb) KM_HIP(c, hipMemcpyAsync(dst, src, wb * rows, hipMemcpyHostToDevice, s));
        else KM_HIP(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, wb, rows, hipMemcpyHostToDevice, s));
        return KM_OK;
    }
    { const int rc = ring_ready(c); if (rc) return rc; }
    km_stage_ring &r = c->ring;
    if (dpitch == wb && spitch == wb) { wb *= rows; rows = 1; dpitch = spitch = wb; }   // contiguous: one long row, cut anywhere
    const char *sp = (const char *)src;
    char *dp = (char *)dst;
    size_t row = 0, col = 0;                                     // col > 0 only while a row longer than a chunk is under way
    while (row < rows) {
        km_ring_slot &sl = r.slot[r.next];
        r.next = (r.next + 1) % KM_RING_SLOTS;
        if (sl.busy) { KM_HIP(c, hipEventSynchronize(sl.done)); sl.busy = false; }
        if (wb > r.chunk) {                                      // piece of one long row
            size_t n = wb - col;
            if (n > r.chunk) n = r.chunk;
            copy_rows((char *)sl.buf, n, sp + row * spitch + col, n, n, 1);
            KM_HIP(c, hipMemcpyAsync(dp + row * dpitch + col, sl.buf, n, hipMemcpyHostToDevice, s));
            col += n;
            if (col == wb) { col = 0; row++; }
        } else {
            size_t nr = r.chunk / wb;
            if (nr > rows - row) nr = rows - row;
            copy_rows((char *)sl.buf, wb, sp + row * spitch, spitch, wb, nr);
            if (dpitch == wb) KM_HIP(c, hipMemcpyAsync(dp + row * dpitch, sl.buf, wb * nr, hipMemcpyHostToDevice, s));
            else KM_HIP(c, hipMemcpy2DAsync(dp + row * dpitch, dpitch, sl.buf, wb, wb, nr, hipMemcpyHostToDevice, s));
            row += nr;
        }
        KM_HIP(c, hipEventRecord(sl.done, s));
        sl.busy = true;
    }
    return KM_OK;
}

// ------------------------------------------------------------------ device -> caller memory
static int land_ready(km_ctx *c)
{
    if (c->land) return KM_OK;
    size_t cap = (size_t)8 << 20;
    KM_HIP(c, hipHostMalloc(&c->land, cap, hipHostMallocDefault));
    c->land_cap = cap; c->land_used = 0;
    return KM_OK;
}

int km_d2h_queue(km_ctx *c, void *dst, const void *d_src, size_t bytes)
{
    if (bytes == 0) return KM_OK;
    { const int rc = land_ready(c); if (rc) return rc; }
    if (bytes > c->land_cap / 2) {
        // large result: two halves of the arena alternate - the DMA of piece k+1 runs while the host copies piece k out
        { const int rc = km_d2h_flush(c); if (rc) return rc; }
        const size_t half = c->land_cap / 2;
        if (!c->land_ev[0]) for (int i = 0; i < 2; i++) KM_HIP(c, hipEventCreateWithFlags(&c->land_ev[i], hipEventDisableTiming));
        const size_t np = (bytes + half - 1) / half;
        auto issue = [&](size_t k) -> int {
            const size_t off = k * half, n = bytes - off < half ? bytes - off : half;
            KM_HIP(c, hipMemcpyAsync((char *)c->land + (k & 1) * half, (const char *)d_src + off, n, hipMemcpyDeviceToHost, c->stream));
            KM_HIP(c, hipEventRecord(c->land_ev[k & 1], c->stream));
            return KM_OK;
        };
        { const int rc = issue(0); if (rc) return rc; }
        for (size_t k = 0; k < np; k++) {
            if (k + 1 < np) { const int rc = issue(k + 1); if (rc) return rc; }
            KM_HIP(c, hipEventSynchronize(c->land_ev[k & 1]));
            const size_t off = k * half, n = bytes - off < half ? bytes - off : half;
            copy_rows((char *)dst + off, n, (const char *)c->land + (k & 1) * half, n, n, 1);
            // (piece k+2 reuses this half: it is issued in the next trip, after this copy-out)
        }
        return KM_OK;
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (c->land_used + need > c->land_cap) { const int rc = km_d2h_flush(c); if (rc) return rc; }
    void *p = (char *)c->land + c->land_used;
    c->land_used += need;
    KM_HIP(c, hipMemcpyAsync(p, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    c->land_jobs.push_back({dst, p, bytes});
    return KM_OK;
}

int km_d2h_flush(km_ctx *c)
{
    KM_HIP(c, hipStreamSynchronize(c->stream));
    for (const km_land_job &j : c->land_jobs) copy_rows((char *)j.dst, j.bytes, (const char *)j.pinned, j.bytes, j.bytes, 1);
    c->land_jobs.clear();
    c->land_used = 0;
    if (c->retired_mark > 0 && !c->retired.empty()) {
        // workspace buffers replaced by larger ones (km_ws) BEFORE this entry point began: the compute stream is idle here; once the other
        // streams are too, nothing of this context can still use them - callers of the blocking entry points alone never reach
        // km_ctx_sync.  Buffers retired INSIDE the running call stay: a pointer the call obtained before the slot was regrown is still
        // valid until the call returns (a flush in the middle of an entry point must not leave it dangling).
        if (c->aux_stream) KM_HIP(c, hipStreamSynchronize(c->aux_stream));
        if (c->chain_stream) KM_HIP(c, hipStreamSynchronize(c->chain_stream));
        if (c->copy_stream) KM_HIP(c, hipStreamSynchronize(c->copy_stream));
        if (c->d2h_stream) KM_HIP(c, hipStreamSynchronize(c->d2h_stream));
        const size_t n = c->retired_mark < c->retired.size() ? c->retired_mark : c->retired.size();
        for (size_t i = 0; i < n; i++) (void)hipFree(c->retired[i]);
        c->retired.erase(c->retired.begin(), c->retired.begin() + (ptrdiff_t)n);
        c->retired_mark = 0;
    }
    return km_upload_check_verify(c);
}

// ------------------------------------------------------------------ KARIOS_HIP_UPLOAD_CHECKSUM: what did the kernels behind an upload see?
#define KM_CHK_ROWS (1 << 17)

static unsigned long long host_row_checksum(const uint8_t *p, size_t n)
{
    unsigned long long s = 0;
    for (size_t i = 0; i < n; i++) s += (unsigned long long)(p[i] + 1u) * (unsigned long long)(2 * i + 1);
    return s;
}

bool km_upload_check_enabled()
{
    static const bool on = getenv("KARIOS_HIP_UPLOAD_CHECKSUM") != nullptr;
    return on;
}

int km_upload_check_arm(km_ctx *c, const char *what, const void *host, size_t elem, int H, int W, ptrdiff_t stride, const void *d)
{
    if (!km_upload_check_enabled() || H <= 0) return KM_OK;
    if (!c->chk_dev) {
        KM_HIP(c, hipMalloc(&c->chk_dev, (size_t)KM_CHK_ROWS * 8));
        KM_HIP(c, hipHostMalloc(&c->chk_host, (size_t)KM_CHK_ROWS * 8, hipHostMallocDefault));
        c->chk_used = 0;
    }
    if (c->chk_used + (size_t)H > KM_CHK_ROWS) return KM_OK;    // arena full until the next verification: this upload goes unchecked
    unsigned long long *dsum = (unsigned long long *)c->chk_dev + c->chk_used, *hsum = (unsigned long long *)c->chk_host + c->chk_used;
    { const int rk = kf_row_checksum(c, d, (size_t)W * elem, H, dsum); if (rk) return rk; }
    KM_HIP(c, hipMemcpyAsync(hsum, dsum, (size_t)H * 8, hipMemcpyDeviceToHost, c->stream));
    c->chk_jobs.push_back({what, host, elem, H, W, stride, d, c->chk_used});
    c->chk_used += (size_t)H;
    c->chk_armed_total++;
    return KM_OK;
}

// Call with c->stream complete and the upload sources still valid (the end of the blocking entry point that uploaded them).
int km_upload_check_verify(km_ctx *c)
{
    if (c->chk_jobs.empty()) return KM_OK;
    for (const km_chk_job &j : c->chk_jobs) {
        const unsigned long long *seen = (const unsigned long long *)c->chk_host + j.off;
        const size_t rb = (size_t)j.W * j.elem;
        int bad = 0, first = -1, last = -1;
        for (int y = 0; y < j.H; y++)
            if (seen[y] != host_row_checksum((const uint8_t *)j.host + (size_t)y * j.stride * j.elem, rb)) { bad++; if (first < 0) first = y; last = y; }
        if (!bad) continue;
        c->chk_miss_total++;
        // the stream is complete now: what does the same kernel see?
        unsigned long long *dsum = (unsigned long long *)c->chk_dev + j.off;
        (void)kf_row_checksum(c, j.d, rb, j.H, dsum);
        std::vector<unsigned long long> again((size_t)j.H);
        unsigned long long *mine = (unsigned long long *)c->chk_host + j.off;   // (this job's own landing words: consumed above)
        (void)hipMemcpyAsync(mine, dsum, (size_t)j.H * 8, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        memcpy(again.data(), mine, (size_t)j.H * 8);
        int still = 0;
        for (int y = 0; y < j.H; y++)
            if (again[y] != host_row_checksum((const uint8_t *)j.host + (size_t)y * j.stride * j.elem, rb)) still++;
        fprintf(stderr, "KARIOS_HIP_UPLOAD_CHECKSUM MISS: %s, %d x %d x %zu B (host stride %td elements): the kernel right behind the upload saw %d rows that differ "
                        "from the source (first %d, last %d); after the stream had drained %d rows differ -> %s\n",
                j.what, j.H, j.W, j.elem, j.stride, bad, first, last, still,
                still == 0 ? "the copy had not finished when the kernel ran (ordering fault below the library)" : "the destination is wrong for good (library or source changed)");
        fflush(stderr);
    }
    c->chk_jobs.clear();
    c->chk_used = 0;
    return KM_OK;
}

void km_upload_check_drop(km_ctx *c)
{
    c->chk_jobs.clear();
    c->chk_used = 0;
}
