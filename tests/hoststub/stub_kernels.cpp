// TEST INFRASTRUCTURE (tests/test_host_asan.py): the stand-in HIP layer of hip/hip_runtime.h and do-little replacements of every
// kernel launcher api.hip / staging.hip call (csrc/common.hpp).  "Device" memory is malloc'ed host memory, so AddressSanitizer
// sees every access: the stand-ins READ their inputs completely and WRITE their outputs completely, at the sizes the launchers
// are entitled to - an undersized workspace slot, a short upload or a read-back beyond a buffer is a sanitizer report.
#include "../../karios_amd/csrc/common.hpp"

#include <algorithm>
#include <cmath>
#include <numeric>

stub_state &stub()
{
    static stub_state s;
    return s;
}

static bool is_pinned(const void *p)
{
    stub_state &s = stub();
    std::lock_guard<std::mutex> g(s.m);
    auto it = s.pinned.upper_bound((const char *)p);
    if (it == s.pinned.begin()) return false;
    --it;
    return (const char *)p < it->first + it->second;
}

hipError_t hipMalloc(void **p, size_t n)
{
    stub_state &s = stub();
    if (s.fail_malloc_after == 0) { s.fail_malloc_after = -1; *p = nullptr; return hipErrorOutOfMemory; }
    if (s.fail_malloc_after > 0) s.fail_malloc_after--;
    *p = malloc(n ? n : 1);
    s.device_allocs++;
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) { if (p) { free(p); stub().device_allocs--; } return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    stub_state &s = stub();
    std::lock_guard<std::mutex> g(s.m);
    s.pinned[(const char *)*p] = n ? n : 1;
    s.host_allocs++;
    return hipSuccess;
}
hipError_t hipHostFree(void *p)
{
    if (!p) return hipSuccess;
    stub_state &s = stub();
    {
        std::lock_guard<std::mutex> g(s.m);
        if (!s.pinned.erase((const char *)p)) return hipErrorInvalidValue;
        s.host_allocs--;
    }
    free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *at, const void *p)
{
    if (!is_pinned(p)) return hipErrorInvalidValue;      // like the runtime: unregistered host memory is an error
    at->type = hipMemoryTypeHost; at->device = 0; at->devicePointer = at->hostPointer = (void *)p;
    return hipSuccess;
}
// the property under test: the library never hands PAGEABLE memory to an asynchronous runtime copy
static void note_copy(const void *host_side)
{
    if (!is_pinned(host_side)) { std::lock_guard<std::mutex> g(stub().m); stub().pageable_async_copies++; }
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t)
{
    if (kind == hipMemcpyHostToDevice) note_copy(src);
    if (kind == hipMemcpyDeviceToHost) note_copy(dst);
    memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind, hipStream_t)
{
    if (dpitch < width || spitch < width) return hipErrorInvalidValue;
    if (kind == hipMemcpyHostToDevice) note_copy(src);
    if (kind == hipMemcpyDeviceToHost) note_copy(dst);
    for (size_t y = 0; y < height; y++) memmove((char *)dst + y * dpitch, (const char *)src + y * spitch, width);
    return hipSuccess;
}

extern "C" {
// counters for the test: {device allocations alive, page-locked allocations alive, asynchronous copies that touched pageable memory}
void stub_counters(long out[3])
{
    out[0] = stub().device_allocs; out[1] = stub().host_allocs; out[2] = stub().pageable_async_copies;
}
void stub_fail_malloc_after(int n) { stub().fail_malloc_after = n; }
}

// ------------------------------------------------------------------ pixel access
static double px(const void *img, int dtype, size_t i)
{
    switch (dtype) {
    case KM_U8: return ((const uint8_t *)img)[i];
    case KM_U16: return ((const uint16_t *)img)[i];
    case KM_I16: return ((const int16_t *)img)[i];
    case KM_F32: return ((const float *)img)[i];
    case KM_F64: return ((const double *)img)[i];
    case KM_I32: return ((const int32_t *)img)[i];
    case KM_U32: return ((const uint32_t *)img)[i];
    default: return 0;
    }
}
static void minmax_of(const void *img, int dtype, int H, int W, ptrdiff_t stride, double *mm)
{
    double mn = INFINITY, mx = -INFINITY;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const double v = px(img, dtype, (size_t)y * stride + x);
            if (v == v) { mn = std::min(mn, v); mx = std::max(mx, v); }
        }
    mm[0] = mn; mm[1] = mx;
}

int kd_minmax(km_ctx *, const void *d, int dtype, int H, int W, ptrdiff_t s, double *mm) { minmax_of(d, dtype, H, W, s, mm); return KM_OK; }
int kd_minmax_pair(km_ctx *, const void *a, const void *b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *mm)
{
    minmax_of(a, dtype, H, W, sa, mm); minmax_of(b, dtype, H, W, sb, mm + 2);
    return KM_OK;
}
int kd_minmax_pair_ws(km_ctx *c, const void *a, const void *b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *mm, int)
{
    return kd_minmax_pair(c, a, b, dtype, H, W, sa, sb, mm);
}
int kd_to_uint8(km_ctx *, const void *d, int dtype, int H, int W, ptrdiff_t s, const double *mm, int invert, uint8_t *out)
{
    const double r = mm[1] - mm[0];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const double v = px(d, dtype, (size_t)y * s + x);
            uint8_t u = dtype == KM_U8 ? (uint8_t)v : (r > 0 && v == v ? (uint8_t)((v - mm[0]) / r * 255.0) : 0);
            out[(size_t)y * W + x] = invert ? 255 - u : u;
        }
    return KM_OK;
}
int kd_auto_mask(km_ctx *, const void *m, const void *r, int dtype, int H, int W, ptrdiff_t sm, ptrdiff_t sr, const double *, const double *, uint8_t *mask,
                 unsigned long long *valid)
{
    unsigned long long n = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const bool ok = px(m, dtype, (size_t)y * sm + x) != 0 && px(r, dtype, (size_t)y * sr + x) != 0;
            mask[(size_t)y * W + x] = ok; n += ok;
        }
    *valid += n;
    return KM_OK;
}
int kd_count_nonzero(km_ctx *, const uint8_t *m, size_t n, unsigned long long *valid)
{
    unsigned long long k = 0;
    for (size_t i = 0; i < n; i++) k += m[i] != 0;
    *valid += k;
    return KM_OK;
}
int kd_laplacian_u8(km_ctx *, const uint8_t *s, int H, int W, int, uint8_t *d) { memmove(d, s, (size_t)H * W); return KM_OK; }
int kd_stretch_laplacian_pair(km_ctx *c, const void *ref, const void *mon, int dtype, int H, int W, ptrdiff_t sr, ptrdiff_t sm, const double *mm, int, int, int inv,
                              const double *nr, const double *nm, uint8_t *lr, uint8_t *lm, uint8_t *mask, unsigned long long *valid)
{
    kd_to_uint8(c, ref, dtype, H, W, sr, mm, 0, lr);
    kd_to_uint8(c, mon, dtype, H, W, sm, mm + 2, inv, lm);
    if (mask) kd_auto_mask(c, mon, ref, dtype, H, W, sm, sr, nm, nr, mask, valid);
    return KM_OK;
}
int kd_min_eigen(km_ctx *, const uint8_t *s, const uint8_t *mask, int H, int W, int, float *eig, unsigned *max_key)
{
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)H * W; i++) { eig[i] = (float)s[i]; if (!mask || mask[i]) mx = std::max(mx, eig[i]); }
    unsigned k; memcpy(&k, &mx, 4);
    *max_key = k | 0x80000000u;
    return KM_OK;
}
int k2_min_eigen(km_ctx *, const uint8_t *, const uint8_t *, int, int, int, float *, unsigned *) { return KM_E_UNSUPPORTED; }
int k2_eig_candidates(km_ctx *, const uint8_t *, const uint8_t *, int, int, int, double, km_scalars *, unsigned long long *, size_t, bool) { return KM_E_UNSUPPORTED; }
int k3_eig_candidates(km_ctx *, const uint8_t *, const uint8_t *, int, int, int, double, km_scalars *, unsigned long long *, size_t) { return KM_E_UNSUPPORTED; }
// candidates: every 7th pixel of every 5th row, keys written up to the capacity the caller reserved
int kd_candidates(km_ctx *, const float *eig, const uint8_t *, int H, int W, double, km_scalars *sc, unsigned long long *keys, size_t cap, bool)
{
    size_t n = 0;
    for (int y = 1; y < H - 1; y += 5)
        for (int x = 1; x < W - 1; x += 7) {
            unsigned b; const float v = eig[(size_t)y * W + x] + 1.f; memcpy(&b, &v, 4);
            if (n < cap) keys[n] = ((unsigned long long)b << 32) | (unsigned)((size_t)y * W + x);
            n++;
        }
    sc->n_cand = (unsigned)n;
    return KM_OK;
}
int ks_topk_prefilter(km_ctx *, const unsigned long long *keys, size_t cap, size_t, km_scalars *sc, double, unsigned long long **kept, size_t *n_kept, size_t *n_total,
                      km_scalars *hs, bool)
{
    *hs = *sc;
    hs->pad0 = 0;
    const size_t n = std::min((size_t)sc->n_cand, cap);
    *kept = const_cast<unsigned long long *>(keys); *n_kept = n; *n_total = n;
    hs->cut[1] = (unsigned)n; hs->cut[3] = (unsigned)n;
    return KM_OK;
}
int ks_sort_keys_desc(km_ctx *, unsigned long long *k, size_t n, unsigned long long **sorted)
{
    std::sort(k, k + n, std::greater<unsigned long long>());
    *sorted = k;
    return KM_OK;
}
int ks_select(km_ctx *, const unsigned long long *sorted, size_t n, int, int W, int max_corners, double, float *xy, int cap, km_scalars *sc, int *found, bool)
{
    int m = 0;
    for (size_t i = 0; i < n && (max_corners <= 0 || m < max_corners) && m < cap; i++, m++) {
        const unsigned idx = (unsigned)(sorted[i] & 0xffffffffu);
        xy[2 * m] = (float)(idx % (unsigned)W); xy[2 * m + 1] = (float)(idx / (unsigned)W);
    }
    sc->n_corners = m;
    if (found) *found = m;
    return KM_OK;
}
int km_sort_u64(km_ctx *, unsigned long long *ka, unsigned long long *kb, unsigned *va, unsigned *vb, size_t n, bool desc)
{
    std::vector<size_t> o(n);
    std::iota(o.begin(), o.end(), (size_t)0);
    std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return desc ? ka[a] > ka[b] : ka[a] < ka[b]; });
    for (size_t i = 0; i < n; i++) { kb[i] = ka[o[i]]; if (va) vb[i] = va[o[i]]; }
    memcpy(ka, kb, n * 8);
    if (va) memcpy(va, vb, n * 4);
    return KM_OK;
}
int km_exclusive_scan(km_ctx *, const unsigned *in, unsigned *out, size_t n, int mode, int)
{
    unsigned s = 0;
    for (size_t i = 0; i < n; i++) { out[i] = s; s += mode == KM_SCAN_IS_ONE ? (in[i] == 1) : in[i]; }
    return KM_OK;
}
size_t kf_kept_capacity(int max_corners) { return (size_t)max_corners * 8; }
int kf_rank(km_ctx *, const unsigned long long *, size_t, int, int, int, double, double, km_scalars *) { return KM_E_UNSUPPORTED; }
int kf_select(km_ctx *, int, int, int, double, float *, int, km_scalars *) { return KM_E_UNSUPPORTED; }
int kd_pyrdown_u8(km_ctx *, const uint8_t *s, int H, int W, uint8_t *d)
{
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) d[(size_t)y * w + x] = s[(size_t)std::min(2 * y, H - 1) * W + std::min(2 * x, W - 1)];
    return KM_OK;
}
int kd_pyrdown_u8_pair(km_ctx *c, const uint8_t *a, const uint8_t *b, int H, int W, uint8_t *da, uint8_t *db) { kd_pyrdown_u8(c, a, H, W, da); return kd_pyrdown_u8(c, b, H, W, db); }
int kd_shift_image(km_ctx *, const void *img, int es, int H, int W, ptrdiff_t stride, int yo, int xo, void *out)
{
    memset(out, 0, (size_t)H * W * es);
    for (int y = 0; y < H; y++) {
        const int sy = y + yo;
        if (sy < 0 || sy >= H) continue;
        for (int x = 0; x < W; x++) {
            const int sx = x + xo;
            if (sx >= 0 && sx < W) memcpy((char *)out + ((size_t)y * W + x) * es, (const char *)img + ((size_t)sy * stride + sx) * es, es);
        }
    }
    return KM_OK;
}
int kl_track(km_ctx *, const km_pyr &A, const km_pyr &B, const float *p, const int *d_n, int n_max, int, int, double, bool back, float *p1, float *p0r, int *)
{
    // touch the last pixel of every pyramid level: the level buffers must be as large as the geometry says
    for (int l = 0; l <= A.levels; l++) {
        const int h = A.Hres[l] ? A.Hres[l] : A.H[l], oy = A.Hres[l] ? A.oy[l] : 0;
        volatile uint8_t t = A.img[l][(size_t)(oy + h - 1) * A.W[l] + A.W[l] - 1] ^ B.img[l][(size_t)(oy + h - 1) * B.W[l] + B.W[l] - 1];
        (void)t;
    }
    const int n = d_n ? std::min(*d_n, n_max) : n_max;
    for (int i = 0; i < 2 * n; i++) { p1[i] = p[i] + 0.25f; if (back && p0r) p0r[i] = p[i] + (i % 8 == 0 ? 0.5f : 0.01f); }
    return KM_OK;
}
int kl_oscillation_probe(km_ctx *, const float *q, int n, uint8_t *out)
{
    for (int i = 0; i < n; i++) out[i] = (double)fabsf(q[4 * i] + q[4 * i + 1]) < 0.01 && (double)fabsf(q[4 * i + 2] + q[4 * i + 3]) < 0.01;
    return KM_OK;
}
int kf_count_kept(km_ctx *, const float *p0, const float *p0r, const int *d_n, int n_max, float thr, int *cnt)
{
    const int n = std::min(*d_n, n_max);
    int k = 0;
    for (int i = 0; i < n; i++) k += std::max(fabsf(p0[2 * i] - p0r[2 * i]), fabsf(p0[2 * i + 1] - p0r[2 * i + 1])) < thr;
    *cnt = k;
    return KM_OK;
}
int kf_dn_keep(km_ctx *, const void *, const void *, int, int H, int W, ptrdiff_t, ptrdiff_t, const float *x0, const float *y0, int n, const double *, int, const double *,
               const double *, uint8_t *keep)
{
    for (int i = 0; i < n; i++) keep[i] = (x0[i] < 0 || y0[i] < 0 || x0[i] >= W || y0[i] >= H) ? 2 : 1;
    return KM_OK;
}
// frame block: 16-byte header + 6 * cap float32 columns (x0 | y0 | dx | dy | score | index bits); every column written to `cap`
int kf_frame(km_ctx *, const float *p0, const float *p1, const float *p0r, const int *d_n, int n_max, int cap, float thr, float xo, float yo, void *out, const km_scalars *hdr, int /*width*/)
{
    const int n = std::min(*d_n, n_max);
    int *h = (int *)out;
    float *f = (float *)((char *)out + 16);
    for (size_t i = 0; i < (size_t)6 * cap; i++) f[i] = 0.f;
    int k = 0;
    for (int i = 0; i < n; i++) {
        const float d = std::max(fabsf(p0[2 * i] - p0r[2 * i]), fabsf(p0[2 * i + 1] - p0r[2 * i + 1]));
        if (!(d < thr)) continue;
        f[k] = p0[2 * i] + xo; f[cap + k] = p0[2 * i + 1] + yo; f[2 * (size_t)cap + k] = p1[2 * i] - p0[2 * i]; f[3 * (size_t)cap + k] = p1[2 * i + 1] - p0[2 * i + 1];
        f[4 * (size_t)cap + k] = 1.f - d / thr;
        k++;
    }
    h[0] = k; h[1] = n; h[2] = hdr ? (int)hdr->flags : 0; h[3] = hdr ? (int)hdr->cut[3] : 0;
    return KM_OK;
}
int kf_row_checksum(km_ctx *, const void *d, size_t rb, int rows, unsigned long long *out)
{
    for (int y = 0; y < rows; y++) {
        const uint8_t *p = (const uint8_t *)d + (size_t)y * rb;
        unsigned long long s = 0;
        for (size_t i = 0; i < rb; i++) s += (unsigned long long)(p[i] + 1u) * (unsigned long long)(2 * i + 1);
        out[y] = s;
    }
    return KM_OK;
}
int kz_zncc(km_ctx *, const void *, const void *, int, int, int, int, int, ptrdiff_t, ptrdiff_t, const float *x0, const float *y0, const float *dx, const float *dy, int n,
            double *out)
{
    for (int i = 0; i < n; i++) out[i] = (double)(x0[i] + y0[i] + dx[i] + dy[i]);
    return KM_OK;
}
int kz_zncc_filtered(km_ctx *, const void *, const void *, int, int, int, int, int, ptrdiff_t, ptrdiff_t, const float *x0, const float *, const float *, const float *, int n,
                     const int *d_n, const float *score, float thr, double *out)
{
    const int m = std::min(*d_n, n);
    for (int i = 0; i < m; i++) out[i] = score[i] >= thr ? (double)x0[i] : NAN;
    return KM_OK;
}
int kz_zncc_windows(km_ctx *, const void *, const void *, int, int, int, int, int, int, ptrdiff_t, ptrdiff_t, const int *uv, int, int count, double *out, uint8_t *fl)
{
    for (int i = 0; i < count; i++) { out[i] = uv[4 * i] + uv[4 * i + 3]; fl[i] = 0; }
    return KM_OK;
}
int kmi_batch(km_ctx *, const void *, const void *, int, int, int, int, int, ptrdiff_t, ptrdiff_t, const float *x0, const float *, const float *, const float *, int n,
              const int *, const float *, float, double *st, double *nmi)
{
    for (int i = 0; i < n; i++) { if (st) st[i] = x0[i]; if (nmi) nmi[i] = -x0[i]; }
    return KM_OK;
}
int kp_phase_shift(km_ctx *c, const void *a, const void *b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double rc[2])
{
    rc[0] = px(a, dtype, (size_t)(H - 1) * sa + W - 1) - px(b, dtype, (size_t)(H - 1) * sb + W - 1);
    rc[1] = 0;
    c->phase_path = 1;
    return KM_OK;
}
void kp_destroy(km_ctx *) {}
int kd_run_valid_sum(km_ctx *c) { c->valid_job_pending = false; return KM_OK; }

// ---- batched units (api_units.hip): every stage touches the first and last byte / element of what the host laid out for it
int kd_minmax_units(km_ctx *, const km_units &U, double *const *out, int)
{
    for (int u = 0; u < U.n; u++) {
        minmax_of(U.ref[u], U.dtype, U.H[u], U.W[u], U.sref[u], out[u]);
        minmax_of(U.mon[u], U.dtype, U.H[u], U.W[u], U.smon[u], out[u] + 2);
    }
    return KM_OK;
}
int kd_stretch_laplacian_units(km_ctx *, const km_units &U, int, int, int, const double *, const double *, km_valid_units *job)
{
    job->n = U.n;
    for (int u = 0; u < U.n; u++) {
        const size_t n = (size_t)U.H[u] * U.W[u];
        for (size_t i = 0; i < n; i++) { U.lap_ref[u][i] = (uint8_t)i; U.lap_mon[u][i] = (uint8_t)(i + 1); U.mask[u][i] = 1; }
        volatile double m = U.mm[u][0] + U.mm[u][3];
        (void)m;
        job->partial[u] = nullptr; job->n_partial[u] = 0; job->out[u] = &U.sc[u]->valid;
    }
    return KM_OK;
}
int kd_valid_sum_units(km_ctx *, const km_valid_units &J)
{
    for (int u = 0; u < J.n; u++) *J.out[u] = 12345ull;
    return KM_OK;
}
int k3_eig_candidates_units(km_ctx *, km_units &U, int, double)
{
    for (int u = 0; u < U.n; u++) { U.keys[u][0] = 1ull; U.keys[u][U.capk - 1] = 2ull; U.eig_partial[u] = nullptr; U.eig_npartial[u] = 0; }
    return KM_OK;
}
int kd_pyrdown_units(km_ctx *, const km_units &U, int level)
{
    for (int u = 0; u < U.n; u++) {
        uint8_t *a = (uint8_t *)U.A[u].img[level], *b = (uint8_t *)U.B[u].img[level];
        const size_t n = (size_t)U.A[u].H[level] * U.A[u].W[level];
        memset(a, 3, n); memset(b, 4, n);
    }
    return KM_OK;
}
int kf_rank_select_units(km_ctx *, const km_units &U, int max_corners, double, double, int cap)
{
    for (int u = 0; u < U.n; u++) {
        const int n = std::min(std::min(max_corners, cap), 40 + u);
        for (int i = 0; i < n; i++) { U.p0[u][2 * i] = (float)(20 + 7 * i % (U.W[u] - 40)); U.p0[u][2 * i + 1] = (float)(20 + 11 * i % (U.H[u] - 40)); }
        U.p0[u][2 * (size_t)cap - 1] = 0.f;
        U.sc[u]->n_corners = n; U.sc[u]->cut[3] = 1000u + (unsigned)u;
    }
    return KM_OK;
}
int kl_jobs_launch(km_ctx *, const km_lk_job *, int, int, int, int, double) { return KM_E_UNSUPPORTED; }   // (the search then runs its trackers one by one)
int kf_count_kept_jobs(km_ctx *, const km_count_jobs &, int, int, float, int *) { return KM_OK; }
static km_units g_lk_units[2];       // (per workspace lane: a pipelined submission prepares its table before the previous one's LK is launched)
int kl_units_prepare(km_ctx *c, const km_units &U, int, int, int, double) { g_lk_units[c->lane & 1] = U; return KM_OK; }
int kl_units_launch(km_ctx *c, int n_units, int n_max, int win)
{
    const km_units &U = g_lk_units[c->lane & 1];
    for (int u = 0; u < n_units; u++) {
        const int rc = kl_track(c, U.A[u], U.B[u], U.p0[u], &U.sc[u]->n_corners, n_max, win, 30, 0.03, true, U.p1[u], U.p0r[u], nullptr);
        if (rc) return rc;
    }
    return KM_OK;
}
int kf_frame_units(km_ctx *c, const km_units &U, int n_max, int cap, float thr)
{
    for (int u = 0; u < U.n; u++) {
        const int rc = kf_frame(c, U.p0[u], U.p1[u], U.p0r[u], &U.sc[u]->n_corners, n_max, cap, thr, U.x_off[u], U.y_off[u], U.frame[u], U.sc[u], U.W[u]);
        if (rc) return rc;
    }
    return KM_OK;
}
int kz_zncc_units(km_ctx *, const km_score_units &A, int n_units, int, int n, float thr)
{
    for (int u = 0; u < n_units; u++) {
        const km_score_unit &s = A.u[u];
        const int m = std::min(*s.d_n, n);
        for (int i = 0; i < m; i++) s.out[i] = s.score[i] >= thr ? (double)s.x0[i] : NAN;
        s.out[n - 1] = s.out[n - 1];
    }
    return KM_OK;
}
int kmi_units(km_ctx *, const km_score_units &A, int n_units, int, int n, float)
{
    for (int u = 0; u < n_units; u++) {
        const km_score_unit &s = A.u[u];
        const int m = std::min(*s.d_n, n);
        for (int i = 0; i < m; i++) { s.out[i] = s.x0[i]; s.out2[i] = -s.x0[i]; }
    }
    return KM_OK;
}
