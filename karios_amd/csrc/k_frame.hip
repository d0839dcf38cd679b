// K8: forward-backward test, score, compaction and (x0, y0) ordering of one tile's tracks
// (reference klt_tracker tail, karios/matcher/klt.py:142-170, and KLT._match_tile, klt.py:341-348).
//
//   d      = max(|p0 - p0r|) per point                       (float32)
//   keep   = d < float32(0.1)            (LK status is ignored by the reference)
//   score  = 1 - d / float32(0.1)
//   x0,y0  = p0 + tile offset ; dx,dy = p1 - p0
//   rows sorted by (x0, y0); the `index` column is the row's position in the un-sorted kept list, i.e. the
//   index labels pandas leaves after `sort_values(inplace=True)`.
// Every arithmetic step is a single correctly rounded float32 operation, as in numpy.
#include <cstring>
#include <string.h>

#include "common.hpp"

#include <type_traits>


// 256-thread workgroups throughout (round 6, see KF_T in k_select2.hip): the frame stage runs beside the dense kernels of the next
// submission, where a 1024-thread workgroup waits for a whole compute unit (fb_place 23 -> 822 us, fb_compact 5 -> 121 us).
#define FB_T 256

// What differs between the units of a batch (blockIdx.y = unit; km_klt_units_frame_submit); the single-unit entry point passes one
struct fb_unit {
    const float *p0, *p1, *p0r;
    const int *d_n;
    unsigned long long *keys;
    unsigned *ranks, *counts;
    float *tmp /* 5*cap */, *out /* 6*cap */;
    int *hdr;
    const km_scalars *sc_hdr;
    float x_off, y_off;
};
struct fb_units_args {
    fb_unit u[KM_UNITS_MAX];
};

// stable compaction of the kept points (rank = position in p0 order): pass 1 counts per workgroup, pass 2 writes
template <bool WRITE>
__global__ __launch_bounds__(FB_T) void fb_compact_kernel(fb_units_args A, int n_max, float back_thr, int cap, int n_sort)
{
    const fb_unit &U = A.u[blockIdx.y];
    const float *__restrict__ p0 = U.p0, *__restrict__ p1 = U.p1, *__restrict__ p0r = U.p0r;
    const int *__restrict__ d_n = U.d_n;
    const float x_off = U.x_off, y_off = U.y_off;
    unsigned long long *__restrict__ keys = U.keys;
    unsigned *__restrict__ ranks = U.ranks, *__restrict__ counts = U.counts;
    float *__restrict__ tmp = U.tmp;
    int *__restrict__ hdr = U.hdr;
    const km_scalars *__restrict__ sc_hdr = U.sc_hdr;
    __shared__ int s_wave[FB_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = min(d_n ? *d_n : n_max, n_max);
    const int i = blockIdx.x * FB_T + tid;
    bool keep = false;
    float x0 = 0, y0 = 0, x1 = 0, y1 = 0, d = 0;
    if (i < n) {
        x0 = p0[2 * i]; y0 = p0[2 * i + 1];
        d = fmaxf(fabsf(__fsub_rn(x0, p0r[2 * i])), fabsf(__fsub_rn(y0, p0r[2 * i + 1])));
        keep = d < back_thr;   // NaN compares false, like numpy
        if (WRITE) { x1 = p1[2 * i]; y1 = p1[2 * i + 1]; }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < FB_T / 64; k++) { const int cnt = s_wave[k]; if (k < wv) off += cnt; tot += cnt; }
    if (!WRITE) {
        if (tid == 0) counts[blockIdx.x] = (unsigned)tot;
        return;
    }
    int base = 0, total = 0;
    for (unsigned b = 0; b < gridDim.x; b++) { const int cb = (int)counts[b]; if (b < blockIdx.x) base += cb; total += cb; }
    if (i >= total && i < n_sort) keys[i] = ~0ull;   // sentinel keys behind the kept ones keep the sort length fixed
    if (keep) {
        const int r = base + off + __popcll(bal & ((1ull << lane) - 1ull));
        if (r < cap) {
            const float gx = __fadd_rn(x0, x_off), gy = __fadd_rn(y0, y_off);
            tmp[r] = gx; tmp[cap + r] = gy;
            tmp[2 * cap + r] = __fsub_rn(x1, x0); tmp[3 * cap + r] = __fsub_rn(y1, y0);
            tmp[4 * cap + r] = __fsub_rn(1.0f, d / back_thr);
            // (x0, y0) are integer-valued and non-negative: the u64 key orders exactly like the float pair
            keys[r] = ((unsigned long long)(unsigned)(int)gx << 32) | (unsigned long long)(unsigned)(int)gy;
            ranks[r] = (unsigned)r;
        }
    }
    // header words 2 / 3 of a tile that went through the synchronisation-free corner path: its flags and the exact candidate count
    // (the host sees them with the frame block)
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        hdr[0] = min(base + tot, cap); hdr[1] = n;
        hdr[2] = sc_hdr ? (int)sc_hdr->flags : 0; hdr[3] = sc_hdr ? (int)sc_hdr->cut[3] : 0;
    }
}

__global__ __launch_bounds__(256) void fb_gather_kernel(fb_units_args A, const unsigned *__restrict__ order, int cap)
{
    const fb_unit &U = A.u[blockIdx.y];
    const float *__restrict__ tmp = U.tmp;
    const int *__restrict__ hdr = U.hdr;
    float *__restrict__ out = U.out;
    const int m = hdr[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const unsigned r = order[i];
#pragma unroll
        for (int c2 = 0; c2 < 5; c2++) out[(size_t)c2 * cap + i] = tmp[(size_t)c2 * cap + r];
        out[(size_t)5 * cap + i] = __uint_as_float(r);   // index label, bit pattern of the int
    }
}

// Row order of the frame without a library sort (eight launches + a gather for 20 000 rows: ~50 us of mostly launch latency).
// Every workgroup buckets ALL kept rows by their x0 (a few tens of columns per bucket; counting sort in LDS: histogram, scan, fill - the
// same in every workgroup, 160 KB of keys from L2, eight loads in flight per thread) and then places its own slice of the
// rows: position = rows in lower buckets + rows of the same bucket with a smaller (x0, y0) key - ties (possible only with
// caller-supplied points) by their position in the kept list, like pandas' stable multi-column sort.  Corners keep a minimum
// distance, so a bucket holds a handful of rows.  One launch, no global synchronisation.
#define FBP_T 256
#define FBP_BINS 2048         // buckets of x0 at most (16 KB of counters + cursors; 2 B per row behind them)
__global__ __launch_bounds__(FBP_T) void fb_place_kernel(fb_units_args A, int cap, int shift, int nbins)
{
    const fb_unit &U = A.u[blockIdx.y];
    const unsigned long long *__restrict__ keys = U.keys;
    const float *__restrict__ tmp = U.tmp;
    const int *__restrict__ hdr = U.hdr;
    float *__restrict__ out = U.out;
    extern __shared__ unsigned fbp_smem[];
    unsigned *cnt = fbp_smem, *start = fbp_smem + nbins;          // start: nbins + 1 entries
    unsigned short *items = (unsigned short *)(start + nbins + 1);
    __shared__ unsigned s_wave[FBP_T / 64];
    const int m = hdr[0], tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = blockIdx.x * FBP_T;                            // rows this workgroup places: [r0, r0 + FBP_T), one per thread
    if (r0 >= m) return;
    const int mine_r = r0 + tid;
    const unsigned long long my_key = keys[min(mine_r, m - 1)];
    float col[5];
#pragma unroll
    for (int c2 = 0; c2 < 5; c2++) col[c2] = tmp[(size_t)c2 * cap + min(mine_r, m - 1)];
    for (int b = tid; b < nbins; b += FBP_T) cnt[b] = 0u;
    __syncthreads();
    // bucket of a row: its column inside the unit (x0 - tile origin) >> shift, clamped - a monotonic map of x0, so the order is exact for any column
    const unsigned xo = (unsigned)(int)U.x_off;
    auto bucket = [&](unsigned long long k) { return min(((unsigned)(k >> 32) - xo) >> shift, (unsigned)nbins - 1u); };
    // 1. histogram of the buckets (all m keys, 160 KB from L2; eight loads in flight per thread)
    for (int r = tid; r < m; r += 8 * FBP_T) {
        unsigned long long k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) k[u] = keys[min(r + u * FBP_T, m - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (r + u * FBP_T < m) atomicAdd(&cnt[bucket(k[u])], 1u);
    }
    __syncthreads();
    // 2. exclusive scan: every thread owns a run of consecutive buckets
    const int per = (nbins + FBP_T - 1) / FBP_T, b0 = min(tid * per, nbins), b1 = min(b0 + per, nbins);
    unsigned mine = 0;
    for (int b = b0; b < b1; b++) mine += cnt[b];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    unsigned run = incl - mine;
    for (int w = 0; w < wv; w++) run += s_wave[w];
    for (int b = b0; b < b1; b++) { const unsigned c2 = cnt[b]; start[b] = run; cnt[b] = run; run += c2; }   // cnt becomes the fill cursor
    if (tid == FBP_T - 1) start[nbins] = run;
    __syncthreads();
    // 3. fill the buckets
    for (int r = tid; r < m; r += 8 * FBP_T) {
        unsigned long long k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) k[u] = keys[min(r + u * FBP_T, m - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (r + u * FBP_T < m) items[atomicAdd(&cnt[bucket(k[u])], 1u)] = (unsigned short)(r + u * FBP_T);
    }
    __syncthreads();
    // 4. rank inside the bucket, write the row to its place
    if (mine_r < m) {
        const unsigned b = bucket(my_key);
        const unsigned lo = start[b], hi = start[b + 1];
        unsigned pos = lo;
        for (unsigned t = lo; t < hi; t += 4) {                 // four dependent (item -> key) loads in flight
            int j[4];
            unsigned long long kj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) j[u] = (int)items[min(t + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 4; u++) kj[u] = keys[j[u]];
#pragma unroll
            for (int u = 0; u < 4; u++) pos += (t + u < hi && (kj[u] < my_key || (kj[u] == my_key && j[u] < mine_r))) ? 1u : 0u;
        }
#pragma unroll
        for (int c2 = 0; c2 < 5; c2++) out[(size_t)c2 * cap + pos] = col[c2];
        out[(size_t)5 * cap + pos] = __uint_as_float((unsigned)mine_r);   // index label, bit pattern of the int
    }
}

// d_out: [header 4 ints: n_rows, n_init, flags, candidates][6 * cap floats: x0 | y0 | dx | dy | score | index bits]
// n units at once (n = 1: kf_frame): scratch = unit k's slices of WS_MISC0 / WS_MISC1 / WS_MISC2 / WS_FRAME_CNT
static int frame_launch(km_ctx *c, int n, const float *const *d_p0, const float *const *d_p1, const float *const *d_p0r, const int *const *d_n, int n_max,
                        int cap, float back_thr, const float *x_off, const float *y_off, void *const *d_out, const km_scalars *const *d_sc_header,
                        const int *width /* columns of each unit (corner columns < width), nullptr: unknown */ = nullptr)
{
    if (cap <= 0 || n_max <= 0) return km_fail(c, KM_E_ARG, "frame: empty capacity");
    // the tile origin is added to the corner coordinates and the sum becomes the (x0, y0) ordering key: a pair of non-negative
    // integers.  The reference's origins are tile offsets (klt.py:341-342); anything else is refused instead of mis-ordered.
    float x_hi = 0.f;
    for (int k = 0; k < n; k++) {
        if (!(x_off[k] >= 0.f && x_off[k] <= 1073741824.f && y_off[k] >= 0.f && y_off[k] <= 1073741824.f))
            return km_fail(c, KM_E_ARG, "frame: tile origin (%g, %g) must be finite and within [0, 2^30]", (double)x_off[k], (double)y_off[k]);
        x_hi = fmaxf(x_hi, x_off[k]);
    }
    const int nblk = (n_max + FB_T - 1) / FB_T;
    const unsigned n_sort = (unsigned)(n_max < cap ? n_max : cap);   // fixed sort length: sentinel keys behind the kept rows
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t kb = up((size_t)cap * 2 * sizeof(unsigned long long)), rb = up((size_t)cap * 2 * sizeof(unsigned)), tb = up((size_t)cap * 5 * sizeof(float)),
                 cb = up((size_t)nblk * sizeof(unsigned));
    char *keys = (char *)km_ws(c, WS_MISC0, kb * n), *ranks = (char *)km_ws(c, WS_MISC1, rb * n), *tmp = (char *)km_ws(c, WS_MISC2, tb * n);
    char *counts = (char *)km_ws(c, WS_FRAME_CNT, cb * n);
    if (!keys || !ranks || !tmp || !counts) return KM_E_NOMEM;
    fb_units_args A;
    for (int k = 0; k < n; k++) {
        fb_unit &u = A.u[k];
        u.p0 = d_p0[k]; u.p1 = d_p1[k]; u.p0r = d_p0r[k]; u.d_n = d_n[k]; u.x_off = x_off[k]; u.y_off = y_off[k];
        u.keys = (unsigned long long *)(keys + kb * k); u.ranks = (unsigned *)(ranks + rb * k); u.tmp = (float *)(tmp + tb * k);
        u.counts = (unsigned *)(counts + cb * k);
        u.hdr = (int *)d_out[k]; u.out = (float *)((char *)d_out[k] + 16); u.sc_hdr = d_sc_header ? d_sc_header[k] : nullptr;
    }
    fb_compact_kernel<false><<<dim3(nblk, n), FB_T, 0, c->stream>>>(A, n_max, back_thr, cap, (int)n_sort);
    KM_LAUNCH_CHECK(c);
    fb_compact_kernel<true><<<dim3(nblk, n), FB_T, 0, c->stream>>>(A, n_max, back_thr, cap, (int)n_sort);
    KM_LAUNCH_CHECK(c);
    if (n_sort <= 32768u) {
        // up to a few 10^4 rows (maxCorners of a tile): every workgroup buckets all rows by x0 in LDS (<= 16 KB of buckets + 2 B per row) and
        // places 256 of them
        unsigned x_max = 65535u;                                 // corner columns (x0 - tile origin) < 65536
        if (width) { x_max = 1u; for (int k = 0; k < n; k++) x_max = (unsigned)width[k] > x_max ? (unsigned)width[k] : x_max; }
        (void)x_hi;
        int shift = 0;
        while (((x_max >> shift) + 1u) > (unsigned)FBP_BINS) shift++;
        const int nbins = (int)(x_max >> shift) + 1;
        const size_t smem = ((size_t)2 * nbins + 1) * sizeof(unsigned) + (size_t)n_sort * sizeof(unsigned short);
        static unsigned long long opted = 0;   // per DEVICE
        if (smem > 48 * 1024 && !(opted & (1ull << (c->device & 63)))) {
            KM_HIP(c, hipFuncSetAttribute((const void *)fb_place_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            opted |= 1ull << (c->device & 63);
        }
        fb_place_kernel<<<dim3((n_sort + FBP_T - 1) / FBP_T, n), FBP_T, smem, c->stream>>>(A, cap, shift, nbins);   // (one row per thread)
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
    // maxCorners = 0 on a large tile: hundreds of thousands of rows - radix sort of (key, rank) pairs (k_sort.hip) + gather
    if (n != 1) return km_fail(c, KM_E_UNSUPPORTED, "frame: batched units hold at most 32768 rows each");
    { const int rs = km_sort_u64(c, A.u[0].keys, A.u[0].keys + cap, A.u[0].ranks, A.u[0].ranks + cap, n_sort, false); if (rs) return rs; }
    fb_gather_kernel<<<32, 256, 0, c->stream>>>(A, A.u[0].ranks, cap);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kf_frame(km_ctx *c, const float *d_p0, const float *d_p1, const float *d_p0r, const int *d_n, int n_max, int cap, float back_thr,
             float x_off, float y_off, void *d_out, const km_scalars *d_sc_header, int width)
{
    return frame_launch(c, 1, &d_p0, &d_p1, &d_p0r, &d_n, n_max, cap, back_thr, &x_off, &y_off, &d_out, d_sc_header ? &d_sc_header : nullptr,
                        width > 0 ? &width : nullptr);
}

// FB test + frame of every unit of a batch: three launches for all of them
int kf_frame_units(km_ctx *c, const km_units &U, int n_max, int cap, float back_thr)
{
    const float *p0[KM_UNITS_MAX], *p1[KM_UNITS_MAX], *p0r[KM_UNITS_MAX];
    const int *dn[KM_UNITS_MAX];
    void *out[KM_UNITS_MAX];
    const km_scalars *sc[KM_UNITS_MAX];
    for (int k = 0; k < U.n; k++) { p0[k] = U.p0[k]; p1[k] = U.p1[k]; p0r[k] = U.p0r[k]; dn[k] = &U.sc[k]->n_corners; out[k] = U.frame[k]; sc[k] = U.sc[k]; }
    return frame_launch(c, U.n, p0, p1, p0r, dn, n_max, cap, back_thr, U.x_off, U.y_off, out, sc, U.W);
}

// ---- K13: DN-value filter of KariosAPI._filter_by_dn_values (reference karios/api/core.py:650-737): a key point is
// dropped when the reference OR monitored pixel under it (x = int(x0), y = int(y0), truncation) equals one of the
// user's `no_values`, or when a pixel equals its own image's no-data value.  One thread per key point; comparisons in
// fp64 like numpy's (`array == value` with the value promoted).  keep[i] = 2 flags a key point outside the image
// (numpy would raise IndexError / wrap a negative index: the caller turns it into an error).
template <typename T>
__global__ __launch_bounds__(256) void dn_keep_kernel(const T *__restrict__ ref, const T *__restrict__ mon, int H, int W, ptrdiff_t sref,
                                                      ptrdiff_t smon, const float *__restrict__ x0, const float *__restrict__ y0, int n,
                                                      const double *__restrict__ no_values, int n_no, int has_ref_nd, double ref_nd,
                                                      int has_mon_nd, double mon_nd, uint8_t *__restrict__ keep)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x = (int)x0[i], y = (int)y0[i];
    if (x < 0 || x >= W || y < 0 || y >= H) { keep[i] = 2; return; }
    const double r = (double)ref[(size_t)y * sref + x], m = (double)mon[(size_t)y * smon + x];
    bool k = true;
    for (int j = 0; j < n_no; j++) k = k && !(r == no_values[j] || m == no_values[j]);
    if (has_ref_nd) k = k && !(r == ref_nd);
    if (has_mon_nd) k = k && !(m == mon_nd);
    keep[i] = k ? 1 : 0;
}

int kf_dn_keep(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref, ptrdiff_t smon, const float *d_x0,
               const float *d_y0, int n, const double *d_no_values, int n_no, const double *ref_nd, const double *mon_nd, uint8_t *d_keep)
{
    if (n <= 0) return KM_OK;
    const int nb = (n + 255) / 256;
    const int hr = ref_nd != nullptr, hm = mon_nd != nullptr;
    const double rv = ref_nd ? *ref_nd : 0.0, mv = mon_nd ? *mon_nd : 0.0;
    switch (dtype) {
#define KM_DN_CASE(CODE, T) \
    case CODE: dn_keep_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, H, W, sref, smon, d_x0, d_y0, n, d_no_values, n_no, hr, rv, hm, mv, d_keep); break;
        KM_DN_CASE(KM_U8, uint8_t) KM_DN_CASE(KM_U16, uint16_t) KM_DN_CASE(KM_I16, int16_t) KM_DN_CASE(KM_F32, float)
#undef KM_DN_CASE
    default: return km_fail(c, KM_E_ARG, "dn filter: bad dtype %d", dtype);
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// ---- forward-backward inlier count of one tracker run (klt.py:142-144), for the auto-ksize search
__global__ __launch_bounds__(256) void fb_count_kernel(const float *__restrict__ p0, const float *__restrict__ p0r, const int *__restrict__ d_n,
                                                       int n_max, float back_thr, int *__restrict__ count)
{
    const int n = min(d_n ? *d_n : n_max, n_max);
    int local = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float d = fmaxf(fabsf(__fsub_rn(p0[2 * i], p0r[2 * i])), fabsf(__fsub_rn(p0[2 * i + 1], p0r[2 * i + 1])));
        local += d < back_thr ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

int kf_count_kept(km_ctx *c, const float *d_p0, const float *d_p0r, const int *d_n, int n_max, float back_thr, int *d_count)
{
    if (n_max <= 0) return KM_OK;
    fb_count_kernel<<<32, 256, 0, c->stream>>>(d_p0, d_p0r, d_n, n_max, back_thr, d_count);   // *d_count zeroed by the caller
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

__global__ __launch_bounds__(256) void fb_count_jobs_kernel(km_count_jobs J, int n_max, float back_thr, int *__restrict__ counts)
{
    const int j = blockIdx.y;
    const float *__restrict__ p0 = J.p0[j], *__restrict__ p0r = J.p0r[j];
    const int n = min(J.d_n[j] ? *J.d_n[j] : n_max, n_max);
    int local = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float d = fmaxf(fabsf(__fsub_rn(p0[2 * i], p0r[2 * i])), fabsf(__fsub_rn(p0[2 * i + 1], p0r[2 * i + 1])));
        local += d < back_thr ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counts[j], local);
}

int kf_count_kept_jobs(km_ctx *c, const km_count_jobs &J, int n_jobs, int n_max, float back_thr, int *d_counts)
{
    if (n_max <= 0 || n_jobs <= 0) return KM_OK;
    fb_count_jobs_kernel<<<dim3(16, n_jobs), 256, 0, c->stream>>>(J, n_max, back_thr, d_counts);   // counts zeroed by the caller
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// ------------------------------------------------------------------ KARIOS_HIP_UPLOAD_CHECKSUM (staging.hip): what does a kernel right behind an upload see?
// out[y] = sum over the bytes b_i of row y of (b_i + 1) * (2 i + 1)  (mod 2^64; the host evaluates the same sum on the source rows)
__global__ __launch_bounds__(256) void row_checksum_kernel(const uint8_t *__restrict__ img, size_t row_bytes, int rows, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[4];
    const int y = blockIdx.x;
    if (y >= rows) return;
    const uint8_t *p = img + (size_t)y * row_bytes;
    unsigned long long s = 0;
    for (size_t i = threadIdx.x; i < row_bytes; i += 256) s += (unsigned long long)(p[i] + 1u) * (unsigned long long)(2 * i + 1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[y] = part[0] + part[1] + part[2] + part[3];
}


int kf_row_checksum(km_ctx *c, const void *d_img, size_t row_bytes, int rows, unsigned long long *d_out)
{
    if (rows <= 0) return KM_OK;
    row_checksum_kernel<<<rows, 256, 0, c->stream>>>((const uint8_t *)d_img, row_bytes, rows, d_out);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
