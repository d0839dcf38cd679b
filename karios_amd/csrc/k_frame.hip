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

#include <rocprim/device/device_radix_sort.hpp>

#define FB_T 1024

// stable compaction of the kept points (rank = position in p0 order): pass 1 counts per workgroup, pass 2 writes
template <bool WRITE>
__global__ __launch_bounds__(FB_T) void fb_compact_kernel(const float *__restrict__ p0, const float *__restrict__ p1,
                                                          const float *__restrict__ p0r, const int *__restrict__ d_n, int n_max,
                                                          float back_thr, float x_off, float y_off, unsigned long long *__restrict__ keys,
                                                          unsigned *__restrict__ ranks, float *__restrict__ tmp /* 5*cap */, int cap,
                                                          int *__restrict__ hdr, unsigned *__restrict__ counts, int n_sort)
{
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
    if (blockIdx.x == gridDim.x - 1 && tid == 0) { hdr[0] = min(base + tot, cap); hdr[1] = n; hdr[2] = 0; hdr[3] = 0; }
}

__global__ __launch_bounds__(256) void fb_gather_kernel(const unsigned *__restrict__ order, const float *__restrict__ tmp, int cap,
                                                        const int *__restrict__ hdr, float *__restrict__ out /* 6*cap */)
{
    const int m = hdr[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const unsigned r = order[i];
#pragma unroll
        for (int c2 = 0; c2 < 5; c2++) out[(size_t)c2 * cap + i] = tmp[(size_t)c2 * cap + r];
        out[(size_t)5 * cap + i] = __uint_as_float(r);   // index label, bit pattern of the int
    }
}

// d_out: [header 4 ints: n_rows, n_init, 0, 0][6 * cap floats: x0 | y0 | dx | dy | score | index bits]
int kf_frame(km_ctx *c, const float *d_p0, const float *d_p1, const float *d_p0r, const int *d_n, int n_max, int cap, float back_thr,
             float x_off, float y_off, void *d_out)
{
    if (cap <= 0 || n_max <= 0) return km_fail(c, KM_E_ARG, "frame: empty capacity");
    int *hdr = (int *)d_out;
    float *out = (float *)((char *)d_out + 16);
    unsigned long long *keys = (unsigned long long *)km_ws(c, WS_MISC0, (size_t)cap * 2 * sizeof(unsigned long long));
    unsigned *ranks = (unsigned *)km_ws(c, WS_MISC1, (size_t)cap * 2 * sizeof(unsigned));
    float *tmp = (float *)km_ws(c, WS_MISC2, (size_t)cap * 5 * sizeof(float));
    if (!keys || !ranks || !tmp) return KM_E_NOMEM;
    unsigned long long *keys_alt = keys + cap;
    unsigned *ranks_alt = ranks + cap;
    const int nblk = (n_max + FB_T - 1) / FB_T;
    const unsigned n_sort = (unsigned)(n_max < cap ? n_max : cap);   // fixed sort length: sentinel keys behind the kept rows
    unsigned *counts = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)nblk * sizeof(unsigned));
    if (!counts) return KM_E_NOMEM;
    fb_compact_kernel<false><<<nblk, FB_T, 0, c->stream>>>(d_p0, d_p1, d_p0r, d_n, n_max, back_thr, x_off, y_off, keys, ranks, tmp, cap, hdr, counts, (int)n_sort);
    KM_LAUNCH_CHECK(c);
    fb_compact_kernel<true><<<nblk, FB_T, 0, c->stream>>>(d_p0, d_p1, d_p0r, d_n, n_max, back_thr, x_off, y_off, keys, ranks, tmp, cap, hdr, counts, (int)n_sort);
    KM_LAUNCH_CHECK(c);
    size_t tmp_bytes = 0;
    KM_HIP(c, rocprim::radix_sort_pairs((void *)nullptr, tmp_bytes, keys, keys_alt, ranks, ranks_alt, n_sort, 0, 64, c->stream));
    void *stmp = km_ws(c, WS_SORT_TMP, tmp_bytes ? tmp_bytes : 16);
    if (!stmp) return KM_E_NOMEM;
    KM_HIP(c, rocprim::radix_sort_pairs(stmp, tmp_bytes, keys, keys_alt, ranks, ranks_alt, n_sort, 0, 64, c->stream));
    fb_gather_kernel<<<32, 256, 0, c->stream>>>(ranks_alt, tmp, cap, hdr, out);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
