// K5: candidate ranking + greedy minimum-distance corner selection
// (cv2.goodFeaturesToTrack steps 6-8; reference call site klt.py:120).
//
// Ranking: one descending u64 radix sort (rocPRIM) of (f32 bits << 32 | raster index) keys
// = OpenCV's greaterThanPtr order (value desc, address desc).
//
// Selection: OpenCV walks the ranked list sequentially and accepts a candidate iff no
// already-accepted corner lies closer than minDistance.  The accepted set only depends on
// higher-ranked accepted corners, so it is reproduced exactly by one persistent 1024-thread
// workgroup that consumes the ranked list in batches:
//   phase 1  every lane tests its candidate against the accepted-corner cell grid (global),
//   phase 2  survivors of the batch are resolved against each other in LDS by rank-ordered
//            fixed-point rounds (accept when no higher-ranked survivor within range is
//            undecided or accepted; reject when one is accepted),
//   phase 3  accepted corners are appended in rank order and inserted in the grid.
// It stops as soon as maxCorners corners are out, which on dense imagery is after a few
// percent of the candidate list.
#include <cstring>
#include <string.h>

#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>

int ks_sort_keys_desc(km_ctx *c, unsigned long long *d_keys, size_t n, unsigned long long **d_sorted)
{
    unsigned long long *alt = (unsigned long long *)km_ws(c, WS_KEYS1, n * sizeof(unsigned long long));
    if (!alt) return KM_E_NOMEM;
    size_t tmp_bytes = 0;
    KM_HIP(c, rocprim::radix_sort_keys_desc((void *)nullptr, tmp_bytes, d_keys, alt, n, 0, 64, c->stream));
    void *tmp = km_ws(c, WS_SORT_TMP, tmp_bytes ? tmp_bytes : 16);
    if (!tmp) return KM_E_NOMEM;
    KM_HIP(c, rocprim::radix_sort_keys_desc(tmp, tmp_bytes, d_keys, alt, n, 0, 64, c->stream));
    *d_sorted = alt;
    return KM_OK;
}

#define SEL_T 1024
#define SEL_SLOTS 4

enum { ST_UNDECIDED = 0, ST_ACCEPT = 1, ST_REJECT = 2 };

__global__ __launch_bounds__(SEL_T) void select_kernel(const unsigned long long *__restrict__ keys, unsigned n, int W, int cell,
                                                       int gw, int gh, double md2, unsigned *grid_cnt, unsigned *grid_pts,
                                                       int max_corners, int cap, float *__restrict__ out_xy, km_scalars *sc)
{
    __shared__ short s_x[SEL_T], s_y[SEL_T];
    __shared__ unsigned char s_state[SEL_T];
    __shared__ int s_wave_cnt[SEL_T / 64];
    __shared__ int s_nacc, s_changed;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_nacc = 0;
    __syncthreads();
    int batches = 0;
    for (unsigned base = 0; base < n; base += SEL_T) {
        const int nacc0 = s_nacc;
        if (max_corners > 0 && nacc0 >= max_corners) break;
        batches++;
        // ---- phase 1: test against the accepted grid
        const unsigned i = base + tid;
        bool alive = i < n;
        int x = 0, y = 0;
        if (alive) {
            const unsigned idx = (unsigned)(keys[i] & 0xffffffffull);
            y = (int)(idx / (unsigned)W);
            x = (int)(idx - (unsigned)y * (unsigned)W);
            const int xc = x / cell, yc = y / cell;
            const int x1 = max(xc - 1, 0), y1 = max(yc - 1, 0), x2 = min(xc + 1, gw - 1), y2 = min(yc + 1, gh - 1);
            for (int yy = y1; yy <= y2 && alive; yy++)
                for (int xx = x1; xx <= x2 && alive; xx++) {
                    const int g = yy * gw + xx;
                    const unsigned cnt = __hip_atomic_load(&grid_cnt[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    for (unsigned k = 0; k < min(cnt, (unsigned)SEL_SLOTS); k++) {
                        const unsigned p = __hip_atomic_load(&grid_pts[g * SEL_SLOTS + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const float dx = (float)x - (float)(p & 0xffffu), dy = (float)y - (float)(p >> 16);
                        if ((double)(dx * dx + dy * dy) < md2) { alive = false; break; }
                    }
                }
        }
        // compact survivors in rank order
        const unsigned long long bal = __ballot(alive);
        if (lane == 0) s_wave_cnt[wv] = __popcll(bal);
        __syncthreads();
        int off = 0, tot = 0;
        for (int k = 0; k < SEL_T / 64; k++) { const int cnt = s_wave_cnt[k]; if (k < wv) off += cnt; tot += cnt; }
        if (alive) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            s_x[pos] = (short)x; s_y[pos] = (short)y; s_state[pos] = ST_UNDECIDED;
        }
        __syncthreads();
        const int ns = tot;
        // ---- phase 2: rank-ordered fixed point among the survivors
        if (ns > 0) {
            const bool mine = tid < ns;
            const int mx = mine ? s_x[tid] : 0, my = mine ? s_y[tid] : 0;
            int st = mine ? ST_UNDECIDED : ST_REJECT;
            for (;;) {
                int nst = st;
                if (st == ST_UNDECIDED) {
                    bool blocked = false, rejected = false;
                    for (int t = 0; t < tid; t++) {
                        const int o = s_state[t];
                        if (o == ST_REJECT) continue;
                        const float dx = (float)(mx - s_x[t]), dy = (float)(my - s_y[t]);
                        if ((double)(dx * dx + dy * dy) < md2) {
                            if (o == ST_ACCEPT) { rejected = true; break; }
                            blocked = true;
                        }
                    }
                    nst = rejected ? ST_REJECT : blocked ? ST_UNDECIDED : ST_ACCEPT;
                }
                __syncthreads();  // all reads of s_state for this round done
                if (tid == 0) s_changed = 0;
                __syncthreads();
                if (nst != st) { s_state[tid] = (unsigned char)nst; st = nst; s_changed = 1; }
                __syncthreads();
                if (!s_changed) break;
            }
            // ---- phase 3: append accepted in rank order, insert in the grid
            const bool acc = mine && st == ST_ACCEPT;
            const unsigned long long ab = __ballot(acc);
            if (lane == 0) s_wave_cnt[wv] = __popcll(ab);
            __syncthreads();
            int aoff = 0, atot = 0;
            for (int k = 0; k < SEL_T / 64; k++) { const int cnt = s_wave_cnt[k]; if (k < wv) aoff += cnt; atot += cnt; }
            if (acc) {
                const int pos = nacc0 + aoff + __popcll(ab & ((1ull << lane) - 1ull));
                if ((max_corners <= 0 || pos < max_corners) && pos < cap) {
                    out_xy[2 * pos] = (float)mx;
                    out_xy[2 * pos + 1] = (float)my;
                }
                // accepted corners in one cell are >= minDistance apart and the cell side is
                // <= minDistance + 0.5, so a cell never holds more than SEL_SLOTS of them
                const int g = (my / cell) * gw + (mx / cell);
                const unsigned slot = atomicAdd(&grid_cnt[g], 1u);
                if (slot < SEL_SLOTS)
                    __hip_atomic_store(&grid_pts[g * SEL_SLOTS + slot], (unsigned)mx | ((unsigned)my << 16), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                else
                    sc->n_cand = 0xffffffffu;  // cannot happen (see above); poison so the host notices
            }
            if (tid == 0) s_nacc = nacc0 + atot;
            __threadfence();  // grid stores visible before the next batch's loads
            __syncthreads();
        }
    }
    if (tid == 0) {
        int na = s_nacc;
        if (max_corners > 0 && na > max_corners) na = max_corners;
        sc->n_corners = na;
        sc->n_batches = batches;
    }
}

// minDistance < 1: the first maxCorners ranked candidates (featureselect.cpp else-branch)
__global__ __launch_bounds__(256) void take_first_kernel(const unsigned long long *__restrict__ keys, unsigned n, int W, int max_corners,
                                                         int cap, float *__restrict__ out_xy, km_scalars *sc)
{
    unsigned lim = n;
    if (max_corners > 0 && lim > (unsigned)max_corners) lim = (unsigned)max_corners;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < lim; i += gridDim.x * blockDim.x) {
        const unsigned idx = (unsigned)(keys[i] & 0xffffffffull);
        const unsigned y = idx / (unsigned)W, x = idx - y * (unsigned)W;
        if (i < (unsigned)cap) { out_xy[2 * i] = (float)x; out_xy[2 * i + 1] = (float)y; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->n_corners = (int)lim; sc->n_batches = 0; }
}

int ks_select(km_ctx *c, const unsigned long long *d_sorted, size_t n, int H, int W, int max_corners, double min_distance,
              float *d_xy, int cap, km_scalars *d_sc)
{
    if (n > 0xffffffffull) return km_fail(c, KM_E_UNSUPPORTED, "too many candidates");
    if (W > 65535 || H > 65535) return km_fail(c, KM_E_UNSUPPORTED, "tile larger than 65535 px per side");
    if (n == 0) {
        KM_HIP(c, hipMemsetAsync(&d_sc->n_corners, 0, 2 * sizeof(int), c->stream));
        return KM_OK;
    }
    if (!(min_distance >= 1)) {
        take_first_kernel<<<64, 256, 0, c->stream>>>(d_sorted, (unsigned)n, W, max_corners, cap, d_xy, d_sc);
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
    const int cell = (int)lrint(min_distance);
    const int gw = (W + cell - 1) / cell, gh = (H + cell - 1) / cell;
    const size_t cells = (size_t)gw * gh;
    // layout: [cells] counts | [cells*SLOTS] packed points (x | y<<16)
    unsigned *grid = (unsigned *)km_ws(c, WS_GRID, cells * (SEL_SLOTS + 1) * sizeof(unsigned));
    if (!grid) return KM_E_NOMEM;
    KM_HIP(c, hipMemsetAsync(grid, 0, cells * sizeof(unsigned), c->stream));
    unsigned *grid_cnt = grid, *grid_pts = grid + cells;
    select_kernel<<<1, SEL_T, 0, c->stream>>>(d_sorted, (unsigned)n, W, cell, gw, gh, min_distance * min_distance, grid_cnt, grid_pts,
                                              max_corners, cap, d_xy, d_sc);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
