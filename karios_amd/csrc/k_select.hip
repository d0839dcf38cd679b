// K5: candidate ranking + greedy minimum-distance corner selection
// (cv2.goodFeaturesToTrack steps 6-8; reference call site klt.py:120).
//
// Ranking: one descending u64 radix sort (rocPRIM) of (f32 bits << 32 | raster index) keys
// = OpenCV's greaterThanPtr order (value desc, address desc).
//
// Selection: OpenCV walks the ranked list sequentially and accepts a candidate iff no
// already-accepted corner lies closer than minDistance, stopping at maxCorners.  A candidate's
// fate depends only on HIGHER-ranked candidates within minDistance, so
//   (1) the decisions on any ranked prefix [0,K) equal the sequential ones, and
//   (2) inside the prefix they are the unique fixed point of
//         accept  <=> every higher-ranked neighbour (< minDistance) is rejected
//         reject  <=> some higher-ranked neighbour is accepted
//       which is reached by data-parallel sweeps over a cell grid (cell = cvRound(minDistance),
//       3x3 cells searched, exactly the buckets OpenCV uses).
// The first maxCorners accepted candidates in rank order (one scan) are the result; if the prefix
// yields fewer, it is enlarged 4x and the sweeps continue from the states already decided.
#include <cstring>
#include <string.h>

#include "common.hpp"

int ks_sort_keys_desc(km_ctx *c, unsigned long long *d_keys, size_t n, unsigned long long **d_sorted)
{
    unsigned long long *alt = (unsigned long long *)km_ws(c, WS_KEYS1, n * sizeof(unsigned long long));
    if (!alt) return KM_E_NOMEM;
    const int rc = km_sort_u64(c, d_keys, alt, nullptr, nullptr, n, true);   // (in place: the second buffer is scratch)
    if (rc) return rc;
    *d_sorted = d_keys;
    return KM_OK;
}

enum { ST_UNDECIDED = 0, ST_ACCEPT = 1, ST_REJECT = 2 };

__device__ __forceinline__ void key_xy(unsigned long long key, int W, int &x, int &y)
{
    const unsigned idx = (unsigned)(key & 0xffffffffull);
    y = (int)(idx / (unsigned)W);
    x = (int)(idx - (unsigned)y * (unsigned)W);
}

// count prefix candidates [k0, k1) per cell
__global__ __launch_bounds__(256) void sel_count_kernel(const unsigned long long *__restrict__ keys, unsigned k0, unsigned k1, int W,
                                                        int cell, int gw, unsigned *__restrict__ cell_cnt, unsigned *__restrict__ state)
{
    const unsigned i = k0 + blockIdx.x * 256 + threadIdx.x;
    if (i >= k1) return;
    state[i] = ST_UNDECIDED;   // new prefix members start undecided (older ones keep their final state)
    int x, y;
    key_xy(keys[i], W, x, y);
    atomicAdd(&cell_cnt[(y / cell) * gw + (x / cell)], 1u);
}

// scatter candidate ranks [0, k1) into their cell's slice (order inside a cell is irrelevant)
__global__ __launch_bounds__(256) void sel_fill_kernel(const unsigned long long *__restrict__ keys, unsigned k1, int W, int cell, int gw,
                                                       const unsigned *__restrict__ cell_off, unsigned *__restrict__ cell_fill,
                                                       unsigned *__restrict__ items)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k1) return;
    int x, y;
    key_xy(keys[i], W, x, y);
    const int g = (y / cell) * gw + (x / cell);
    items[cell_off[g] + atomicAdd(&cell_fill[g], 1u)] = i;
}

// one launch = up to SWEEPS relaxation sweeps over the prefix; states only move UNDECIDED -> final and
// every final state is the sequential algorithm's decision, so concurrent in-place updates are safe.
#define SEL_SWEEPS 4
__global__ __launch_bounds__(256) void sel_sweep_kernel(const unsigned long long *__restrict__ keys, unsigned k1, int W, int cell, int gw,
                                                        int gh, double md2, const unsigned *__restrict__ cell_off,
                                                        const unsigned *__restrict__ cell_cnt, const unsigned *__restrict__ items,
                                                        unsigned *state, unsigned *n_undecided)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    bool undecided = false;
    if (i < k1 && __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ST_UNDECIDED) {
        int x, y;
        key_xy(keys[i], W, x, y);
        const int xc = x / cell, yc = y / cell;
        const int x1 = max(xc - 1, 0), y1 = max(yc - 1, 0), x2 = min(xc + 1, gw - 1), y2 = min(yc + 1, gh - 1);
        undecided = true;
        for (int sweep = 0; sweep < SEL_SWEEPS && undecided; sweep++) {
            bool blocked = false, rejected = false;
            for (int yy = y1; yy <= y2 && !rejected; yy++) {
                // the (up to) three cells of a grid row are neighbours in the cell order, so their candidates form ONE
                // contiguous range of `items`: two offset loads per row instead of two per cell
                const int g0 = yy * gw + x1;
                const unsigned o = cell_off[g0], e = cell_off[g0 + (x2 - x1 + 1)];
                for (unsigned k = o; k < e; k++) {
                    const unsigned j = items[k];
                    if (j >= i) continue;  // only higher-ranked candidates matter
                    int xj, yj;
                    key_xy(keys[j], W, xj, yj);
                    const float dx = (float)x - (float)xj, dy = (float)y - (float)yj;
                    if (!((double)(dx * dx + dy * dy) < md2)) continue;
                    const unsigned sj = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (sj == ST_ACCEPT) { rejected = true; break; }
                    if (sj == ST_UNDECIDED) blocked = true;
                }
            }
            if (rejected) { __hip_atomic_store(&state[i], (unsigned)ST_REJECT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
            else if (!blocked) { __hip_atomic_store(&state[i], (unsigned)ST_ACCEPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
        }
    }
    const unsigned long long bal = __ballot(undecided);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_undecided, (unsigned)__popcll(bal));
}

// pos = exclusive scan of the accept flags = index in OpenCV's output order
__global__ __launch_bounds__(256) void sel_emit_kernel(const unsigned long long *__restrict__ keys, const unsigned *__restrict__ state,
                                                       const unsigned *__restrict__ pos, unsigned k1, int W, int max_corners, int cap,
                                                       float *__restrict__ out_xy, km_scalars *sc, int rounds)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k1) return;
    const bool acc = state[i] == ST_ACCEPT;
    const unsigned p = pos[i];
    if (acc && (max_corners <= 0 || p < (unsigned)max_corners) && p < (unsigned)cap) {
        int x, y;
        key_xy(keys[i], W, x, y);
        out_xy[2 * p] = (float)x;
        out_xy[2 * p + 1] = (float)y;
    }
    if (i == k1 - 1) {
        int total = (int)(p + (acc ? 1u : 0u));
        if (max_corners > 0 && total > max_corners) total = max_corners;
        sc->n_corners = total;
        sc->n_batches = rounds;
        sc->und[7] = (unsigned)total;   // travels to the host together with the undecided counters
    }
}

// ---- top-K pre-filter: only the strongest candidates can reach the first maxCorners corners, so the full
// candidate list (several 10^6 keys) is cut down before the radix sort.  Values are binned by the distance of
// their top 18 float bits from the maximum's (bins of ~0.2 %); the smallest bin bound D with at least K_target
// candidates above it defines the kept set, which is a rank prefix of the full list.
#define TK_NB KM_TK_NB
#define TK_SHIFT 14

__device__ __forceinline__ unsigned tk_bin(unsigned long long key, unsigned top)
{
    const unsigned b = (unsigned)(key >> (32 + TK_SHIFT));
    const unsigned d = top > b ? top - b : 0u;
    return d < TK_NB - 1 ? d : TK_NB - 1;
}

// exact TOZERO threshold of goodFeaturesToTrack: thr = (float)(maxVal * qualityLevel); candidates need value > thr
__device__ __forceinline__ float tk_threshold(const km_scalars *sc, double quality)
{
    const unsigned mk = sc->max_eig_key;
    const unsigned b = (mk & 0x80000000u) ? (mk & 0x7fffffffu) : ~mk;
    const float maxv = mk ? __uint_as_float(b) : 0.f;
    return (float)__dmul_rn((double)maxv, quality);
}
__device__ __forceinline__ bool tk_above(unsigned long long key, float thr) { return __uint_as_float((unsigned)(key >> 32)) > thr; }

__global__ __launch_bounds__(1024) void tk_hist_kernel(const unsigned long long *__restrict__ keys, unsigned cap, const km_scalars *sc,
                                                       double quality, unsigned *__restrict__ hist)
{
    // the key buffer is KM_NSHARD regions of cap / KM_NSHARD slots; blockIdx.y selects the region
    const unsigned cap_s = cap / KM_NSHARD;
    const unsigned n = min(sc->shard_cnt[blockIdx.y], cap_s);
    keys += (size_t)blockIdx.y * cap_s;
    const float thr = tk_threshold(sc, quality);
    __shared__ unsigned h[TK_NB];
    for (int i = threadIdx.x; i < TK_NB; i += 1024) h[i] = 0;
    __syncthreads();
    const unsigned top = (sc->max_eig_key & 0x7fffffffu) >> TK_SHIFT;   // max eig > 0: ordered key = bits | 0x80000000
    // the clamp bin collects the bulk of the (weak) candidates: count it per wave, not per lane
    unsigned tail = 0;
    for (unsigned b = blockIdx.x * 1024; b < n; b += gridDim.x * 1024) {
        const unsigned i = b + threadIdx.x;
        unsigned d = 0xffffffffu;
        if (i < n) { const unsigned long long kk = keys[i]; if (tk_above(kk, thr)) d = tk_bin(kk, top); }
        const bool is_tail = d == TK_NB - 1;
        tail += (unsigned)__popcll(__ballot(is_tail));
        if (d < TK_NB - 1) atomicAdd(&h[d], 1u);
    }
    if ((threadIdx.x & 63) == 0 && tail) atomicAdd(&h[TK_NB - 1], tail);
    __syncthreads();
    for (int i = threadIdx.x; i < TK_NB; i += 1024)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// hist[TK_NB] -> cut[0] = D (largest kept bin), cut[1] = number of kept keys
__global__ __launch_bounds__(1024) void tk_cut_kernel(const unsigned *__restrict__ hist, unsigned k_target, unsigned *__restrict__ cut,
                                                      km_scalars *sc, double quality)
{
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_first;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned a = hist[2 * t], b = hist[2 * t + 1];
    // inclusive scan of the pair sums across the workgroup
    unsigned v = a + b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) s_wave[wv] = v;
    if (t == 0) s_first = 0xffffffffu;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < 16; w++) { if (w < wv) base += s_wave[w]; total += s_wave[w]; }
    const unsigned incl_b = base + v, incl_a = incl_b - b;     // cumulative counts through bins 2t and 2t+1
    // first bin whose cumulative count reaches k_target
    unsigned mine = 0xffffffffu;
    if (incl_a >= k_target) mine = 2 * t;
    else if (incl_b >= k_target) mine = 2 * t + 1;
    if (mine != 0xffffffffu) atomicMin(&s_first, mine);
    __syncthreads();
    const unsigned D = s_first == 0xffffffffu ? TK_NB - 1 : s_first;
    if (t == 0) {
        cut[3] = total;                      // exact number of candidates (value > thr)
        sc->thr = tk_threshold(sc, quality);
        const unsigned mk = sc->max_eig_key;
        const unsigned bb = (mk & 0x80000000u) ? (mk & 0x7fffffffu) : ~mk;
        sc->max_eig = mk ? __uint_as_float(bb) : 0.f;
    }
    if (D == (unsigned)(2 * t)) { cut[0] = D; cut[1] = incl_a; cut[2] = 0; }
    else if (D == (unsigned)(2 * t + 1)) { cut[0] = D; cut[1] = s_first == 0xffffffffu ? total : incl_b; cut[2] = 0; }
}

__global__ __launch_bounds__(1024) void tk_compact_kernel(const unsigned long long *__restrict__ keys, unsigned cap, const km_scalars *sc,
                                                          double quality, unsigned *cut, unsigned long long *__restrict__ out)
{
    const unsigned cap_s = cap / KM_NSHARD;
    const unsigned n = min(sc->shard_cnt[blockIdx.y], cap_s);
    keys += (size_t)blockIdx.y * cap_s;
    const float thr = tk_threshold(sc, quality);
    __shared__ unsigned s_wave[16];
    __shared__ unsigned s_base;
    const unsigned top = (sc->max_eig_key & 0x7fffffffu) >> TK_SHIFT;
    const unsigned D = cut[0];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // pass 1: the workgroup's total over all its trips -> ONE reservation in the output (an atomic with return on the
    // shared cursor costs microseconds at device scope; one per trip made this kernel latency-bound)
    unsigned mine = 0;
    for (unsigned b = blockIdx.x * 1024; b < n; b += gridDim.x * 1024) {
        const unsigned i = b + threadIdx.x;
        if (i < n) { const unsigned long long k = keys[i]; mine += (tk_above(k, thr) && tk_bin(k, top) <= D) ? 1u : 0u; }
    }
    for (int o = 32; o > 0; o >>= 1) mine += (unsigned)__shfl_xor((int)mine, o);
    if (lane == 0) s_wave[wv] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0;
        for (int w = 0; w < 16; w++) tot += s_wave[w];
        s_base = tot ? atomicAdd(&cut[2], tot) : 0u;
    }
    __syncthreads();
    unsigned base = s_base;
    // pass 2: write (the keys come from L2 this time)
    for (unsigned b = blockIdx.x * 1024; b < n; b += gridDim.x * 1024) {
        const unsigned i = b + threadIdx.x;
        unsigned long long k = 0;
        bool keep = false;
        if (i < n) { k = keys[i]; keep = tk_above(k, thr) && tk_bin(k, top) <= D; }
        const unsigned long long bal = __ballot(keep);
        __syncthreads();                                  // s_wave of the previous trip fully consumed
        if (lane == 0) s_wave[wv] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned off = 0, tot = 0;
        for (int w = 0; w < 16; w++) { const unsigned cn = s_wave[w]; if (w < wv) off += cn; tot += cn; }
        if (keep) out[base + off + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = k;
        base += tot;
    }
}

// the histogram pass alone (k_select2.hip continues on the device)
int ks_topk_hist(km_ctx *c, const unsigned long long *d_keys, size_t cap_keys, km_scalars *d_sc, double quality)
{
    tk_hist_kernel<<<dim3(16, KM_NSHARD), 1024, 0, c->stream>>>(d_keys, (unsigned)cap_keys, d_sc, quality, d_sc->hist);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// keeps (at least) the k_target strongest keys ABOVE the exact threshold (k_target = 0: all of them).  The number of keys
// in d_keys is read on the device (sc->shard_cnt, clamped to the shard capacity).  The histogram and the cut live in the
// scalar block, which the caller zeroed at the start of the call (`rezero` for a repeated pass).  One host
// synchronisation brings the whole block to the host: *n_kept, *n_total = exact candidate count, *hs.
int ks_topk_prefilter(km_ctx *c, const unsigned long long *d_keys, size_t cap_keys, size_t k_target, km_scalars *d_sc, double quality,
                      unsigned long long **d_kept, size_t *n_kept, size_t *n_total, km_scalars *hs, bool rezero)
{
    unsigned *hist = d_sc->hist, *cut = d_sc->cut;
    if (rezero) {
        KM_HIP(c, hipMemsetAsync(cut, 0, sizeof d_sc->cut, c->stream));
        KM_HIP(c, hipMemsetAsync(hist, 0, sizeof d_sc->hist, c->stream));
    }
    tk_hist_kernel<<<dim3(16, KM_NSHARD), 1024, 0, c->stream>>>(d_keys, (unsigned)cap_keys, d_sc, quality, hist);
    KM_LAUNCH_CHECK(c);
    tk_cut_kernel<<<1, 1024, 0, c->stream>>>(hist, k_target ? (unsigned)k_target : 0xffffffffu, cut, d_sc, quality);
    KM_LAUNCH_CHECK(c);
    km_scalars *land = (km_scalars *)km_pinned_rb(c, sizeof *hs);   // pinned: the copy is asynchronous, the deferred job queues right behind it
    if (!land) return KM_E_NOMEM;
    KM_HIP(c, hipMemcpyAsync(land, d_sc, sizeof *hs, hipMemcpyDeviceToHost, c->stream));
    { const int rcw = km_wait_readback(c); if (rcw) return rcw; }
    *hs = *land;
    const size_t kept = hs->cut[1];
    *n_total = hs->cut[3];
    *n_kept = kept;
    *d_kept = nullptr;
    // total keys emitted; an overflowing shard is reported as n_cand > cap_keys so that the caller regrows and repeats
    size_t emitted = 0, worst = 0;
    for (int i = 0; i < KM_NSHARD; i++) { emitted += hs->shard_cnt[i]; if (hs->shard_cnt[i] > worst) worst = hs->shard_cnt[i]; }
    hs->n_cand = (unsigned)emitted;
    if (worst > cap_keys / KM_NSHARD) { hs->n_cand = (unsigned)(worst * KM_NSHARD > cap_keys ? worst * KM_NSHARD : cap_keys + 1); return KM_OK; }
    unsigned long long *out = (unsigned long long *)km_ws(c, WS_MISC3, (kept + 16) * sizeof(unsigned long long));
    if (!out) return KM_E_NOMEM;
    if (kept > 0) {
        tk_compact_kernel<<<dim3(64, KM_NSHARD), 1024, 0, c->stream>>>(d_keys, (unsigned)cap_keys, d_sc, quality, cut, out);
        KM_LAUNCH_CHECK(c);
    }
    *d_kept = out;
    return KM_OK;
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
              float *d_xy, int cap, km_scalars *d_sc, int *n_found, bool fresh_scalars)
{
    if (n_found) *n_found = -1;  // -1: not read back
    if (n > 0xfffffff0ull) return km_fail(c, KM_E_UNSUPPORTED, "too many candidates");
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
    const double md2 = min_distance * min_distance;
    const unsigned N = (unsigned)n;
    // workspace: [cells+1] counts | [cells] fill cursors  (zeroed by ONE memset) | [cells+1] offsets ; items / state / pos per candidate
    unsigned *grid = (unsigned *)km_ws(c, WS_GRID, (3 * cells + 4) * sizeof(unsigned));
    unsigned *per = (unsigned *)km_ws(c, WS_MISC2, (size_t)N * 3 * sizeof(unsigned));
    if (!grid || !per) return KM_E_NOMEM;
    unsigned *cell_cnt = grid, *cell_fill = grid + cells + 1, *cell_off = grid + 2 * cells + 2;
    unsigned *items = per, *state = per + N, *pos = per + 2 * (size_t)N;
    KM_HIP(c, hipMemsetAsync(cell_cnt, 0, (2 * cells + 1) * sizeof(unsigned), c->stream));   // counts + fill cursors
    const size_t first = c->opt_select_first > 0 ? (size_t)c->opt_select_first : (size_t)max_corners * 3;
    unsigned k0 = 0, k1 = (max_corners > 0 || c->opt_select_first > 0) ? (unsigned)(first < n ? first : n) : N;
    int rounds = 0;
    bool und_fresh = fresh_scalars;   // d_sc->und[] was zeroed with the scalar block at the start of the call
    for (;;) {
        // (re)build the cell lists for the prefix [0, k1): counts only need the new part [k0, k1)
        sel_count_kernel<<<(k1 - k0 + 255) / 256, 256, 0, c->stream>>>(d_sorted, k0, k1, W, cell, gw, cell_cnt, state);
        KM_LAUNCH_CHECK(c);
        { const int rs = km_exclusive_scan(c, cell_cnt, cell_off, cells + 1, KM_SCAN_PLAIN, WS_SORT_TMP); if (rs) return rs; }
        if (k0 > 0) KM_HIP(c, hipMemsetAsync(cell_fill, 0, cells * sizeof(unsigned), c->stream));
        sel_fill_kernel<<<(k1 + 255) / 256, 256, 0, c->stream>>>(d_sorted, k1, W, cell, gw, cell_off, cell_fill, items);
        KM_LAUNCH_CHECK(c);
        int got = 0;
        for (;;) {
            if (!und_fresh) KM_HIP(c, hipMemsetAsync(d_sc->und, 0, sizeof d_sc->und, c->stream));
            und_fresh = false;
            for (int g = 0; g < 4; g++) {  // 4 launches (16 sweeps), each with its own undecided counter
                sel_sweep_kernel<<<(k1 + 255) / 256, 256, 0, c->stream>>>(d_sorted, k1, W, cell, gw, gh, md2, cell_off, cell_cnt, items, state,
                                                                          &d_sc->und[g]);
                KM_LAUNCH_CHECK(c);
                rounds++;
            }
            // optimistic tail: rank the accepted corners right away and fetch (undecided, corner count) together
            { const int rs = km_exclusive_scan(c, state, pos, (size_t)k1, KM_SCAN_IS_ONE, WS_SORT_TMP); if (rs) return rs; }
            sel_emit_kernel<<<(k1 + 255) / 256, 256, 0, c->stream>>>(d_sorted, state, pos, k1, W, max_corners, cap, d_xy, d_sc, rounds);
            KM_LAUNCH_CHECK(c);
            unsigned *back = (unsigned *)km_pinned_rb(c, sizeof(km_scalars));   // pinned landing zone (see km_pinned_rb)
            if (!back) return KM_E_NOMEM;
            KM_HIP(c, hipMemcpyAsync(back, d_sc->und, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            { const int rcw = km_wait_readback(c); if (rcw) return rcw; }
            got = (int)back[7];
            if (back[3] == 0) break;   // every prefix member decided: the emitted list is final for this prefix
            if (rounds > 100000) return km_fail(c, KM_E_INTERNAL, "corner selection did not converge");
        }
        if (n_found) *n_found = got;
        if (k1 == N) break;
        if (max_corners > 0 && got >= max_corners) break;
        k0 = k1;
        k1 = (unsigned)((size_t)k1 * 4 < n ? (size_t)k1 * 4 : n);
        c->stats.path_flags |= KM_PATH_PREFIX_GROWN;
    }
    return KM_OK;
}
