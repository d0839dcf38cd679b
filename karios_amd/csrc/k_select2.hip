// K5, speculative form: candidate ranking + greedy minimum-distance selection of cv2.goodFeaturesToTrack (reference call site
// klt.py:120) WITHOUT a host synchronisation and without library sorts.
//
// k_select.hip sizes its launches and buffers from counts it reads back to the host (twice per tile) and ranks with rocPRIM.
// Here every buffer has a fixed capacity, every kernel reads its element count from device memory, and anything that does not
// fit - a key-buffer shard or the fused kernel's stage overflowed, more keys kept than KF_CAP, a value bin too large for one
// workgroup, a grid cell with more than KF_CELL candidates, sweeps that did not converge, too few corners in the top slice -
// only raises a bit in sc->flags.  The flags travel with the tile's result; a flagged tile is repeated through the exact
// synchronising path.  An unflagged result IS the sequential algorithm's (same argument as k_select.hip: decisions on a
// rank prefix are final and the sweeps' fixed point is unique).
//
// Ranking without a sort: OpenCV's order (value descending, address descending) is the order of the u64 keys
// (value bits << 32 | raster index).  The selection only ever asks "does j outrank i?" - a key comparison - so the kept keys are
// merely grouped by the 2048 value bins of the top-K pre-filter (scan of the bin populations in the cut kernel, scatter).  A
// total order is needed for the ACCEPTED corners alone (OpenCV's output order and the cut at maxCorners): they are grouped by
// bin again, and inside a bin every accepted corner counts the accepted ones with a larger key - a few thousand comparisons
// per corner instead of sorting 160 000 candidates (a per-bin bitonic sort in LDS, the first version, took 0.2 ms: the
// quantised eigenvalues of a Laplacian image put > 10 000 equal-valued keys into single bins).
#include "common.hpp"

#define KF_NB KM_TK_NB
#define KF_SHIFT 14
#define KF_CELL 64            // candidates a grid cell can hold (a 10 x 10 cell of a textured image holds up to ~25 local maxima)
#define KF_SWEEPS 4            // sweeps per launch of the cell-walking form (more neighbours than KF_NBR)
#define KF_POLLS 32           // state polls per launch of the register form: a poll is one round of <= KF_NBR parallel loads
#define KF_SLICE 4u           // the ranked top slice holds (at least) KF_SLICE * maxCorners keys (the exact path starts from 8x and can grow)
// Workgroup size of every kernel of the chain.  256, not 1024 (round 6): the chain is latency-bound and is meant to run BESIDE the dense
// kernels of another submission (second context), whose long-lived waves hold every VGPR of a SIMD.  A 1024-thread workgroup needs a
// whole compute unit drained at once and starved there (f_hist_cut 85 -> 551 us, f_acc_count_scan 8 -> 661 us beside a dense kernel of
// another queue; the 256-thread f_sweep 6 -> 18 us: tools/investigations/overlap_stats.py); a 256-thread workgroup takes the slots ONE retiring dense
// workgroup leaves.
#define KF_T 256
#define KF_NW (KF_T / 64)
#define KF_BPT (KF_NB / KF_T)  // value bins per thread in the one-workgroup scans

namespace {

enum { S_UNDECIDED = 0, S_ACCEPT = 1, S_REJECT = 2 };

// Every kernel of the chain serves a BATCH of independent units (blockIdx.z = unit; km_klt_units_frame_submit) - the chain is bound by
// the latency of dependent loads and leaves the GPU idle (VALU 4 % busy), so U units' chains run side by side in the time of one.  What
// differs between units travels by value in the kernel arguments; the single-unit entry points pass a table of one.
struct kf_unit {
    const unsigned long long *keys;       // candidate keys the fused eigenvalue pass emitted (KM_NSHARD regions)
    km_scalars *sc;
    const unsigned *max_partial;          // its per-wave maxima (nullptr: sc->max_eig_key is final)
    unsigned long long *kept, *acc_keys;
    uint2 *cell_rec;
    unsigned *cell_items, *state, *acc_cnt, *acc_cur, *acc_off, *chunk_off;
    float *out_xy;
    size_t n_zero16;
    unsigned n_partial;
    int W, gw, gh;
};
struct kf_units_args {
    kf_unit u[KM_UNITS_MAX];
};

__device__ __forceinline__ unsigned kf_bin(unsigned long long key, unsigned top)
{
    const unsigned b = (unsigned)(key >> (32 + KF_SHIFT));
    const unsigned d = top > b ? top - b : 0u;
    return d < KF_NB - 1 ? d : KF_NB - 1;
}
__device__ __forceinline__ float kf_threshold(const km_scalars *sc, double quality)
{
    const unsigned mk = sc->max_eig_key;
    const unsigned b = (mk & 0x80000000u) ? (mk & 0x7fffffffu) : ~mk;
    const float maxv = mk ? __uint_as_float(b) : 0.f;
    return (float)__dmul_rn((double)maxv, quality);
}
// candidates the selection works on: none when the kept list overflowed (its contents are then undefined)
__device__ __forceinline__ unsigned kf_count(const km_scalars *sc, unsigned kept_cap) { return sc->cut[1] > kept_cap ? 0u : sc->cut[1]; }
__device__ __forceinline__ bool kf_above(unsigned long long key, float thr) { return __uint_as_float((unsigned)(key >> 32)) > thr; }
__device__ __forceinline__ void kf_xy(unsigned long long key, int W, int &x, int &y)
{
    const unsigned idx = (unsigned)(key & 0xffffffffull);
    y = (int)(idx / (unsigned)W);
    x = (int)(idx - (unsigned)y * (unsigned)W);
}

// "Last workgroup finishes the job": every workgroup of a launch takes a ticket when its contribution has been performed; the one
// that draws the last ticket runs the one-workgroup step that used to be the next launch (a launch of a 3-us kernel costs a
// pipeline drain + dispatch on a chain where nothing overlaps).  The contributions are device-scope ATOMICS (performed at the
// memory side, beyond the per-XCD L2s) and the last workgroup reads them with device-scope atomic loads, so the hand-over needs
// no cache maintenance: only "my atomics are done" (s_waitcnt) before the ticket.  (A __threadfence() here writes back every
// dirty line of the XCD's L2 - the pyramid levels the second stream is producing at that moment included: + 90 us.)
__device__ __forceinline__ bool kf_last_workgroup(unsigned *ticket, unsigned n_workgroups)
{
    __shared__ unsigned s_ticket;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    __syncthreads();
    return s_ticket == n_workgroups - 1;
}

// Launch 1 of the ranking.  Every workgroup: (side jobs) zero its share of the selection's cell records and of the accepted-corner
// counters; reduce the per-wave maxima of the fused eigenvalue pass (a few thousand words from L2 - cheaper than a launch in
// between; workgroup 0 publishes the result); histogram of its keys by value bin.  The LAST workgroup then cuts:
// hist[KF_NB] -> cut[0] = D (last kept bin), cut[1] = kept keys, cut[3] = exact candidate count, bin_off[b] = first slot of
// bin b in the kept list; overflow flags of the emission stage.
__global__ __launch_bounds__(KF_T) void f_hist_cut_kernel(kf_units_args A, unsigned cap, double quality, unsigned k_target, unsigned kept_cap,
                                                          unsigned test_flags)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ keys = U.keys;
    km_scalars *sc = U.sc;
    const unsigned *__restrict__ max_partial = U.max_partial;
    const unsigned n_partial = U.n_partial;
    uint4 *__restrict__ zero16 = (uint4 *)U.cell_rec;
    const size_t n_zero16 = U.n_zero16;
    unsigned *__restrict__ acc_zero = U.acc_cnt;
    __shared__ unsigned h[KF_NB];
    __shared__ unsigned s_wave[KF_NW];
    __shared__ unsigned s_first, s_maxkey;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x, n_wg = gridDim.x * gridDim.y;
    for (size_t i = (size_t)wg * KF_T + t; i < n_zero16; i += (size_t)n_wg * KF_T) zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    for (unsigned i = wg * KF_T + t; i < 2 * KF_NB; i += n_wg * KF_T) acc_zero[i] = 0u;      // acc_cnt + acc_cur of the ranking behind the sweeps
    for (int i = t; i < KF_NB; i += KF_T) h[i] = 0;
    unsigned maxkey;
    if (max_partial) {
        unsigned m = 0;
        for (unsigned i = t; i < n_partial; i += KF_T) m = max(m, max_partial[i]);
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (lane == 0) s_wave[wv] = m;
        __syncthreads();
        if (t == 0) {
            unsigned mm = 0;
            for (int i = 0; i < KF_NW; i++) mm = max(mm, s_wave[i]);
            s_maxkey = mm;
            if (wg == 0) sc->max_eig_key = mm;         // for the launches that follow (every workgroup here has its own copy)
        }
        __syncthreads();
        maxkey = s_maxkey;
    } else {
        maxkey = sc->max_eig_key;
        __syncthreads();
    }
    const unsigned bm = (maxkey & 0x80000000u) ? (maxkey & 0x7fffffffu) : ~maxkey;
    const float maxv = maxkey ? __uint_as_float(bm) : 0.f;
    const float thr = (float)__dmul_rn((double)maxv, quality);
    const unsigned top = (maxkey & 0x7fffffffu) >> KF_SHIFT;   // max eig > 0: ordered key = bits | 0x80000000
    // the key buffer is KM_NSHARD regions of cap / KM_NSHARD slots; blockIdx.y selects the region
    {
        const unsigned cap_s = cap / KM_NSHARD;
        const unsigned n = min(sc->shard_cnt[blockIdx.y], cap_s);
        const unsigned long long *kk = keys + (size_t)blockIdx.y * cap_s;
        unsigned tail = 0;                              // the clamp bin collects the bulk of the (weak) candidates: counted per wave
        for (unsigned b = blockIdx.x * KF_T; b < n; b += gridDim.x * KF_T) {
            const unsigned i = b + t;
            unsigned d = 0xffffffffu;
            if (i < n) { const unsigned long long k = kk[i]; if (kf_above(k, thr)) d = kf_bin(k, top); }
            tail += (unsigned)__popcll(__ballot(d == KF_NB - 1));
            if (d < KF_NB - 1) atomicAdd(&h[d], 1u);
        }
        if (lane == 0 && tail) atomicAdd(&h[KF_NB - 1], tail);
    }
    __syncthreads();
    for (int i = t; i < KF_NB; i += KF_T)
        if (h[i]) atomicAdd(&sc->hist[i], h[i]);
    if (!kf_last_workgroup(&sc->tickets[0], n_wg)) return;
    // ---- the cut (one workgroup, every histogram complete): thread t owns the KF_BPT consecutive bins from KF_BPT * t
    unsigned hv[KF_BPT], v = 0;
#pragma unroll
    for (int j = 0; j < KF_BPT; j++) { hv[j] = __hip_atomic_load(&sc->hist[KF_BPT * t + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v += hv[j]; }
    const unsigned mine_sum = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    if (lane == 63) s_wave[wv] = v;                     // (the maximum's use of s_wave ended two barriers ago)
    if (t == 0) s_first = 0xffffffffu;
    __syncthreads();
    unsigned base = 0, total = 0;
    for (int w = 0; w < KF_NW; w++) { if (w < wv) base += s_wave[w]; total += s_wave[w]; }
    unsigned run = base + v - mine_sum;                 // keys in the bins before this thread's
    unsigned mine = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < KF_BPT; j++) {
        sc->bin_off[KF_BPT * t + j] = run;
        run += hv[j];
        if (mine == 0xffffffffu && run >= k_target) mine = KF_BPT * t + j;     // first bin whose inclusive count reaches the target
    }
    if (mine != 0xffffffffu) atomicMin(&s_first, mine);
    __syncthreads();
    const unsigned D = s_first == 0xffffffffu ? KF_NB - 1 : s_first;
    if (D / KF_BPT == (unsigned)t) {
        unsigned kept = base + v - mine_sum;
        for (unsigned j = 0; j <= D % KF_BPT; j++) kept += hv[j];              // inclusive count at bin D (= total when no bin reaches the target)
        sc->cut[0] = D; sc->cut[1] = kept; sc->cut[2] = 0;
        if (kept > kept_cap) atomicOr(&sc->flags, KM_FLAG_KEPT_OVERFLOW);
    }
    if (t == 0) {
        sc->cut[3] = total;
        sc->thr = thr;
        sc->max_eig = maxv;
        unsigned fl = (sc->pad0 ? KM_FLAG_STAGE_OVERFLOW : 0u) | test_flags;
        for (int s2 = 0; s2 < KM_NSHARD; s2++) if (sc->shard_cnt[s2] > cap / KM_NSHARD) fl |= KM_FLAG_SHARD_OVERFLOW;
        if (fl) atomicOr(&sc->flags, fl);
    }
}

// Launch 2: kept keys -> their bin's range of `out` (order inside a bin is irrelevant: the selection compares keys) AND into their
// cell of the minimum-distance grid (fixed capacity), state undecided.  A workgroup reads its ~14 000 keys ONCE: the ~4 % above the
// cut are counted per bin and stashed in LDS; it then reserves one range per non-empty bin (a device-scope atomic with return costs
// microseconds and same-address ones serialise: one per key made this the slowest kernel of the stage) and places the stashed keys
// one per thread - the cell insertion is a device-scope atomic WITH return, and inside the key loop its latency was paid per trip
// (+ 90 us).  A cell's record holds its population AND its first candidate: nearly every occupied cell holds exactly one (160 000
// candidates on 1.2 million cells), and the sweeps then reach a neighbour's key with two dependent loads instead of three.
// (The cell records were zeroed by launch 1.)
#define KF_STASH 1024         // (a workgroup reads ~3 500 keys, ~4 % of them above the cut)
__global__ __launch_bounds__(KF_T) void f_scatter_cells_kernel(kf_units_args A, unsigned cap, double quality, unsigned kept_cap, int cell,
                                                               unsigned stash_cap /* <= KF_STASH (test knob: small values force the second read) */)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ keys = U.keys;
    km_scalars *sc = U.sc;
    unsigned long long *__restrict__ out = U.kept;
    const int W = U.W, gw = U.gw;
    uint2 *__restrict__ cell_rec = U.cell_rec;
    unsigned *__restrict__ cell_items = U.cell_items, *__restrict__ state = U.state;
    __shared__ unsigned s_cnt[KF_NB], s_base[KF_NB];
    __shared__ unsigned long long s_stash[KF_STASH];
    __shared__ unsigned s_n;
    const unsigned cap_s = cap / KM_NSHARD;
    const unsigned n = min(sc->shard_cnt[blockIdx.y], cap_s);
    keys += (size_t)blockIdx.y * cap_s;
    const float thr = kf_threshold(sc, quality);
    const unsigned top = (sc->max_eig_key & 0x7fffffffu) >> KF_SHIFT;
    const unsigned D = sc->cut[0];
    if (sc->cut[1] > kept_cap) return;                   // flagged: the exact path takes over
    for (int i = threadIdx.x; i < KF_NB; i += KF_T) s_cnt[i] = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (unsigned i = blockIdx.x * KF_T + threadIdx.x; i < n; i += gridDim.x * KF_T) {
        const unsigned long long k = keys[i];
        if (!kf_above(k, thr)) continue;
        const unsigned b = kf_bin(k, top);
        if (b > D) continue;
        atomicAdd(&s_cnt[b], 1u);
        const unsigned e = atomicAdd(&s_n, 1u);
        if (e < stash_cap) s_stash[e] = k;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KF_NB; i += KF_T) {
        const unsigned cnt = s_cnt[i];
        s_base[i] = cnt ? sc->bin_off[i] + atomicAdd(&sc->bin_cur[i], cnt) : 0u;
        s_cnt[i] = 0;
    }
    __syncthreads();
    auto place = [&](unsigned long long k) {
        const unsigned pos = s_base[kf_bin(k, top)] + atomicAdd(&s_cnt[kf_bin(k, top)], 1u);
        out[pos] = k;
        state[pos] = S_UNDECIDED;
        int x, y;
        kf_xy(k, W, x, y);
        const unsigned g = (unsigned)(y / cell) * (unsigned)gw + (unsigned)(x / cell);
        const unsigned slot = atomicAdd(&cell_rec[g].x, 1u);
        if (slot == 0) cell_rec[g].y = pos;
        else if (slot < KF_CELL) cell_items[(size_t)g * KF_CELL + slot] = pos;
        else atomicOr(&sc->flags, KM_FLAG_CELL_OVERFLOW);
    };
    const unsigned kept_here = s_n;
    if (kept_here <= stash_cap) {
        for (unsigned e = threadIdx.x; e < kept_here; e += KF_T) place(s_stash[e]);
    } else {
        // more keys above the cut than the stash holds (a tile whose strong corners crowd into one part of the key buffer): second read
        for (unsigned i = blockIdx.x * KF_T + threadIdx.x; i < n; i += gridDim.x * KF_T) {
            const unsigned long long k = keys[i];
            if (kf_above(k, thr) && kf_bin(k, top) <= D) place(k);
        }
    }
}

// One launch = up to KF_SWEEPS relaxation sweeps of every undecided candidate.  The launch is bound by the LATENCY of dependent
// loads (2.4 waves per SIMD, every load a cache miss somewhere in a 10 MB grid), so a thread first fetches the nine cell records
// of its neighbourhood at once, then the keys of their first candidates at once, keeps the few higher-ranked neighbours within the
// minimum distance (0.2 on average) in registers, and the sweeps only poll those neighbours' states.
#define KF_NBR 6
__global__ __launch_bounds__(256) void f_sweep_kernel(kf_units_args A, int cell, double md2, int und_slot, unsigned kept_cap)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ keys = U.kept;
    km_scalars *sc = U.sc;
    const int W = U.W, gw = U.gw, gh = U.gh;
    const uint2 *__restrict__ cell_rec = U.cell_rec;
    const unsigned *__restrict__ cell_items = U.cell_items;
    unsigned *state = U.state, *n_undecided = &sc->und[und_slot];
    const unsigned n = kf_count(sc, kept_cap);
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    bool undecided = false;
    if (i < n && __hip_atomic_load(&state[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == S_UNDECIDED) {
        int x, y;
        const unsigned long long ki = keys[i];
        kf_xy(ki, W, x, y);
        const int xc = x / cell, yc = y / cell;
        unsigned nb[KF_NBR];
        int nnb = 0;
        bool overflow = false;
        auto consider = [&](unsigned j, unsigned long long kj) {
            if (kj <= ki) return;                              // only higher-ranked candidates matter (rank = key order)
            int xj, yj;
            kf_xy(kj, W, xj, yj);
            const float dx = (float)x - (float)xj, dy = (float)y - (float)yj;
            if (!((double)(dx * dx + dy * dy) < md2)) return;
#pragma unroll
            for (int t = 0; t < KF_NBR; t++) if (nnb == t) nb[t] = j;
            overflow |= nnb >= KF_NBR;
            nnb += 1;
        };
        uint2 rec[9];
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) {
            const int yy = yc - 1 + c9 / 3, xx = xc - 1 + c9 % 3;
            const bool in = yy >= 0 && yy < gh && xx >= 0 && xx < gw;
            rec[c9] = in ? cell_rec[(unsigned)yy * (unsigned)gw + (unsigned)xx] : make_uint2(0u, 0u);
        }
        unsigned long long k0[9];
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) k0[c9] = rec[c9].x ? keys[rec[c9].y] : 0ull;
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) if (rec[c9].x) consider(rec[c9].y, k0[c9]);
#pragma unroll
        for (int c9 = 0; c9 < 9; c9++) {
            const unsigned cnt = min(rec[c9].x, (unsigned)KF_CELL);
            if (cnt > 1) {                                     // rare: the cell's further candidates
                const int yy = yc - 1 + c9 / 3, xx = xc - 1 + c9 % 3;
                const size_t base = ((size_t)yy * (size_t)gw + (size_t)xx) * KF_CELL;
                for (unsigned k = 1; k < cnt; k++) {
                    const unsigned j = cell_items[base + k];
                    consider(j, keys[j]);
                }
            }
        }
        undecided = true;
        if (!overflow) {
            for (int sweep = 0; sweep < KF_POLLS && undecided; sweep++) {
                unsigned sj[KF_NBR];
#pragma unroll
                for (int t = 0; t < KF_NBR; t++)
                    sj[t] = t < nnb ? __hip_atomic_load(&state[nb[t]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)S_REJECT;
                bool blocked = false, rejected = false;
#pragma unroll
                for (int t = 0; t < KF_NBR; t++) { rejected |= sj[t] == S_ACCEPT; blocked |= sj[t] == S_UNDECIDED; }
                if (rejected) { __hip_atomic_store(&state[i], (unsigned)S_REJECT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
                else if (!blocked) { __hip_atomic_store(&state[i], (unsigned)S_ACCEPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
            }
        } else {
            // more higher-ranked neighbours than registers: walk the cells every sweep
            for (int sweep = 0; sweep < KF_SWEEPS && undecided; sweep++) {
                bool blocked = false, rejected = false;
                for (int c9 = 0; c9 < 9 && !rejected; c9++) {
                    const int yy = yc - 1 + c9 / 3, xx = xc - 1 + c9 % 3;
                    if (yy < 0 || yy >= gh || xx < 0 || xx >= gw) continue;
                    const size_t g = (size_t)yy * (size_t)gw + (size_t)xx;
                    const unsigned cnt = min(cell_rec[g].x, (unsigned)KF_CELL);
                    for (unsigned k = 0; k < cnt; k++) {
                        const unsigned j = k == 0 ? cell_rec[g].y : cell_items[g * KF_CELL + k];
                        const unsigned long long kj = keys[j];
                        if (kj <= ki) continue;
                        int xj, yj;
                        kf_xy(kj, W, xj, yj);
                        const float dx = (float)x - (float)xj, dy = (float)y - (float)yj;
                        if (!((double)(dx * dx + dy * dy) < md2)) continue;
                        const unsigned s2 = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (s2 == S_ACCEPT) { rejected = true; break; }
                        if (s2 == S_UNDECIDED) blocked = true;
                    }
                }
                if (rejected) { __hip_atomic_store(&state[i], (unsigned)S_REJECT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
                else if (!blocked) { __hip_atomic_store(&state[i], (unsigned)S_ACCEPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); undecided = false; }
            }
        }
    }
    const unsigned long long bal = __ballot(undecided);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_undecided, (unsigned)__popcll(bal));
}

// accepted corners per value bin.  Workgroup-aggregated like the scatter: thousands of accepted corners share a handful of bins
// (equal eigenvalues), and one device-scope atomic per corner on the same address took 300 us.  The LAST workgroup then scans:
// acc_off = exclusive scan of acc_cnt, chunk_off = exclusive scan of the bins' 64-corner chunks; corner count, flags.
__global__ __launch_bounds__(KF_T) void f_acc_count_scan_kernel(kf_units_args A, unsigned kept_cap, int max_corners, unsigned und_slot)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ keys = U.kept;
    const unsigned *__restrict__ state = U.state;
    km_scalars *sc = U.sc;
    unsigned *__restrict__ acc_cnt = U.acc_cnt, *__restrict__ acc_off = U.acc_off, *__restrict__ chunk_off = U.chunk_off;
    __shared__ unsigned s_cnt[KF_NB];
    __shared__ unsigned s_wave[KF_NW], s_wave2[KF_NW];
    const unsigned n = kf_count(sc, kept_cap);
    const unsigned top = (sc->max_eig_key & 0x7fffffffu) >> KF_SHIFT;
    for (int i = threadIdx.x; i < KF_NB; i += KF_T) s_cnt[i] = 0;
    __syncthreads();
    for (unsigned i = blockIdx.x * KF_T + threadIdx.x; i < n; i += gridDim.x * KF_T)
        if (state[i] == S_ACCEPT) atomicAdd(&s_cnt[kf_bin(keys[i], top)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < KF_NB; i += KF_T)
        if (s_cnt[i]) atomicAdd(&acc_cnt[i], s_cnt[i]);
    if (!kf_last_workgroup(&sc->tickets[1], gridDim.x)) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned cv[KF_BPT], v = 0, w = 0;                 // thread t owns the KF_BPT consecutive bins from KF_BPT * t
#pragma unroll
    for (int j = 0; j < KF_BPT; j++) {
        cv[j] = __hip_atomic_load(&acc_cnt[KF_BPT * t + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v += cv[j];
        w += (cv[j] + 63) / 64;
    }
    const unsigned mine_v = v, mine_w = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(v, o), u2 = __shfl_up(w, o);
        if (lane >= o) { v += u; w += u2; }
    }
    if (lane == 63) { s_wave[wv] = v; s_wave2[wv] = w; }
    __syncthreads();
    unsigned base = 0, total = 0, base2 = 0, total2 = 0;
    for (int k = 0; k < KF_NW; k++) { if (k < wv) { base += s_wave[k]; base2 += s_wave2[k]; } total += s_wave[k]; total2 += s_wave2[k]; }
    unsigned run = base + v - mine_v, run2 = base2 + w - mine_w;
#pragma unroll
    for (int j = 0; j < KF_BPT; j++) {
        acc_off[KF_BPT * t + j] = run;
        chunk_off[KF_BPT * t + j] = run2;
        run += cv[j];
        run2 += (cv[j] + 63) / 64;
    }
    if (t == 0) {
        acc_off[KF_NB] = total;
        chunk_off[KF_NB] = total2;
        unsigned fl = 0;
        if (sc->und[und_slot]) fl |= KM_FLAG_NOT_CONVERGED;
        if (total < (unsigned)max_corners && sc->cut[1] < sc->cut[3]) fl |= KM_FLAG_SLICE_SHORT;   // the top slice ran dry: all candidates needed
        if (fl) atomicOr(&sc->flags, fl);
        sc->n_corners = (int)min(total, (unsigned)max_corners);
        sc->n_batches = KF_SWEEPS * 4;
    }
}

// accepted keys grouped by bin (any order inside a bin)
__global__ __launch_bounds__(KF_T) void f_acc_fill_kernel(kf_units_args A, unsigned kept_cap)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ keys = U.kept;
    const unsigned *__restrict__ state = U.state;
    const km_scalars *sc = U.sc;
    const unsigned *__restrict__ acc_off = U.acc_off;
    unsigned *__restrict__ acc_cur = U.acc_cur;
    unsigned long long *__restrict__ acc_keys = U.acc_keys;
    __shared__ unsigned s_cnt[KF_NB], s_base[KF_NB];
    const unsigned n = kf_count(sc, kept_cap);
    const unsigned top = (sc->max_eig_key & 0x7fffffffu) >> KF_SHIFT;
    for (int i = threadIdx.x; i < KF_NB; i += KF_T) s_cnt[i] = 0;
    __syncthreads();
    for (unsigned i = blockIdx.x * KF_T + threadIdx.x; i < n; i += gridDim.x * KF_T)
        if (state[i] == S_ACCEPT) atomicAdd(&s_cnt[kf_bin(keys[i], top)], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < KF_NB; i += KF_T) {
        const unsigned cnt = s_cnt[i];
        s_base[i] = cnt ? acc_off[i] + atomicAdd(&acc_cur[i], cnt) : 0u;
        s_cnt[i] = 0;
    }
    __syncthreads();
    for (unsigned i = blockIdx.x * KF_T + threadIdx.x; i < n; i += gridDim.x * KF_T)
        if (state[i] == S_ACCEPT) {
            const unsigned long long k = keys[i];
            const unsigned b = kf_bin(k, top);
            acc_keys[s_base[b] + atomicAdd(&s_cnt[b], 1u)] = k;
        }
}

// Position of an accepted corner in OpenCV's output order = accepted corners in stronger bins + accepted corners of its own bin
// with a larger key; the first maxCorners positions are the result.  One WORKGROUP per chunk of 64 corners of ONE bin: each of
// its KF_NW wavefronts compares the 64 corners (one per lane) with its share of the bin - keys at wave-uniform addresses, scalar
// loads, eight per instruction - and the partial counts meet in LDS.  (One wavefront walking a 10 000-corner bin alone - equal
// eigenvalues are common in a quantised Laplacian image - took 73 us.)
__global__ __launch_bounds__(KF_T) void f_acc_emit_kernel(kf_units_args A, int max_corners, int cap)
{
    const kf_unit &U = A.u[blockIdx.z];
    const unsigned long long *__restrict__ acc_keys = U.acc_keys;
    const unsigned *__restrict__ acc_off = U.acc_off, *__restrict__ chunk_off = U.chunk_off;
    const int W = U.W;
    float *__restrict__ out_xy = U.out_xy;
    __shared__ unsigned s_part[KF_NW][64];
    const unsigned chunk = blockIdx.x;
    if (chunk >= chunk_off[KF_NB]) return;
    unsigned lo_b = 0, hi_b = KF_NB;                           // bin of this chunk: last b with chunk_off[b] <= chunk
    while (hi_b - lo_b > 1) {
        const unsigned mid = (lo_b + hi_b) >> 1;
        if (chunk_off[mid] <= chunk) lo_b = mid; else hi_b = mid;
    }
    const unsigned b = lo_b;
    const unsigned lo = acc_off[b], hi = acc_off[b + 1];
    if (lo >= (unsigned)max_corners) return;                 // the whole bin lies behind the cut
    const unsigned lane = threadIdx.x & 63, wv = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned s = lo + (chunk - chunk_off[b]) * 64 + lane;
    const bool live = s < hi;
    const unsigned long long k = live ? acc_keys[s] : 0ull;
    const unsigned len = hi - lo, per = (len + KF_NW - 1) / KF_NW;
    const unsigned t0 = lo + min(wv * per, len), t1 = lo + min((wv + 1) * per, len);
    unsigned p = 0;
    for (unsigned t = t0; t < t1; t++) p += acc_keys[t] > k ? 1u : 0u;      // uniform index: scalar loads
    s_part[wv][lane] = p;
    __syncthreads();
    if (wv == 0) {
        unsigned pos = lo;
#pragma unroll
        for (int w = 0; w < KF_NW; w++) pos += s_part[w][lane];
        if (live && pos < (unsigned)max_corners && pos < (unsigned)cap) {
            int x, y;
            kf_xy(k, W, x, y);
            out_xy[2 * pos] = (float)x;
            out_xy[2 * pos + 1] = (float)y;
        }
    }
}

}  // namespace

size_t kf_kept_capacity(int max_corners) { return (size_t)max_corners * 2 * KF_SLICE; }

// Workspace of ONE unit's chain (byte sizes: kf_bytes): kept keys + accepted keys | cell records + further slots | states, counters
struct kf_sizes {
    size_t kept, grid, per;
    unsigned kept_cap;
    int cell;
};
static int kf_sizes_of(int max_corners, double min_distance, int H_max, int W_max, kf_sizes *z)
{
    if (!(max_corners > 0) || !(min_distance >= 1)) return KM_E_UNSUPPORTED;
    z->kept_cap = (unsigned)kf_kept_capacity(max_corners);
    z->cell = (int)lrint(min_distance);
    const size_t cells = (size_t)((W_max + z->cell - 1) / z->cell) * (size_t)((H_max + z->cell - 1) / z->cell);
    if (cells * KF_CELL > 0x7fffffffull) return KM_E_UNSUPPORTED;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    z->kept = up((2 * (size_t)z->kept_cap + 32) * sizeof(unsigned long long));
    z->grid = up((cells + 1) * (2 + KF_CELL) * sizeof(unsigned));
    z->per = up(((size_t)z->kept_cap + 4 * KF_NB + 16) * sizeof(unsigned));
    return KM_OK;
}
static void kf_carve(const kf_sizes &z, char *kept, char *grid, char *per, int H, int W, kf_unit *u)
{
    u->gw = (W + z.cell - 1) / z.cell; u->gh = (H + z.cell - 1) / z.cell; u->W = W;
    const size_t cells = (size_t)u->gw * u->gh;
    u->kept = (unsigned long long *)kept; u->acc_keys = u->kept + z.kept_cap + 16;
    u->cell_rec = (uint2 *)grid; u->cell_items = (unsigned *)(u->cell_rec + cells);
    u->n_zero16 = (cells + 1) / 2;   // (cell records of 8 bytes in a 16-byte aligned buffer; with an odd cell count the first two item slots behind them are zeroed too - they are written later)
    unsigned *p = (unsigned *)per;
    u->state = p; u->acc_cnt = p + z.kept_cap; u->acc_cur = u->acc_cnt + KF_NB; u->acc_off = u->acc_cur + KF_NB;   // acc_off, chunk_off: KF_NB + 1 entries
    u->chunk_off = u->acc_off + KF_NB + 4;
}

// the tables of a batch: unit k's slices of WS_MISC3 / WS_GRID / WS_MISC2 (sized for the largest unit)
static int kf_layout_units(km_ctx *c, int n, const int *H, const int *W, int max_corners, double min_distance, kf_units_args *A, kf_sizes *z)
{
    int Hm = 0, Wm = 0;
    for (int k = 0; k < n; k++) { Hm = H[k] > Hm ? H[k] : Hm; Wm = W[k] > Wm ? W[k] : Wm; }
    const int rc = kf_sizes_of(max_corners, min_distance, Hm, Wm, z);
    if (rc) return rc;
    char *kept = (char *)km_ws(c, WS_MISC3, z->kept * n), *grid = (char *)km_ws(c, WS_GRID, z->grid * n), *per = (char *)km_ws(c, WS_MISC2, z->per * n);
    if (!kept || !grid || !per) return KM_E_NOMEM;
    for (int k = 0; k < n; k++) kf_carve(*z, kept + z->kept * k, grid + z->grid * k, per + z->per * k, H[k], W[k], &A->u[k]);
    return KM_OK;
}

static int kf_rank_launch(km_ctx *c, const kf_units_args &A, int n, size_t cap_keys, int max_corners, double quality, const kf_sizes &z)
{
    static const int gx = [] { const char *e = km_dev_env("KARIOS_HIP_RANK_GRID"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? v : 16; }();   // tuning override
    // (a batch keeps ~256 workgroups per launch in flight: 16 x 16 per unit alone, fewer columns per unit as the units multiply)
    const int gxu = (n >= 8 ? (gx + 3) / 4 : n >= 3 ? (gx + 1) / 2 : gx) * (1024 / KF_T);
    f_hist_cut_kernel<<<dim3(gxu, KM_NSHARD, n), KF_T, 0, c->stream>>>(A, (unsigned)cap_keys, quality, (unsigned)max_corners * KF_SLICE, z.kept_cap,
                                                                     (unsigned)c->opt_spec_flag);
    KM_LAUNCH_CHECK(c);
    f_scatter_cells_kernel<<<dim3(gxu, KM_NSHARD, n), KF_T, 0, c->stream>>>(A, (unsigned)cap_keys, quality, z.kept_cap, z.cell,
                                                                          c->opt_stash_cap > 0 && c->opt_stash_cap < KF_STASH ? (unsigned)c->opt_stash_cap : (unsigned)KF_STASH);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

#define KF_LAUNCHES 4          // sweep launches (42 + 6 + 2 + 2 us: the last two find nothing left to do on a GPU of their own; with several
                               // contexts sharing the GPU the polling sweeps give up earlier and three launches left tiles unconverged - flagged, repeated)
static int kf_select_launch(km_ctx *c, const kf_units_args &A, int n, int max_corners, double min_distance, int cap, const kf_sizes &z)
{
    const double md2 = min_distance * min_distance;
    const unsigned g256 = (z.kept_cap + 255) / 256;
    for (int g = 0; g < KF_LAUNCHES; g++) {
        f_sweep_kernel<<<dim3(g256, 1, n), 256, 0, c->stream>>>(A, z.cell, md2, g, z.kept_cap);
        KM_LAUNCH_CHECK(c);
    }
    const int gb = (n >= 8 ? 8 : n >= 3 ? 16 : 32) * (1024 / KF_T);
    f_acc_count_scan_kernel<<<dim3(gb, 1, n), KF_T, 0, c->stream>>>(A, z.kept_cap, max_corners, (unsigned)(KF_LAUNCHES - 1));
    KM_LAUNCH_CHECK(c);
    f_acc_fill_kernel<<<dim3(gb, 1, n), KF_T, 0, c->stream>>>(A, z.kept_cap);
    KM_LAUNCH_CHECK(c);
    // chunks of 64 accepted corners: at most kept_cap / 64 + one partial chunk per bin
    f_acc_emit_kernel<<<dim3(z.kept_cap / 64 + KF_NB, 1, n), KF_T, 0, c->stream>>>(A, max_corners, cap);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// Ranking of the keys the fused kernel emitted: (maximum, histogram, cut) -> (scatter by value bin + grid cells).  Two launches;
// everything is enqueued, nothing is read back.  Requires max_corners > 0, min_distance >= 1 and a scalar block zeroed at the
// start of the call.
int kf_rank(km_ctx *c, const unsigned long long *d_keys, size_t cap_keys, int H, int W, int max_corners, double quality, double min_distance, km_scalars *sc)
{
    kf_units_args A;
    kf_sizes z;
    const int rc = kf_layout_units(c, 1, &H, &W, max_corners, min_distance, &A, &z);
    if (rc) return rc;
    A.u[0].keys = d_keys; A.u[0].sc = sc; A.u[0].max_partial = c->eig_partial; A.u[0].n_partial = c->eig_npartial; A.u[0].out_xy = nullptr;
    c->eig_partial = nullptr; c->eig_npartial = 0;
    return kf_rank_launch(c, A, 1, cap_keys, max_corners, quality, z);
}

// Greedy minimum-distance selection on the ranked keys of kf_rank: corners in d_xy, their count in sc->n_corners.
int kf_select(km_ctx *c, int H, int W, int max_corners, double min_distance, float *d_xy, int cap, km_scalars *sc)
{
    kf_units_args A;
    kf_sizes z;
    const int rc = kf_layout_units(c, 1, &H, &W, max_corners, min_distance, &A, &z);
    if (rc) return rc;
    A.u[0].keys = nullptr; A.u[0].sc = sc; A.u[0].max_partial = nullptr; A.u[0].n_partial = 0; A.u[0].out_xy = d_xy;
    return kf_select_launch(c, A, 1, max_corners, min_distance, cap, z);
}

// The whole chain for a batch of units (api_units.hip): nine launches for ALL of them.  KM_E_UNSUPPORTED: see kf_sizes_of.
int kf_rank_select_units(km_ctx *c, const km_units &U, int max_corners, double quality, double min_distance, int cap)
{
    kf_units_args A;
    kf_sizes z;
    const int rc = kf_layout_units(c, U.n, U.H, U.W, max_corners, min_distance, &A, &z);
    if (rc) return rc;
    for (int k = 0; k < U.n; k++) {
        A.u[k].keys = U.keys[k]; A.u[k].sc = U.sc[k]; A.u[k].max_partial = U.eig_partial[k]; A.u[k].n_partial = U.eig_npartial[k]; A.u[k].out_xy = U.p0[k];
    }
    int r = kf_rank_launch(c, A, U.n, U.capk, max_corners, quality, z);
    if (r) return r;
    return kf_select_launch(c, A, U.n, max_corners, min_distance, cap, z);
}
