// Device-wide ordering primitives of the EXACT fallback paths (no library behind them):
//   km_sort_u64        stable LSD radix sort of 64-bit keys (+ an optional 32-bit payload), ascending or descending
//   km_exclusive_scan  exclusive prefix sum of 32-bit counts, or of the "accepted" flags of the selection states
// The default (synchronisation-free) corner path never comes here (k_select2.hip ranks by bins, k_frame.hip places up to 32 768 rows
// in one workgroup); these serve `goodFeaturesToTrack` with maxCorners = 0 / the flagged-unit repeat (k_select.hip: every
// candidate ranked in `greaterThanPtr` order, reference klt.py:112-125 via cv.goodFeaturesToTrack) and the (x0, y0) ordering of
// frames of more than 32 768 rows (k_frame.hip; reference klt.py:187 `sort_values`-free raster order of `np.nonzero`).
//
// Sort: 8 passes of 8 bits.  A pass is three launches: (1) every workgroup histograms the digit of its 2048-key tile in LDS and
// leaves a row-major [digit][tile] count table + per-digit totals; (2) workgroup d turns row d of the table into global offsets
// (base of digit d = sum of the totals below it, then a running sum over the tiles); (3) every workgroup ranks its tile again -
// a wave owns a contiguous run of the tile and takes it 64 keys at a time: the lanes holding the same digit find each other with
// eight ballots, their order among themselves is the population count below the lane, the run's count so far sits in a per-wave
// LDS counter - and scatters key (+ payload) to offset[digit][tile] + keys of the digit in earlier waves + rank.  Equal digits keep
// their input order in every pass, which is what makes the least-significant-digit-first order correct.
#include "common.hpp"

namespace {

constexpr int RS_T = 256, RS_ITEMS = 8, RS_TILE = RS_T * RS_ITEMS, RS_WAVES = RS_T / 64;

template <bool DESC> __device__ __forceinline__ unsigned rs_digit(unsigned long long k, int shift)
{
    const unsigned d = (unsigned)(k >> shift) & 0xffu;
    return DESC ? 255u - d : d;
}

template <bool DESC>
__global__ __launch_bounds__(RS_T) void rs_count_kernel(const unsigned long long *__restrict__ keys, unsigned n, int shift, unsigned ntiles,
                                                        unsigned *__restrict__ table /* [256][ntiles] */, unsigned *__restrict__ totals /* [256] */)
{
    __shared__ unsigned hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const unsigned i = base + r * RS_T + threadIdx.x;
        if (i < n) atomicAdd(&hist[rs_digit<DESC>(keys[i], shift)], 1u);
    }
    __syncthreads();
    const unsigned h = hist[threadIdx.x];
    table[(size_t)threadIdx.x * ntiles + blockIdx.x] = h;
    if (h) atomicAdd(&totals[threadIdx.x], h);
}

// workgroup d: table[d][*] -> exclusive global offsets
__global__ __launch_bounds__(RS_T) void rs_offsets_kernel(unsigned *__restrict__ table, const unsigned *__restrict__ totals, unsigned ntiles)
{
    __shared__ unsigned part[RS_T];
    __shared__ unsigned carry;
    const unsigned d = blockIdx.x, t = threadIdx.x;
    // base of the digit: totals of the digits below it
    part[t] = t < d ? totals[t] : 0u;
    __syncthreads();
    for (int o = RS_T / 2; o > 0; o >>= 1) {
        if (t < (unsigned)o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) carry = part[0];
    __syncthreads();
    unsigned *row = table + (size_t)d * ntiles;
    for (unsigned b0 = 0; b0 < ntiles; b0 += RS_T) {
        const unsigned b = b0 + t;
        const unsigned v = b < ntiles ? row[b] : 0u;
        // inclusive scan of the 256 values of this chunk (Hillis-Steele in LDS)
        part[t] = v;
        __syncthreads();
        for (int o = 1; o < RS_T; o <<= 1) {
            const unsigned add = t >= (unsigned)o ? part[t - o] : 0u;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        const unsigned incl = part[t], c0 = carry;
        if (b < ntiles) row[b] = c0 + incl - v;
        __syncthreads();
        if (t == RS_T - 1) carry = c0 + incl;
        __syncthreads();
    }
}

template <bool DESC, bool PAIRS>
__global__ __launch_bounds__(RS_T) void rs_scatter_kernel(const unsigned long long *__restrict__ keys, unsigned long long *__restrict__ keys_out,
                                                          const unsigned *__restrict__ vals, unsigned *__restrict__ vals_out, unsigned n, int shift,
                                                          unsigned ntiles, const unsigned *__restrict__ table)
{
    __shared__ unsigned run[RS_WAVES][256];          // keys of a digit seen so far in the wave's run; then: destination of the wave's first
    const unsigned t = threadIdx.x, lane = t & 63u, w = t >> 6;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) run[k][t] = 0u;
    __syncthreads();
    const unsigned base = blockIdx.x * RS_TILE + w * (64u * RS_ITEMS);
    unsigned long long key[RS_ITEMS];
    unsigned rank[RS_ITEMS];
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const unsigned i = base + r * 64u + lane;
        const bool valid = i < n;
        key[r] = valid ? keys[i] : 0ull;
        const unsigned d = rs_digit<DESC>(key[r], shift);
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const bool one = (d >> bit) & 1u;
            const unsigned long long bal = __ballot(one);
            peers &= one ? bal : ~bal;
        }
        // (every lane of the group reads the counter before its lowest lane advances it: one wave, program order)
        volatile unsigned *counter = &run[w][d];
        const unsigned seen = valid ? *counter : 0u;
        rank[r] = seen + (unsigned)__popcll(peers & lt);
        if (valid && (peers & lt) == 0ull) *counter = seen + (unsigned)__popcll(peers);
    }
    __syncthreads();
    {
        // thread = digit: destination of each wave's first key of the digit
        unsigned at = table[(size_t)t * ntiles + blockIdx.x];
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) {
            const unsigned cnt = run[k][t];
            run[k][t] = at;
            at += cnt;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const unsigned i = base + r * 64u + lane;
        if (i < n) {
            const unsigned dst = run[w][rs_digit<DESC>(key[r], shift)] + rank[r];
            keys_out[dst] = key[r];
            if (PAIRS) vals_out[dst] = vals[i];
        }
    }
}

// ---- exclusive scan: tile sums -> scan of the sums (one workgroup) -> tiles
constexpr int SC_T = 256, SC_ITEMS = 8, SC_TILE = SC_T * SC_ITEMS;

template <int MODE> __device__ __forceinline__ unsigned sc_value(unsigned v) { return MODE == KM_SCAN_IS_ONE ? (v == 1u ? 1u : 0u) : v; }

__device__ __forceinline__ unsigned sc_block_exclusive(unsigned v, unsigned *lds /* [SC_T / 64 + 1] */, unsigned &total)
{
    // wave inclusive scan by shuffles, then the wave sums through LDS
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    unsigned s = v;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)s, o);
        if (lane >= (unsigned)o) s += u;
    }
    __syncthreads();
    if (lane == 63u) lds[w] = s;
    __syncthreads();
    unsigned before = 0u, all = 0u;
#pragma unroll
    for (int k = 0; k < SC_T / 64; k++) {
        const unsigned x = lds[k];
        before += (unsigned)k < w ? x : 0u;
        all += x;
    }
    total = all;
    return before + s - v;
}

template <int MODE>
__global__ __launch_bounds__(SC_T) void sc_sums_kernel(const unsigned *__restrict__ in, size_t n, unsigned *__restrict__ sums)
{
    __shared__ unsigned lds[SC_T / 64 + 1];
    const size_t base = (size_t)blockIdx.x * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
    unsigned v = 0u;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++)
        if (base + k < n) v += sc_value<MODE>(in[base + k]);
    unsigned total;
    (void)sc_block_exclusive(v, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SC_T) void sc_scan_sums_kernel(unsigned *__restrict__ sums, unsigned nsums)
{
    __shared__ unsigned lds[SC_T / 64 + 1];
    unsigned carry = 0u;
    for (unsigned b0 = 0; b0 < nsums; b0 += SC_T) {
        const unsigned b = b0 + threadIdx.x;
        const unsigned v = b < nsums ? sums[b] : 0u;
        unsigned total;
        const unsigned ex = sc_block_exclusive(v, lds, total);
        if (b < nsums) sums[b] = carry + ex;
        carry += total;
    }
}

template <int MODE>
__global__ __launch_bounds__(SC_T) void sc_tiles_kernel(const unsigned *__restrict__ in, unsigned *__restrict__ out, size_t n,
                                                        const unsigned *__restrict__ sums)
{
    __shared__ unsigned lds[SC_T / 64 + 1];
    const size_t base = (size_t)blockIdx.x * SC_TILE + (size_t)threadIdx.x * SC_ITEMS;
    unsigned x[SC_ITEMS], v = 0u;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        x[k] = base + k < n ? sc_value<MODE>(in[base + k]) : 0u;
        v += x[k];
    }
    unsigned total;
    unsigned at = sums[blockIdx.x] + sc_block_exclusive(v, lds, total);
#pragma unroll
    for (int k = 0; k < SC_ITEMS; k++) {
        if (base + k < n) out[base + k] = at;
        at += x[k];
    }
}

template <bool DESC>
int sort_passes(km_ctx *c, unsigned long long *k0, unsigned long long *k1, unsigned *v0, unsigned *v1, unsigned n, unsigned ntiles, unsigned *table,
                unsigned *totals)
{
    const bool pairs = v0 != nullptr;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = 8 * pass;
        unsigned *tot = totals + 256 * pass;
        rs_count_kernel<DESC><<<ntiles, RS_T, 0, c->stream>>>(k0, n, shift, ntiles, table, tot);
        KM_LAUNCH_CHECK(c);
        rs_offsets_kernel<<<256, RS_T, 0, c->stream>>>(table, tot, ntiles);
        KM_LAUNCH_CHECK(c);
        if (pairs) rs_scatter_kernel<DESC, true><<<ntiles, RS_T, 0, c->stream>>>(k0, k1, v0, v1, n, shift, ntiles, table);
        else rs_scatter_kernel<DESC, false><<<ntiles, RS_T, 0, c->stream>>>(k0, k1, nullptr, nullptr, n, shift, ntiles, table);
        KM_LAUNCH_CHECK(c);
        unsigned long long *tk = k0; k0 = k1; k1 = tk;
        unsigned *tv = v0; v0 = v1; v1 = tv;
    }
    return KM_OK;
}

}  // namespace

// Sorts keys_a[0, n) (and vals_a alongside, when given); keys_b / vals_b are the second buffers of the ping-pong.  After the eight
// passes the result lies in keys_a / vals_a again.
int km_sort_u64(km_ctx *c, unsigned long long *keys_a, unsigned long long *keys_b, unsigned *vals_a, unsigned *vals_b, size_t n, bool descending)
{
    if (n == 0) return KM_OK;
    if (n > 0xfffff000ull) return km_fail(c, KM_E_ARG, "sort: %zu keys exceed the 32-bit positions of this sort", n);
    if ((vals_a == nullptr) != (vals_b == nullptr)) return km_fail(c, KM_E_ARG, "sort: payload buffers must come in pairs");
    const unsigned ntiles = (unsigned)((n + RS_TILE - 1) / RS_TILE);
    unsigned *tmp = (unsigned *)km_ws(c, WS_SORT_TMP, ((size_t)256 * ntiles + 8 * 256) * sizeof(unsigned));
    if (!tmp) return KM_E_NOMEM;
    unsigned *totals = tmp, *table = tmp + 8 * 256;
    KM_HIP(c, hipMemsetAsync(totals, 0, 8 * 256 * sizeof(unsigned), c->stream));
    return descending ? sort_passes<true>(c, keys_a, keys_b, vals_a, vals_b, (unsigned)n, ntiles, table, totals)
                      : sort_passes<false>(c, keys_a, keys_b, vals_a, vals_b, (unsigned)n, ntiles, table, totals);
}

// out[i] = sum of value(in[j]) for j < i; value = the count itself (KM_SCAN_PLAIN) or 1 where the word equals 1 (KM_SCAN_IS_ONE: the
// "accepted" state of the corner selection).  in == out is allowed.  `tmp_slot`: workspace slot of the tile sums.
int km_exclusive_scan(km_ctx *c, const unsigned *in, unsigned *out, size_t n, int mode, int tmp_slot)
{
    if (n == 0) return KM_OK;
    const size_t ntiles = (n + SC_TILE - 1) / SC_TILE;
    if (ntiles > 0x7fffffffull) return km_fail(c, KM_E_ARG, "scan: %zu values are too many", n);
    unsigned *sums = (unsigned *)km_ws(c, tmp_slot, ntiles * sizeof(unsigned));
    if (!sums) return KM_E_NOMEM;
    if (mode == KM_SCAN_IS_ONE) sc_sums_kernel<KM_SCAN_IS_ONE><<<(unsigned)ntiles, SC_T, 0, c->stream>>>(in, n, sums);
    else sc_sums_kernel<KM_SCAN_PLAIN><<<(unsigned)ntiles, SC_T, 0, c->stream>>>(in, n, sums);
    KM_LAUNCH_CHECK(c);
    sc_scan_sums_kernel<<<1, SC_T, 0, c->stream>>>(sums, (unsigned)ntiles);
    KM_LAUNCH_CHECK(c);
    if (mode == KM_SCAN_IS_ONE) sc_tiles_kernel<KM_SCAN_IS_ONE><<<(unsigned)ntiles, SC_T, 0, c->stream>>>(in, out, n, sums);
    else sc_tiles_kernel<KM_SCAN_PLAIN><<<(unsigned)ntiles, SC_T, 0, c->stream>>>(in, out, n, sums);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
