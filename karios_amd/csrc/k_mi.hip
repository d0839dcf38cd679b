// K12 (next row): per-key-point mutual-information scores on the 57x57 chips of the RAW images
//   * `MutualInfoService._mutual_info`  (karios/matcher/mutual_info_service.py:32-63): Studholme NMI
//         (H(X)+H(Y)) / H(X,Y), natural log, NaN when H(X,Y) == 0
//   * `ZNCCService._mutual_information` (karios/matcher/zncc_service.py:129-151): 2*(H(X)+H(Y)-H(X,Y)) / (H(X)+H(Y)),
//         log2, NaN when H(X)+H(Y) == 0
// Both come from the same 32x32 joint histogram `np.histogram2d(chip_ref, chip_mon, bins=32)`: per-chip bin edges
// linspace(min, max, 33) (min-0.5 / max+0.5 for a constant chip), bin = searchsorted(edges, x, 'right') - 1 with the
// maximum folded into the last bin.  One wavefront per key point; the joint histogram lives in LDS (ds atomics).
// Key-point rounding / bounds rules are those of the ZNCC kernel (`_compute_mutual_info` = `_compute_zncc`).
#include "common.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define LK_LIKE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define MI_CHIP 57
#define MI_MARGIN 28
#define MI_NPX (MI_CHIP * MI_CHIP)
#define MI_BINS 32
#define MI_PER_LANE ((MI_NPX + 63) / 64)

__device__ __forceinline__ double mi_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double mi_wave_min(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double mi_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// np.linspace(lo, hi, 33)[i] as numpy evaluates it (step = (hi-lo)/32; i*step + lo; last element = hi)
__device__ __forceinline__ double mi_edge(int i, double lo, double hi, double step)
{
    return i == MI_BINS ? hi : __dadd_rn(__dmul_rn((double)i, step), lo);
}

__device__ __forceinline__ int mi_bin(double x, double lo, double hi, double step, const double *edges)
{
    int b = (int)((x - lo) / step);
    b = min(max(b, 0), MI_BINS - 1);
    while (b < MI_BINS - 1 && x >= edges[b + 1]) b++;
    while (b > 0 && x < edges[b]) b--;
    return b;   // x == hi lands in the last bin (numpy's on_edge correction)
}

template <typename T>
__global__ __launch_bounds__(256) void mi_kernel(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon,
                                                 int Wmon, ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0,
                                                 const float *__restrict__ y0, const float *__restrict__ dx,
                                                 const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                                 const float *__restrict__ score, float score_thr, double *__restrict__ out_studholme,
                                                 double *__restrict__ out_nmi, km_window win)
{
    __shared__ unsigned s_hist[4][MI_BINS * MI_BINS];
    __shared__ double s_edges[4][2][MI_BINS + 1];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wv;
    if (k >= (d_n ? min(*d_n, n) : n)) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    auto give_up = [&]() { if (lane == 0) { if (out_studholme) out_studholme[k] = nan; if (out_nmi) out_nmi[k] = nan; } };
    if (score && !(score[k] >= score_thr)) { give_up(); return; }
    const float fx0 = x0[k], fy0 = y0[k];
    int X0 = (int)fx0, Y0 = (int)fy0;
    const float sx = __fadd_rn(fx0, dx[k]), sy = __fadd_rn(fy0, dy[k]);
    bool ok = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    int X1 = 0, Y1 = 0;
    if (ok) {
        X1 = __float2int_rn(sx); Y1 = __float2int_rn(sy);
        const int Wr = win.H ? win.W : Wref, Hr = win.H ? win.H : Href, Wm = win.H ? win.W : Wmon, Hm = win.H ? win.H : Hmon;   // see k_zncc.hip
        ok = !(X0 - MI_MARGIN < 0 || Y0 - MI_MARGIN < 0 || X1 - MI_MARGIN < 0 || Y1 - MI_MARGIN < 0) &&
             !(X0 >= Wr - MI_MARGIN || Y0 >= Hr - MI_MARGIN || X1 >= Wm - MI_MARGIN || Y1 >= Hm - MI_MARGIN);
    }
    if (!ok) { give_up(); return; }
    if (win.H) {
        X0 -= win.ox; X1 -= win.ox; Y0 -= win.oy; Y1 -= win.oy;
        if (X0 - MI_MARGIN < 0 || Y0 - MI_MARGIN < 0 || X1 - MI_MARGIN < 0 || Y1 - MI_MARGIN < 0 || X0 + MI_MARGIN >= Wref || Y0 + MI_MARGIN >= Href ||
            X1 + MI_MARGIN >= Wmon || Y1 + MI_MARGIN >= Hmon) {
            const double miss = __longlong_as_double((long long)KM_NAN_OUTSIDE_WINDOW);
            if (lane == 0) { if (out_studholme) out_studholme[k] = miss; if (out_nmi) out_nmi[k] = miss; }
            return;
        }
    }
    unsigned *hist = s_hist[wv];
    double *e1 = s_edges[wv][0], *e2 = s_edges[wv][1];
    for (int i = lane; i < MI_BINS * MI_BINS; i += 64) hist[i] = 0;
    // chips -> registers, per-chip min / max
    double va[MI_PER_LANE], vb[MI_PER_LANE];
    double mn1 = INFINITY, mx1 = -INFINITY, mn2 = INFINITY, mx2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < MI_PER_LANE; i++) {
        const int idx = i * 64 + lane;
        va[i] = 0; vb[i] = 0;
        if (idx < MI_NPX) {
            const int r = idx / MI_CHIP, cx = idx - r * MI_CHIP;
            va[i] = (double)ref[(size_t)(Y0 - MI_MARGIN + r) * sref + (X0 - MI_MARGIN + cx)];
            vb[i] = (double)mon[(size_t)(Y1 - MI_MARGIN + r) * smon + (X1 - MI_MARGIN + cx)];
            mn1 = fmin(mn1, va[i]); mx1 = fmax(mx1, va[i]); mn2 = fmin(mn2, vb[i]); mx2 = fmax(mx2, vb[i]);
        }
    }
    mn1 = mi_wave_min(mn1); mx1 = mi_wave_max(mx1); mn2 = mi_wave_min(mn2); mx2 = mi_wave_max(mx2);
    if (mn1 == mx1) { mn1 -= 0.5; mx1 += 0.5; }   // numpy _get_outer_edges for a constant sample
    if (mn2 == mx2) { mn2 -= 0.5; mx2 += 0.5; }
    const double st1 = (mx1 - mn1) / MI_BINS, st2 = (mx2 - mn2) / MI_BINS;
    if (lane <= MI_BINS) { e1[lane] = mi_edge(lane, mn1, mx1, st1); e2[lane] = mi_edge(lane, mn2, mx2, st2); }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int i = 0; i < MI_PER_LANE; i++) {
        if (i * 64 + lane < MI_NPX) {
            const int b1 = mi_bin(va[i], mn1, mx1, st1, e1), b2 = mi_bin(vb[i], mn2, mx2, st2, e2);
            atomicAdd(&hist[b1 * MI_BINS + b2], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // entropies: joint over 1024 cells (16 per lane); marginals: lane b < 32 sums row b / column b
    const double inv_n = 1.0 / (double)MI_NPX;
    double hxy = 0;
    for (int i = lane; i < MI_BINS * MI_BINS; i += 64) {
        const unsigned cnt = hist[i];
        if (cnt) { const double p = (double)cnt / (double)MI_NPX; hxy -= p * log(p); }
    }
    double hx = 0, hy = 0;
    if (lane < MI_BINS) {
        unsigned rx = 0, ry = 0;
        for (int j = 0; j < MI_BINS; j++) { rx += hist[lane * MI_BINS + j]; ry += hist[j * MI_BINS + lane]; }
        if (rx) { const double p = (double)rx / (double)MI_NPX; hx = -p * log(p); }
        if (ry) { const double p = (double)ry / (double)MI_NPX; hy = -p * log(p); }
    }
    (void)inv_n;
    hxy = mi_wave_sum(hxy); hx = mi_wave_sum(hx); hy = mi_wave_sum(hy);
    if (lane == 0) {
        if (out_studholme) out_studholme[k] = hxy == 0.0 ? nan : (hx + hy) / hxy;
        if (out_nmi) {
            // entropies in bits for this variant (log2); the ratio is base-independent but the zero test is not rescaled
            const double denom = hx + hy;
            out_nmi[k] = denom == 0.0 ? nan : 2.0 * (hx + hy - hxy) / denom;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Integer pixel types (uint8 / uint16 / int16 - everything the reference reads from a raster except float32).  Round 4: the same
// scores from the same histogram, with the arithmetic the data allows:
//   * bins.  numpy's edges are linspace(lo, hi, 33) = lo + i * (R / 32), R = hi - lo.  For integer samples R / 32 is a dyadic
//     rational, exact in float64, and so is every edge; "edge_i <= x" is the integer statement i * R <= 32 (x - lo): the bin is
//     floor(32 (x - lo) / R), the maximum folded into bin 31 (numpy's on_edge rule) - one multiplication by a per-chip constant
//     (see the kernel).  No float64 division per pixel, no edge table.
//   * entropies.  A cell holds c of N = 3249 samples: -sum p ln p = ln N - (1 / N) sum c ln c, and c ln c comes from a 3250-entry
//     float64 table (host libm, uploaded once per context) - the first form spent most of its instructions in 18 float64 log()
//     calls per lane.  The two "undefined" cases of the reference (H(X,Y) == 0; H(X) + H(Y) == 0) both mean ONE occupied cell
//     and are detected on the counts, never on a rounded entropy.
//   * chips stay in registers as packed 16-bit pairs (57 VGPRs instead of 204 for float64 copies; lane = chip column, one chip row per
//     step), key points are dealt to the XCDs in contiguous eighths of the (x0, y0)-ordered rows like the ZNCC kernel's (neighbouring
//     chips overlap in one L2).
// Results agree with the first form (and the oracle's numpy) to ~1e-15; the gate is 1e-9.
// (144 - 168 VGPRs = 3 waves per SIMD; 128 for a fourth wave spills ~70 bytes - measured as built: 0.097 ms at 20 000 points)
template <typename T>
__device__ __forceinline__ void mi_int_item(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon, int Wmon,
                                            ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0, const float *__restrict__ y0,
                                            const float *__restrict__ dx, const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                            const float *__restrict__ score, float score_thr, double *__restrict__ out_studholme,
                                            double *__restrict__ out_nmi, const km_window &win, const double *__restrict__ clogc,
                                            unsigned (&s_hist)[4][MI_BINS * MI_BINS])
{
    const int n_rows = d_n ? min(*d_n, n) : n;
    const unsigned per = ((unsigned)(n_rows + 3) / 4 + KM_XCDS - 1) / KM_XCDS, blk = (blockIdx.x % KM_XCDS) * per + blockIdx.x / KM_XCDS;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = (int)blk * 4 + wv;
    if (blockIdx.x / KM_XCDS >= per || k >= n_rows) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    auto give = [&](double v) { if (lane == 0) { if (out_studholme) out_studholme[k] = v; if (out_nmi) out_nmi[k] = v; } };
    if (score && !(score[k] >= score_thr)) { give(nan); return; }
    const float fx0 = x0[k], fy0 = y0[k];
    int X0 = (int)fx0, Y0 = (int)fy0;
    const float sx = __fadd_rn(fx0, dx[k]), sy = __fadd_rn(fy0, dy[k]);
    bool ok = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    int X1 = 0, Y1 = 0;
    if (ok) {
        X1 = __float2int_rn(sx); Y1 = __float2int_rn(sy);
        const int Wr = win.H ? win.W : Wref, Hr = win.H ? win.H : Href, Wm = win.H ? win.W : Wmon, Hm = win.H ? win.H : Hmon;   // see k_zncc.hip
        ok = !(X0 - MI_MARGIN < 0 || Y0 - MI_MARGIN < 0 || X1 - MI_MARGIN < 0 || Y1 - MI_MARGIN < 0) &&
             !(X0 >= Wr - MI_MARGIN || Y0 >= Hr - MI_MARGIN || X1 >= Wm - MI_MARGIN || Y1 >= Hm - MI_MARGIN);
    }
    if (!ok) { give(nan); return; }
    if (win.H) {
        X0 -= win.ox; X1 -= win.ox; Y0 -= win.oy; Y1 -= win.oy;
        if (X0 - MI_MARGIN < 0 || Y0 - MI_MARGIN < 0 || X1 - MI_MARGIN < 0 || Y1 - MI_MARGIN < 0 || X0 + MI_MARGIN >= Wref || Y0 + MI_MARGIN >= Href ||
            X1 + MI_MARGIN >= Wmon || Y1 + MI_MARGIN >= Hmon) {
            give(__longlong_as_double((long long)KM_NAN_OUTSIDE_WINDOW));
            return;
        }
    }
    unsigned *hist = s_hist[wv];
#pragma unroll
    for (int i = 0; i < MI_BINS * MI_BINS / 64; i++) hist[i * 64 + lane] = 0;
    // chips -> registers.  Lane = chip column (57 of 64 lanes work), one chip row per step: no index arithmetic, the row base is a
    // scalar; a pixel pair (ref, mon) is one register of two 16-bit halves (biased to unsigned: the bias cancels in x - lo), and the
    // per-chip minimum / maximum run in packed 16-bit arithmetic (the PMC pass of the first integer form: 3 058 VALU instructions per
    // key point at 82 % pipe occupancy - the kernel is bound by instruction issue, not by its LDS atomics)
    constexpr unsigned BIAS2 = std::is_signed<T>::value ? 0x80008000u : 0u;
    const int bx0 = __builtin_amdgcn_readfirstlane(X0 - MI_MARGIN), by0 = __builtin_amdgcn_readfirstlane(Y0 - MI_MARGIN);
    const int bx1 = __builtin_amdgcn_readfirstlane(X1 - MI_MARGIN), by1 = __builtin_amdgcn_readfirstlane(Y1 - MI_MARGIN);
    const T *pr = ref + (ptrdiff_t)by0 * sref + bx0;
    const T *pm = mon + (ptrdiff_t)by1 * smon + bx1;
    const __amdgpu_buffer_rsrc_t bra = __builtin_amdgcn_make_buffer_rsrc((void *)pr, 0, (int)(((MI_CHIP - 1) * sref + MI_CHIP) * (ptrdiff_t)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t brb = __builtin_amdgcn_make_buffer_rsrc((void *)pm, 0, (int)(((MI_CHIP - 1) * smon + MI_CHIP) * (ptrdiff_t)sizeof(T)), 0x00020000);
    const unsigned rowa = (unsigned)(sref * (ptrdiff_t)sizeof(T)), rowb = (unsigned)(smon * (ptrdiff_t)sizeof(T));
    const bool col_on = lane < MI_CHIP;
    const unsigned cl = (col_on ? (unsigned)lane : 0u) * (unsigned)sizeof(T);
    typedef unsigned short mi_us2 __attribute__((ext_vector_type(2)));
    unsigned pk[MI_CHIP];
    mi_us2 vmin = {0xffff, 0xffff}, vmax = {0, 0};
#pragma unroll
    for (int i = 0; i < MI_CHIP; i++) {
        // (buffer descriptors: scalar row offsets, one lane offset - no vector instruction goes into addressing)
        const unsigned a = km_chip_px<T>(bra, cl, (unsigned)i * rowa), b = km_chip_px<T>(brb, cl, (unsigned)i * rowb);
        pk[i] = (a | (b << 16)) ^ BIAS2;             // (int16: two's complement + 0x8000 = the value + 32768, for both halves at once)
        const mi_us2 v = __builtin_bit_cast(mi_us2, pk[i]);
        vmin = __builtin_elementwise_min(vmin, v);
        vmax = __builtin_elementwise_max(vmax, v);
    }
    if (!col_on) { vmin = mi_us2{0xffff, 0xffff}; vmax = mi_us2{0, 0}; }
    int mn1 = vmin.x, mx1 = vmax.x, mn2 = vmin.y, mx2 = vmax.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn1 = min(mn1, __shfl_xor(mn1, o)); mx1 = max(mx1, __shfl_xor(mx1, o));
        mn2 = min(mn2, __shfl_xor(mn2, o)); mx2 = max(mx2, __shfl_xor(mx2, o));
    }
    const int R1 = mx1 - mn1, R2 = mx2 - mn2;
    // floor(32 d / R) for 0 <= d <= R < 2^16 by ONE multiplication: with L = ceil(log2 R), s = 21 + L and M = ceil(2^s / R),
    // M R = 2^s + e with 0 <= e < R <= 2^L, hence 32 d M / 2^s = 32 d / R + 32 d e / (R 2^s) and the excess is below
    // 2^21 2^L / (R 2^s) = 1 / R: it cannot carry a quotient with fractional part <= (R - 1) / R over the next integer.  M <= 2^22, the
    // product d M < 2^38: a 64-bit multiply-add and a 64-bit shift by s - 5.  (M from a float64 division: 2^s / R is an integer only
    // for R a power of two, where the division is exact; otherwise it lies >= 1 / R from the integers, far beyond its 2^-31 error.)
    auto magic = [](int R, unsigned &M, int &sh) {
        const int L = R > 1 ? 32 - __clz(R - 1) : 0;
        M = R ? (unsigned)ceil(ldexp(1.0, 21 + L) / (double)R) : 0u;
        sh = 21 + L - 5;
    };
    unsigned M1, M2;
    int sh1, sh2;
    magic(R1, M1, sh1); magic(R2, M2, sh2);
    LK_LIKE_SYNC();
    if (col_on) {
#pragma unroll
        for (int i = 0; i < MI_CHIP; i++) {
            const unsigned d1 = (pk[i] & 0xffffu) - (unsigned)mn1, d2 = (pk[i] >> 16) - (unsigned)mn2;
            const int q1 = (int)(unsigned)(((unsigned long long)d1 * M1) >> sh1), q2 = (int)(unsigned)(((unsigned long long)d2 * M2) >> sh2);
            // the maximum folds into the last bin (numpy's on_edge rule); a constant chip: numpy widens the range to [v - 0.5, v + 0.5],
            // the samples sit in the middle bin 16
            const int b1 = R1 ? min(q1, MI_BINS - 1) : MI_BINS / 2, b2 = R2 ? min(q2, MI_BINS - 1) : MI_BINS / 2;
            atomicAdd(&hist[b1 * MI_BINS + b2], 1u);
        }
    }
    LK_LIKE_SYNC();
    // sum c ln c over the joint cells (16 per lane) and over the two marginals (lane b < 32: row b / column b)
    double sxy = 0;
    int occupied = 0;
#pragma unroll
    for (int i = 0; i < MI_BINS * MI_BINS / 64; i++) {
        const unsigned c = hist[i * 64 + lane];
        occupied += c != 0;
        sxy += clogc[c];
    }
    double sxm = 0, sym = 0;
    if (lane < MI_BINS) {
        unsigned rx = 0, ry = 0;
#pragma unroll 8
        for (int j = 0; j < MI_BINS; j++) { rx += hist[lane * MI_BINS + ((j + lane) & (MI_BINS - 1))]; ry += hist[j * MI_BINS + lane]; }
        sxm = clogc[rx]; sym = clogc[ry];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) occupied += __shfl_xor(occupied, o);
    sxy = mi_wave_sum(sxy); sxm = mi_wave_sum(sxm); sym = mi_wave_sum(sym);
    if (lane == 0) {
        const double lnN = clogc[MI_NPX] / (double)MI_NPX, invN = 1.0 / (double)MI_NPX;
        const double hxy = lnN - sxy * invN, hx = lnN - sxm * invN, hy = lnN - sym * invN;
        // one occupied cell: H(X,Y) = H(X) = H(Y) = 0 exactly in the reference -> both scores undefined (NaN)
        if (out_studholme) out_studholme[k] = occupied == 1 ? nan : (hx + hy) / hxy;
        if (out_nmi) out_nmi[k] = occupied == 1 ? nan : 2.0 * (hx + hy - hxy) / (hx + hy);
    }
}

// c ln c for c = 0 .. 57^2 (float64, host libm), in a workspace slot of the context
template <typename T>
__global__ __launch_bounds__(256) void mi_int_kernel(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon, int Wmon,
                                                     ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0, const float *__restrict__ y0,
                                                     const float *__restrict__ dx, const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                                     const float *__restrict__ score, float score_thr, double *__restrict__ out_studholme,
                                                     double *__restrict__ out_nmi, km_window win, const double *__restrict__ clogc)
{
    __shared__ unsigned s_hist[4][MI_BINS * MI_BINS];
    mi_int_item<T>(ref, mon, Href, Wref, Hmon, Wmon, sref, smon, x0, y0, dx, dy, n, d_n, score, score_thr, out_studholme, out_nmi, win, clogc, s_hist);
}
// batched units: blockIdx.y = unit
template <typename T>
__global__ __launch_bounds__(256) void mi_int_units_kernel(km_score_units A, int n, float score_thr, const double *__restrict__ clogc)
{
    __shared__ unsigned s_hist[4][MI_BINS * MI_BINS];
    const km_score_unit &U = A.u[blockIdx.y];
    mi_int_item<T>((const T *)U.ref, (const T *)U.mon, U.Href, U.Wref, U.Hmon, U.Wmon, U.sref, U.smon, U.x0, U.y0, U.dx, U.dy, n, U.d_n, U.score, score_thr,
                   U.out, U.out2, U.win, clogc, s_hist);
}

static const double *mi_table(km_ctx *c)
{
    // (a table that outlives the call: always the context's lane-0 slot - the lanes of pipelined batched submissions share it)
    const int lane = c->lane;
    c->lane = 0;
    double *d = (double *)km_ws(c, WS_MI_TABLE, (size_t)(MI_NPX + 1) * sizeof(double));
    c->lane = lane;
    if (!d) return nullptr;
    if (!c->mi_table_ready) {
        std::vector<double> h((size_t)MI_NPX + 1);
        h[0] = 0.0;
        for (int i = 1; i <= MI_NPX; i++) h[(size_t)i] = (double)i * log((double)i);
        if (km_h2d_staged(c, c->stream, d, h.size() * 8, h.data(), h.size() * 8, h.size() * 8, 1) != KM_OK) return nullptr;
        c->mi_table_ready = true;
    }
    return d;
}

// both mutual-information scores of the confident rows of every unit's frame (integer pixels): one launch
int kmi_units(km_ctx *c, const km_score_units &A, int n_units, int dtype, int n, float score_thr)
{
    if (n <= 0 || n_units <= 0) return KM_OK;
    const double *tab = mi_table(c);
    if (!tab) return KM_E_NOMEM;
    const dim3 grid(km_xcd_grid((unsigned)((n + 3) / 4)), n_units);
    switch (dtype) {
    case KM_U8: mi_int_units_kernel<uint8_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr, tab); break;
    case KM_U16: mi_int_units_kernel<uint16_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr, tab); break;
    case KM_I16: mi_int_units_kernel<int16_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr, tab); break;
    default: return KM_E_UNSUPPORTED;
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kmi_batch(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
              ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, const int *d_n,
              const float *d_score, float score_thr, double *d_studholme, double *d_nmi)
{
    if (n <= 0) return KM_OK;
    if (dtype != KM_F32) {
        const double *tab = mi_table(c);
        if (!tab) return KM_E_NOMEM;
        const int nbx = (int)km_xcd_grid((unsigned)((n + 3) / 4));
#define KM_MII(T) mi_int_kernel<T><<<nbx, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_studholme, d_nmi, c->window, tab)
        switch (dtype) {
        case KM_U8: KM_MII(uint8_t); break;
        case KM_U16: KM_MII(uint16_t); break;
        case KM_I16: KM_MII(int16_t); break;
        default: return km_fail(c, KM_E_ARG, "mi: bad dtype %d", dtype);
        }
#undef KM_MII
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
    const int nb = (n + 3) / 4;
#define KM_MI(T) mi_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_studholme, d_nmi, c->window)
    KM_MI(float);          // (integer pixels: integer bins and table entropies above)
#undef KM_MI
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
