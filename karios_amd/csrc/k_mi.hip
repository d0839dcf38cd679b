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

int kmi_batch(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
              ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, const int *d_n,
              const float *d_score, float score_thr, double *d_studholme, double *d_nmi)
{
    if (n <= 0) return KM_OK;
    const int nb = (n + 3) / 4;
#define KM_MI(T) mi_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_studholme, d_nmi, c->window)
    switch (dtype) {
    case KM_U8: KM_MI(uint8_t); break;
    case KM_U16: KM_MI(uint16_t); break;
    case KM_I16: KM_MI(int16_t); break;
    case KM_F32: KM_MI(float); break;
    default: return km_fail(c, KM_E_ARG, "mi: bad dtype %d", dtype);
    }
#undef KM_MI
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
