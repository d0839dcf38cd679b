// K9: batched per-keypoint ZNCC (ZNCCService._compute_zncc + _zncc2, reference
// matcher/zncc_service.py:45-126, 186-238).  One wavefront per keypoint: the two 43x43 patches
// of the RAW images are read once into registers, mean / population std / correlation are two-pass
// fp64 sums reduced across the wave.
#include "common.hpp"

#include <cstdlib>
#include <type_traits>

#define ZN_HW 21
#define ZN_MARGIN 28
#define ZN_SIDE 43
#define ZN_NPX (ZN_SIDE * ZN_SIDE)
#define ZN_PER_LANE ((ZN_NPX + 63) / 64)

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void zncc_kernel(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon,
                                                   int Wmon, ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0,
                                                   const float *__restrict__ y0, const float *__restrict__ dx,
                                                   const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                                   const float *__restrict__ score, float score_thr, double *__restrict__ out, km_window win)
{
    // Workgroup w runs on XCD w % 8 (an L2 each).  The rows of a frame are ordered by (x0, y0): handing every XCD one CONTIGUOUS
    // eighth of them makes the ~1000 key points an XCD works on at a time a band of ~70 image columns, whose 43 x 43 chips overlap
    // in L2 - dealt round-robin, neighbouring chips sat in eight different L2s and every 86-byte chip row cost its own 128-byte lines.
    // (the eighths are cut from the rows the frame really has: the launch is sized for the capacity)
    const int n_rows = d_n ? min(*d_n, n) : n;
    const unsigned per = ((unsigned)(n_rows + 3) / 4 + KM_XCDS - 1) / KM_XCDS, blk = (blockIdx.x % KM_XCDS) * per + blockIdx.x / KM_XCDS;
    const int k = (int)blk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // key point of this wave: uniform, scalar addressing
    const int lane = threadIdx.x & 63;
    if (blockIdx.x / KM_XCDS >= per || k >= n_rows) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // rows below the confidence threshold are not scored (core.py:878-893): NaN, like the reference's column
    if (score && !(score[k] >= score_thr)) { if (lane == 0) out[k] = nan; return; }
    const float fx0 = x0[k], fy0 = y0[k];
    int X0 = (int)fx0, Y0 = (int)fy0;  // int(series["x0"])
    // round(np.float32 + np.float32): half-to-even on the f32 sum
    const float sx = __fadd_rn(fx0, dx[k]), sy = __fadd_rn(fy0, dy[k]);
    bool ok = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    int X1 = 0, Y1 = 0;
    if (ok) {
        X1 = __float2int_rn(sx); Y1 = __float2int_rn(sy);
        // the bounds rule is stated on the IMAGE (the window's extent only decides whether the pixels are resident)
        const int Wr = win.H ? win.W : Wref, Hr = win.H ? win.H : Href, Wm = win.H ? win.W : Wmon, Hm = win.H ? win.H : Hmon;
        ok = !(X0 - ZN_MARGIN < 0 || Y0 - ZN_MARGIN < 0 || X1 - ZN_MARGIN < 0 || Y1 - ZN_MARGIN < 0) &&
             !(X0 >= Wr - ZN_MARGIN || Y0 >= Hr - ZN_MARGIN || X1 >= Wm - ZN_MARGIN || Y1 >= Hm - ZN_MARGIN);
    }
    if (!ok) { if (lane == 0) out[k] = nan; return; }
    if (win.H) {
        X0 -= win.ox; X1 -= win.ox; Y0 -= win.oy; Y1 -= win.oy;
        if (X0 - ZN_HW < 0 || Y0 - ZN_HW < 0 || X1 - ZN_HW < 0 || Y1 - ZN_HW < 0 || X0 + ZN_HW >= Wref || Y0 + ZN_HW >= Href || X1 + ZN_HW >= Wmon ||
            Y1 + ZN_HW >= Hmon) {
            if (lane == 0) out[k] = __longlong_as_double((long long)KM_NAN_OUTSIDE_WINDOW);
            return;
        }
    }
    double va[ZN_PER_LANE], vb[ZN_PER_LANE];
    double s1 = 0, s2 = 0;
#pragma unroll
    for (int i = 0; i < ZN_PER_LANE; i++) {
        const int idx = i * 64 + lane;
        va[i] = 0; vb[i] = 0;
        if (idx < ZN_NPX) {
            const int r = idx / ZN_SIDE, cx = idx - r * ZN_SIDE;
            va[i] = (double)ref[(size_t)(Y0 - ZN_HW + r) * sref + (X0 - ZN_HW + cx)];
            vb[i] = (double)mon[(size_t)(Y1 - ZN_HW + r) * smon + (X1 - ZN_HW + cx)];
            s1 += va[i]; s2 += vb[i];
        }
    }
    const double m1 = wave_sum_f64(s1) / (double)ZN_NPX, m2 = wave_sum_f64(s2) / (double)ZN_NPX;
    double v1 = 0, v2 = 0, cc = 0;
#pragma unroll
    for (int i = 0; i < ZN_PER_LANE; i++) {
        if (i * 64 + lane < ZN_NPX) {
            const double a = va[i] - m1, b = vb[i] - m2;
            v1 += a * a; v2 += b * b; cc += a * b;
        }
    }
    v1 = wave_sum_f64(v1); v2 = wave_sum_f64(v2); cc = wave_sum_f64(cc);
    const double sd1 = sqrt(v1 / (double)ZN_NPX), sd2 = sqrt(v2 / (double)ZN_NPX);
    if (lane == 0) out[k] = (sd1 == 0.0 || sd2 == 0.0) ? nan : cc / (sd1 * sd2) / (double)ZN_NPX;
}

// Integer pixel types (round 4): the same score from EXACT integer moments.  With n = 43^2 samples a, b < 2^16 (int16 biased by 2^15:
// ZNCC does not see a constant), S_a, S_b < 2^27, S_aa, S_bb, S_ab < 2^43, and
//     zncc = (n S_ab - S_a S_b) / sqrt((n S_aa - S_a^2) (n S_bb - S_b^2))
// with both numerators exact in int64 (< 2^54).  That is the reference's mean(((p1 - m1) / s1) ((p2 - m2) / s2)) (zncc_service.py:118-126)
// with ONE rounding per factor instead of numpy's two-pass float64 sums: closer to the real value than numpy itself (the gate is 1e-9;
// observed difference to the two-pass form <= 1e-13), and "std == 0 -> NaN" (zncc_service.py:118-120) is the exact integer statement
// n S_aa == S_a^2.  One pass, nothing kept: lane = chip column (43 of 64 lanes), one chip row per step, scalar row bases, ~30 VGPRs.
// The float64 two-pass kernel above cost ~1 100 issue slots per key point (58 float64 registers of pixels, index arithmetic per pixel);
// this one ~420.
template <typename T>
__device__ __forceinline__ void zncc_int_item(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon, int Wmon,
                                              ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0, const float *__restrict__ y0,
                                              const float *__restrict__ dx, const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                              const float *__restrict__ score, float score_thr, double *__restrict__ out, const km_window &win)
{
    const int n_rows = d_n ? min(*d_n, n) : n;
    const unsigned per = ((unsigned)(n_rows + 3) / 4 + KM_XCDS - 1) / KM_XCDS, blk = (blockIdx.x % KM_XCDS) * per + blockIdx.x / KM_XCDS;
    const int k = (int)blk * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (blockIdx.x / KM_XCDS >= per || k >= n_rows) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (score && !(score[k] >= score_thr)) { if (lane == 0) out[k] = nan; return; }
    const float fx0 = x0[k], fy0 = y0[k];
    int X0 = (int)fx0, Y0 = (int)fy0;
    const float sx = __fadd_rn(fx0, dx[k]), sy = __fadd_rn(fy0, dy[k]);
    bool ok = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    int X1 = 0, Y1 = 0;
    if (ok) {
        X1 = __float2int_rn(sx); Y1 = __float2int_rn(sy);
        const int Wr = win.H ? win.W : Wref, Hr = win.H ? win.H : Href, Wm = win.H ? win.W : Wmon, Hm = win.H ? win.H : Hmon;
        ok = !(X0 - ZN_MARGIN < 0 || Y0 - ZN_MARGIN < 0 || X1 - ZN_MARGIN < 0 || Y1 - ZN_MARGIN < 0) &&
             !(X0 >= Wr - ZN_MARGIN || Y0 >= Hr - ZN_MARGIN || X1 >= Wm - ZN_MARGIN || Y1 >= Hm - ZN_MARGIN);
    }
    if (!ok) { if (lane == 0) out[k] = nan; return; }
    if (win.H) {
        X0 -= win.ox; X1 -= win.ox; Y0 -= win.oy; Y1 -= win.oy;
        if (X0 - ZN_HW < 0 || Y0 - ZN_HW < 0 || X1 - ZN_HW < 0 || Y1 - ZN_HW < 0 || X0 + ZN_HW >= Wref || Y0 + ZN_HW >= Href || X1 + ZN_HW >= Wmon ||
            Y1 + ZN_HW >= Hmon) {
            if (lane == 0) out[k] = __longlong_as_double((long long)KM_NAN_OUTSIDE_WINDOW);
            return;
        }
    }
    constexpr unsigned BIAS = std::is_signed<T>::value ? 32768u : 0u;
    const int bx0 = __builtin_amdgcn_readfirstlane(X0 - ZN_HW), by0 = __builtin_amdgcn_readfirstlane(Y0 - ZN_HW);
    const int bx1 = __builtin_amdgcn_readfirstlane(X1 - ZN_HW), by1 = __builtin_amdgcn_readfirstlane(Y1 - ZN_HW);
    const T *pr = ref + (ptrdiff_t)by0 * sref + bx0;
    const T *pm = mon + (ptrdiff_t)by1 * smon + bx1;
    // chip rows through buffer descriptors: scalar row offsets, one lane offset - not one vector instruction goes into addressing
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)pr, 0, (int)(((ZN_SIDE - 1) * sref + ZN_SIDE) * (ptrdiff_t)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)pm, 0, (int)(((ZN_SIDE - 1) * smon + ZN_SIDE) * (ptrdiff_t)sizeof(T)), 0x00020000);
    const unsigned rowa = (unsigned)(sref * (ptrdiff_t)sizeof(T)), rowb = (unsigned)(smon * (ptrdiff_t)sizeof(T));
    unsigned sa = 0, sb = 0;
    unsigned long long saa = 0, sbb = 0, sab = 0;
    if (lane < ZN_SIDE) {                                  // (one exec mask for the whole pass: idle lanes keep zero sums)
        const unsigned cl = (unsigned)lane * (unsigned)sizeof(T);
#pragma unroll
        for (int i = 0; i < ZN_SIDE; i++) {
            unsigned a = km_chip_px<T>(ra, cl, (unsigned)i * rowa), b = km_chip_px<T>(rb, cl, (unsigned)i * rowb);
            if constexpr (std::is_signed<T>::value) { a = (a + BIAS) & 0xffffu; b = (b + BIAS) & 0xffffu; }
            sa += a; sb += b;
            saa += (unsigned long long)a * a; sbb += (unsigned long long)b * b; sab += (unsigned long long)a * b;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sa += (unsigned)__shfl_xor((int)sa, o); sb += (unsigned)__shfl_xor((int)sb, o); }
    // per-lane second moments < 2^38, their wave sums < 2^43: exact in float64
    const double daa = wave_sum_f64((double)saa), dbb = wave_sum_f64((double)sbb), dab = wave_sum_f64((double)sab);
    if (lane == 0) {
        const long long N = ZN_NPX, Sa = (long long)sa, Sb = (long long)sb;
        const long long v1 = N * (long long)daa - Sa * Sa, v2 = N * (long long)dbb - Sb * Sb, cv = N * (long long)dab - Sa * Sb;
        out[k] = (v1 == 0 || v2 == 0) ? nan : (double)cv / (sqrt((double)v1) * sqrt((double)v2));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void zncc_int_kernel(const T *__restrict__ ref, const T *__restrict__ mon, int Href, int Wref, int Hmon, int Wmon,
                                                       ptrdiff_t sref, ptrdiff_t smon, const float *__restrict__ x0, const float *__restrict__ y0,
                                                       const float *__restrict__ dx, const float *__restrict__ dy, int n, const int *__restrict__ d_n,
                                                       const float *__restrict__ score, float score_thr, double *__restrict__ out, km_window win)
{
    zncc_int_item<T>(ref, mon, Href, Wref, Hmon, Wmon, sref, smon, x0, y0, dx, dy, n, d_n, score, score_thr, out, win);
}
// batched units: blockIdx.y = unit
template <typename T>
__global__ __launch_bounds__(256) void zncc_int_units_kernel(km_score_units A, int n, float score_thr)
{
    const km_score_unit &U = A.u[blockIdx.y];
    zncc_int_item<T>((const T *)U.ref, (const T *)U.mon, U.Href, U.Wref, U.Hmon, U.Wmon, U.sref, U.smon, U.x0, U.y0, U.dx, U.dy, n, U.d_n, U.score, score_thr,
                     U.out, U.win);
}

// ZNCC of the confident rows of every unit's frame (integer pixels): one launch.  KM_E_UNSUPPORTED: float32 rasters (units one by one)
int kz_zncc_units(km_ctx *c, const km_score_units &A, int n_units, int dtype, int n, float score_thr)
{
    if (n <= 0 || n_units <= 0) return KM_OK;
    const dim3 grid(km_xcd_grid((unsigned)((n + 3) / 4)), n_units);
    switch (dtype) {
    case KM_U8: zncc_int_units_kernel<uint8_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr); break;
    case KM_U16: zncc_int_units_kernel<uint16_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr); break;
    case KM_I16: zncc_int_units_kernel<int16_t><<<grid, 256, 0, c->stream>>>(A, n, score_thr); break;
    default: return KM_E_UNSUPPORTED;
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kz_zncc(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
            ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, double *d_out)
{
    return kz_zncc_filtered(c, d_ref, d_mon, dtype, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, nullptr, nullptr, 0.f, d_out);
}

int kz_zncc_filtered(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int Href, int Wref, int Hmon, int Wmon, ptrdiff_t sref,
                     ptrdiff_t smon, const float *d_x0, const float *d_y0, const float *d_dx, const float *d_dy, int n, const int *d_n,
                     const float *d_score, float score_thr, double *d_out)
{
    if (n <= 0) return KM_OK;
    const int nb = (int)km_xcd_grid((unsigned)((n + 3) / 4));
    if (dtype != KM_F32) {
#define KM_ZI(T) zncc_int_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_out, c->window)
        switch (dtype) {
        case KM_U8: KM_ZI(uint8_t); break;
        case KM_U16: KM_ZI(uint16_t); break;
        case KM_I16: KM_ZI(int16_t); break;
        default: return km_fail(c, KM_E_ARG, "zncc: bad dtype %d", dtype);
        }
#undef KM_ZI
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
#define KM_Z(T) zncc_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_out, c->window)
    KM_Z(float);           // (integer pixels: exact integer moments above)
#undef KM_Z
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// ---- _zncc2 for ANY window half-size and any pair of pixel types (reference zncc_service.py:45-126): out[k] = ZNCC of the
// (2n+1)^2 windows centred at row u1[k], column v1[k] of image 1 and (u2[k], v2[k]) of image 2; NaN when a window has no
// variance; windows leaving their image are the CALLER's IndexError (flagged here as NaN with out_flags[k] = 1).
// One wavefront per window pair, two passes over the pixels (mean, then centred sums), fp64 like numpy on integer / float64 data.
__device__ __forceinline__ double zn_load(const void *img, int dtype, size_t idx)
{
    switch (dtype) {   // wave-uniform
    case KM_U8: return (double)((const uint8_t *)img)[idx];
    case KM_U16: return (double)((const uint16_t *)img)[idx];
    case KM_I16: return (double)((const int16_t *)img)[idx];
    case KM_F32: return (double)((const float *)img)[idx];
    case KM_F64: return ((const double *)img)[idx];
    case KM_I32: return (double)((const int32_t *)img)[idx];
    default: return (double)((const uint32_t *)img)[idx];
    }
}

__global__ __launch_bounds__(256) void zncc_win_kernel(const void *__restrict__ img1, const void *__restrict__ img2, int dt1, int dt2, int H1, int W1,
                                                       int H2, int W2, ptrdiff_t s1, ptrdiff_t s2, const int *__restrict__ uv /* 4 x count */,
                                                       int half, int count, double *__restrict__ out, uint8_t *__restrict__ flags)
{
    const int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= count) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int u1 = uv[k], v1 = uv[count + k], u2 = uv[2 * count + k], v2 = uv[3 * count + k];
    const bool outside = u1 - half < 0 || u1 + half >= H1 || v1 - half < 0 || v1 + half >= W1 || u2 - half < 0 || u2 + half >= H2 ||
                         v2 - half < 0 || v2 + half >= W2;
    if (outside) { if (lane == 0) { out[k] = nan; if (flags) flags[k] = 1; } return; }
    const int side = 2 * half + 1, npx = side * side;
    double a_sum = 0, b_sum = 0;
    for (int i = lane; i < npx; i += 64) {
        const int r = i / side, cx = i - r * side;
        a_sum += zn_load(img1, dt1, (size_t)(u1 - half + r) * s1 + (v1 - half + cx));
        b_sum += zn_load(img2, dt2, (size_t)(u2 - half + r) * s2 + (v2 - half + cx));
    }
    const double m1 = wave_sum_f64(a_sum) / (double)npx, m2 = wave_sum_f64(b_sum) / (double)npx;
    double q1 = 0, q2 = 0, cc = 0;
    for (int i = lane; i < npx; i += 64) {
        const int r = i / side, cx = i - r * side;
        const double a = zn_load(img1, dt1, (size_t)(u1 - half + r) * s1 + (v1 - half + cx)) - m1;
        const double b = zn_load(img2, dt2, (size_t)(u2 - half + r) * s2 + (v2 - half + cx)) - m2;
        q1 += a * a; q2 += b * b; cc += a * b;
    }
    q1 = wave_sum_f64(q1); q2 = wave_sum_f64(q2); cc = wave_sum_f64(cc);
    const double sd1 = sqrt(q1 / (double)npx), sd2 = sqrt(q2 / (double)npx);
    if (lane == 0) { out[k] = (sd1 == 0.0 || sd2 == 0.0) ? nan : cc / (sd1 * sd2) / (double)npx; if (flags) flags[k] = 0; }
}

int kz_zncc_windows(km_ctx *c, const void *d_img1, const void *d_img2, int dt1, int dt2, int H1, int W1, int H2, int W2, ptrdiff_t s1, ptrdiff_t s2,
                    const int *d_uv, int half, int count, double *d_out, uint8_t *d_flags)
{
    if (count <= 0) return KM_OK;
    zncc_win_kernel<<<(count + 3) / 4, 256, 0, c->stream>>>(d_img1, d_img2, dt1, dt2, H1, W1, H2, W2, s1, s2, d_uv, half, count, d_out, d_flags);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
