// K9: batched per-keypoint ZNCC (ZNCCService._compute_zncc + _zncc2, reference
// matcher/zncc_service.py:45-126, 186-238).  One wavefront per keypoint: the two 43x43 patches
// of the RAW images are read once into registers, mean / population std / correlation are two-pass
// fp64 sums reduced across the wave.
#include "common.hpp"

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
                                                   const float *__restrict__ score, float score_thr, double *__restrict__ out)
{
    const int k = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // key point of this wave: uniform, scalar addressing
    const int lane = threadIdx.x & 63;
    if (k >= (d_n ? min(*d_n, n) : n)) return;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // rows below the confidence threshold are not scored (core.py:878-893): NaN, like the reference's column
    if (score && !(score[k] >= score_thr)) { if (lane == 0) out[k] = nan; return; }
    const float fx0 = x0[k], fy0 = y0[k];
    const int X0 = (int)fx0, Y0 = (int)fy0;  // int(series["x0"])
    // round(np.float32 + np.float32): half-to-even on the f32 sum
    const float sx = __fadd_rn(fx0, dx[k]), sy = __fadd_rn(fy0, dy[k]);
    bool ok = isfinite(sx) && isfinite(sy) && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
    int X1 = 0, Y1 = 0;
    if (ok) {
        X1 = __float2int_rn(sx); Y1 = __float2int_rn(sy);
        ok = !(X0 - ZN_MARGIN < 0 || Y0 - ZN_MARGIN < 0 || X1 - ZN_MARGIN < 0 || Y1 - ZN_MARGIN < 0) &&
             !(X0 >= Wref - ZN_MARGIN || Y0 >= Href - ZN_MARGIN || X1 >= Wmon - ZN_MARGIN || Y1 >= Hmon - ZN_MARGIN);
    }
    if (!ok) { if (lane == 0) out[k] = nan; return; }
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
    const int nb = (n + 3) / 4;
#define KM_Z(T) zncc_kernel<T><<<nb, 256, 0, c->stream>>>((const T *)d_ref, (const T *)d_mon, Href, Wref, Hmon, Wmon, sref, smon, d_x0, d_y0, d_dx, d_dy, n, d_n, d_score, score_thr, d_out)
    switch (dtype) {
    case KM_U8: KM_Z(uint8_t); break;
    case KM_U16: KM_Z(uint16_t); break;
    case KM_I16: KM_Z(int16_t); break;
    case KM_F32: KM_Z(float); break;
    default: return km_fail(c, KM_E_ARG, "zncc: bad dtype %d", dtype);
    }
#undef KM_Z
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
