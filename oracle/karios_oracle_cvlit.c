/*
 * karios_oracle_cvlit.c -- a SECOND, independent CPU path for the OpenCV-defined arithmetic: "OpenCV-literal" float32.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as karios_oracle.c).
 *
 * karios_oracle.c DEFINES the structure tensor of goodFeaturesToTrack and the normal matrix / mismatch vector of
 * calcOpticalFlowPyrLK through exact integer sums converted once to float32; the HIP kernels follow that definition bit for
 * bit.  OpenCV itself (opencv=4.8.*, reference environment.yml:8, absent here) rounds earlier and more often:
 *
 *   cornerMinEigenVal   Sobel output CV_32F with scale = 1 / (2^(ksize-1) * blockSize * 255) folded into the SMOOTHING taps
 *                       ([1,2,1]*scale as float32, difference taps [-1,0,1] unscaled), float32 products Dx*Dx, Dx*Dy, Dy*Dy,
 *                       boxFilter with double running sums cast to float32, then (a+c) - sqrt((a-c)^2 + b^2) in float32;
 *   LKTrackerInvoker    A11, A12, A22 accumulated as float32 products of the int16 derivatives in four SIMD lanes (pixel x goes
 *                       to lane x mod 4), b1, b2 as int32 products converted to float32 and accumulated in four lanes, the lanes
 *                       summed at the end (SSE2 / universal-intrinsic path of lkpyramid.cpp; the last winSize mod 4 / mod 8
 *                       pixels of a row go to scalar float32 accumulators).
 *
 * Neither formulation is "the" reference bit for bit: OpenCV's result also depends on its build (FMA contraction in the
 * AVX2 dispatch, SIMD width).  `fma` = 1 evaluates the places where an AVX2/FMA3 build would contract with fmaf.
 * tools/investigations/oracle_sensitivity.py runs both formulations on the same images and reports how far the results can move apart:
 * that spread, not either value, is what "parity unpinned" costs.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int r101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do { p = p < 0 ? -p : 2 * len - 2 - p; } while ((unsigned)p >= (unsigned)len);
    return p;
}

static inline float mul_add(float a, float b, float c, int fma) { return fma ? fmaf(a, b, c) : a * b + c; }

/* cornerMinEigenVal(u8, blockSize, ksize = 3, BORDER_DEFAULT) as OpenCV evaluates it in float32. */
int kl_min_eigen_cv(const uint8_t *src, int H, int W, int block, float *eig, int fma)
{
    if (block < 1) return -1;
    const size_t n = (size_t)H * W;
    float *cxx = (float *)malloc(n * sizeof(float)), *cxy = (float *)malloc(n * sizeof(float)), *cyy = (float *)malloc(n * sizeof(float));
    double *rxx = (double *)malloc(n * sizeof(double)), *rxy = (double *)malloc(n * sizeof(double)), *ryy = (double *)malloc(n * sizeof(double));
    if (!cxx || !cxy || !cyy || !rxx || !rxy || !ryy) { free(cxx); free(cxy); free(cyy); free(rxx); free(rxy); free(ryy); return -2; }
    const double scale_d = 1.0 / ((double)(1 << 2) * block * 255.0);
    const float k1 = (float)scale_d, k0 = (float)(2.0 * scale_d);       /* smoothing taps [1,2,1] * scale as float32 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const uint8_t *r0 = src + (size_t)r101(y - 1, H) * W, *r1 = src + (size_t)y * W, *r2 = src + (size_t)r101(y + 1, H) * W;
        for (int x = 0; x < W; x++) {
            const int xm = r101(x - 1, W), xp = r101(x + 1, W);
            /* Dx: rows differenced horizontally (exact small integers as float), columns smoothed with the scaled taps */
            const float d0 = (float)(r0[xp] - r0[xm]), d1 = (float)(r1[xp] - r1[xm]), d2 = (float)(r2[xp] - r2[xm]);
            const float dx = mul_add(d1, k0, (d0 + d2) * k1, fma);
            /* Dy: rows smoothed horizontally with the scaled taps (float32), columns differenced */
            const float s0 = mul_add((float)r0[x], k0, (float)(r0[xm] + r0[xp]) * k1, fma);
            const float s2 = mul_add((float)r2[x], k0, (float)(r2[xm] + r2[xp]) * k1, fma);
            const float dy = s2 - s0;
            const size_t i = (size_t)y * W + x;
            cxx[i] = dx * dx; cxy[i] = dx * dy; cyy[i] = dy * dy;
        }
    }
    const int anchor = block / 2;
    /* boxFilter, normalize = false: RowSum<float, double> (running sum along the row) ... */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        const float *a = cxx + (size_t)y * W, *b = cxy + (size_t)y * W, *c = cyy + (size_t)y * W;
        double sa = 0, sb = 0, sc = 0;
        for (int i = 0; i < block; i++) { const int j = r101(i - anchor, W); sa += a[j]; sb += b[j]; sc += c[j]; }
        for (int x = 0; x < W; x++) {
            const size_t o = (size_t)y * W + x;
            rxx[o] = sa; rxy[o] = sb; ryy[o] = sc;
            const int jn = r101(x + 1 - anchor + block - 1, W), jo = r101(x - anchor, W);
            sa += (double)a[jn] - (double)a[jo]; sb += (double)b[jn] - (double)b[jo]; sc += (double)c[jn] - (double)c[jo];
        }
    }
    /* ... ColumnSum<double, float>: double sums over the rows, cast to float32; then calcMinEigenVal in float32 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double sa = 0, sb = 0, sc = 0;
            for (int i = 0; i < block; i++) {
                const size_t j = (size_t)r101(y - anchor + i, H) * W + x;
                sa += rxx[j]; sb += rxy[j]; sc += ryy[j];
            }
            const float a = (float)sa * 0.5f, b = (float)sb, c = (float)sc * 0.5f;
            const float t = a - c;
            const float q = mul_add(t, t, b * b, fma);
            eig[(size_t)y * W + x] = (a + c) - sqrtf(q);
        }
    free(cxx); free(cxy); free(cyy); free(rxx); free(rxy); free(ryy);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct { const uint8_t *img; int H, W; } lvl_t;

static inline int pxl(const lvl_t *L, int y, int x) { return L->img[(size_t)r101(y, L->H) * L->W + r101(x, L->W)]; }

static inline void scharr_at(const lvl_t *L, int y, int x, int *ix, int *iy)
{
    if ((unsigned)x >= (unsigned)L->W || (unsigned)y >= (unsigned)L->H) { *ix = 0; *iy = 0; return; }
    const int a00 = pxl(L, y - 1, x - 1), a01 = pxl(L, y - 1, x), a02 = pxl(L, y - 1, x + 1);
    const int a10 = pxl(L, y, x - 1), a12 = pxl(L, y, x + 1);
    const int a20 = pxl(L, y + 1, x - 1), a21 = pxl(L, y + 1, x), a22 = pxl(L, y + 1, x + 1);
    *ix = ((a02 + a22) * 3 + a12 * 10) - ((a00 + a20) * 3 + a10 * 10);
    *iy = ((a20 + a22) * 3 + a21 * 10) - ((a00 + a02) * 3 + a01 * 10);
}

#define DSC(x, n) (((x) + (1 << ((n)-1))) >> (n))

static inline void weights(float a, float b, int *w00, int *w01, int *w10, int *w11)
{
    const float oma = 1.f - a, omb = 1.f - b;
    *w00 = (int)lrintf(oma * omb * 16384.f);
    *w01 = (int)lrintf(a * omb * 16384.f);
    *w10 = (int)lrintf(oma * b * 16384.f);
    *w11 = 16384 - *w00 - *w01 - *w10;
}

int ko_pyrdown_u8(const uint8_t *src, int H, int W, uint8_t *dst);   /* karios_oracle.c: integer, identical in both paths */

/* Oscillation stop of LKTrackerInvoker (OpenCV 4.8 lkpyramid.cpp): `std::abs(delta.x + prevDelta.x) < 0.01 && std::abs(delta.y +
 * prevDelta.y) < 0.01` - a float32 sum and magnitude promoted to double against the DOUBLE literal (0.01f = 0.00999999977... passes).
 * Exported for the known-answer test. */
int kl_lk_oscillates(float ddx, float pdx, float ddy, float pdy)
{
    return (double)fabsf(ddx + pdx) < 0.01 && (double)fabsf(ddy + pdy) < 0.01;
}

/* calcOpticalFlowPyrLK with OpenCV's float32 lane accumulation (see the file header); same interface as ko_pyrlk. */
int kl_pyrlk_cv(const uint8_t *prev, const uint8_t *next, int H, int W, const float *pts, int n, int win, int max_level, int max_count,
                double eps, float *out_pts)
{
    if (win <= 2 || max_level < 0) return -1;
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    const double epsilon = eps * eps;
    enum { MAXL = 8 };
    lvl_t P[MAXL + 1], N[MAXL + 1];
    uint8_t *own[2 * (MAXL + 1)];
    int nown = 0, levels = 0;
    if (max_level > MAXL) max_level = MAXL;
    P[0].img = prev; P[0].H = H; P[0].W = W;
    N[0].img = next; N[0].H = H; N[0].W = W;
    for (int l = 0, w = W, h = H; l < max_level; l++) {
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        if (nw <= win || nh <= win) break;
        uint8_t *a = (uint8_t *)malloc((size_t)nw * nh), *b = (uint8_t *)malloc((size_t)nw * nh);
        if (!a || !b) return -2;
        ko_pyrdown_u8(P[l].img, P[l].H, P[l].W, a);
        ko_pyrdown_u8(N[l].img, N[l].H, N[l].W, b);
        own[nown++] = a; own[nown++] = b;
        P[l + 1].img = a; P[l + 1].H = nh; P[l + 1].W = nw;
        N[l + 1].img = b; N[l + 1].H = nh; N[l + 1].W = nw;
        levels = l + 1; w = nw; h = nh;
    }
    const float half = (float)(win - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    for (int i = 0; i < 2 * n; i++) out_pts[i] = pts[i];
#pragma omp parallel
    {
        short *Ibuf = (short *)malloc((size_t)win * win * 3 * sizeof(short));
        short *dIx = Ibuf + (size_t)win * win, *dIy = dIx + (size_t)win * win;
        for (int level = levels; level >= 0; level--) {
            const lvl_t *I = &P[level], *J = &N[level];
#pragma omp for schedule(dynamic, 64)
            for (int p = 0; p < n; p++) {
                const float sc = (float)(1. / (1 << level));
                float prx = pts[2 * p] * sc, pry = pts[2 * p + 1] * sc, nx, ny;
                if (level == levels) { nx = prx; ny = pry; }
                else { nx = out_pts[2 * p] * 2.f; ny = out_pts[2 * p + 1] * 2.f; }
                out_pts[2 * p] = nx; out_pts[2 * p + 1] = ny;
                prx -= half; pry -= half;
                const int ipx = (int)floorf(prx), ipy = (int)floorf(pry);
                if (ipx < -win || ipx >= I->W || ipy < -win || ipy >= I->H) continue;
                float a = prx - (float)ipx, b = pry - (float)ipy;
                int w00, w01, w10, w11;
                weights(a, b, &w00, &w01, &w10, &w11);
                float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0}, sA11 = 0, sA12 = 0, sA22 = 0;
                const int simd4 = win & ~3;                  /* pixels of a row that go through the 4-wide loop */
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int gy = ipy + y, gx = ipx + x;
                        const int ival = DSC(pxl(I, gy, gx) * w00 + pxl(I, gy, gx + 1) * w01 + pxl(I, gy + 1, gx) * w10 + pxl(I, gy + 1, gx + 1) * w11, 14 - 5);
                        int x00, y00, x01, y01, x10, y10, x11, y11;
                        scharr_at(I, gy, gx, &x00, &y00); scharr_at(I, gy, gx + 1, &x01, &y01);
                        scharr_at(I, gy + 1, gx, &x10, &y10); scharr_at(I, gy + 1, gx + 1, &x11, &y11);
                        const int ixv = DSC(x00 * w00 + x01 * w01 + x10 * w10 + x11 * w11, 14);
                        const int iyv = DSC(y00 * w00 + y01 * w01 + y10 * w10 + y11 * w11, 14);
                        Ibuf[y * win + x] = (short)ival; dIx[y * win + x] = (short)ixv; dIy[y * win + x] = (short)iyv;
                        const float fx = (float)ixv, fy = (float)iyv;
                        if (x < simd4) { const int l = x & 3; qA11[l] += fx * fx; qA12[l] += fx * fy; qA22[l] += fy * fy; }
                        else { sA11 += (float)(ixv * ixv); sA12 += (float)(ixv * iyv); sA22 += (float)(iyv * iyv); }
                    }
                float A11 = (sA11 + (qA11[0] + qA11[1] + qA11[2] + qA11[3])) * FLT_SCALE;
                float A12 = (sA12 + (qA12[0] + qA12[1] + qA12[2] + qA12[3])) * FLT_SCALE;
                float A22 = (sA22 + (qA22[0] + qA22[1] + qA22[2] + qA22[3])) * FLT_SCALE;
                float D = A11 * A22 - A12 * A12;
                const float dA = A11 - A22, q = dA * dA + 4.f * A12 * A12;
                const float minEig = (A22 + A11 - sqrtf(q)) / (float)(2 * win * win);
                if (minEig < 1e-4f || D < FLT_EPSILON) continue;
                D = 1.f / D;
                nx -= half; ny -= half;
                float pdx = 0.f, pdy = 0.f;
                const int simd8 = win & ~7;                  /* the mismatch loop is 8 pixels wide */
                for (int j = 0; j < max_count; j++) {
                    const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                    if (inx < -win || inx >= J->W || iny < -win || iny >= J->H) break;
                    a = nx - (float)inx; b = ny - (float)iny;
                    weights(a, b, &w00, &w01, &w10, &w11);
                    /* qb0 = (d0*Ix0, d0*Iy0, d1*Ix1, d1*Iy1), qb1 = the next two pixels; b1 = lanes 0 + 2, b2 = lanes 1 + 3 */
                    float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0}, sb1 = 0, sb2 = 0;
                    for (int y = 0; y < win; y++)
                        for (int x = 0; x < win; x++) {
                            const int gy = iny + y, gx = inx + x;
                            const int diff = DSC(pxl(J, gy, gx) * w00 + pxl(J, gy, gx + 1) * w01 + pxl(J, gy + 1, gx) * w10 + pxl(J, gy + 1, gx + 1) * w11, 14 - 5) -
                                             Ibuf[y * win + x];
                            const float px_ = (float)(diff * dIx[y * win + x]), py_ = (float)(diff * dIy[y * win + x]);
                            if (x < simd8) {
                                float *q4 = (x & 2) ? qb1 : qb0;
                                q4[(x & 1) * 2] += px_; q4[(x & 1) * 2 + 1] += py_;
                            } else { sb1 += px_; sb2 += py_; }
                        }
                    const float l0 = qb0[0] + qb1[0], l1 = qb0[1] + qb1[1], l2 = qb0[2] + qb1[2], l3 = qb0[3] + qb1[3];
                    const float b1 = (sb1 + (l0 + l2)) * FLT_SCALE, b2 = (sb2 + (l1 + l3)) * FLT_SCALE;
                    const float ddx = (A12 * b2 - A22 * b1) * D, ddy = (A12 * b1 - A11 * b2) * D;
                    nx += ddx; ny += ddy;
                    out_pts[2 * p] = nx + half; out_pts[2 * p + 1] = ny + half;
                    if ((double)ddx * ddx + (double)ddy * ddy <= epsilon) break;
                    /* OpenCV: std::abs(delta.x + prevDelta.x) < 0.01 - float32 magnitude against the DOUBLE literal */
                    if (j > 0 && kl_lk_oscillates(ddx, pdx, ddy, pdy)) {
                        out_pts[2 * p] -= ddx * 0.5f; out_pts[2 * p + 1] -= ddy * 0.5f;
                        break;
                    }
                    pdx = ddx; pdy = ddy;
                }
            }
        }
        free(Ibuf);
    }
    for (int i = 0; i < nown; i++) free(own[i]);
    return 0;
}
