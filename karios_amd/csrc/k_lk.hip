// K7: pyramidal Lucas-Kanade tracker, forward and backward pass fused
// (cv2.calcOpticalFlowPyrLK x2, reference klt.py:134-140; algorithm SURVEY App. A.3).
//
// One 64-lane wavefront per keypoint.  Per pyramid level the (w+3)^2 neighbourhood of the
// template image is staged in LDS, the Scharr derivative is evaluated there (no dense
// derivative image is ever written to HBM), and every lane keeps its <= NPL window pixels
// (Q5 intensity, Ix, Iy) in registers for all iterations.  The normal matrix and mismatch
// vector are exact integer sums (per-lane int32, wave-reduced in int64) converted once to
// f32, so the result does not depend on the reduction order; all f32 steps of the 2x2
// solve are rounded individually (this file is compiled with -ffp-contract=off).
// The backward pass starts from the forward result of the same keypoint, so both passes run
// back to back in the same wavefront.
#include "common.hpp"

#ifndef KM_LK_WAVES
#define KM_LK_WAVES 0   // occupancy target (waves per SIMD); 0 = leave it to the register allocator
#endif
#if KM_LK_WAVES
#define KM_LK_OCC __attribute__((amdgpu_waves_per_eu(KM_LK_WAVES, KM_LK_WAVES)))
#else
#define KM_LK_OCC
#endif

#include <float.h>
#include <type_traits>

struct lk_args {
    km_pyr A, B;
    const float *pts_in;
    const int *d_n;
    int n_max, win, max_count, backward;
    int general_templates;   // (development build, KARIOS_HIP_LK_GENERAL: every template through the general two-row form - A/B of the degenerate-weight forms)
    double epsilon;
    float *p1, *p0r;
    int *left_band;   // row-band mode: raised when a point's window needs rows outside the resident band
};

// rows [gy0, gy0 + rows) of level `level`, clipped to the image (REFLECT_101 only folds rows back INTO that range), resident?
__device__ __forceinline__ bool lk_rows_resident(const km_pyr &P, int level, int gy0, int rows)
{
    if (P.Hres[level] == 0) return true;
    const int lo = max(gy0, 0), hi = min(gy0 + rows, P.H[level]);
    // a window that overhangs the top / bottom of the image mirrors rows within `rows` of that border
    const int need_lo = gy0 < 0 ? 0 : lo, need_hi = gy0 + rows > P.H[level] ? P.H[level] : hi;
    return need_lo >= P.oy[level] && need_hi <= P.oy[level] + P.Hres[level] && (gy0 >= 0 || P.oy[level] == 0) &&
           (gy0 + rows <= P.H[level] || P.oy[level] + P.Hres[level] == P.H[level]);
}

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    const float oma = 1.f - a, omb = 1.f - b;
    const float t00 = oma * omb, t01 = a * omb, t10 = oma * b;
    w00 = __float2int_rn(t00 * 16384.f);
    w01 = __float2int_rn(t01 * 16384.f);
    w10 = __float2int_rn(t10 * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

#define LK_M 3  // margin (px) of the cached search-image patch around the current window

// Stage a rows x cols byte patch whose top-left pixel is (gx0, gy0) into LDS (row pitch `pitch`, multiple of 4).
// Inside the image: (unaligned) dword loads, ALL issued before the first one is consumed - a loop that loads and stores
// per trip exposes the memory latency once per trip (9 trips per level and pass at winSize 25).  Addresses are a uniform
// base + 32-bit lane offset, so a slot in flight costs one register.  Otherwise byte loads with REFLECT_101 (OpenCV pads
// every pyramid level by winSize with that border).
template <int MAXIT> struct lk_patch_regs {
    uint32_t v[MAXIT];
};

__device__ __forceinline__ bool lk_patch_inside(int IW, int IH, int gx0, int gy0, int rows, int cols, int maxit)
{
    return gx0 >= 0 && gy0 >= 0 && gx0 + cols + 4 <= IW && gy0 + rows <= IH && rows * ((cols + 3) >> 2) <= 64 * maxit && rows * ((cols + 3) >> 2) < 1024 &&
           cols <= 64;
}

template <int MAXIT>
__device__ __forceinline__ void stage_patch_issue(const uint8_t *__restrict__ img, int IW, int gx0, int gy0, int rows, int cols, lk_patch_regs<MAXIT> &rg)
{
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // opaque: keeps the per-slot index arithmetic local (hoisted out of the level loop it costs ~30 VGPRs)
    const int nw = (cols + 3) >> 2, total = rows * nw;
    const unsigned inv_nw = 65536u / (unsigned)nw + 1u;   // i / nw == (i * inv_nw) >> 16 for i < 1024, nw <= 16 (a runtime division costs ~25 instructions)
    const unsigned base = (unsigned)gy0 * (unsigned)IW + (unsigned)gx0;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int i = min(lane + 64 * it, total - 1);     // surplus trips re-read the last word (never stored)
        const int r = (int)(((unsigned)i * inv_nw) >> 16), d = i - r * nw;
        const unsigned off = base + (unsigned)r * (unsigned)IW + 4u * (unsigned)d;
        __builtin_memcpy(&rg.v[it], img + off, 4);
    }
}

template <int MAXIT>
__device__ __forceinline__ void stage_patch_commit(int rows, int cols, uint8_t *lds, int pitch, const lk_patch_regs<MAXIT> &rg)
{
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int nw = (cols + 3) >> 2, total = rows * nw;
    const unsigned inv_nw = 65536u / (unsigned)nw + 1u;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int i = lane + 64 * it;
        if (i < total) {
            const int r = (int)(((unsigned)i * inv_nw) >> 16), d = i - r * nw;
            *(uint32_t *)(lds + r * pitch + 4 * d) = rg.v[it];
        }
    }
}

__device__ __forceinline__ void stage_patch_border(const uint8_t *__restrict__ img, int IW, int IH, int gx0, int gy0, int rows, int cols,
                                                   uint8_t *lds, int pitch)
{
    for (int i = threadIdx.x & 63; i < rows * cols; i += 64) {
        const int r = i / cols, cx = i - r * cols;
        lds[r * pitch + cx] = img[(size_t)km_reflect101(gy0 + r, IH) * IW + km_reflect101(gx0 + cx, IW)];
    }
}

template <int MAXIT>
__device__ __forceinline__ void stage_patch(const uint8_t *__restrict__ img, int IW, int IH, int gx0, int gy0, int rows, int cols,
                                            uint8_t *lds, int pitch)
{
    if (lk_patch_inside(IW, IH, gx0, gy0, rows, cols, MAXIT)) {
        lk_patch_regs<MAXIT> rg;
        stage_patch_issue<MAXIT>(img, IW, gx0, gy0, rows, cols, rg);
        stage_patch_commit<MAXIT>(rows, cols, lds, pitch, rg);
    } else {
        stage_patch_border(img, IW, IH, gx0, gy0, rows, cols, lds, pitch);
    }
}

// exact wave sum of an int64 whose per-lane magnitude is < 2^31: two int32 DPP reductions (low 16 bits / rest)
__device__ __forceinline__ int wave_sum_i32_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ long long wave_sum_split(int v)
{
    const int lo = v & 0xffff, hi = v >> 16;
    return ((long long)wave_sum_i32_dpp(hi) << 16) + (long long)wave_sum_i32_dpp(lo);
}
// the same for a per-lane magnitude < 2^29 (the mismatch sums: <= 10 pixels x 8160 x 4080 = 3.3e8): the first two steps of the scan add
// at most four lanes - still an int32 - so they run once, unsplit; only the remaining four steps need the two 16-bit halves
__device__ __forceinline__ long long wave_sum_split4(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    int lo = v & 0xffff, hi = v >> 16;
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false);
    lo += __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false);
    hi += __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false);
    return ((long long)__builtin_amdgcn_readlane(hi, 63) << 16) + (long long)__builtin_amdgcn_readlane(lo, 63);
}

// Eight patch bytes from a byte-aligned LDS address as three ALIGNED dwords + two v_alignbyte.  A ds_read_b64 whose address is
// not a multiple of 8 takes the LDS unit's unaligned path: ~55 cycles per wave instruction against ~4 aligned (PMC, round 5:
// SQ_LDS_UNALIGNED_STALL = 96 % of SQ_LDS_IDX_ACTIVE, the LDS unit busy 83 % of the kernel - the tracker was LDS-bound on it).
// Reads up to 3 bytes in front of p (same patch: its base is 16-byte aligned) and 4 behind p + 8 (the pitch / tail slack of lk2_geo).
__device__ __forceinline__ uint2 lk_lds_read8(const uint8_t *p, unsigned sh)
{
    const uint32_t *q = (const uint32_t *)(p - sh);
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    uint2 r;
    r.x = __builtin_amdgcn_alignbyte(d1, d0, sh);
    r.y = __builtin_amdgcn_alignbyte(d2, d1, sh);
    return r;
}
__device__ __forceinline__ unsigned lk_lds_shift(const uint8_t *p) { return (unsigned)(size_t)p & 3u; }

// Window pixels are dealt to the lanes as horizontal RUNS of LK_RUN pixels (run r = t * 64 + lane covers columns
// [x0, x0 + n) of window row y).  A run needs LK_RUN + 1 consecutive bytes from each of two rows of the search patch:
// two (unaligned) 8-byte LDS reads per run and iteration instead of four byte reads per pixel, the byte pairs are cut out
// with v_perm and the bilinear interpolation is two v_dot2_i32_i16 per pixel; the mismatch vector is accumulated with
// v_dot2_i32_i16 over pixel pairs as well.  All arithmetic is the exact integer arithmetic of cv::calcOpticalFlowPyrLK.
#define LK_RUN 5
#define LK_WPB 1   // key points (= wavefronts) per workgroup (4 measured slower: 0.363 vs 0.328 ms - a workgroup only retires with its slowest wave)
// every wavefront works on its own LDS region: only the compiler must be kept from reordering LDS traffic across the phases
#define LK_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
typedef short lk_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lk_s2 lk_as_s2(uint32_t v) { return __builtin_bit_cast(lk_s2, v); }
__device__ __forceinline__ uint32_t lk_as_u(lk_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t lk_pack16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

// Track one point from image pyramid I to J (all lanes hold identical scalars).
// run_desc[t] = y | x0 << 8 | n << 16 (n = 0: the lane has no t-th run).
// OpenCV's oscillation stop (lkpyramid.cpp, LKTrackerInvoker): `std::abs(delta.x + prevDelta.x) < 0.01 && std::abs(delta.y +
// prevDelta.y) < 0.01` - the float32 sum, its float32 magnitude, promoted to double against the DOUBLE literal 0.01.  0.01f is
// 0.00999999977...: a sum of exactly 0.01f passes OpenCV's test and would fail `< 0.01f` (tests/test_gpu_parity.py::test_lk_oscillation_literal).
__device__ __host__ __forceinline__ bool lk_oscillates(float ddx, float pdx, float ddy, float pdy)
{
    return (double)fabsf(ddx + pdx) < 0.01 && (double)fabsf(ddy + pdy) < 0.01;
}

template <int NR>
__device__ void lk_track_point(const km_pyr &I, const km_pyr &J, float px, float py, int win, int max_count, double epsilon,
                               const int (&run_desc)[NR], uint8_t *raw, short *derx, uint8_t *jp, float &outx, float &outy, bool &left_band)
{
    const int lane = threadIdx.x & 63;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int RW = win + 3, DW = win + 1;
    short *dery = derx + DW * DW + 8;                 // second derivative plane (8 shorts of slack behind each plane)
    const int RP = (RW + 3 + 3) & ~3;                 // raw patch pitch
    const int JS = win + 1 + 2 * LK_M, JP = (JS + 3 + 3) & ~3;
    float resx = px, resy = py;
    for (int level = I.levels; level >= 0; level--) {
        const uint8_t *Iimg = I.img[level], *Jimg = J.img[level];
        const int IW = I.W[level], IH = I.H[level], JW = J.W[level], JH = J.H[level];
        const float sc = (float)(1. / (1 << level));
        float prx = px * sc, pry = py * sc;
        float nx, ny;
        if (level == I.levels) { nx = prx; ny = pry; }
        else { nx = resx * 2.f; ny = resy * 2.f; }
        resx = nx; resy = ny;
        prx -= half; pry -= half;
        const int ipx = (int)floorf(prx), ipy = (int)floorf(pry);
        if (ipx < -win || ipx >= IW || ipy < -win || ipy >= IH) continue;
        float a = prx - (float)ipx, b = pry - (float)ipy;
        int w00, w01, w10, w11;
        lk_weights(a, b, w00, w01, w10, w11);

        // stage the template neighbourhood and (speculatively) the search neighbourhood around the start position
        LK_WAVE_SYNC();
        int jx0 = (int)floorf(nx - half) - LK_M, jy0 = (int)floorf(ny - half) - LK_M;
        if (!lk_rows_resident(I, level, ipy - 1, RW) || !lk_rows_resident(J, level, jy0, JS)) { left_band = true; break; }
        {
            constexpr int MAXIT = 2 * NR;                 // covers (win + 7)^2 / 4 words for every winSize served by NR runs per lane
            const bool in_i = lk_patch_inside(IW, IH, ipx - 1, ipy - 1, RW, RW, MAXIT), in_j = lk_patch_inside(JW, JH, jx0, jy0, JS, JS, MAXIT);
            if (in_i && in_j) {                           // template and search patch travel together
                lk_patch_regs<MAXIT> ri, rj;
                stage_patch_issue<MAXIT>(Iimg, IW, ipx - 1, ipy - 1, RW, RW, ri);
                stage_patch_issue<MAXIT>(Jimg, JW, jx0, jy0, JS, JS, rj);
                stage_patch_commit<MAXIT>(RW, RW, raw, RP, ri);
                stage_patch_commit<MAXIT>(JS, JS, jp, JP, rj);
            } else {
                stage_patch<MAXIT>(Iimg, IW, IH, ipx - 1, ipy - 1, RW, RW, raw, RP);
                stage_patch<MAXIT>(Jimg, JW, JH, jx0, jy0, JS, JS, jp, JP);
            }
        }
        LK_WAVE_SYNC();
        // Scharr derivative on the (w+1)^2 bilinear support; zero outside the image.  Four adjacent positions per lane
        // and step share their 3x6 neighbourhood (column sums s = 3*(a0+a2)+10*a1 and d = a2-a0 per column).
        {
            const int ngrp = (DW + 3) >> 2;
            const unsigned inv_ngrp = 65536u / (unsigned)ngrp + 1u;   // exact quotient for i < 1024, ngrp <= 16
            for (int i = lane; i < DW * ngrp; i += 64) {
                const int r = (int)(((unsigned)i * inv_ngrp) >> 16), g = i - r * ngrp;
                const int cx0 = 4 * g;
                const uint8_t *p = raw + (r + 1) * RP + cx0;      // column cx0-1 of the centre row is p[0]
                int sv[6], dv[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int a0 = p[k - RP], a1 = p[k], a2 = p[k + RP];
                    sv[k] = (a0 + a2) * 3 + a1 * 10;   // vertical smoothing  [3 10 3]
                    dv[k] = a2 - a0;                   // vertical difference [-1 0 1]
                }
                const int gy = ipy + r;
                const bool row_in = (unsigned)gy < (unsigned)IH;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int cx = cx0 + k;
                    if (cx < DW) {
                        int ix = 0, iy = 0;
                        if (row_in && (unsigned)(ipx + cx) < (unsigned)IW) {
                            ix = sv[k + 2] - sv[k];
                            iy = (dv[k] + dv[k + 2]) * 3 + dv[k + 1] * 10;
                        }
                        derx[r * DW + cx] = (short)ix;      // two 16-bit planes: a run reads its 6 consecutive values per
                        dery[r * DW + cx] = (short)iy;      // row with one (unaligned) 16-byte LDS read
                    }
                }
            }
        }
        LK_WAVE_SYNC();
        // per-lane window pixels -> registers (16-bit pairs along the run: Q5 intensity, Ix, Iy); exact integer normal matrix
        uint32_t IvP[NR][3], IxP[NR][3], IyP[NR][3];
        int sA11 = 0, sA12 = 0, sA22 = 0;  // per-lane partial sums stay below 2^31 (<= 25 pixels per lane: 25 * 4080^2 = 4.2e8)
        {
            const lk_s2 wt0 = lk_as_s2(lk_pack16(w00, w01)), wt1 = lk_as_s2(lk_pack16(w10, w11));   // signed: w11 may be -1
#pragma unroll
            for (int t = 0; t < NR; t++) {
                int rd = run_desc[t];
                asm volatile("" : "+v"(rd));   // opaque: the LDS addresses of the run are recomputed here instead of living in registers across the kernel
                const int y = rd & 0xff, x0 = (rd >> 8) & 0xff, n = rd >> 16;
                // template intensity (Q5): bytes x0+1 .. x0+6 of raw rows y+1, y+2
                const uint8_t *pr = raw + (y + 1) * RP + (x0 + 1);
                uint2 r0, r1;
                __builtin_memcpy(&r0, pr, 8);
                __builtin_memcpy(&r1, pr + RP, 8);
                // derivatives: values x0 .. x0+5 of rows y, y+1 of both planes
                uint4 gx0, gx1, gy0, gy1;
                __builtin_memcpy(&gx0, derx + y * DW + x0, 16);
                __builtin_memcpy(&gx1, derx + (y + 1) * DW + x0, 16);
                __builtin_memcpy(&gy0, dery + y * DW + x0, 16);
                __builtin_memcpy(&gy1, dery + (y + 1) * DW + x0, 16);
                const uint32_t ax0[4] = {gx0.x, gx0.y, gx0.z, gx0.w}, ax1[4] = {gx1.x, gx1.y, gx1.z, gx1.w};
                const uint32_t ay0[4] = {gy0.x, gy0.y, gy0.z, gy0.w}, ay1[4] = {gy1.x, gy1.y, gy1.z, gy1.w};
                int iv[LK_RUN + 1], ixv[LK_RUN + 1], iyv[LK_RUN + 1];
                iv[LK_RUN] = 0; ixv[LK_RUN] = 0; iyv[LK_RUN] = 0;
#pragma unroll
                for (int k = 0; k < LK_RUN; k++) {
                    const uint32_t sel = 0x0c000c00u | (uint32_t)k | ((uint32_t)(k + 1) << 16);
                    const lk_s2 c0 = lk_as_s2(__builtin_amdgcn_perm(r0.y, r0.x, sel)), c1 = lk_as_s2(__builtin_amdgcn_perm(r1.y, r1.x, sel));
                    iv[k] = __builtin_amdgcn_sdot2(c0, wt0, __builtin_amdgcn_sdot2(c1, wt1, 1 << (14 - 5 - 1), false), false) >> (14 - 5);
                    // 16-bit pair (value k, value k+1) of a row held as 4 dwords
                    auto pair = [&](const uint32_t (&a)[4]) -> lk_s2 {
                        return lk_as_s2((k & 1) ? __builtin_amdgcn_alignbyte(a[(k + 1) / 2], a[k / 2], 2) : a[k / 2]);
                    };
                    ixv[k] = __builtin_amdgcn_sdot2(pair(ax0), wt0, __builtin_amdgcn_sdot2(pair(ax1), wt1, 1 << 13, false), false) >> 14;
                    iyv[k] = __builtin_amdgcn_sdot2(pair(ay0), wt0, __builtin_amdgcn_sdot2(pair(ay1), wt1, 1 << 13, false), false) >> 14;
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    // pixels beyond the run's length (window edge, surplus lanes) contribute nothing
                    const uint32_t pm = n >= 2 * q + 2 ? 0xffffffffu : (n == 2 * q + 1 ? 0x0000ffffu : 0u);
                    IvP[t][q] = lk_pack16(iv[2 * q], iv[2 * q + 1]) & pm;
                    IxP[t][q] = lk_pack16(ixv[2 * q], ixv[2 * q + 1]) & pm;
                    IyP[t][q] = lk_pack16(iyv[2 * q], iyv[2 * q + 1]) & pm;
                    sA11 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IxP[t][q]), sA11, false);
                    sA12 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IyP[t][q]), sA12, false);
                    sA22 = __builtin_amdgcn_sdot2(lk_as_s2(IyP[t][q]), lk_as_s2(IyP[t][q]), sA22, false);
                }
            }
        }
        const long long iA11 = wave_sum_split(sA11), iA12 = wave_sum_split(sA12), iA22 = wave_sum_split(sA22);
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dA = A11 - A22;
        const float q = dA * dA + 4.f * A12 * A12;
        const float minEig = (A22 + A11 - sqrtf(q)) / (float)(2 * win * win);
        if (minEig < 1e-4f || D < FLT_EPSILON) continue;
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= JW || iny < -win || iny >= JH) break;
            if (inx < jx0 || iny < jy0 || inx > jx0 + 2 * LK_M || iny > jy0 + 2 * LK_M) {
                // the window left the cached neighbourhood: re-centre it
                LK_WAVE_SYNC();
                jx0 = inx - LK_M; jy0 = iny - LK_M;
                if (!lk_rows_resident(J, level, jy0, JS)) { left_band = true; break; }
                stage_patch<2 * NR>(Jimg, JW, JH, jx0, jy0, JS, JS, jp, JP);
                LK_WAVE_SYNC();
            }
            a = nx - (float)inx; b = ny - (float)iny;
            lk_weights(a, b, w00, w01, w10, w11);
            const lk_s2 wr0 = lk_as_s2(lk_pack16(w00, w01)), wr1 = lk_as_s2(lk_pack16(w10, w11));   // signed: w11 may be -1
            const uint8_t *jb = jp + (iny - jy0) * JP + (inx - jx0);
            int sb1 = 0, sb2 = 0;
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int y = run_desc[t] & 0xff, x0 = (run_desc[t] >> 8) & 0xff;
                const uint8_t *p0 = jb + y * JP + x0;
                const unsigned jsh = lk_lds_shift(p0);
                const uint2 r0 = lk_lds_read8(p0, jsh), r1 = lk_lds_read8(p0 + JP, jsh);   // bytes x0 .. x0+7 of the two patch rows
                int val[LK_RUN + 1];
                val[LK_RUN] = 0;
#pragma unroll
                for (int k = 0; k < LK_RUN; k++) {
                    const uint32_t sel = 0x0c000c00u | (uint32_t)k | ((uint32_t)(k + 1) << 16);   // (byte k, byte k+1) as 16-bit values
                    const lk_s2 c0 = lk_as_s2(__builtin_amdgcn_perm(r0.y, r0.x, sel)), c1 = lk_as_s2(__builtin_amdgcn_perm(r1.y, r1.x, sel));
                    val[k] = __builtin_amdgcn_sdot2(c0, wr0, __builtin_amdgcn_sdot2(c1, wr1, 1 << (14 - 5 - 1), false), false) >> (14 - 5);
                }
#pragma unroll
                for (int q2 = 0; q2 < 3; q2++) {
                    const lk_s2 diff = lk_as_s2(lk_pack16(val[2 * q2], val[2 * q2 + 1])) - lk_as_s2(IvP[t][q2]);
                    sb1 = __builtin_amdgcn_sdot2(diff, lk_as_s2(IxP[t][q2]), sb1, false);
                    sb2 = __builtin_amdgcn_sdot2(diff, lk_as_s2(IyP[t][q2]), sb2, false);
                }
            }
            const long long ib1 = wave_sum_split(sb1), ib2 = wave_sum_split(sb2);
            const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
            const float ddx = (A12 * b2 - A22 * b1) * D;
            const float ddy = (A12 * b1 - A11 * b2) * D;
            nx += ddx; ny += ddy;
            resx = nx + half; resy = ny + half;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= epsilon) break;
            if (j > 0 && lk_oscillates(ddx, pdx, ddy, pdy)) {
                resx -= ddx * 0.5f; resy -= ddy * 0.5f;
                break;
            }
            pdx = ddx; pdy = ddy;
        }
        if (left_band) break;
    }
    outx = resx; outy = resy;
}

template <int NR>
__global__ __launch_bounds__(64 * LK_WPB) KM_LK_OCC void lk_kernel(lk_args g, int sm_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * LK_WPB + wv;
    const int n = g.d_n ? min(*g.d_n, g.n_max) : g.n_max;
    if (p >= n) return;                                  // (no workgroup barrier anywhere: waves are independent)
    unsigned char *smem = smem_all + (size_t)wv * sm_per_wave;
    const int win = g.win;
    const int RP = (win + 3 + 3 + 3) & ~3, JS = win + 1 + 2 * LK_M, JP = (JS + 3 + 3) & ~3;
    uint8_t *raw = smem;
    uint8_t *jp = smem + (((win + 3) * RP + 15) & ~15);
    short *derx = (short *)(jp + ((JS * JP + 15) & ~15));   // two int16 planes of (win+1)^2 values, 8 shorts of slack each
    const int rpr = (win + LK_RUN - 1) / LK_RUN, total = win * rpr;
    int run_desc[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) {
        const int r = t * 64 + (int)(threadIdx.x & 63);
        const int y = r / rpr, x0 = (r - y * rpr) * LK_RUN;
        run_desc[t] = r < total ? (y | (x0 << 8) | (min(LK_RUN, win - x0) << 16)) : 0;
    }
    const float px = g.pts_in[2 * p], py = g.pts_in[2 * p + 1];
    float fx, fy;
    bool left_band = false;
    lk_track_point<NR>(g.A, g.B, px, py, win, g.max_count, g.epsilon, run_desc, raw, derx, jp, fx, fy, left_band);
    if ((threadIdx.x & 63) == 0) { g.p1[2 * p] = fx; g.p1[2 * p + 1] = fy; }
    if (g.backward) {
        float rx, ry;
        lk_track_point<NR>(g.B, g.A, fx, fy, win, g.max_count, g.epsilon, run_desc, raw, derx, jp, rx, ry, left_band);
        if ((threadIdx.x & 63) == 0) { g.p0r[2 * p] = rx; g.p0r[2 * p + 1] = ry; }
    }
    if (left_band && g.left_band && (threadIdx.x & 63) == 0) atomicOr(g.left_band, 1);
}


// ================================================================================================================
// K7, second form (two-level pyramids = the reference's maxLevel 1, klt.py:128-140; whole levels resident).
//
// What the first form pays per key point besides the window arithmetic: FOUR round trips to HBM (template + search
// neighbourhood per level and direction - the backward pass stages exactly the neighbourhoods the forward pass just had,
// with the roles of the two images swapped), a Scharr pass that writes two int16 planes to LDS through byte reads
// (two thirds of the kernel's LDS instructions) and reads them back, and index arithmetic on run-time sizes.
// Here every wavefront stages ONE (win + 7)^2 neighbourhood per image and level around its key point - four patches, all
// loads of the key point in flight together - and both directions work on them: the forward search patch of J is the
// backward template patch, the forward template patch of I is the backward search patch.  A lane evaluates the Scharr
// derivative of its own runs straight from the patch bytes in packed 16-bit arithmetic (4 unaligned 8-byte LDS reads per
// run, no derivative planes, no LDS write / barrier / read-back), so a patch is never modified and stays valid for the
// other direction.  A patch is re-centred (re-read from HBM) only when a window leaves its margin (displacements > 2 px
// per level; the reference's data is pre-aligned to less).  The arithmetic is the first form's, bit for bit.
#define LK2_M 3                       // margin (px) of a patch around the window it was centred on
#define LK2_INVALID (-(1 << 28))      // origin of a patch that holds nothing yet

template <int WIN> struct lk2_geo {
    int win;
    __device__ __host__ explicit lk2_geo(int w) : win(WIN ? WIN : w) {}
    __device__ __host__ int ps() const { return (WIN ? WIN : win) + 1 + 2 * LK2_M; }          // rows = columns of a patch
    __device__ __host__ int pp() const { return (ps() + 4 + 3) & ~3; }                       // row pitch (8-byte run reads end <= 4 bytes behind a row)
    __device__ __host__ int bytes() const { return (ps() * pp() + 16 + 15) & ~15; }          // + slack behind the last row
};

typedef unsigned short lk_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lk_us2 lk_as_us2(uint32_t v) { return __builtin_bit_cast(lk_us2, v); }
__device__ __forceinline__ uint32_t lk_us_as_u(lk_us2 v) { return __builtin_bit_cast(uint32_t, v); }

// (re)load the patch of image level (img, IW, IH) with origin (ox, oy); caller brackets with LK_WAVE_SYNC
template <int MAXIT, int WIN>
__device__ __forceinline__ void lk2_restage(const uint8_t *__restrict__ img, int IW, int IH, int ox, int oy, const lk2_geo<WIN> &geo, uint8_t *patch)
{
    stage_patch<MAXIT>(img, IW, IH, ox, oy, geo.ps(), geo.ps(), patch, geo.pp());
}

// One direction of the tracker on the resident patches.  X = template image (patches px[0..1] = level 0, 1), Y = search
// image.  ox/oy: patch origins per (image slot, level), updated when a patch is re-centred.
// a * b (two int16 products) + k with the addend in a register that stays live: VOP3P form, no copy of the constant.  The result
// is only ever the accumulator (src C) of the next dot of the same family, which the hardware forwards (tools/hazard_scan.py
// checks the generated code: the compiler cannot see a DOT behind inline assembly).
__device__ __forceinline__ int lk_dot2_k(lk_s2 a, lk_s2 b, int k)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(k));
    return d;
}

template <int NR, int WIN, int MAXIT>
__device__ void lk2_track_point(const km_pyr &I, const km_pyr &J, uint8_t *pI, uint8_t *pJ, int (&oI)[2][2], int (&oJ)[2][2], float px, float py,
                                const lk2_geo<WIN> &geo, int max_count, double epsilon, const int (&run_desc)[NR], float &outx, float &outy, int general_templates)
{
    const int win = geo.win;
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int PP = geo.pp(), PB = geo.bytes();
    float resx = px, resy = py;
    // rounding constants live in registers: the three-operand v_dot2_i32_i16 (lk_dot2_k) takes them as its addend, where the
    // compiler's accumulating v_dot2c needs a v_mov of the literal in front of every interpolated value
    int k_half9 = 1 << (14 - 5 - 1), k_half14 = 1 << 13;
    asm volatile("" : "+v"(k_half9), "+v"(k_half14));
#pragma unroll 1
    for (int level = 1; level >= 0; level--) {
        const int IW = I.W[level], IH = I.H[level], JW = J.W[level], JH = J.H[level];
        const float sc = level ? 0.5f : 1.f;
        float prx = px * sc, pry = py * sc;
        float nx, ny;
        if (level == 1) { nx = prx; ny = pry; }
        else { nx = resx * 2.f; ny = resy * 2.f; }
        resx = nx; resy = ny;
        prx -= half; pry -= half;
        const int ipx = (int)floorf(prx), ipy = (int)floorf(pry);
        if (ipx < -win || ipx >= IW || ipy < -win || ipy >= IH) continue;
        float a = prx - (float)ipx, b = pry - (float)ipy;
        int w00, w01, w10, w11;
        lk_weights(a, b, w00, w01, w10, w11);
        uint8_t *X = pI + level * PB, *Y = pJ + level * PB;
        // the template needs patch rows / columns [t - 1, t + win + 1] with t = window origin - patch origin: 1 <= t <= 2 M - 1
        int tx = ipx - oI[level][0], ty = ipy - oI[level][1];
        if (tx < 1 || tx > 2 * LK2_M - 1 || ty < 1 || ty > 2 * LK2_M - 1) {
            LK_WAVE_SYNC();
            oI[level][0] = ipx - LK2_M; oI[level][1] = ipy - LK2_M;
            lk2_restage<MAXIT>(I.img[level], IW, IH, oI[level][0], oI[level][1], geo, X);
            LK_WAVE_SYNC();
            tx = ty = LK2_M;
        }
        // derivatives are zero outside the image (OpenCV pads the derivative image with zeros, the intensities with REFLECT_101)
        const bool need_mask = !(ipx >= 0 && ipy >= 0 && ipx + win <= IW - 1 && ipy + win <= IH - 1);
        uint32_t IvP[NR][3], IxP[NR][3], IyP[NR][3];
        int sA11 = 0, sA12 = 0, sA22 = 0;
        // The template's share of the mismatch vector does not change over the iterations: b = sum (J - I) (Ix, Iy) = sum J (Ix, Iy) - cI,
        // cI = sum I (Ix, Iy) per lane, once per pass (round 6: three packed subtractions per run and iteration less; exact integers either way)
        int cI1 = 0, cI2 = 0;
        // The bilinear weights of the TEMPLATE are often degenerate: a key point of the forward pass is an integer position (a corner
        // of goodFeaturesToTrack) and winSize is odd, so at level 0 the weights are exactly (16384, 0, 0, 0) and at level 1 multiples of
        // 4096 - I = 32 p, Ix / Iy = the Scharr values themselves, CV_DESCALE changes nothing.  MODE 2 (w01 = w10 =
        // w11 = 0): no interpolation at all, one bilinear row; MODE 1 (w10 = w11 = 0): one bilinear row, horizontal interpolation only;
        // MODE 0: the general two-row form (the backward pass, whose start positions are the forward pass's sub-pixel results).  The
        // values are the general form's bit for bit: a zero weight contributes a zero to an exact integer sum.
        auto build_templates = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            constexpr int ROWS = MODE == 0 ? 4 : 3, ND = MODE == 0 ? 2 : 1;
            const lk_s2 wt0 = lk_as_s2(lk_pack16(w00, w01)), wt1 = lk_as_s2(lk_pack16(w10, w11));   // signed: w11 may be -1
            const uint8_t *xb = X + (ty - 1) * PP + (tx - 1);
#pragma unroll
            for (int t = 0; t < NR; t++) {
                int rd = run_desc[t];
                asm volatile("" : "+v"(rd));
                const int y = rd & 0xff, x0 = (rd >> 8) & 0xff, n = rd >> 16;
                // patch rows y-1 .. y+2 (window coordinates), bytes x0-1 .. x0+6
                const uint8_t *pr = xb + y * PP + x0;
                uint2 R[4];
                const unsigned rsh = lk_lds_shift(pr);          // (the pitch is a multiple of 4: one shift for all rows)
#pragma unroll
                for (int k = 0; k < ROWS; k++) R[k] = lk_lds_read8(pr + k * PP, rsh);
                if constexpr (ROWS == 3) R[3] = R[2];
                // columns as 16-bit pairs (c0,c1) (c2,c3) (c4,c5) (c6,c7)
                lk_us2 E[4][4];
#pragma unroll
                for (int k = 0; k < ROWS; k++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        E[k][j] = lk_as_us2(__builtin_amdgcn_perm(R[k].y, R[k].x, 0x0c000c00u | (uint32_t)(2 * j) | ((uint32_t)(2 * j + 1) << 16)));
                uint32_t gx[2][3], gy[2][3];      // derivative pairs of the two bilinear rows: positions (0,1) (2,3) (4,5)
#pragma unroll
                for (int d = 0; d < ND; d++) {
                    lk_us2 S[4], V[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        S[j] = (E[d][j] + E[d + 2][j]) * (unsigned short)3 + E[d + 1][j] * (unsigned short)10;   // [3 10 3] down the column
                        V[j] = E[d + 2][j] - E[d][j];                                                               // [-1 0 1] down the column
                    }
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        gx[d][q] = lk_us_as_u(S[q + 1] - S[q]);                                                     // s(c+1) - s(c-1)
                        const lk_us2 vo = lk_as_us2(__builtin_amdgcn_alignbyte(lk_us_as_u(V[q + 1]), lk_us_as_u(V[q]), 2));   // (v(c), v(c+1)) of the odd columns
                        gy[d][q] = lk_us_as_u((V[q] + V[q + 1]) * (unsigned short)3 + vo * (unsigned short)10);     // 3 (v(c-1) + v(c+1)) + 10 v(c)
                    }
                }
                if (need_mask) {
                    const int gy0 = ipy + y, gx0 = ipx + x0;
#pragma unroll
                    for (int d = 0; d < ND; d++) {
                        const uint32_t mrow = (unsigned)(gy0 + d) < (unsigned)IH ? 0xffffffffu : 0u;
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            const uint32_t m = (((unsigned)(gx0 + 2 * q) < (unsigned)IW ? 0x0000ffffu : 0u) |
                                                ((unsigned)(gx0 + 2 * q + 1) < (unsigned)IW ? 0xffff0000u : 0u)) & mrow;
                            gx[d][q] &= m; gy[d][q] &= m;
                        }
                    }
                }
                if constexpr (MODE == 2) {
                    // weights (16384, 0, 0, 0): I = 32 p (bytes 1 .. 6 of patch row y), Ix / Iy = the Scharr pairs as they are
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        const uint32_t pm = n >= 2 * q + 2 ? 0xffffffffu : (n == 2 * q + 1 ? 0x0000ffffu : 0u);
                        const uint32_t pp2 = __builtin_amdgcn_perm(R[1].y, R[1].x, 0x0c000c00u | (uint32_t)(2 * q + 1) | ((uint32_t)(2 * q + 2) << 16));
                        IvP[t][q] = (pp2 << 5) & pm;                    // (two values <= 255: the shift does not cross the halves)
                        IxP[t][q] = gx[0][q] & pm;
                        IyP[t][q] = gy[0][q] & pm;
                        sA11 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IxP[t][q]), sA11, false);
                        sA12 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IyP[t][q]), sA12, false);
                        sA22 = __builtin_amdgcn_sdot2(lk_as_s2(IyP[t][q]), lk_as_s2(IyP[t][q]), sA22, false);
                        cI1 = __builtin_amdgcn_sdot2(lk_as_s2(IvP[t][q]), lk_as_s2(IxP[t][q]), cI1, false);
                        cI2 = __builtin_amdgcn_sdot2(lk_as_s2(IvP[t][q]), lk_as_s2(IyP[t][q]), cI2, false);
                    }
                } else {
                int iv[LK_RUN + 1], ixv[LK_RUN + 1], iyv[LK_RUN + 1];
                iv[LK_RUN] = 0; ixv[LK_RUN] = 0; iyv[LK_RUN] = 0;
#pragma unroll
                for (int k = 0; k < LK_RUN; k++) {
                    // intensities: bytes k+1, k+2 of patch rows y, y+1
                    const uint32_t sel = 0x0c000c00u | (uint32_t)(k + 1) | ((uint32_t)(k + 2) << 16);
                    const lk_s2 c0 = lk_as_s2(__builtin_amdgcn_perm(R[1].y, R[1].x, sel));
                    auto pair = [&](const uint32_t (&g)[3]) -> lk_s2 {
                        return lk_as_s2((k & 1) ? __builtin_amdgcn_alignbyte(g[(k + 1) / 2], g[k / 2], 2) : g[k / 2]);
                    };
                    if constexpr (MODE == 1) {
                        // (the compiler's own dot: its result is read by a shift, and a DOT result behind inline assembly gets no hazard padding)
                        iv[k] = __builtin_amdgcn_sdot2(c0, wt0, k_half9, false) >> (14 - 5);
                        ixv[k] = __builtin_amdgcn_sdot2(pair(gx[0]), wt0, k_half14, false) >> 14;
                        iyv[k] = __builtin_amdgcn_sdot2(pair(gy[0]), wt0, k_half14, false) >> 14;
                    } else {
                        const lk_s2 c1 = lk_as_s2(__builtin_amdgcn_perm(R[2].y, R[2].x, sel));
                        iv[k] = __builtin_amdgcn_sdot2(c0, wt0, lk_dot2_k(c1, wt1, k_half9), false) >> (14 - 5);
                        ixv[k] = __builtin_amdgcn_sdot2(pair(gx[0]), wt0, lk_dot2_k(pair(gx[1]), wt1, k_half14), false) >> 14;
                        iyv[k] = __builtin_amdgcn_sdot2(pair(gy[0]), wt0, lk_dot2_k(pair(gy[1]), wt1, k_half14), false) >> 14;
                    }
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    const uint32_t pm = n >= 2 * q + 2 ? 0xffffffffu : (n == 2 * q + 1 ? 0x0000ffffu : 0u);
                    IvP[t][q] = lk_pack16(iv[2 * q], iv[2 * q + 1]) & pm;
                    IxP[t][q] = lk_pack16(ixv[2 * q], ixv[2 * q + 1]) & pm;
                    IyP[t][q] = lk_pack16(iyv[2 * q], iyv[2 * q + 1]) & pm;
                    sA11 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IxP[t][q]), sA11, false);
                    sA12 = __builtin_amdgcn_sdot2(lk_as_s2(IxP[t][q]), lk_as_s2(IyP[t][q]), sA12, false);
                    sA22 = __builtin_amdgcn_sdot2(lk_as_s2(IyP[t][q]), lk_as_s2(IyP[t][q]), sA22, false);
                    cI1 = __builtin_amdgcn_sdot2(lk_as_s2(IvP[t][q]), lk_as_s2(IxP[t][q]), cI1, false);
                    cI2 = __builtin_amdgcn_sdot2(lk_as_s2(IvP[t][q]), lk_as_s2(IyP[t][q]), cI2, false);
                }
                }
            }
        };
        {
            // (wave-uniform: every lane holds the key point's scalars; readfirstlane makes the branch a scalar one)
            const int z01 = __builtin_amdgcn_readfirstlane(w01), z1 = __builtin_amdgcn_readfirstlane(w10 | w11) | general_templates;
            if (z1 == 0 && z01 == 0) build_templates(std::integral_constant<int, 2>{});
            else if (z1 == 0) build_templates(std::integral_constant<int, 1>{});
            else build_templates(std::integral_constant<int, 0>{});
        }
        const long long iA11 = wave_sum_split(sA11), iA12 = wave_sum_split(sA12), iA22 = wave_sum_split(sA22);
        const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dA = A11 - A22;
        const float q = dA * dA + 4.f * A12 * A12;
        const float minEig = (A22 + A11 - sqrtf(q)) / (float)(2 * win * win);
        if (minEig < 1e-4f || D < FLT_EPSILON) continue;
        D = 1.f / D;
        nx -= half; ny -= half;
        float pdx = 0.f, pdy = 0.f;
        int jx0 = oJ[level][0], jy0 = oJ[level][1];
        for (int j = 0; j < max_count; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= JW || iny < -win || iny >= JH) break;
            if (inx < jx0 || iny < jy0 || inx > jx0 + 2 * LK2_M || iny > jy0 + 2 * LK2_M) {
                // the window left the cached neighbourhood: re-centre it
                LK_WAVE_SYNC();
                jx0 = inx - LK2_M; jy0 = iny - LK2_M;
                lk2_restage<MAXIT>(J.img[level], JW, JH, jx0, jy0, geo, Y);
                LK_WAVE_SYNC();
            }
            a = nx - (float)inx; b = ny - (float)iny;
            lk_weights(a, b, w00, w01, w10, w11);
            const lk_s2 wc0 = lk_as_s2(lk_pack16(w00, w10)), wc1 = lk_as_s2(lk_pack16(w01, w11));   // (column k, column k + 1); signed: w11 may be -1
            const uint8_t *jb = Y + (iny - jy0) * PP + (inx - jx0);
            int sb1 = -cI1, sb2 = -cI2;
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int y = run_desc[t] & 0xff, x0 = (run_desc[t] >> 8) & 0xff;
                const uint8_t *p0 = jb + y * PP + x0;
                const unsigned jsh = lk_lds_shift(p0);
                const uint2 r0 = lk_lds_read8(p0, jsh), r1 = lk_lds_read8(p0 + PP, jsh);   // bytes x0 .. x0+7 of the two patch rows
                int val[LK_RUN + 1];
                val[LK_RUN] = 0;
                // VERTICAL byte pairs (row y, row y + 1) of columns 0 .. 5: six v_perm serve the five interpolated values (the horizontal
                // pairs (k, k + 1) of either row took ten); the integer sum w00 p(k) + w10 q(k) + w01 p(k+1) + w11 q(k+1) is the same
                lk_s2 V[LK_RUN + 1];
#pragma unroll
                for (int k = 0; k <= LK_RUN; k++) {
                    const uint32_t sel = 0x0c000c00u | (uint32_t)(k & 3) | ((uint32_t)(4 + (k & 3)) << 16);
                    V[k] = lk_as_s2(k < 4 ? __builtin_amdgcn_perm(r1.x, r0.x, sel) : __builtin_amdgcn_perm(r1.y, r0.y, sel));
                }
#pragma unroll
                for (int k = 0; k < LK_RUN; k++)
                    val[k] = __builtin_amdgcn_sdot2(V[k], wc0, lk_dot2_k(V[k + 1], wc1, k_half9), false) >> (14 - 5);
#pragma unroll
                for (int q2 = 0; q2 < 3; q2++) {
                    const lk_s2 jv = lk_as_s2(lk_pack16(val[2 * q2], val[2 * q2 + 1]));      // (positions beyond the run meet a zero Ix / Iy)
                    sb1 = __builtin_amdgcn_sdot2(jv, lk_as_s2(IxP[t][q2]), sb1, false);
                    sb2 = __builtin_amdgcn_sdot2(jv, lk_as_s2(IyP[t][q2]), sb2, false);
                }
            }
            const long long ib1 = wave_sum_split4(sb1), ib2 = wave_sum_split4(sb2);
            const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
            const float ddx = (A12 * b2 - A22 * b1) * D;
            const float ddy = (A12 * b1 - A11 * b2) * D;
            nx += ddx; ny += ddy;
            resx = nx + half; resy = ny + half;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= epsilon) break;
            if (j > 0 && lk_oscillates(ddx, pdx, ddy, pdy)) {
                resx -= ddx * 0.5f; resy -= ddy * 0.5f;
                break;
            }
            pdx = ddx; pdy = ddy;
        }
        oJ[level][0] = jx0; oJ[level][1] = jy0;
    }
    outx = resx; outy = resy;
}

template <int NR, int WIN, int MAXIT>
__device__ __forceinline__ void lk2_body(const lk_args &g, const int *__restrict__ order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int n = g.d_n ? min(*g.d_n, g.n_max) : g.n_max;
    // workgroup w runs on XCD w % 8: every XCD takes one contiguous eighth of the (spatially ordered) point list
    const unsigned per = ((unsigned)n + KM_XCDS - 1) / KM_XCDS, slot = (blockIdx.x % KM_XCDS) * per + blockIdx.x / KM_XCDS;
    if (blockIdx.x / KM_XCDS >= per || slot >= (unsigned)n) return;
    const int p = order ? order[slot] : (int)slot;
    const lk2_geo<WIN> geo(g.win);
    const int win = geo.win, PS = geo.ps(), PP = geo.pp(), PB = geo.bytes();
    uint8_t *pA = smem_all, *pB = smem_all + 2 * PB;      // patches: [image][level]
    const int rpr = (win + LK_RUN - 1) / LK_RUN, total = win * rpr;
    int run_desc[NR];
#pragma unroll
    for (int t = 0; t < NR; t++) {
        const int r = t * 64 + (int)(threadIdx.x & 63);
        const int y = r / rpr, x0 = (r - y * rpr) * LK_RUN;
        run_desc[t] = r < total ? (y | (x0 << 8) | (min(LK_RUN, win - x0) << 16)) : 0;
    }
    const float px = g.pts_in[2 * p], py = g.pts_in[2 * p + 1];
    const float half = (float)(win - 1) * 0.5f;
    // one neighbourhood per image and level, centred on the key point's window: all loads of the key point in flight together
    int oA[2][2], oB[2][2];
    bool want[2], inside = true;
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const float sc = l ? 0.5f : 1.f;
        const int ipx = (int)floorf(px * sc - half), ipy = (int)floorf(py * sc - half);
        want[l] = !(ipx < -win || ipx >= g.A.W[l] || ipy < -win || ipy >= g.A.H[l]);      // (else the level is skipped: nothing to stage)
        oA[l][0] = oB[l][0] = want[l] ? ipx - LK2_M : LK2_INVALID;
        oA[l][1] = oB[l][1] = want[l] ? ipy - LK2_M : LK2_INVALID;
        inside = inside && want[l] && lk_patch_inside(g.A.W[l], g.A.H[l], oA[l][0], oA[l][1], PS, PS, MAXIT);
    }
    if (inside) {
        lk_patch_regs<MAXIT> r0, r1, r2, r3;
        stage_patch_issue<MAXIT>(g.A.img[1], g.A.W[1], oA[1][0], oA[1][1], PS, PS, r0);
        stage_patch_issue<MAXIT>(g.B.img[1], g.B.W[1], oB[1][0], oB[1][1], PS, PS, r1);
        stage_patch_issue<MAXIT>(g.A.img[0], g.A.W[0], oA[0][0], oA[0][1], PS, PS, r2);
        stage_patch_issue<MAXIT>(g.B.img[0], g.B.W[0], oB[0][0], oB[0][1], PS, PS, r3);
        stage_patch_commit<MAXIT>(PS, PS, pA + PB, PP, r0);
        stage_patch_commit<MAXIT>(PS, PS, pB + PB, PP, r1);
        stage_patch_commit<MAXIT>(PS, PS, pA, PP, r2);
        stage_patch_commit<MAXIT>(PS, PS, pB, PP, r3);
    } else {
#pragma unroll 1
        for (int l = 1; l >= 0; l--)
            if (want[l]) {
                lk2_restage<MAXIT>(g.A.img[l], g.A.W[l], g.A.H[l], oA[l][0], oA[l][1], geo, pA + l * PB);
                lk2_restage<MAXIT>(g.B.img[l], g.B.W[l], g.B.H[l], oB[l][0], oB[l][1], geo, pB + l * PB);
            }
    }
    LK_WAVE_SYNC();
    float fx, fy;
    lk2_track_point<NR, WIN, MAXIT>(g.A, g.B, pA, pB, oA, oB, px, py, geo, g.max_count, g.epsilon, run_desc, fx, fy, g.general_templates);
    if ((threadIdx.x & 63) == 0) { g.p1[2 * p] = fx; g.p1[2 * p + 1] = fy; }
    if (g.backward) {
        float rx, ry;
        lk2_track_point<NR, WIN, MAXIT>(g.B, g.A, pB, pA, oB, oA, fx, fy, geo, g.max_count, g.epsilon, run_desc, rx, ry, g.general_templates);
        if ((threadIdx.x & 63) == 0) { g.p0r[2 * p] = rx; g.p0r[2 * p + 1] = ry; }
    }
}

template <int NR, int WIN, int MAXIT>
__global__ __launch_bounds__(64) KM_LK_OCC void lk2_kernel(lk_args g, const int *__restrict__ order)
{
    lk2_body<NR, WIN, MAXIT>(g, order);
}

// the reference's window (matching_winsize 25, processing_configuration.json:14) with every size a compile-time constant.
// 6 waves per SIMD: the register allocator would take 82 VGPRs (5 waves) for two fewer copies.
#ifndef LK2_25_WAVES
#define LK2_25_WAVES 6
#endif
#if LK2_25_WAVES
#define LK2_25_OCC __attribute__((amdgpu_waves_per_eu(LK2_25_WAVES, LK2_25_WAVES)))
#else
#define LK2_25_OCC
#endif
__global__ __launch_bounds__(64) LK2_25_OCC void lk2_kernel_win25(lk_args g, const int *__restrict__ order)
{
    lk2_body<2, 25, 4>(g, order);
}


// ---- batched units (api_units.hip): ONE launch tracks the corners of every unit (blockIdx.y = unit): a launch pays one wave
// lifetime of fill + drain (~30 us) whatever its size, and sixteen units' 320 000 corners keep the GPU in steady state throughout
// The per-unit arguments are the single-unit kernel's own `lk_args`, in a table in device memory (sixteen of them exceed the 4 KB of
// kernel arguments): the tracker indexes the pyramids by level at run time, which wants them in addressable memory, not in registers.
template <int NR, int WIN, int MAXIT>
__global__ __launch_bounds__(64) KM_LK_OCC void lk2_units_kernel(const lk_args *__restrict__ table)
{
    lk2_body<NR, WIN, MAXIT>(table[blockIdx.y], nullptr);
}
__global__ __launch_bounds__(64) LK2_25_OCC void lk2_units_kernel_win25(const lk_args *__restrict__ table)
{
    lk2_body<2, 25, 4>(table[blockIdx.y], nullptr);
}

// kl_units_prepare: validate + upload the table (early in the call: the small copy is then off the critical path);
// kl_units_launch: the launch itself.  KM_E_UNSUPPORTED (no message): a unit without a level-1 pyramid (the first LK form serves it:
// units one by one)
int kl_units_prepare(km_ctx *c, const km_units &U, int n_max, int win, int max_count, double epsilon)
{
    if (n_max <= 0 || U.n <= 0) return KM_OK;
    if (win <= 2) return km_fail(c, KM_E_ARG, "winSize %d must be > 2", win);
    if (win > 40) return km_fail(c, KM_E_UNSUPPORTED, "winSize %d (supported 3..40)", win);
    if (!c->opt_lk2) return KM_E_UNSUPPORTED;
    lk_args host[KM_UNITS_MAX];
    const double e = epsilon < 0 ? 0 : epsilon > 10 ? 10 : epsilon;
    for (int u = 0; u < U.n; u++) {
        if (U.A[u].levels != 1 || U.B[u].levels != 1) return KM_E_UNSUPPORTED;
        lk_args &g = host[u];
        g.A = U.A[u]; g.B = U.B[u]; g.pts_in = U.p0[u]; g.d_n = &U.sc[u]->n_corners; g.n_max = n_max; g.win = win;
        g.max_count = max_count < 0 ? 0 : max_count > 100 ? 100 : max_count;
        g.backward = 1; g.epsilon = e * e; g.p1 = U.p1[u]; g.p0r = U.p0r[u]; g.left_band = nullptr;
        g.general_templates = km_dev_env("KARIOS_HIP_LK_GENERAL") ? 1 : 0;
    }
    lk_args *table = (lk_args *)km_ws(c, WS_UNITS_LK, sizeof(lk_args) * KM_UNITS_MAX);
    if (!table) return KM_E_NOMEM;
    return km_h2d_small(c, table, host, sizeof(lk_args) * (size_t)U.n);
}

int kl_units_launch(km_ctx *c, int n_units, int n_max, int win)
{
    if (n_max <= 0 || n_units <= 0) return KM_OK;
    const lk_args *table = (const lk_args *)km_ws_peek(c, WS_UNITS_LK);
    const int runs = win * ((win + LK_RUN - 1) / LK_RUN), nr = (runs + 63) / 64;
    const lk2_geo<0> geo(win);
    const size_t sm2 = (size_t)4 * geo.bytes();
    const dim3 grid(km_xcd_grid((unsigned)n_max), n_units);
    if (win == 25) lk2_units_kernel_win25<<<grid, 64, sm2, c->stream>>>(table);
    else switch (nr) {
    case 1: lk2_units_kernel<1, 0, 3><<<grid, 64, sm2, c->stream>>>(table); break;
    case 2: lk2_units_kernel<2, 0, 4><<<grid, 64, sm2, c->stream>>>(table); break;
    case 3: lk2_units_kernel<3, 0, 6><<<grid, 64, sm2, c->stream>>>(table); break;
    case 4: lk2_units_kernel<4, 0, 8><<<grid, 64, sm2, c->stream>>>(table); break;
    default: lk2_units_kernel<5, 0, 9><<<grid, 64, sm2, c->stream>>>(table); break;
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// Independent tracker runs in ONE launch (unit = blockIdx.y): the 25 (mon kernel, ref kernel) runs of the Laplacian kernel-size search
// (klt.py:465-545) - every job its own pyramids, corner list and outputs.  Forward + backward, two-level pyramids (the lk2 form).
// KM_E_UNSUPPORTED (no message): a job without a level-1 pyramid, lk2 off, a window the form does not hold - the caller runs kl_track
// job by job.
int kl_jobs_launch(km_ctx *c, const km_lk_job *jobs, int n_jobs, int n_max, int win, int max_count, double epsilon)
{
    if (n_max <= 0 || n_jobs <= 0) return KM_OK;
    if (n_jobs > KM_LK_JOBS_MAX || win <= 2 || win > 40 || !c->opt_lk2) return KM_E_UNSUPPORTED;
    lk_args host[KM_LK_JOBS_MAX];
    const double e = epsilon < 0 ? 0 : epsilon > 10 ? 10 : epsilon;
    for (int j = 0; j < n_jobs; j++) {
        if (jobs[j].A.levels != 1 || jobs[j].B.levels != 1) return KM_E_UNSUPPORTED;
        lk_args &g = host[j];
        g.A = jobs[j].A; g.B = jobs[j].B; g.pts_in = jobs[j].pts_in; g.d_n = jobs[j].d_n; g.n_max = n_max; g.win = win;
        g.max_count = max_count < 0 ? 0 : max_count > 100 ? 100 : max_count;
        g.backward = 1; g.epsilon = e * e; g.p1 = jobs[j].p1; g.p0r = jobs[j].p0r; g.left_band = nullptr;
        g.general_templates = km_dev_env("KARIOS_HIP_LK_GENERAL") ? 1 : 0;
    }
    lk_args *table = (lk_args *)km_ws(c, WS_UNITS_LK, sizeof(lk_args) * KM_LK_JOBS_MAX);
    if (!table) return KM_E_NOMEM;
    { const int rc = km_h2d_small(c, table, host, sizeof(lk_args) * (size_t)n_jobs); if (rc) return rc; }
    return kl_units_launch(c, n_jobs, n_max, win);
}

int kl_track(km_ctx *c, const km_pyr &A, const km_pyr &B, const float *d_pts_in, const int *d_n, int n_max, int win, int max_count,
             double epsilon, bool backward_too, float *d_p1, float *d_p0r, int *d_left_band)
{
    if (n_max <= 0) return KM_OK;
    if (win <= 2) return km_fail(c, KM_E_ARG, "winSize %d must be > 2", win);
    if (win > 40) return km_fail(c, KM_E_UNSUPPORTED, "winSize %d (supported 3..40)", win);
    lk_args g;
    g.A = A; g.B = B; g.pts_in = d_pts_in; g.d_n = d_n; g.n_max = n_max; g.win = win;
    g.max_count = max_count < 0 ? 0 : max_count > 100 ? 100 : max_count;
    g.backward = backward_too ? 1 : 0;
    g.general_templates = km_dev_env("KARIOS_HIP_LK_GENERAL") ? 1 : 0;
    double e = epsilon < 0 ? 0 : epsilon > 10 ? 10 : epsilon;
    g.epsilon = e * e;
    g.p1 = d_p1; g.p0r = d_p0r; g.left_band = d_left_band;
    const int runs = win * ((win + LK_RUN - 1) / LK_RUN), nr = (runs + 63) / 64;   // runs per lane (win <= 40: <= 5)
    if (c->opt_lk2 && A.levels == 1 && B.levels == 1 && A.Hres[0] == 0 && B.Hres[0] == 0 && !d_left_band) {
        // second form: four resident patches per key point, both directions on them
        const lk2_geo<0> geo(win);
        const size_t sm2 = (size_t)4 * geo.bytes();
        const unsigned nblk2 = km_xcd_grid((unsigned)n_max);
        const int *order = nullptr;          // (processing order of the points: the batched form passes a unit-major list)
        if (win == 25) lk2_kernel_win25<<<nblk2, 64, sm2, c->stream>>>(g, order);
        else switch (nr) {
        case 1: lk2_kernel<1, 0, 3><<<nblk2, 64, sm2, c->stream>>>(g, order); break;
        case 2: lk2_kernel<2, 0, 4><<<nblk2, 64, sm2, c->stream>>>(g, order); break;
        case 3: lk2_kernel<3, 0, 6><<<nblk2, 64, sm2, c->stream>>>(g, order); break;
        case 4: lk2_kernel<4, 0, 8><<<nblk2, 64, sm2, c->stream>>>(g, order); break;
        default: lk2_kernel<5, 0, 9><<<nblk2, 64, sm2, c->stream>>>(g, order); break;
        }
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
    const int RP = (win + 3 + 3 + 3) & ~3, JS = win + 1 + 2 * LK_M, JP = (JS + 3 + 3) & ~3;
    const size_t sm = (((size_t)(win + 3) * RP + 15) & ~(size_t)15) + (((size_t)JS * JP + 15) & ~(size_t)15) + ((size_t)(win + 1) * (win + 1) + 8) * 2 * sizeof(short) + 16;
    const size_t smw = (sm + 15) & ~(size_t)15, sm_all = smw * LK_WPB;
    const int nblk = (n_max + LK_WPB - 1) / LK_WPB;
    switch (nr) {
    case 1: lk_kernel<1><<<nblk, 64 * LK_WPB, sm_all, c->stream>>>(g, (int)smw); break;
    case 2: lk_kernel<2><<<nblk, 64 * LK_WPB, sm_all, c->stream>>>(g, (int)smw); break;
    case 3: lk_kernel<3><<<nblk, 64 * LK_WPB, sm_all, c->stream>>>(g, (int)smw); break;
    case 4: lk_kernel<4><<<nblk, 64 * LK_WPB, sm_all, c->stream>>>(g, (int)smw); break;
    default: lk_kernel<5><<<nblk, 64 * LK_WPB, sm_all, c->stream>>>(g, (int)smw); break;
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// test hook of the predicate above (km_lk_oscillation_probe): one thread per quadruple
__global__ void lk_oscillation_probe_kernel(const float *__restrict__ q, int n, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = lk_oscillates(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]) ? 1 : 0;
}
int kl_oscillation_probe(km_ctx *c, const float *d_q, int n, uint8_t *d_out)
{
    if (n <= 0) return KM_OK;
    lk_oscillation_probe_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(d_q, n, d_out);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
