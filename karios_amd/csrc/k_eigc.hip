// K3+K4 fused, 4 pixels per lane: minimum-eigenvalue map of cv2.goodFeaturesToTrack and its candidate corners in one
// pass, without ever writing the map (reference call site karios/matcher/klt.py:120, 494; algorithm SURVEY.md App. A.2:
// Sobel 3x3 REFLECT_101 -> products -> box filter blockSize x blockSize with its own REFLECT_101 border on the product
// images -> lambda_min -> threshold quality*max -> 3x3 dilation equality -> candidates).
//
// One wavefront owns a strip of 256 columns (4 per lane, loaded as one dword per row) and marches down `rows_per_item`
// candidate rows:
//   * vertical box sum  V += P(lead row) - P(trail row): the products of the row entering the window AND of the row
//     leaving it are recomputed from the source (two 3-row register windows; the trailing rows come from L2), so
//     there is no blockSize-deep ring in registers - 12 accumulators per lane whatever the block size;
//   * derivatives in packed 16-bit arithmetic (v_pk_*), products accumulated with v_mad_i32_i24;
//   * horizontal box sum = difference of two pixel-prefix sums: in-lane partial sums + one DPP wave scan per product,
//     the two ends fetched from the neighbour lanes with ds_bpermute;
//   * lambda_min per pixel exactly as the map kernel (k_dense.hip eig_march_kernel): fp64 scaling, separately rounded
//     float32 operations, correctly rounded sqrt;
//   * three lambda rows stay in registers; a pixel that is a 3x3 local maximum, lies off the image border, passes the
//     mask and exceeds a RUNNING lower bound of the final threshold is appended to a per-wave LDS stage and flushed to the
//     sharded key buffer.  The exact threshold is applied by the top-K pre-filter (k_select.hip tk_*).
// All sums are exact integers; results are bit-identical to eig map + candidate kernel.
#include <cstring>
#include <string.h>

#include "common.hpp"

#include <type_traits>

namespace {

typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned eigc_key(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float eigc_unkey(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dppz(int v)   // DPP move, 0 where no source lane
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// bound_ctrl: a lane without source reads 0, so no register has to be cleared first
__device__ __forceinline__ int lane_m1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // value of lane-1 (wave_shr:1)
__device__ __forceinline__ int lane_p1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }   // value of lane+1 (wave_shl:1)
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += dppz<0x111, 0xf>(v);  // row_shr:1
    v += dppz<0x112, 0xf>(v);  // row_shr:2
    v += dppz<0x114, 0xf>(v);  // row_shr:4
    v += dppz<0x118, 0xf>(v);  // row_shr:8
    v += dppz<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dppz<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }
__device__ __forceinline__ uint32_t as_u(short2v v) { return __builtin_bit_cast(uint32_t, v); }

// four source pixels (bytes of one dword) of three consecutive rows -> Sobel column sums as 16-bit pairs
//   t0 = a0 + 2*a1 + a2 (smoothing, for dx), t1 = a2 - a0 (difference, for dy); pair 0 = pixels 0,1, pair 1 = pixels 2,3
struct vsob {
    uint32_t t0[2], t1[2];
};
__device__ __forceinline__ void unpack4(uint32_t w, uint32_t &lo, uint32_t &hi)
{
    lo = __builtin_amdgcn_perm(0u, w, 0x0c010c00u);   // (b0, b1) as 16-bit values
    hi = __builtin_amdgcn_perm(0u, w, 0x0c030c02u);   // (b2, b3)
}
__device__ __forceinline__ vsob sobel_cols(const uint32_t (&a0)[2], const uint32_t (&a1)[2], const uint32_t (&a2)[2])
{
    vsob r;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const short2v s = as_s2(a0[h]) + as_s2(a2[h]);
        r.t0[h] = as_u(s + as_s2(a1[h]) + as_s2(a1[h]));
        r.t1[h] = as_u(as_s2(a2[h]) - as_s2(a0[h]));
    }
    return r;
}
// horizontal parts: dx[x] = t0[x+1] - t0[x-1], dy[x] = t1[x-1] + 2*t1[x] + t1[x+1]  (pairs again)
__device__ __forceinline__ void sobel_rows(const vsob &v, uint32_t (&dx)[2], uint32_t (&dy)[2])
{
    const uint32_t l0 = (uint32_t)lane_m1((int)v.t0[1]), r0 = (uint32_t)lane_p1((int)v.t0[0]);
    const uint32_t l1 = (uint32_t)lane_m1((int)v.t1[1]), r1 = (uint32_t)lane_p1((int)v.t1[0]);
    // shifted pairs: m = (x-1, x), c = (x+1, x+2) for the low pair; the high pair reuses c as its (x-1, x)
    const uint32_t t0_m = __builtin_amdgcn_alignbyte(v.t0[0], l0, 2), t0_c = __builtin_amdgcn_alignbyte(v.t0[1], v.t0[0], 2),
                   t0_p = __builtin_amdgcn_alignbyte(r0, v.t0[1], 2);
    const uint32_t t1_m = __builtin_amdgcn_alignbyte(v.t1[0], l1, 2), t1_c = __builtin_amdgcn_alignbyte(v.t1[1], v.t1[0], 2),
                   t1_p = __builtin_amdgcn_alignbyte(r1, v.t1[1], 2);
    dx[0] = as_u(as_s2(t0_c) - as_s2(t0_m));
    dx[1] = as_u(as_s2(t0_p) - as_s2(t0_c));
    dy[0] = as_u(as_s2(t1_m) + as_s2(v.t1[0]) + as_s2(v.t1[0]) + as_s2(t1_c));
    dy[1] = as_u(as_s2(t1_c) + as_s2(v.t1[1]) + as_s2(v.t1[1]) + as_s2(t1_p));
}
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(short)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)v >> 16; }

#ifndef EIGC_ABL
#define EIGC_ABL 0   // compile-time ablation mask (timing experiments only)
#endif
#define EIGC_STAGE 512   // keys per wave in LDS (+ one dummy slot per lane behind them)
#define EIGC_FLUSH_AT 128  // flush between row segments once this many keys are staged (a segment of 8 rows may add 512 more)
#define EIGC_PF 4        // source / mask rows in flight per stream (register FIFOs)
#define EIGC_SEG 2       // row groups between two looks at the candidate stage (flush, running threshold)

template <int BLOCK>
__global__ __launch_bounds__(256) void eigc_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, double scale2,
                                                   unsigned *__restrict__ max_partial, int nstrips, int rows_per_item, int nitems, double quality,
                                                   km_scalars *sc, unsigned long long *__restrict__ keys, size_t cap)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L;
    // margins (pixels): Sobel 1 + window + candidate neighbour 1; the left one rounded up to 4 and the stride kept a
    // multiple of 4, so that every lane's dword load is aligned whenever the row pitch is (W % 4 == 0)
    constexpr int ML = (L + 2 + 3) & ~3, STRIDE = (256 - ML - (Rr + 2)) & ~3;
    constexpr int PF = EIGC_PF;
    __shared__ unsigned long long stage[4][EIGC_STAGE + 64];   // [EIGC_STAGE + lane] = dummy slot of a lane without candidate
    __shared__ int xs_scratch[4][3][256];           // border strips only: pixel-prefix sums of the three products
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, kept in a scalar register
    const int wave_id = blockIdx.x * 4 + wv;
    if (wave_id >= nitems) { if (lane == 0) max_partial[wave_id] = 0u; return; }
    const int rowblock = wave_id / nstrips, strip = wave_id - rowblock * nstrips;
    const int xs = strip * STRIDE - ML;              // image column of strip pixel 0
    const int c0 = xs + 4 * lane;                    // image column of this lane's pixel 0
    const bool border = xs < 0 || xs + 255 > W - 1;  // wave-uniform: some strip pixel lies outside the image
    // candidate rows [ya, yb); lambda rows ya-1 .. yb
    const int ya = 1 + rowblock * rows_per_item, yb = min(H - 1, ya + rows_per_item);
    const int ye0 = ya - 1, ye1 = yb;                // first / last lambda row of this item
    const int m_first = ye0 - L, m_last = ye1 + Rr;  // product rows marched by this item (may lie outside: mirrored)

    // per-pixel constants
    uint32_t inimg_pair[2];                          // 0xffff per pixel whose column lies inside the image
    bool cand_px[4], e_px[4];                        // pixel may emit a candidate / lambda is meaningful and inside the image
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int i = 4 * lane + p, c = c0 + p;
        const bool in = c >= 0 && c <= W - 1;
        if (p & 1) inimg_pair[p >> 1] |= in ? 0xffff0000u : 0u; else inimg_pair[p >> 1] = in ? 0x0000ffffu : 0u;
        e_px[p] = in && i >= ML - 1 && i <= ML + STRIDE;
        cand_px[p] = i >= ML && i < ML + STRIDE && c >= 1 && c <= W - 2;
    }
    // every lane loads ONE dword per row: lanes (partly) outside the image load the nearest 4 in-image columns and a byte
    // permute puts REFLECT_101 values where the Sobel needs them (columns -1 and W; other outside columns never matter)
    const int c_load = min(max(c0, 0), W - 4);
    uint32_t load_sel = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int idx = km_reflect101(c0 + p, W) - c_load;
        load_sel |= (uint32_t)((idx >= 0 && idx <= 3) ? idx : 0) << (8 * p);
    }
    const uint8_t *mptr = mask ? mask : src;         // no mask: the loads still happen (fixed number of memory operations per row)
    const uint32_t mask_or = mask ? 0u : 0x01010101u;
    auto clamp_row = [&](int r) { return min(max(r, 0), H - 1); };

    auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;   // interior strip: no column border handling at all
    auto load_raw = [&](const uint8_t *base, int r) -> uint32_t {   // row r must lie inside the image
        const uint8_t *rowp = base + (size_t)r * W;
        if (FAST) return *(const uint32_t *)(rowp + (unsigned)c0);
        return *(const uint32_t *)(rowp + (unsigned)c_load);
    };
    auto unpack_src = [&](uint32_t w, uint32_t (&out)[2]) {
        if (!FAST) w = __builtin_amdgcn_perm(w, w, load_sel);
        unpack4(w, out[0], out[1]);
    };
    struct win3 { uint32_t a0[2], a1[2], a2[2]; };
    // product row m of the (REFLECT_101-extended) product images = products of image row r = reflect(m), whose Sobel
    // reads the source rows reflect(r - 1), r, reflect(r + 1)
    auto window_reload = [&](int m, win3 &w) {
        const int r = km_reflect101(m, H);
        unpack_src(load_raw(src, km_reflect101(r - 1, H)), w.a0);
        unpack_src(load_raw(src, r), w.a1);
        unpack_src(load_raw(src, km_reflect101(r + 1, H)), w.a2);
    };
    // Marching from product row m - 1 to m changes the 3-row source window by at most ONE row, also across the mirrored
    // zones above / below the image, where the window moves back up: entering_row(m) names that row (-1: none), so every
    // source row reaches the wave through one prefetched stream and the loop body has a fixed set of memory operations.
    auto entering_row = [&](int m) -> int {
        if (m >= 1 && m <= H - 2) return m + 1;       // inside: the window slides down
        if (m < 0) return -m - 1;                     // mirrored above the image: it slides up
        if (m >= H) return 2 * (H - 1) - m - 1;       // mirrored below: it slides up again
        return -1;                                    // m == 0 or m == H - 1: a permutation of the rows already held
    };
    auto window_step = [&](int m, win3 &w, uint32_t entering) {   // window of product row m from that of m - 1
        uint32_t e[2];
        unpack_src(entering, e);
        if (m >= 1 && m <= H - 2) {                   // (a0, a1, a2) <- (a1, a2, new)
            w.a0[0] = w.a1[0]; w.a0[1] = w.a1[1]; w.a1[0] = w.a2[0]; w.a1[1] = w.a2[1]; w.a2[0] = e[0]; w.a2[1] = e[1];
        } else if (m == 0) {                          // rows (0, 1, 2) -> (1, 0, 1)
            const uint32_t t0 = w.a0[0], t1 = w.a0[1];
            w.a0[0] = w.a1[0]; w.a0[1] = w.a1[1]; w.a2[0] = w.a1[0]; w.a2[1] = w.a1[1]; w.a1[0] = t0; w.a1[1] = t1;
        } else if (m == H - 1) {                      // rows (H-3, H-2, H-1) -> (H-2, H-1, H-2)
            w.a0[0] = w.a1[0]; w.a0[1] = w.a1[1]; w.a1[0] = w.a2[0]; w.a1[1] = w.a2[1]; w.a2[0] = w.a0[0]; w.a2[1] = w.a0[1];
        } else {                                      // (a0, a1, a2) <- (new, a0, a1)
            w.a2[0] = w.a1[0]; w.a2[1] = w.a1[1]; w.a1[0] = w.a0[0]; w.a1[1] = w.a0[1]; w.a0[0] = e[0]; w.a0[1] = e[1];
        }
    };
    auto derivs = [&](const win3 &w, uint32_t (&dx)[2], uint32_t (&dy)[2]) {
        const vsob v = sobel_cols(w.a0, w.a1, w.a2);
        sobel_rows(v, dx, dy);
        if (!FAST) { dx[0] &= inimg_pair[0]; dx[1] &= inimg_pair[1]; dy[0] &= inimg_pair[0]; dy[1] &= inimg_pair[1]; }   // products of outside columns are 0
    };

    int V[3][4];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int p = 0; p < 4; p++) V[q][p] = 0;
    auto accumulate = [&](const uint32_t (&dx)[2], const uint32_t (&dy)[2], bool subtract) {
        // V += (dx*dx, dx*dy, dy*dy) or V -= ...: the second factor is negated pairwise for the row leaving the window
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const short2v x = as_s2(dx[h]), y = as_s2(dy[h]);
            const short2v sx = subtract ? -x : x, sy = subtract ? -y : y;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                V[0][2 * h + k] += (int)x[k] * (int)sx[k];
                V[1][2 * h + k] += (int)x[k] * (int)sy[k];
                V[2][2 * h + k] += (int)y[k] * (int)sy[k];
            }
        }
    };

    // ---- horizontal window: W(i) = S(i + Rr) - S(i - L - 1), S = inclusive pixel prefix over the strip
    constexpr int UO[4] = {(0 + Rr) / 4, (1 + Rr) / 4, (2 + Rr) / 4, (3 + Rr) / 4};                    // lane offsets of the upper end
    constexpr int UJ[4] = {(0 + Rr) % 4, (1 + Rr) % 4, (2 + Rr) % 4, (3 + Rr) % 4};
    // lower end i - L - 1 = 4*lane + p - (L + 1): lane offset -ceil((L + 1 - p) / 4), in-lane index (p - L - 1) mod 4
    constexpr int LO[4] = {-((L + 1 - 0 + 3) / 4), -((L + 1 - 1 + 3) / 4), -((L + 1 - 2 + 3) / 4), -((L + 1 - 3 + 3) / 4)};
    constexpr int LJ[4] = {((0 - L - 1) % 4 + 4) % 4, ((1 - L - 1) % 4 + 4) % 4, ((2 - L - 1) % 4 + 4) % 4, ((3 - L - 1) % 4 + 4) % 4};
    auto bperm_from = [&](int lane_off, int v) { if (EIGC_ABL & 4) return v + lane_off; return __builtin_amdgcn_ds_bpermute(((lane + lane_off) & 63) * 4, v); };
    int *xsw = &xs_scratch[wv][0][0];
    auto windows = [&](int (&Wd)[3][4]) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int p1 = V[q][0] + V[q][1], p2 = p1 + V[q][2], qsum = p2 + V[q][3];
            const int A = wave_scan_incl(qsum), Ap = lane_m1(A);
            int X[4] = {Ap + V[q][0], Ap + p1, Ap + p2, A};
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const int up = UO[p] == 0 ? X[UJ[p]] : bperm_from(UO[p], X[UJ[p]]);
                const int lw = LO[p] == 0 ? X[LJ[p]] : bperm_from(LO[p], X[LJ[p]]);
                Wd[q][p] = up - lw;
            }
            if (!FAST) {
                // box filter's REFLECT_101 on the product images: add the products mirrored in from outside
                *(int4 *)(xsw + q * 256 + 4 * lane) = make_int4(X[0], X[1], X[2], X[3]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const int c = c0 + p;
                    int ia = -1, ib = -1;
                    if (c >= 0 && c < L) { ia = (L - c) - xs; ib = 0 - xs; }                                  // S'(L - c) - S'(0)
                    else if (c <= W - 1 && c + Rr > W - 1) { ia = (W - 2) - xs; ib = (2 * W - 3 - c - Rr) - xs; }  // S'(W-2) - S'(2(W-1) - c - Rr - 1)
                    if (ia >= 0) Wd[q][p] += xsw[q * 256 + ia] - xsw[q * 256 + ib];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    auto lambda_min = [&](int sxx, int sxy, int syy) {
        const float cxx = (float)__dmul_rn((double)sxx, scale2);
        const float cxy = (float)__dmul_rn((double)sxy, scale2);
        const float cyy = (float)__dmul_rn((double)syy, scale2);
        const float a = __fmul_rn(cxx, 0.5f), b = cxy, cc = __fmul_rn(cyy, 0.5f);
        const float t = __fsub_rn(a, cc);
        const float sq = __fadd_rn(__fmul_rn(t, t), __fmul_rn(b, b));
        if (EIGC_ABL & 2) return __fsub_rn(__fadd_rn(a, cc), sq);
        return __fsub_rn(__fadd_rn(a, cc), sqrtf(sq));
    };

    // ---- candidate staging
    unsigned long long *st = stage[wv];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned cnt = 0;
    const unsigned shard = (unsigned)wave_id % KM_NSHARD;
    const size_t cap_s = cap / KM_NSHARD;
    // The stage is only inspected BETWEEN row groups (a flush inside the row body would put conditional global stores
    // into the unrolled loop and make the compiler drain the prefetched loads every row).  Within a group the append is
    // bounded by the stage size; a group that would overflow it - more than ~55 % of its pixels being candidates, i.e. a
    // plateau image - raises sc->pad0 and the caller repeats the pass with the eig-map + candidate kernels.
    auto flush_if = [&](unsigned threshold) {
        if (cnt <= threshold) return;
        if (cnt > EIGC_STAGE) { if (lane == 0) atomicOr(&sc->pad0, 1u); cnt = EIGC_STAGE; }
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sc->shard_cnt[shard], cnt);
        base = __shfl(base, 0);
        for (unsigned i = lane; i < cnt; i += 64)
            if ((size_t)base + i < cap_s) keys[shard * cap_s + base + i] = st[i];
        cnt = 0;
    };
    float best = -INFINITY;                           // max of lambda over the masked image pixels seen by this lane
    float thr_run = 0.f;
    unsigned published = 0u;
    auto refresh_threshold = [&]() {
        unsigned wk = best > -INFINITY ? eigc_key(best) : 0u;
        for (int o = 32; o > 0; o >>= 1) wk = max(wk, (unsigned)__shfl_xor((int)wk, o));
        const unsigned gk = __hip_atomic_load(&sc->run_max_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wk > gk && published == 0u && lane == 0) atomicMax(&sc->run_max_key, wk);   // one early publication per wave
        if (wk > gk) published = 1u;
        const unsigned mk = max(wk, gk);
        thr_run = mk ? (float)__dmul_rn((double)eigc_unkey(mk), quality) : 0.f;
    };

    float e2[4] = {0.f, 0.f, 0.f, 0.f}, e1[4] = {0.f, 0.f, 0.f, 0.f};   // lambda rows y-2, y-1
    uint32_t mk1 = 0;                                                  // mask bytes of row y-1
    win3 lead, trail;
    // One marching step = product row m.  `nl` / `nt` = the source rows entering the lead / trail windows, `mkraw` = mask
    // row m - Rr.  INTERIOR (compile time): both windows slide down and the step is past the warm-up, i.e. the body is
    // straight-line code apart from the rare stage flush.
    auto row_step = [&](int m, uint32_t nl, uint32_t nt, uint32_t mkraw, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        const int step = m - m_first;
        const uint32_t mk = (FAST ? mkraw : __builtin_amdgcn_perm(mkraw, mkraw, load_sel)) | mask_or;   // mask bytes of lambda row m - Rr
        uint32_t dx[2], dy[2];
        if (INTERIOR) {
            lead.a0[0] = lead.a1[0]; lead.a0[1] = lead.a1[1]; lead.a1[0] = lead.a2[0]; lead.a1[1] = lead.a2[1];
            unpack_src(nl, lead.a2);
            trail.a0[0] = trail.a1[0]; trail.a0[1] = trail.a1[1]; trail.a1[0] = trail.a2[0]; trail.a1[1] = trail.a2[1];
            unpack_src(nt, trail.a2);
        } else if (step > 0) {
            window_step(m, lead, nl);
            window_step(m - BLOCK, trail, nt);
        }
        derivs(lead, dx, dy);
        accumulate(dx, dy, false);
        if (!(EIGC_ABL & 8) && (INTERIOR || step >= BLOCK)) {
            derivs(trail, dx, dy);
            accumulate(dx, dy, true);
        }
        if (!INTERIOR && step < BLOCK - 1) return;
        const int y = m - Rr;                            // lambda row completed by this step (ye0 <= y <= ye1)
        int Wd[3][4];
        if (EIGC_ABL & 16) { for (int q = 0; q < 3; q++) for (int p = 0; p < 4; p++) Wd[q][p] = V[q][p]; } else windows(Wd);
        float e0[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            e0[p] = (EIGC_ABL & 32) ? __int_as_float(Wd[0][p] + Wd[1][p] + Wd[2][p]) : lambda_min(Wd[0][p], Wd[1][p], Wd[2][p]);
            const bool counts = e_px[p] && ((mk >> (8 * p)) & 0xffu) != 0u;
            best = fmaxf(best, counts ? e0[p] : -INFINITY);
        }
        if (!(EIGC_ABL & 1) && (INTERIOR || y >= ye0 + 2)) {
            // candidate test of row y-1 against lambda rows y-2, y-1, y
            float m3[4];
#pragma unroll
            for (int p = 0; p < 4; p++) m3[p] = fmaxf(fmaxf(e2[p], e1[p]), e0[p]);
            const float m3l = __int_as_float(lane_m1(__float_as_int(m3[3]))), m3r = __int_as_float(lane_p1(__float_as_int(m3[0])));
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float left = p == 0 ? m3l : m3[p - 1], right = p == 3 ? m3r : m3[p + 1];
                const float nb = fmaxf(fmaxf(left, right), fmaxf(e2[p], e0[p]));
                const bool is = cand_px[p] && e1[p] > thr_run && e1[p] != 0.f && e1[p] >= nb && ((mk1 >> (8 * p)) & 0xffu) != 0u;
                const unsigned long long bal = __ballot(is);
                const unsigned slot = cnt + (unsigned)__popcll(bal & lt_mask);
                // unconditional store (no branch in the row body): non-candidates and overflow go to the lane's dummy slot
                st[(is && slot < EIGC_STAGE) ? slot : EIGC_STAGE + (unsigned)lane] =
                    ((unsigned long long)__float_as_uint(e1[p]) << 32) | (unsigned long long)((unsigned)(y - 1) * (unsigned)W + (unsigned)(c0 + p));
                cnt += (unsigned)__popcll(bal);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) { e2[p] = e1[p]; e1[p] = e0[p]; }
        mk1 = mk;
    };

    window_reload(m_first, lead);
    window_reload(m_first - BLOCK, trail);              // the trail window follows BLOCK product rows behind from the start
    // interior steps [mi_lo, mi_hi]: past the warm-up (step >= BLOCK + 1: candidates are being tested) and both windows in
    // the plain sliding regime for the step itself and for every row prefetched from it
    const int mi_lo = max(m_first + BLOCK + 1, BLOCK + 1), mi_hi = min(m_last, H - 2 - PF);
    int m = m_first;
    // general steps (image top / warm-up): rows are loaded when they are needed
    auto general_until = [&](int m_end) {
        for (; m <= m_end; m++) {
            if (((m - m_first) & 7) == 0) refresh_threshold();
            const uint32_t nl = load_raw(src, max(entering_row(m), 0)), nt = load_raw(src, max(entering_row(m - BLOCK), 0));
            const uint32_t mkraw = load_raw(mptr, clamp_row(m - Rr));
            row_step(m, nl, nt, mkraw, std::false_type{});
            flush_if(EIGC_FLUSH_AT);
        }
    };
    general_until(min(mi_lo - 1, m_last));
    if (m <= mi_hi) {
        // register FIFO with static slots: slot k holds the rows of step m + k and is refilled for step m + k + PF right
        // after use, so a load has PF - 1 whole steps to arrive and nothing is ever copied
        uint32_t ql[PF], qt[PF], qm[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
            ql[k] = load_raw(src, m + k + 1);
            qt[k] = load_raw(src, m + k - BLOCK + 1);
            qm[k] = load_raw(mptr, m + k - Rr);
        }
        while (m + PF - 1 <= mi_hi) {
            // between segments: the only places of the interior march with conditional global memory traffic
            flush_if(EIGC_FLUSH_AT);
            refresh_threshold();
            const int seg_end = min(mi_hi, m + EIGC_SEG * PF - 1);
            for (; m + PF - 1 <= seg_end; m += PF) {
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    row_step(m + k, ql[k], qt[k], qm[k], std::true_type{});
                    ql[k] = load_raw(src, m + k + PF + 1);           // <= H - 1 by the choice of mi_hi
                    qt[k] = load_raw(src, m + k + PF - BLOCK + 1);
                    qm[k] = load_raw(mptr, m + k + PF - Rr);
                }
            }
        }
    }
    general_until(m_last);                                // image bottom / the last few steps of the item
    flush_if(0u);
    unsigned key = best > -INFINITY ? eigc_key(best) : 0u;
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    if (lane == 0) {
        max_partial[wave_id] = key;
        if (key > __hip_atomic_load(&sc->run_max_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&sc->run_max_key, key);
    }
    };  // run
    if (!border) run(std::true_type{});
    else run(std::false_type{});
}

__global__ __launch_bounds__(1024) void eigc_max_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned *out)
{
    unsigned m = 0;
    for (unsigned i = threadIdx.x; i < n; i += 1024) m = max(m, partial[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh[16];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int i = 0; i < 16; i++) t = max(t, sh[i]);
        *out = t;
    }
}

template <int BLOCK>
int launch_eigc(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, double scale2, double quality, km_scalars *sc,
                unsigned long long *d_keys, size_t cap)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L, STRIDE = (256 - ((L + 2 + 3) & ~3) - (Rr + 2)) & ~3;
    const int nstrips = (W - 1 + STRIDE - 1) / STRIDE;   // strips tile the columns 0 .. W-2 (candidates: 1 .. W-2)
    int wg_per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, (const void *)eigc_kernel<BLOCK>, 256, 0) != hipSuccess || wg_per_cu < 1)
        wg_per_cu = 3;
    int rows = km_pick_rows(H - 2, nstrips, BLOCK + 1, (long)c->n_cu * 4 * wg_per_cu, 64, 384);
    if (const char *e = getenv("KARIOS_HIP_EIG_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 8192) rows = v; }   // tuning override
    const int nitems = nstrips * ((H - 2 + rows - 1) / rows);
    const unsigned nblk = (unsigned)((nitems + 3) / 4);
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)nblk * 4 * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    eigc_kernel<BLOCK><<<nblk, 256, 0, c->stream>>>(d_src, d_mask, H, W, scale2, partial, nstrips, rows, nitems, quality, sc, d_keys, cap);
    KM_LAUNCH_CHECK(c);
    eigc_max_kernel<<<1, 1024, 0, c->stream>>>(partial, nblk * 4, &sc->max_eig_key);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

}  // namespace


// Fused minimum-eigenvalue + candidate pass.  Expects sc->run_max_key and sc->shard_cnt[] zeroed (`rezero` does it).
// Returns KM_E_UNSUPPORTED (without an error message) when this kernel does not cover the case.
int ke_eig_candidates(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, double quality, km_scalars *sc,
                      unsigned long long *d_keys, size_t cap, bool rezero)
{
    if (block < 1 || block > 15) return KM_E_UNSUPPORTED;
    if (!(W >= 2 * block + 8 && H >= 2 * block + 8)) return KM_E_UNSUPPORTED;   // mirrored columns / rows stay near their border
    if (rezero) KM_HIP(c, hipMemsetAsync(&sc->run_max_key, 0, (2 + KM_NSHARD) * sizeof(unsigned), c->stream));   // run_max_key, pad, shard counters
    const double scale = 1.0 / (4.0 * (double)block * 255.0), s2 = scale * scale;
    switch (block) {
#define KM_EIGC_CASE(B) case B: return launch_eigc<B>(c, d_src, d_mask, H, W, s2, quality, sc, d_keys, cap);
        KM_EIGC_CASE(1) KM_EIGC_CASE(2) KM_EIGC_CASE(3) KM_EIGC_CASE(4) KM_EIGC_CASE(5) KM_EIGC_CASE(7) KM_EIGC_CASE(9) KM_EIGC_CASE(11)
        KM_EIGC_CASE(13) KM_EIGC_CASE(15)
#undef KM_EIGC_CASE
    default: return KM_E_UNSUPPORTED;
    }
}
