// K3 + K4 fused, EIGHT pixels per lane: minimum eigenvalue of cv2.goodFeaturesToTrack and its candidate test in one pass
// (reference call site karios/matcher/klt.py:120, 494; algorithm SURVEY.md App. A.2), same arithmetic and same results as
// eig2_item (eig2_item.hpp: exact integer sums, fp64 scaling, individually rounded float32 formula, correctly rounded sqrt).
//
// Why a second formulation.  The 2-px kernel is bound by VALU issue (>= 94 % busy, 171 instructions per 128-pixel row step),
// and most of those instructions move data between lanes: every pixel pair pays 2 DPP moves + 4 byte alignments per Sobel
// window, a 6-step DPP wave scan + 4 ds_bpermute per product for the horizontal box sum, and one LDS append with ballot /
// mbcnt arithmetic per pixel for the 3 % of the pixels that are candidates; 18 of its 128 columns and 16 of its 64 rows
// are halo.  Here a lane owns 8 ADJACENT columns (one 8-byte load per row) and a wavefront 512:
//   * Sobel: the odd-aligned pixel pairs are shared between neighbouring pairs of the same lane (5 byte alignments and
//     2 DPP moves per 4 pairs and operand instead of 8 + 8);
//   * the box filter (<= 15 wide) of a pixel spans this lane and its two neighbours only: it slides along the lane's pixels,
//     W(p+1) = W(p) + V(p+1+R) - V(p-L), with the neighbours' vertical sums as DPP operands of the add / subtract itself -
//     no scan, no LDS;
//   * a lane seldom holds more than two candidates per row: the last candidate of the lane is selected while the pixels are
//     tested and appended once per row, a second append takes the first candidates of the lanes that hold two (their tests
//     stay in scalar registers as wave masks); rows where some lane holds three or more take a per-pixel path (tie_rows);
//   * 24 of 512 columns are halo, and an item is ~120 rows tall (2-3 waves per SIMD hide the latencies: eight independent
//     pixels per lane give the scheduler what four more resident waves gave the 2-px kernel).
// The strips that touch the left / right image border (12 columns each) run eig2_item inside the same launch.
#include <cstring>
#include <string.h>

#include "eig2_item.hpp"

namespace {

#define EIG3_STAGE 1024     // candidate keys per wave in LDS (+ one dummy slot per lane behind them)
#define EIG3_FLUSH_AT 384   // flush between row segments once this many keys are staged
#define EIG3_SEG 3          // row groups (of 3 rows) between two looks at the stage / the running threshold
#define EIG3_MARGIN 12      // halo columns either side of a strip's outputs (>= L + 2, multiple of 4)
#define EIG3_STRIDE (512 - 2 * EIG3_MARGIN)

__device__ __forceinline__ int e3_mad_lo(uint32_t a, uint32_t b, int c)
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int e3_mad_hi(uint32_t a, uint32_t b, int c)
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t e3_pk_mad2(uint32_t a, uint32_t c)   // a * 2 + c on both 16-bit halves: one v_pk_mad_u16
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(0x00020002u), "v"(c));
    return d;
}
// w + v[lane + 1] / w - v[lane - 1] (0 beyond the wave) in ONE instruction.  Written as assembly because the optimiser
// re-associates w + a - b into (a - b) + w, which needs a separate DPP move for one of the two operands; the callers pad
// the producers of v (see the s_nop in compute), the hazard recogniser cannot see in here.
__device__ __forceinline__ int e3_add_next(int w, int v)
{
    int d;
    asm("v_add_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
__device__ __forceinline__ int e3_add_prev(int w, int v)
{
    int d;
    asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
// v[lane + 1] - w / v[lane - 1] - w.  (The mirrored form, w - v[lane +- 1] = v_subrev_u32_dpp, is NOT used: written in assembly
// it computed dpp(src1) - src0 on this toolchain - tools/ubench/asm_check.hip checks every helper here against plain arithmetic.)
__device__ __forceinline__ int e3_next_minus(int v, int w)
{
    int d;
    asm("v_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w));
    return d;
}
__device__ __forceinline__ int e3_prev_minus(int v, int w)
{
    int d;
    asm("v_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w));
    return d;
}

template <class F, int... I> __device__ __forceinline__ void e3_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void e3_for(F &&f) { e3_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(0_c) .. f((N-1)_c)

template <int BLOCK>
__device__ __forceinline__ void eig3_item(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, double scale2,
                                          unsigned *__restrict__ max_partial, double quality, km_scalars *sc, unsigned long long *__restrict__ keys,
                                          size_t cap, unsigned stage_cap, int wave_id, int xs, int col_lo, int col_hi, int ye0, int ye1,
                                          unsigned long long *st /* [EIG3_STAGE + 64] */, int count_skips = 0)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L;
    static_assert(L <= 7 && Rr <= 7, "the box window must stay within the two neighbouring lanes");
    const int lane = threadIdx.x & 63;
    const int c0 = xs + 8 * lane;                    // image column of this lane's pixel 0 (the whole strip lies inside the image)
    const int m_first = ye0 - L, m_last = ye1 + Rr;  // product rows marched (may lie outside the image: mirrored)

    // bytes 0xff where the pixel's column may emit a candidate / counts for the maximum
    uint32_t colmask[2] = {0u, 0u};
#pragma unroll
    for (int p = 0; p < 8; p++)
        if (c0 + p >= col_lo && c0 + p < col_hi) colmask[p >> 2] |= 0xffu << (8 * (p & 3));
    const uint8_t *mptr = mask ? mask : src;         // no mask: the loads still happen (fixed set of memory operations per row)
    const uint32_t mask_or = mask ? 0u : 0x01010101u;

    auto load8 = [&](const uint8_t *base, int r) -> uint2 {
        const uint8_t *rowp = base + (size_t)r * W;  // wave-uniform row base + opaque 32-bit lane offset
        uint2 v;
        __builtin_memcpy(&v, rowp + e2_opaque((unsigned)c0), 8);
        return v;
    };
    struct row4 { uint32_t q[4]; };                  // 8 pixels as four 16-bit pairs
    auto unpack = [&](uint2 w) -> row4 {
        row4 r;
        r.q[0] = __builtin_amdgcn_perm(0u, w.x, 0x0c010c00u);
        r.q[1] = __builtin_amdgcn_perm(0u, w.x, 0x0c030c02u);
        r.q[2] = __builtin_amdgcn_perm(0u, w.y, 0x0c010c00u);
        r.q[3] = __builtin_amdgcn_perm(0u, w.y, 0x0c030c02u);
        return r;
    };
    // Sobel derivatives of the product row whose source rows are r0 (above), r1, r2 (below); NEG: also their negatives
    auto derivs = [&](const row4 &r0, const row4 &r1, const row4 &r2, uint32_t (&dx)[4], uint32_t (&dy)[4]) {
        uint32_t t0[4], t1[4], m0[5], m1[5];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            t0[i] = e3_pk_mad2(r1.q[i], e2u(e2s(r0.q[i]) + e2s(r2.q[i])));   // column sums (for dx)
            t1[i] = e2u(e2s(r2.q[i]) - e2s(r0.q[i]));                         // column differences (for dy)
        }
        // t0 comes out of inline assembly, which the compiler's hazard recogniser cannot see into: a VGPR written by a VALU
        // instruction must not be read through DPP in the next two wait states
        asm("s_nop 1" : "+v"(t0[0]), "+v"(t0[3]));
        const uint32_t l0 = (uint32_t)e2_lane_m1((int)t0[3]), r0n = (uint32_t)e2_lane_p1((int)t0[0]);
        const uint32_t l1 = (uint32_t)e2_lane_m1((int)t1[3]), r1n = (uint32_t)e2_lane_p1((int)t1[0]);
        // m[i] = (pixel 2i-1, pixel 2i): the odd-aligned pairs, shared by the two even-aligned pairs either side
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t lo0 = i == 0 ? l0 : t0[i - 1], hi0 = i == 4 ? r0n : t0[i];
            const uint32_t lo1 = i == 0 ? l1 : t1[i - 1], hi1 = i == 4 ? r1n : t1[i];
            m0[i] = __builtin_amdgcn_alignbyte(hi0, lo0, 2);
            m1[i] = __builtin_amdgcn_alignbyte(hi1, lo1, 2);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            dx[i] = e2u(e2s(m0[i + 1]) - e2s(m0[i]));
            dy[i] = e2u(e2s(e3_pk_mad2(t1[i], m1[i])) + e2s(m1[i + 1]));
        }
    };
    // vertical box sums of the three products, per pixel
    int V[3][8];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int p = 0; p < 8; p++) V[q][p] = 0;
    auto accumulate = [&](const uint32_t (&dx)[4], const uint32_t (&dy)[4], bool subtract) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t x = dx[i], y = dy[i];
            const uint32_t sx = subtract ? e2u(-e2s(x)) : x, sy = subtract ? e2u(-e2s(y)) : y;
            V[0][2 * i] = e3_mad_lo(x, sx, V[0][2 * i]);
            V[0][2 * i + 1] = e3_mad_hi(x, sx, V[0][2 * i + 1]);
            V[1][2 * i] = e3_mad_lo(x, sy, V[1][2 * i]);
            V[1][2 * i + 1] = e3_mad_hi(x, sy, V[1][2 * i + 1]);
            V[2][2 * i] = e3_mad_lo(y, sy, V[2][2 * i]);
            V[2][2 * i + 1] = e3_mad_hi(y, sy, V[2][2 * i + 1]);
        }
    };
    // horizontal box sums: window of pixel p = strip pixels p - L .. p + Rr, indices < 0 / > 7 live in the neighbouring lanes
    // (the three products advance in lockstep: three independent dependency chains for the scheduler instead of one)
    auto windows = [&](auto &&emit /* (pixel, sxx, sxy, syy): called as soon as the three sums of a pixel exist */) {
        // w + the vertical sum of strip pixel k of this lane's window (k < 0: left neighbour's pixel k + 8, k > 7: right neighbour's k - 8)
        auto add_ext = [&](int w, auto q_tag, auto kt) -> int {
            constexpr int q = decltype(q_tag)::value, k = decltype(kt)::value;
            if constexpr (k < 0) return e3_add_prev(w, V[q][k + 8]);
            else if constexpr (k > 7) return e3_add_next(w, V[q][k - 8]);
            else return w + V[q][k];
        };
        // vext(k) - w
        auto ext_minus = [&](auto q_tag, auto kt, int w) -> int {
            constexpr int q = decltype(q_tag)::value, k = decltype(kt)::value;
            if constexpr (k < 0) return e3_prev_minus(V[q][k + 8], w);
            else if constexpr (k > 7) return e3_next_minus(V[q][k - 8], w);
            else return V[q][k] - w;
        };
        int w[3];
        e3_for<3>([&](auto qt) {
            constexpr int q = decltype(qt)::value;
            if constexpr (L == 7 && Rr == 7) {
                // the window of pixel 0 = the whole lane + pixels 1..7 of the left neighbour
                const int T = ((V[q][0] + V[q][1]) + (V[q][2] + V[q][3])) + ((V[q][4] + V[q][5]) + (V[q][6] + V[q][7]));
                w[q] = T + e2_lane_m1(T - V[q][0]);
            } else {
                w[q] = 0;
                e3_for<BLOCK>([&](auto kt) { w[q] = add_ext(w[q], qt, std::integral_constant<int, decltype(kt)::value - L>{}); });
            }
        });
        emit(std::integral_constant<int, 0>{}, w[0], w[1], w[2]);
        e3_for<7>([&](auto pt) {
            constexpr int P = decltype(pt)::value;
            // w + entering - leaving as two "operand minus accumulator" steps: leaving - w, then entering - (leaving - w)
            int t[3];
            e3_for<3>([&](auto qt) { t[decltype(qt)::value] = ext_minus(qt, std::integral_constant<int, P - L>{}, w[decltype(qt)::value]); });
            e3_for<3>([&](auto qt) {
                constexpr int q = decltype(qt)::value;
                w[q] = ext_minus(qt, std::integral_constant<int, P + 1 + Rr>{}, t[q]);
            });
            emit(std::integral_constant<int, P + 1>{}, w[0], w[1], w[2]);
        });
    };
    const double scale2h = scale2 * 0.5;              // (exact)
    auto lambda_min = [&](int sxx, int sxy, int syy) {
        // a = cxx * 0.5f, cc = cyy * 0.5f of the reference formula: halving commutes with both roundings (no value here is
        // anywhere near the subnormal range: the smallest non-zero product is scale2 ~ 4e-9 / BLOCK^2), so it is folded into the scale
        const float a = (float)__dmul_rn((double)sxx, scale2h);
        const float b = (float)__dmul_rn((double)sxy, scale2);
        const float cc = (float)__dmul_rn((double)syy, scale2h);
        const float t = __fsub_rn(a, cc);
        const float sq = __fadd_rn(__fmul_rn(t, t), __fmul_rn(b, b));
        return __fsub_rn(__fadd_rn(a, cc), e2_sqrt(sq));
    };

    float best = -INFINITY;
    unsigned tie_rows = 0u;                           // (wave-uniform: a scalar register)
#ifdef KM_DEV
    unsigned skip_total = 0u, skip_hit = 0u;          // "eig3_count" (VERDICT r4 item 5): how often could the eigenvalue formula be skipped?
#endif
    // ---- candidate staging and the running threshold (as in eig2_item)
    unsigned cnt = 0;
    const unsigned shard = (unsigned)wave_id % KM_NSHARD;
    const size_t cap_s = cap / KM_NSHARD;
    auto flush_if = [&](unsigned threshold) {
        if (cnt <= threshold) return;
        if (cnt > stage_cap) { if (lane == 0) atomicOr(&sc->pad0, 1u); cnt = min(cnt, (unsigned)EIG3_STAGE); }
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sc->shard_cnt[shard], cnt);
        base = __shfl(base, 0);
        for (unsigned i = lane; i < cnt; i += 64)
            if ((size_t)base + i < cap_s) keys[shard * cap_s + base + i] = st[i];
        cnt = 0;
    };
    float thr_run = 0.f;
    unsigned published = 0u, gk_seen = 0u;
    auto refresh_threshold = [&](bool global) {
        unsigned wk = best > -INFINITY ? e2_key(best) : 0u;
        for (int o = 32; o > 0; o >>= 1) wk = max(wk, (unsigned)__shfl_xor((int)wk, o));
        if (global) {
            gk_seen = __hip_atomic_load(&sc->run_max_shard[wave_id & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (published == 0u && wk > gk_seen && (gk_seen == 0u || e2_unkey(wk) > e2_unkey(gk_seen) * 1.0625f)) {
            if (lane == 0) atomicMax(&sc->run_max_shard[wave_id & 63], wk);
            published = 1u;
        }
        const unsigned mk2 = max(wk, gk_seen);
        thr_run = mk2 ? (float)__dmul_rn((double)e2_unkey(mk2), quality) : 0.f;
    };

    // rotating state, slot = (step + k) % 3: source rows of the lead / trail Sobel windows, lambda rows y-2, y-1, y
    row4 LW[3], TW[3];
    float E[3][8];
    float EM[8];                                      // lambda row y-1 with -inf where the mask / the column range excludes the pixel
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int p = 0; p < 8; p++) E[k][p] = 0.f;
#pragma unroll
    for (int p = 0; p < 8; p++) EM[p] = -INFINITY;

    // the part of a marching step behind the windows: product row m is accumulated, lambda row y = m - Rr completed,
    // candidate row y - 1 tested.  PH = step % 3; TRAIL / OUT / CAND: warm-up steps do less
    auto compute = [&](auto ph_tag, int m, uint2 mkraw, bool trail, bool out, bool cand) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int S0 = PH, S1 = (PH + 1) % 3, S2 = (PH + 2) % 3;   // window rows: oldest, middle, newest
        uint32_t dx[4], dy[4];
        derivs(LW[S0], LW[S1], LW[S2], dx, dy);
        accumulate(dx, dy, false);
        if (trail) {
            derivs(TW[S0], TW[S1], TW[S2], dx, dy);
            accumulate(dx, dy, true);
        }
        if (!out) return;
        const int y = m - Rr;                        // lambda row completed by this step
        constexpr int C = PH, P1 = (PH + 2) % 3, P2 = (PH + 1) % 3;    // lambda slots: row y, y-1, y-2
        {
            // the vertical sums were written by inline assembly (v_mad_i32_i16) and are read through DPP - also inline assembly -
            // below: 2 wait states after a VALU write of the operand, 5 after a VALU write of EXEC
            asm("s_nop 4" : "+v"(V[0][0]), "+v"(V[0][1]), "+v"(V[0][2]), "+v"(V[0][3]), "+v"(V[0][4]), "+v"(V[0][5]), "+v"(V[0][6]), "+v"(V[0][7]),
                            "+v"(V[1][0]), "+v"(V[1][1]), "+v"(V[1][2]), "+v"(V[1][3]), "+v"(V[1][4]), "+v"(V[1][5]), "+v"(V[1][6]), "+v"(V[1][7]),
                            "+v"(V[2][0]), "+v"(V[2][1]), "+v"(V[2][2]), "+v"(V[2][3]), "+v"(V[2][4]), "+v"(V[2][5]), "+v"(V[2][6]), "+v"(V[2][7]));
            // (a pixel's eigenvalue is formed as soon as its three window sums exist: the 24 sums of a row never coexist)
#ifdef KM_DEV
            if (count_skips) {
                // lambda_min <= min(c_xx, c_yy) = min(S_xx, S_yy) scale^2: when that bound is <= the running lower bound of the threshold
                // for ALL 64 lanes, TOZERO zeroes the pixel slot whatever the formula returns (the bound only rises): the fp64-scaled,
                // individually rounded formula with its correctly rounded sqrt (44 % of the kernel's instructions) could be skipped
                windows([&](auto pt, int sxx, int sxy, int syy) {
                    E[C][decltype(pt)::value] = lambda_min(sxx, sxy, syy);
                    const float bound = (float)__dmul_rn((double)min(sxx, syy), scale2);
                    skip_total++;
                    skip_hit += __ballot(!(bound <= thr_run)) == 0ull ? 1u : 0u;
                });
            } else
#endif
            windows([&](auto pt, int sxx, int sxy, int syy) { E[C][decltype(pt)::value] = lambda_min(sxx, sxy, syy); });
        }
        if (cand) {
            // candidate test of row y-1: own value >= the 3x3 maximum (itself included)
            float m3[8];
#pragma unroll
            for (int p = 0; p < 8; p++) m3[p] = fmaxf(fmaxf(E[P2][p], E[P1][p]), E[C][p]);
            const float m3l = __int_as_float(e2_lane_m1(__float_as_int(m3[7]))), m3r = __int_as_float(e2_lane_p1(__float_as_int(m3[0])));
            auto is_cand = [&](int p) {
                const float left = p == 0 ? m3l : m3[(p + 7) & 7], right = p == 7 ? m3r : m3[(p + 1) & 7];
                const float nb = fmaxf(fmaxf(left, right), m3[p]);
                return EM[p] > thr_run && EM[p] >= nb;   // (thr_run >= +0 - qualityLevel > 0 is checked on entry - so OpenCV's `val != 0` is implied)
            };
            float selval = 0.f;
            unsigned selp = 0u, ncand = 0u;
            bool isv[8];                                  // (wave masks: scalar registers)
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const bool is = isv[p] = is_cand(p);
                selval = is ? EM[p] : selval;
                selp = is ? (unsigned)p : selp;
                ncand += is ? 1u : 0u;
            }
            const unsigned rowidx = (unsigned)(y - 1) * (unsigned)W + (unsigned)c0;
            const unsigned long long multi = __ballot(ncand > 1u);
            auto append = [&](unsigned long long bal, bool mine, float v, unsigned p) {
                const unsigned slot = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                st[min(mine ? slot : ~0u, EIG3_STAGE + (unsigned)lane)] = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(rowidx + p);
                cnt += (unsigned)__popcll(bal);
            };
            if (__builtin_expect(multi == 0ull, 1)) {
                append(__ballot(ncand != 0u), ncand != 0u, selval, selp);
            } else if (__ballot(ncand > 2u) == 0ull) {
                // some lane holds two candidates (local maxima two or more pixels apart inside its 8: the common case on textured
                // content - 83 % of the row steps of the headline pair): the lanes' LAST candidates, then the first ones of those lanes
                float fval = selval;
                unsigned fp = selp;
#pragma unroll
                for (int p = 6; p >= 0; p--) {
                    fval = isv[p] ? EM[p] : fval;
                    fp = isv[p] ? (unsigned)p : fp;
                }
                append(__ballot(ncand != 0u), ncand != 0u, selval, selp);
                append(multi, ncand > 1u, fval, fp);
            } else {
                // three or more candidates in one lane
                tie_rows++;
#pragma unroll
                for (int p = 0; p < 8; p++) {
                    const bool is = isv[p];
                    const unsigned long long bal = __ballot(is);
                    const unsigned slot = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    st[min(is ? slot : ~0u, EIG3_STAGE + (unsigned)lane)] =
                        ((unsigned long long)__float_as_uint(EM[p]) << 32) | (unsigned long long)(rowidx + (unsigned)p);
                    cnt += (unsigned)__popcll(bal);
                }
            }
        }
        // row y: masked copy (the next step's candidates) and the running maximum
        const uint32_t mk[2] = {(mkraw.x | mask_or) & colmask[0], (mkraw.y | mask_or) & colmask[1]};
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const bool counts = ((mk[p >> 2] >> (8 * (p & 3))) & 0xffu) != 0u;
            EM[p] = counts ? E[C][p] : -INFINITY;
            best = fmaxf(best, EM[p]);
        }
    };

    // ---- general step: both windows reloaded from their (mirrored) source rows; used for the warm-up and near the top / bottom
    auto entering3 = [&](int m, row4 (&win)[3], auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        const int r = km_reflect101(m, H);
        win[PH] = unpack(load8(src, km_reflect101(r - 1, H)));
        win[(PH + 1) % 3] = unpack(load8(src, r));
        win[(PH + 2) % 3] = unpack(load8(src, km_reflect101(r + 1, H)));
    };
    auto clamp_row = [&](int r) { return min(max(r, 0), H - 1); };
    auto general_step = [&](int m, auto ph_tag) {
        const int step = m - m_first;
        entering3(m, LW, ph_tag);
        const bool trail = step >= BLOCK;
        if (trail) entering3(m - BLOCK, TW, ph_tag);
        const bool out = step >= BLOCK - 1;
        const uint2 mkraw = load8(mptr, clamp_row(m - Rr));
        compute(ph_tag, m, mkraw, trail, out, out && (m - Rr) >= ye0 + 2);
    };
    int m = m_first;
    auto general_until = [&](int m_end, bool align3) {
        // align3: stop early only at a step index that is a multiple of 3 (the unrolled interior loop starts in phase 0)
        for (; m <= m_last; m++) {
            const int step = m - m_first;
            if (m > m_end && (!align3 || step % 3 == 0)) break;
            if ((step & 7) == 0) refresh_threshold(m == m_first);
            switch (step % 3) {
            case 0: general_step(m, std::integral_constant<int, 0>{}); break;
            case 1: general_step(m, std::integral_constant<int, 1>{}); break;
            default: general_step(m, std::integral_constant<int, 2>{}); break;
            }
            flush_if(EIG3_FLUSH_AT);
        }
    };
    // interior: rows m-1 .. m+1 (and m-BLOCK-1 .. m-BLOCK+1 once the trailing window is in use) plain, 3 rows prefetched.  An item
    // whose rows all lie below the top border enters the sliding march after three steps and runs its warm-up there (the
    // general steps wait for their loads one by one: sixteen of them cost a tenth of a 96-row item)
    const bool early = m_first >= 1 && BLOCK >= 5;   // every trailing row inside the image; the trail slides in before its first use
    const int mi_lo = early ? m_first + 1 : max(m_first + BLOCK + 1, BLOCK + 1), mi_hi = min(m_last, H - 2 - 3);
    general_until(min(mi_lo - 1, m_last), true);
    if (m <= mi_hi && m + 2 <= mi_hi) {
        uint2 ql[3], qt[3], qm[3];                   // static FIFO slots: slot k serves step m + k, refilled for m + k + 3
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ql[k] = load8(src, m + k + 1);
            qt[k] = load8(src, max(m + k - BLOCK + 1, 0));      // (clamped rows are loaded before the trail / the mask are in use)
            qm[k] = load8(mptr, max(m + k - Rr, 0));
        }
        // warm-up groups: not every part of a step is active yet
        while (m + 2 <= mi_hi && m - m_first < BLOCK + 1) {
            e3_for<3>([&](auto kt) {
                constexpr int K = decltype(kt)::value;
                const int step = m + K - m_first;
                LW[(K + 2) % 3] = unpack(ql[K]);
                TW[(K + 2) % 3] = unpack(qt[K]);
                compute(kt, m + K, qm[K], step >= BLOCK, step >= BLOCK - 1, step >= BLOCK - 1 && m + K - Rr >= ye0 + 2);
                ql[K] = load8(src, m + K + 3 + 1);
                qt[K] = load8(src, max(m + K + 3 - BLOCK + 1, 0));
                qm[K] = load8(mptr, max(m + K + 3 - Rr, 0));
            });
            m += 3;
        }
        while (m + 2 <= mi_hi) {
            flush_if(EIG3_FLUSH_AT);
            refresh_threshold(false);
            const int seg_end = min(mi_hi, m + EIG3_SEG * 3 - 1);
            for (; m + 2 <= seg_end; m += 3) {
                e3_for<3>([&](auto kt) {
                    constexpr int K = decltype(kt)::value;
                    LW[(K + 2) % 3] = unpack(ql[K]);
                    TW[(K + 2) % 3] = unpack(qt[K]);
                    compute(kt, m + K, qm[K], true, true, true);
                    ql[K] = load8(src, m + K + 3 + 1);
                    qt[K] = load8(src, m + K + 3 - BLOCK + 1);
                    qm[K] = load8(mptr, m + K + 3 - Rr);
                });
            }
        }
    }
    general_until(m_last, false);
    flush_if(0u);
    unsigned key = best > -INFINITY ? e2_key(best) : 0u;
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    if (lane == 0) max_partial[wave_id] = key;
    if (lane == 0 && tie_rows) atomicAdd(&sc->tie_rows, tie_rows);   // (diagnostics: km_klt_stats.tie_rows)
#ifdef KM_DEV
    if (lane == 0 && count_skips) { atomicAdd(&sc->skip_total, (unsigned long long)skip_total); atomicAdd(&sc->skip_hit, (unsigned long long)skip_hit); }
#endif
}

// items [0, n_border): eig2_item on the two border strips (left: columns 0 .. MARGIN-1, right: W-MARGIN .. W-1), rows2 rows each;
// items [n_border, nitems): eig3_item, rowblock-major over nstrips strips of rows3 rows
template <int BLOCK>
__global__ __launch_bounds__(256, 3) void eig3_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, double scale2,
                                                   unsigned *__restrict__ max_partial, int n_border, int rows2, int nstrips, int rows3, int nitems,
                                                   double quality, km_scalars *sc, unsigned long long *__restrict__ keys, size_t cap, unsigned stage_cap2,
                                                   unsigned stage_cap3, int count_skips)
{
    __shared__ int xs_scratch[4][3][128];
    __shared__ unsigned long long stage[4][EIG3_STAGE + 64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned tile;
    if (!km_xcd_tile((unsigned)(nitems + 3) / 4u, tile)) return;
    const int wave_id = (int)tile * 4 + wv;
    if (wave_id >= nitems) { if (lane == 0) max_partial[wave_id] = 0u; return; }
    if (wave_id < n_border) {
        const int rowblock = wave_id >> 1;
        const bool right = wave_id & 1;
        constexpr int ML2 = eig2_geom<BLOCK, true>::ML;
        eig2_item<BLOCK, true>(src, mask, H, W, scale2, nullptr, max_partial, rows2, quality, sc, keys, cap, stage_cap2, wave_id, rowblock,
                               right ? W - EIG3_MARGIN - ML2 : -ML2, right ? W - EIG3_MARGIN : 0, right ? W : EIG3_MARGIN, &xs_scratch[wv][0][0], stage[wv]);
        return;
    }
    const int j = wave_id - n_border;
    const int rowblock = j / nstrips, strip = j - rowblock * nstrips;
    const int xs = min(strip * EIG3_STRIDE, W - 512);
    const int col_lo = EIG3_MARGIN + strip * EIG3_STRIDE, col_hi = min(col_lo + EIG3_STRIDE, W - EIG3_MARGIN);
    const int ye0 = rowblock * rows3, ye1 = min(H - 1, ye0 + rows3 + 1);
    eig3_item<BLOCK>(src, mask, H, W, scale2, max_partial, quality, sc, keys, cap, stage_cap3, wave_id, xs, col_lo, col_hi, ye0, ye1, stage[wv], count_skips);
}

// batched units: one linear item space over the units (unit u owns [item0[u], item0[u + 1]): its border items first, then its strips,
// exactly the items of the single-unit launch); shards, running-threshold words and per-wave maxima are the unit's own
struct e3_units_args {
    const uint8_t *src[KM_UNITS_MAX], *mask[KM_UNITS_MAX];
    unsigned *max_partial[KM_UNITS_MAX];
    km_scalars *sc[KM_UNITS_MAX];
    unsigned long long *keys[KM_UNITS_MAX];
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX], n_border[KM_UNITS_MAX], nstrips[KM_UNITS_MAX];
    int item0[KM_UNITS_MAX + 1];
    int n, rows2, rows3;
};
template <int BLOCK>
__global__ __launch_bounds__(256, 3) void eig3_units_kernel(e3_units_args U, double scale2, double quality, size_t cap, unsigned stage_cap2, unsigned stage_cap3)
{
    __shared__ int xs_scratch[4][3][128];
    __shared__ unsigned long long stage[4][EIG3_STAGE + 64];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = U.item0[U.n];
    unsigned tile;
    if (!km_xcd_tile((unsigned)(total + 3) / 4u, tile)) return;
    const int wave_lin = (int)tile * 4 + wv;
    if (wave_lin >= total) return;
    int u = 0;
    while (u + 1 < U.n && wave_lin >= U.item0[u + 1]) u++;
    const int wave_id = wave_lin - U.item0[u];
    const uint8_t *src = U.src[u], *mask = U.mask[u];
    const int H = U.H[u], W = U.W[u], n_border = U.n_border[u], nstrips = U.nstrips[u];
    unsigned *max_partial = U.max_partial[u];
    km_scalars *sc = U.sc[u];
    unsigned long long *keys = U.keys[u];
    if (wave_id < n_border) {
        const int rowblock = wave_id >> 1;
        const bool right = wave_id & 1;
        constexpr int ML2 = eig2_geom<BLOCK, true>::ML;
        eig2_item<BLOCK, true>(src, mask, H, W, scale2, nullptr, max_partial, U.rows2, quality, sc, keys, cap, stage_cap2, wave_id, rowblock,
                               right ? W - EIG3_MARGIN - ML2 : -ML2, right ? W - EIG3_MARGIN : 0, right ? W : EIG3_MARGIN, &xs_scratch[wv][0][0], stage[wv]);
        return;
    }
    const int j = wave_id - n_border;
    const int rowblock = j / nstrips, strip = j - rowblock * nstrips;
    const int xs = min(strip * EIG3_STRIDE, W - 512);
    const int col_lo = EIG3_MARGIN + strip * EIG3_STRIDE, col_hi = min(col_lo + EIG3_STRIDE, W - EIG3_MARGIN);
    const int ye0 = rowblock * U.rows3, ye1 = min(H - 1, ye0 + U.rows3 + 1);
    eig3_item<BLOCK>(src, mask, H, W, scale2, max_partial, quality, sc, keys, cap, stage_cap3, wave_id, xs, col_lo, col_hi, ye0, ye1, stage[wv]);
}

template <int BLOCK>
int launch_eig3_units(km_ctx *c, km_units &U, double scale2, double quality)
{
    e3_units_args A;
    A.n = U.n; A.rows2 = 64;
    long border_total = 0;
    for (int u = 0; u < U.n; u++) {
        A.src[u] = U.lap_ref[u]; A.mask[u] = U.mask[u]; A.sc[u] = U.sc[u]; A.keys[u] = U.keys[u];
        A.H[u] = U.H[u]; A.W[u] = U.W[u];
        A.nstrips[u] = (U.W[u] - 2 * EIG3_MARGIN + EIG3_STRIDE - 1) / EIG3_STRIDE;
        A.n_border[u] = 2 * ((U.H[u] - 2 + A.rows2 - 1) / A.rows2);
        border_total += A.n_border[u];
    }
    // rows per strip item: what the single-unit launch picks for the LARGEST unit alone (one round of 3 resident waves per SIMD: ~96 rows
    // for a 10980^2 tile), clamped to [48, 96].  A batch is several rounds anyway, and measured at four 10980^2 units the item height, not
    // the number of rounds, decides: 0.357 ms per unit at 96 rows, 0.347 at 64, 0.381 at 128, 0.423 at the ~192 rows a rounds x height
    // cost model prefers (long items drain the last round slowly); 16 units of 5490^2 are flat between 48 and 128 rows.
    (void)border_total;
    int rows3 = 48;
    for (int u = 0; u < U.n; u++) {
        const int r = km_pick_rows(U.H[u] - 2, A.nstrips[u], 5, 1024L * 3 - A.n_border[u], 32, 192);
        rows3 = r > rows3 ? r : rows3;
    }
    rows3 = rows3 > 96 ? 96 : rows3;
    if (const char *e = km_dev_env("KARIOS_HIP_EIG3_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 8192) rows3 = v; }   // tuning override
    A.rows3 = rows3;
    A.item0[0] = 0;
    for (int u = 0; u < U.n; u++) A.item0[u + 1] = A.item0[u] + A.n_border[u] + A.nstrips[u] * ((U.H[u] - 2 + rows3 - 1) / rows3);
    const int total = A.item0[U.n];
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, ((size_t)total + 16) * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    for (int u = 0; u < U.n; u++) {
        A.max_partial[u] = partial + A.item0[u];
        U.eig_partial[u] = A.max_partial[u]; U.eig_npartial[u] = (unsigned)(A.item0[u + 1] - A.item0[u]);
    }
    const unsigned cap2 = c->opt_stage_cap > 0 && c->opt_stage_cap < EIG2_STAGE ? (unsigned)c->opt_stage_cap : (unsigned)EIG2_STAGE;
    const unsigned cap3 = c->opt_stage_cap > 0 && c->opt_stage_cap < EIG3_STAGE ? (unsigned)c->opt_stage_cap : (unsigned)EIG3_STAGE;
    eig3_units_kernel<BLOCK><<<km_xcd_grid((unsigned)(total + 3) / 4u), 256, 0, c->stream>>>(A, scale2, quality, U.capk, cap2, cap3);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

__global__ __launch_bounds__(1024) void eig3_max_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned *out)
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
int launch_eig3(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, double scale2, double quality, km_scalars *sc,
                unsigned long long *d_keys, size_t cap)
{
    const int nstrips = (W - 2 * EIG3_MARGIN + EIG3_STRIDE - 1) / EIG3_STRIDE;
    const int rows2 = 64;
    const int n_border = 2 * ((H - 2 + rows2 - 1) / rows2);
    // 3 waves per SIMD are resident (168 VGPRs); the border items hold a slot each while they run.  Measured at 10980^2: 0.336 ms with
    // 96-row items (2645 + 344 items: one round), 0.341 at 48 (two rounds), 0.373 at 64 (1.4 rounds), 0.424 at 128 (slots left empty)
    int rows3 = km_pick_rows(H - 2, nstrips, 5, 1024L * 3 - n_border, 32, 192);   // (a warm-up row of an item costs about a third of a full row)
    if (const char *e = km_dev_env("KARIOS_HIP_EIG3_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 8192) rows3 = v; }   // tuning override
    const int nrowblocks = (H - 2 + rows3 - 1) / rows3;
    const int nitems = n_border + nstrips * nrowblocks;
    const unsigned ntiles = (unsigned)(nitems + 3) / 4u;
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)ntiles * 4 * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    // "stage_cap" (test knob): usable slots of the per-wave key stage
    const unsigned cap2 = c->opt_stage_cap > 0 && c->opt_stage_cap < EIG2_STAGE ? (unsigned)c->opt_stage_cap : (unsigned)EIG2_STAGE;
    const unsigned cap3 = c->opt_stage_cap > 0 && c->opt_stage_cap < EIG3_STAGE ? (unsigned)c->opt_stage_cap : (unsigned)EIG3_STAGE;
    eig3_kernel<BLOCK><<<km_xcd_grid(ntiles), 256, 0, c->stream>>>(d_src, d_mask, H, W, scale2, partial, n_border, rows2, nstrips, rows3, nitems, quality, sc,
                                                                   d_keys, cap, cap2, cap3, c->opt_eig3_count ? 1 : 0);
    KM_LAUNCH_CHECK(c);
    if (c->eig_defer_max) {          // speculative corner path: kf_rank's first launch takes the maximum of the partials itself
        c->eig_partial = partial; c->eig_npartial = ntiles * 4;
        return KM_OK;
    }
    eig3_max_kernel<<<1, 1024, 0, c->stream>>>(partial, ntiles * 4, &sc->max_eig_key);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

}  // namespace

// Batched units (api_units.hip): the fused pass of every unit in ONE launch; the per-wave maxima stay in U.eig_partial[u] for the
// ranking's first launch.  KM_E_UNSUPPORTED (no message): a unit the 8-px kernel does not cover.
int k3_eig_candidates_units(km_ctx *c, km_units &U, int block, double quality)
{
    if (block < 1 || block > 15 || (block & 1) == 0 || !c->opt_eig3) return KM_E_UNSUPPORTED;
    for (int u = 0; u < U.n; u++)
        if (U.W[u] < 512 || U.H[u] < 2 * block + 8) return KM_E_UNSUPPORTED;
    const double scale = 1.0 / (4.0 * (double)block * 255.0), s2 = scale * scale;
    switch (block) {
#define KM_EIG3_CASE(B) case B: return launch_eig3_units<B>(c, U, s2, quality);
        KM_EIG3_CASE(1) KM_EIG3_CASE(3) KM_EIG3_CASE(5) KM_EIG3_CASE(7) KM_EIG3_CASE(9) KM_EIG3_CASE(11) KM_EIG3_CASE(13) KM_EIG3_CASE(15)
#undef KM_EIG3_CASE
    default: return KM_E_UNSUPPORTED;
    }
}

// Fused minimum-eigenvalue + candidate pass, 8 pixels per lane.  Same contract as k2_eig_candidates (k_eig2.hip), which
// forwards here; KM_E_UNSUPPORTED (no message) when the case is not covered (narrow images, even block sizes).
int k3_eig_candidates(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, double quality, km_scalars *sc,
                      unsigned long long *d_keys, size_t cap)
{
    if (block < 1 || block > 15 || (block & 1) == 0) return KM_E_UNSUPPORTED;
    if (W < 512 || H < 2 * block + 8) return KM_E_UNSUPPORTED;
    const double scale = 1.0 / (4.0 * (double)block * 255.0), s2 = scale * scale;
    switch (block) {
#define KM_EIG3_CASE(B) case B: return launch_eig3<B>(c, d_src, d_mask, H, W, s2, quality, sc, d_keys, cap);
        KM_EIG3_CASE(1) KM_EIG3_CASE(3) KM_EIG3_CASE(5) KM_EIG3_CASE(7) KM_EIG3_CASE(9) KM_EIG3_CASE(11) KM_EIG3_CASE(13) KM_EIG3_CASE(15)
#undef KM_EIG3_CASE
    default: return KM_E_UNSUPPORTED;
    }
}
