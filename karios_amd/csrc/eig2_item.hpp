// Shared by k_eig2.hip and k_eig3.hip: the 2-pixels-per-lane marching item of the minimum-eigenvalue map / the fused
// minimum-eigenvalue + candidate pass (see k_eig2.hip for the description) as a device function: one wavefront, one
// 128-column strip, one block of rows.  k_eig3.hip runs it on the strips that touch the left / right image border.
#pragma once
#include "common.hpp"

#include <type_traits>

namespace {

template <int BLOCK, bool EMIT> struct eig2_geom {
    static constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L, XM = EMIT ? 1 : 0;
    static constexpr int ML = (L + 1 + XM + 1) & ~1, STRIDE = (128 - ML - (Rr + 1 + XM)) & ~1;   // even margins / stride: 2-byte aligned loads when W is even
};

typedef short e2_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ e2_s2 e2s(uint32_t v) { return __builtin_bit_cast(e2_s2, v); }
__device__ __forceinline__ uint32_t e2u(e2_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ unsigned e2_key(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float e2_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int e2_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned e2_opaque(unsigned x)
{
    asm volatile("" : "+v"(x));
    return x;
}
// lo16(a) * lo16(b) + c  (v_mad_i32_i16: the halves are selected by the instruction, no extraction)
__device__ __forceinline__ int e2_mad_lo(uint32_t a, uint32_t b, int c)
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int e2_lane_m1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }   // lane-1, 0 at lane 0
__device__ __forceinline__ int e2_lane_p1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }   // lane+1, 0 at lane 63
__device__ __forceinline__ int e2_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += e2_dpp<0x142, 0xa>(v);                                       // row_bcast:15 -> rows 1,3
    v += e2_dpp<0x143, 0xc>(v);                                       // row_bcast:31 -> rows 2,3
    return v;
}
// correctly rounded float32 square root for x == 0 or x >= 2^-96 (see sqrt_rn_normal in k_dense.hip)
__device__ __forceinline__ float e2_sqrt(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const float r_dn = __int_as_float(__float_as_int(r) - 1), r_up = __int_as_float(__float_as_int(r) + 1);
    const float e_dn = __builtin_fmaf(-r_dn, r, x), e_up = __builtin_fmaf(-r_up, r, x);
    float res = e_dn <= 0.f ? r_dn : r;
    res = e_up > 0.f ? r_up : res;
    return res;
}

#ifndef EIG2_PF
#define EIG2_PF 3   // rows in flight per stream
#endif

#define EIG2_STAGE 512     // EMIT: candidate keys per wave in LDS (+ one dummy slot per lane behind them)
#define EIG2_FLUSH_AT 128  // EMIT: flush between row segments once this many keys are staged
#ifndef EIG2_SEG
#define EIG2_SEG 3         // EMIT: row groups (of EIG2_PF rows) between two looks at the stage / the running threshold
#endif

// EMIT = false: writes the eig map.  EMIT = true: K3 + K4 fused - the map is never written; three lambda rows stay in
// registers and every pixel that is a 3x3 local maximum, lies off the image border, passes the mask and exceeds a RUNNING
// lower bound of the final threshold is appended to a per-wave LDS stage (flushed between row segments to the sharded key
// buffer; a segment that would overflow the stage raises sc->pad0 and the caller falls back to map + candidate kernel).
// The exact threshold is applied afterwards by the top-K pre-filter (k_select.hip tk_*).
template <int BLOCK, bool EMIT>
__device__ __forceinline__ void eig2_item(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, double scale2,
                                          float *__restrict__ eig, unsigned *__restrict__ max_partial, int rows_per_item, double quality, km_scalars *sc,
                                          unsigned long long *__restrict__ keys, size_t cap, unsigned stage_cap,
                                          int wave_id, int rowblock, int xs /* image column of strip pixel 0 */, int col_lo, int col_hi /* EMIT: candidate columns */,
                                          int *xsw /* [3][128], border strips only: pixel-prefix sums of the three products */,
                                          unsigned long long *st /* EMIT: [EIG2_STAGE + 64] */)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L;
    constexpr int XM = EMIT ? 1 : 0;                 // EMIT: one more margin pixel per side (the candidates' neighbours)
    constexpr int ML = eig2_geom<BLOCK, EMIT>::ML, STRIDE = eig2_geom<BLOCK, EMIT>::STRIDE;
    constexpr int PF = EIG2_PF;
    const int lane = threadIdx.x & 63;
    const int c0 = xs + 2 * lane;                    // image column of this lane's pixel 0
    const bool border = xs < 0 || xs + 127 > W - 1;  // wave-uniform: some strip pixel lies outside the image
    // map: lambda rows [ye0, ye1] = the item's rows; EMIT: candidate rows [ye0 + 1, ye1 - 1] inside 1 .. H-2
    const int ye0 = rowblock * rows_per_item, ye1 = EMIT ? min(H - 1, ye0 + rows_per_item + 1) : min(H, ye0 + rows_per_item) - 1;
    const int m_first = ye0 - L, m_last = ye1 + Rr;  // product rows marched (may lie outside: mirrored)

    uint32_t inimg_pair = 0;                         // 0xffff per pixel whose column lies inside the image
    bool out_px[2], cand_px[2];                      // pixel is an output of this strip and inside the image / may emit a candidate
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int i = 2 * lane + p, c = c0 + p;
        const bool in = c >= 0 && c <= W - 1;
        inimg_pair |= in ? (p ? 0xffff0000u : 0x0000ffffu) : 0u;
        out_px[p] = in && i >= ML - XM && i < ML + STRIDE + XM;       // lambda is meaningful here (EMIT: the candidates' neighbours too)
        cand_px[p] = i >= ML && i < ML + STRIDE && c >= 1 && c <= W - 2 && c >= col_lo && c < col_hi;
    }
    // lanes (partly) outside the image load the nearest two in-image columns; a byte permute puts REFLECT_101 values where
    // the Sobel needs them (columns -1 and W; other outside columns never matter)
    const int c_load = min(max(c0, 0), W - 2);
    uint32_t load_sel = 0x0c0c0000u;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int idx = km_reflect101(c0 + p, W) - c_load;
        load_sel |= (uint32_t)((idx >= 0 && idx <= 1) ? idx : 0) << (8 * p);
    }
    const uint8_t *mptr = mask ? mask : src;         // no mask: the loads still happen (fixed set of memory operations per row)
    const uint32_t mask_or = mask ? 0u : 0x0101u;
    const bool w_even = (W & 1) == 0 && ((uintptr_t)eig % 8 == 0);   // 8-byte aligned float2 stores
    auto clamp_row = [&](int r) { return min(max(r, 0), H - 1); };

    auto run = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;   // interior strip: no column border handling at all
    auto load_raw = [&](const uint8_t *base, int r) -> uint32_t {   // two bytes of row r (inside the image)
        const uint8_t *rowp = base + (size_t)r * W;   // wave-uniform row base + opaque 32-bit lane offset: no per-lane 64-bit arithmetic
        unsigned short v;
        __builtin_memcpy(&v, rowp + e2_opaque((unsigned)(FAST ? c0 : c_load)), 2);
        return (uint32_t)v;
    };
    auto unpack_src = [&](uint32_t w) -> uint32_t {      // bytes (b0, b1) -> 16-bit pair
        if (!FAST) w = __builtin_amdgcn_perm(w, w, load_sel) & 0xffffu;
        return __builtin_amdgcn_perm(0u, w, 0x0c010c00u);
    };
    struct win3 { uint32_t a0, a1, a2; };
    auto window_reload = [&](int m, win3 &w) {
        const int r = km_reflect101(m, H);
        w.a0 = unpack_src(load_raw(src, km_reflect101(r - 1, H)));
        w.a1 = unpack_src(load_raw(src, r));
        w.a2 = unpack_src(load_raw(src, km_reflect101(r + 1, H)));
    };
    // marching from product row m - 1 to m changes the window by at most one source row (see k_eigc.hip)
    auto entering_row = [&](int m) -> int {
        if (m >= 1 && m <= H - 2) return m + 1;
        if (m < 0) return -m - 1;
        if (m >= H) return 2 * (H - 1) - m - 1;
        return -1;
    };
    auto window_step = [&](int m, win3 &w, uint32_t entering) {
        const uint32_t e = unpack_src(entering);
        if (m >= 1 && m <= H - 2) { w.a0 = w.a1; w.a1 = w.a2; w.a2 = e; }
        else if (m == 0) { const uint32_t t = w.a0; w.a0 = w.a1; w.a2 = w.a1; w.a1 = t; }
        else if (m == H - 1) { w.a0 = w.a1; w.a1 = w.a2; w.a2 = w.a0; }
        else { w.a2 = w.a1; w.a1 = w.a0; w.a0 = e; }
    };
    auto derivs = [&](const win3 &w, uint32_t &dx, uint32_t &dy) {
        const e2_s2 t0 = e2s(w.a0) + e2s(w.a2) + e2s(w.a1) + e2s(w.a1);      // column sums (for dx)
        const e2_s2 t1 = e2s(w.a2) - e2s(w.a0);                              // column differences (for dy)
        const uint32_t t0u = e2u(t0), t1u = e2u(t1);
        const uint32_t l0 = (uint32_t)e2_lane_m1((int)t0u), r0 = (uint32_t)e2_lane_p1((int)t0u);
        const uint32_t l1 = (uint32_t)e2_lane_m1((int)t1u), r1 = (uint32_t)e2_lane_p1((int)t1u);
        const uint32_t t0_m = __builtin_amdgcn_alignbyte(t0u, l0, 2), t0_p = __builtin_amdgcn_alignbyte(r0, t0u, 2);   // (x-1, x), (x+1, x+2)
        const uint32_t t1_m = __builtin_amdgcn_alignbyte(t1u, l1, 2), t1_p = __builtin_amdgcn_alignbyte(r1, t1u, 2);
        dx = e2u(e2s(t0_p) - e2s(t0_m));
        dy = e2u(e2s(t1_m) + t1 + t1 + e2s(t1_p));
        if (!FAST) { dx &= inimg_pair; dy &= inimg_pair; }   // products of outside columns are 0
    };
    // vertical box sums of the three products: VP = pixel 0 + pixel 1 of the lane's pair (one v_dot2_i32_i16 per product and
    // row), V0 = pixel 0 alone (one v_mad_i32_i16 on the low halves); the windows need exactly these two
    int VP[3] = {0, 0, 0}, V0[3] = {0, 0, 0};
    auto accumulate = [&](uint32_t dx, uint32_t dy, bool subtract) {
        const e2_s2 x = e2s(dx), y = e2s(dy);
        const e2_s2 sx = subtract ? -x : x, sy = subtract ? -y : y;
        VP[0] = __builtin_amdgcn_sdot2(x, sx, VP[0], false);
        VP[1] = __builtin_amdgcn_sdot2(x, sy, VP[1], false);
        VP[2] = __builtin_amdgcn_sdot2(y, sy, VP[2], false);
        V0[0] = e2_mad_lo(dx, e2u(sx), V0[0]);
        V0[1] = e2_mad_lo(dx, e2u(sy), V0[1]);
        V0[2] = e2_mad_lo(dy, e2u(sy), V0[2]);
    };
    // horizontal window W(i) = S(i + Rr) - S(i - L - 1), S = inclusive pixel prefix over the strip (i = 2*lane + p)
    constexpr int UO[2] = {(0 + Rr) / 2, (1 + Rr) / 2}, UJ[2] = {(0 + Rr) % 2, (1 + Rr) % 2};
    constexpr int LO[2] = {-((L + 1 - 0 + 1) / 2), -((L + 1 - 1 + 1) / 2)};
    constexpr int LJ[2] = {((0 - L - 1) % 2 + 2) % 2, ((1 - L - 1) % 2 + 2) % 2};
    auto bperm_from = [&](int lane_off, int v) { return __builtin_amdgcn_ds_bpermute(((lane + lane_off) & 63) * 4, v); };
    auto windows = [&](int (&Wd)[3][2]) {
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const int A = e2_scan(VP[q]), Ap = e2_lane_m1(A);
            const int X[2] = {Ap + V0[q], A};
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int up = UO[p] == 0 ? X[UJ[p]] : bperm_from(UO[p], X[UJ[p]]);
                const int lw = LO[p] == 0 ? X[LJ[p]] : bperm_from(LO[p], X[LJ[p]]);
                Wd[q][p] = up - lw;
            }
            if (!FAST) {
                // box filter's REFLECT_101 on the product images: add the products mirrored in from outside
                *(int2 *)(xsw + q * 128 + 2 * lane) = make_int2(X[0], X[1]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int c = c0 + p;
                    int ia = -1, ib = -1;
                    if (c >= 0 && c < L) { ia = (L - c) - xs; ib = 0 - xs; }                                  // S'(L - c) - S'(0)
                    else if (c <= W - 1 && c + Rr > W - 1) { ia = (W - 2) - xs; ib = (2 * W - 3 - c - Rr) - xs; }  // S'(W-2) - S'(2(W-1) - c - Rr - 1)
                    if (ia >= 0) Wd[q][p] += xsw[q * 128 + ia] - xsw[q * 128 + ib];
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
        return __fsub_rn(__fadd_rn(a, cc), e2_sqrt(sq));
    };

    float best = -INFINITY;
    // ---- EMIT: candidate staging and the running threshold (see k_eigc.hip for the derivation)
    unsigned cnt = 0;
    const unsigned shard = (unsigned)wave_id % KM_NSHARD;
    const size_t cap_s = cap / KM_NSHARD;
    auto flush_if = [&](unsigned threshold) {
        if (cnt <= threshold) return;
        if (cnt > stage_cap) { if (lane == 0) atomicOr(&sc->pad0, 1u); cnt = min(cnt, (unsigned)EIG2_STAGE); }   // stage_cap = EIG2_STAGE unless a test shrank it
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sc->shard_cnt[shard], cnt);
        base = __shfl(base, 0);
        for (unsigned i = lane; i < cnt; i += 64)
            if ((size_t)base + i < cap_s) keys[shard * cap_s + base + i] = st[i];
        cnt = 0;
    };
    float thr_run = 0.f;
    unsigned published = 0u;
    // running lower bound of the final threshold: quality * max(own wave so far, global running maximum).  The global
    // words live at device scope (all XCDs): every access costs microseconds and same-address accesses serialise (158 000
    // loads of ONE word once cost 1.6 ms), so there are 64 of them (any subset maximum is a valid lower bound), each is read
    // ONCE per item and written only on a clear improvement;
    // between segments only the wave's own maximum is refreshed (register shuffles).
    unsigned gk_seen = 0u;
    auto refresh_threshold = [&](bool global) {
        unsigned wk = best > -INFINITY ? e2_key(best) : 0u;
        for (int o = 32; o > 0; o >>= 1) wk = max(wk, (unsigned)__shfl_xor((int)wk, o));
        if (global) {
            gk_seen = __hip_atomic_load(&sc->run_max_shard[wave_id & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (published == 0u && wk > gk_seen && (gk_seen == 0u || e2_unkey(wk) > e2_unkey(gk_seen) * 1.0625f)) {
            if (lane == 0) atomicMax(&sc->run_max_shard[wave_id & 63], wk);   // at most one publication per item, only >= 1/16 above what was seen
            published = 1u;
        }
        const unsigned mk2 = max(wk, gk_seen);
        thr_run = mk2 ? (float)__dmul_rn((double)e2_unkey(mk2), quality) : 0.f;
    };
    float e2r[2] = {0.f, 0.f}, e1r[2] = {0.f, 0.f};   // lambda rows y-2, y-1
    uint32_t mk1 = 0;                                  // mask bytes of row y-1
    win3 lead, trail;
    // one marching step = product row m; nl / nt = source rows entering the lead / trail windows, mkraw = mask bytes of row m - Rr
    auto row_step = [&](int m, uint32_t nl, uint32_t nt, uint32_t mkraw, auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        const int step = m - m_first;
        uint32_t dx, dy;
        if (INTERIOR) {
            lead.a0 = lead.a1; lead.a1 = lead.a2; lead.a2 = unpack_src(nl);
            trail.a0 = trail.a1; trail.a1 = trail.a2; trail.a2 = unpack_src(nt);
        } else if (step > 0) {
            window_step(m, lead, nl);
            window_step(m - BLOCK, trail, nt);
        }
        derivs(lead, dx, dy);
        accumulate(dx, dy, false);
        if (INTERIOR || step >= BLOCK) {
            derivs(trail, dx, dy);
            accumulate(dx, dy, true);
        }
        if (!INTERIOR && step < BLOCK - 1) return;
        const int y = m - Rr;                            // lambda row completed by this step (ye0 <= y <= ye1)
        const uint32_t mk = (FAST ? mkraw : (__builtin_amdgcn_perm(mkraw, mkraw, load_sel) & 0xffffu)) | mask_or;
        int Wd[3][2];
        windows(Wd);
        float e[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            e[p] = lambda_min(Wd[0][p], Wd[1][p], Wd[2][p]);
            const bool counts = out_px[p] && ((mk >> (8 * p)) & 0xffu) != 0u;
            best = fmaxf(best, counts ? e[p] : -INFINITY);
        }
        if constexpr (!EMIT) {
            float *orow = eig + (size_t)y * W;
            if (out_px[0] || out_px[1]) {              // margins and stride are even: the two pixels of a FAST lane go together
                if (FAST && w_even) {
                    *(float2 *)(orow + (unsigned)c0) = make_float2(e[0], e[1]);
                } else {
#pragma unroll
                    for (int p = 0; p < 2; p++) if (out_px[p]) orow[c0 + p] = e[p];
                }
            }
        } else {
            if (INTERIOR || y >= ye0 + 2) {
                // candidate test of row y-1 against lambda rows y-2, y-1, y
                const float m3[2] = {fmaxf(fmaxf(e2r[0], e1r[0]), e[0]), fmaxf(fmaxf(e2r[1], e1r[1]), e[1])};
                const float m3l = __int_as_float(e2_lane_m1(__float_as_int(m3[1]))), m3r = __int_as_float(e2_lane_p1(__float_as_int(m3[0])));
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const float left = p == 0 ? m3l : m3[0], right = p == 1 ? m3r : m3[1];
                    const float nb = fmaxf(fmaxf(left, right), fmaxf(e2r[p], e[p]));
                    const bool is = cand_px[p] && e1r[p] > thr_run && e1r[p] != 0.f && e1r[p] >= nb && ((mk1 >> (8 * p)) & 0xffu) != 0u;
                    const unsigned long long bal = __ballot(is);
                    const unsigned slot = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    // unconditional store (no branch in the row body): non-candidates go to the lane's dummy slot, an overflowing
                    // candidate to some dummy slot (min: never beyond the array; the overflow flag is raised by flush_if)
                    st[min(is ? slot : ~0u, EIG2_STAGE + (unsigned)lane)] =
                        ((unsigned long long)__float_as_uint(e1r[p]) << 32) | (unsigned long long)((unsigned)(y - 1) * (unsigned)W + (unsigned)(c0 + p));
                    cnt += (unsigned)__popcll(bal);
                }
            }
            e2r[0] = e1r[0]; e2r[1] = e1r[1]; e1r[0] = e[0]; e1r[1] = e[1];
            mk1 = mk;
        }
    };

    window_reload(m_first, lead);
    window_reload(m_first - BLOCK, trail);
    const int mi_lo = max(m_first + BLOCK + (EMIT ? 1 : 0), BLOCK + 1), mi_hi = min(m_last, H - 2 - PF);
    int m = m_first;
    auto general_until = [&](int m_end) {
        for (; m <= m_end; m++) {
            const uint32_t nl = load_raw(src, max(entering_row(m), 0)), nt = load_raw(src, max(entering_row(m - BLOCK), 0));
            const uint32_t mkraw = load_raw(mptr, clamp_row(m - Rr));
            if (EMIT && ((m - m_first) & 7) == 0) refresh_threshold(m == m_first);
            row_step(m, nl, nt, mkraw, std::false_type{});
            if (EMIT) flush_if(EIG2_FLUSH_AT);
        }
    };
    general_until(min(mi_lo - 1, m_last));
    if (m <= mi_hi) {
        uint32_t ql[PF], qt[PF], qm[PF];             // static FIFO slots: slot k serves step m + k, refilled for m + k + PF
#pragma unroll
        for (int k = 0; k < PF; k++) {
            ql[k] = load_raw(src, m + k + 1);
            qt[k] = load_raw(src, m + k - BLOCK + 1);
            qm[k] = load_raw(mptr, m + k - Rr);
        }
        while (m + PF - 1 <= mi_hi) {
            // between segments: the only places of the interior march with conditional global memory traffic
            if (EMIT) { flush_if(EIG2_FLUSH_AT); refresh_threshold(false); }
            const int seg_end = EMIT ? min(mi_hi, m + EIG2_SEG * PF - 1) : mi_hi;
            for (; m + PF - 1 <= seg_end; m += PF) {
#pragma unroll
                for (int k = 0; k < PF; k++) {
                    row_step(m + k, ql[k], qt[k], qm[k], std::true_type{});
                    ql[k] = load_raw(src, m + k + PF + 1);
                    qt[k] = load_raw(src, m + k + PF - BLOCK + 1);
                    qm[k] = load_raw(mptr, m + k + PF - Rr);
                }
            }
        }
    }
    general_until(m_last);
    if (EMIT) flush_if(0u);
    unsigned key = best > -INFINITY ? e2_key(best) : 0u;
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    if (lane == 0) {
        max_partial[wave_id] = key;
        // (no final publication: the exact maximum comes from max_partial; the running key only has to be a lower bound)
    }
    };  // run
    if (!border) run(std::true_type{});
    else run(std::false_type{});
}

}  // namespace
