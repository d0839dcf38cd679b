// K3, two pixels per lane: minimum-eigenvalue map of cv2.goodFeaturesToTrack (reference call site
// karios/matcher/klt.py:120, 494; algorithm SURVEY.md App. A.2: Sobel 3x3 REFLECT_101 -> products -> box filter
// blockSize x blockSize with its own REFLECT_101 border on the product images -> lambda_min) + its maximum over the mask.
//
// Same arithmetic as eig_march_kernel (k_dense.hip) - exact integer sums, fp64 scaling, individually rounded float32
// operations, correctly rounded sqrt - in the register-light formulation found with the fused experiment (k_eigc.hip):
//   * one wavefront owns 128 columns (2 per lane, one 16-bit load per row) and marches down its rows;
//   * vertical box sum  V += P(row entering) - P(row leaving): the products of BOTH rows are recomputed from the source
//     (two 3-row windows in registers; the trailing rows come from L2), so there is no blockSize-deep ring - six
//     accumulators per lane whatever the block size, ~60 VGPRs, 8 waves per SIMD;
//   * derivatives in packed 16-bit arithmetic on the pixel pair, products accumulated with v_mad_i32_i24;
//   * horizontal box sum = difference of two pixel-prefix sums: one DPP wave scan per product + 4 ds_bpermute;
//   * the source / mask rows travel through register FIFOs with static slots (unrolled by the FIFO depth).
// Image borders: lanes (partly) outside load their nearest in-image pixels and a byte permute supplies the Sobel's
// REFLECT_101 values; the products of outside columns are zeroed and the box filter's own REFLECT_101 terms are added from
// the pixel-prefix sums through an LDS scratch (border strips only); mirrored rows are handled by the window stepping.
#include <cstring>
#include <string.h>


#include "eig2_item.hpp"

namespace {

template <int BLOCK, bool EMIT>
__global__ __launch_bounds__(256) void eig2_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, double scale2,
                                                   float *__restrict__ eig, unsigned *__restrict__ max_partial, int nstrips, int rows_per_item,
                                                   int nitems, double quality, km_scalars *sc, unsigned long long *__restrict__ keys, size_t cap, unsigned stage_cap)
{
    __shared__ int xs_scratch[4][3][128];            // border strips only: pixel-prefix sums of the three products
    __shared__ unsigned long long stage[EMIT ? 4 : 1][EMIT ? EIG2_STAGE + 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned tile;
    if (!km_xcd_tile((unsigned)(nitems + 3) / 4u, tile)) return;
    const int wave_id = (int)tile * 4 + wv;
    if (wave_id >= nitems) { if (lane == 0) max_partial[wave_id] = 0u; return; }
    const int rowblock = wave_id / nstrips, strip = wave_id - rowblock * nstrips;
    eig2_item<BLOCK, EMIT>(src, mask, H, W, scale2, eig, max_partial, rows_per_item, quality, sc, keys, cap, stage_cap, wave_id, rowblock,
                           strip * eig2_geom<BLOCK, EMIT>::STRIDE - eig2_geom<BLOCK, EMIT>::ML, 0, W, &xs_scratch[wv][0][0], stage[EMIT ? wv : 0]);
}

__global__ __launch_bounds__(1024) void eig2_max_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned *out)
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

template <int BLOCK, bool EMIT>
int launch_eig2(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, double scale2, float *d_eig, unsigned *d_max_key,
                double quality, km_scalars *sc, unsigned long long *d_keys, size_t cap)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L, XM = EMIT ? 1 : 0, ML = (L + 1 + XM + 1) & ~1, STRIDE = (128 - ML - (Rr + 1 + XM)) & ~1;
    // map: strips tile the columns 0 .. W-1, items the rows 0 .. H-1; EMIT: candidate columns / rows 1 .. W-2 / 1 .. H-2
    const int nstrips = EMIT ? (W - 1 + STRIDE - 1) / STRIDE : (W + STRIDE - 1) / STRIDE;
    // short items (several rounds of resident waves) balance best: map 0.352 ms at 48 rows, 0.38 at 128, 0.49 at 384; fused 0.464 at 64
    int rows = EMIT ? 64 : 48;
    if (const char *e = km_dev_env("KARIOS_HIP_EIG2_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 8192) rows = v; }   // tuning override
    const int nrowblocks = EMIT ? (H - 2 + rows - 1) / rows : (H + rows - 1) / rows;
    const int nitems = nstrips * nrowblocks;
    const unsigned ntiles = (unsigned)(nitems + 3) / 4u;
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)ntiles * 4 * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    const unsigned stage_cap = c->opt_stage_cap > 0 && c->opt_stage_cap < EIG2_STAGE ? (unsigned)c->opt_stage_cap : (unsigned)EIG2_STAGE;
    eig2_kernel<BLOCK, EMIT><<<km_xcd_grid(ntiles), 256, 0, c->stream>>>(d_src, d_mask, H, W, scale2, d_eig, partial, nstrips, rows, nitems, quality, sc,
                                                                         d_keys, cap, stage_cap);
    KM_LAUNCH_CHECK(c);
    eig2_max_kernel<<<1, 1024, 0, c->stream>>>(partial, ntiles * 4, d_max_key);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

}  // namespace

// Minimum-eigenvalue map + masked maximum, 2 pixels per lane.  KM_E_UNSUPPORTED (no message) when the case is not covered.
int k2_min_eigen(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, float *d_eig, unsigned *d_max_key)
{
    if (block < 1 || block > 15) return KM_E_UNSUPPORTED;
    if (!(W >= 2 * block + 8 && H >= 2 * block + 8)) return KM_E_UNSUPPORTED;   // mirrored columns / rows stay near their border
    const double scale = 1.0 / (4.0 * (double)block * 255.0), s2 = scale * scale;
    switch (block) {
#define KM_EIG2_CASE(B) case B: return launch_eig2<B, false>(c, d_src, d_mask, H, W, s2, d_eig, d_max_key, 0.0, nullptr, nullptr, 0);
        KM_EIG2_CASE(1) KM_EIG2_CASE(2) KM_EIG2_CASE(3) KM_EIG2_CASE(4) KM_EIG2_CASE(5) KM_EIG2_CASE(7) KM_EIG2_CASE(9) KM_EIG2_CASE(11)
        KM_EIG2_CASE(13) KM_EIG2_CASE(15)
#undef KM_EIG2_CASE
    default: return KM_E_UNSUPPORTED;
    }
}

// Fused minimum-eigenvalue + candidate pass, 2 pixels per lane (no eig map).  Expects sc->run_max_key, sc->pad0 and
// sc->shard_cnt[] zeroed (`rezero` does it).  KM_E_UNSUPPORTED (no message) when the case is not covered.
int k2_eig_candidates(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, double quality, km_scalars *sc,
                      unsigned long long *d_keys, size_t cap, bool rezero)
{
    if (block < 1 || block > 15) return KM_E_UNSUPPORTED;
    if (!(W >= 2 * block + 8 && H >= 2 * block + 8)) return KM_E_UNSUPPORTED;
    if (rezero) {
        KM_HIP(c, hipMemsetAsync(&sc->run_max_key, 0, (2 + KM_NSHARD) * sizeof(unsigned), c->stream));   // run_max_key, pad, shard counters
        KM_HIP(c, hipMemsetAsync(sc->run_max_shard, 0, sizeof sc->run_max_shard, c->stream));
    }
    if (c->opt_eig3) {   // wide images: 8 pixels per lane (k_eig3.hip), the 2-px item only on the two border strips
        const int r = k3_eig_candidates(c, d_src, d_mask, H, W, block, quality, sc, d_keys, cap);
        if (r != KM_E_UNSUPPORTED) return r;
    }
    const double scale = 1.0 / (4.0 * (double)block * 255.0), s2 = scale * scale;
    switch (block) {
#define KM_EIG2_CASE(B) case B: return launch_eig2<B, true>(c, d_src, d_mask, H, W, s2, nullptr, &sc->max_eig_key, quality, sc, d_keys, cap);
        KM_EIG2_CASE(1) KM_EIG2_CASE(3) KM_EIG2_CASE(5) KM_EIG2_CASE(7) KM_EIG2_CASE(9) KM_EIG2_CASE(11) KM_EIG2_CASE(13) KM_EIG2_CASE(15)
#undef KM_EIG2_CASE
    default: return KM_E_UNSUPPORTED;
    }
}
