// Dense per-pixel kernels of the KARIOS matching path (gfx950):
//   K1  NaN-aware min/max reduction            (reference klt.py:46)
//   K2  uint8 stretch -> Laplacian -> auto mask (klt.py:42-49, 268-273, 433-434)
//   K3  Sobel -> structure tensor -> box sum -> min eigenvalue (+ masked max)
//   K4  threshold + 3x3 local maxima + mask -> candidate keys
//   K6  pyrDown 5x5
//   K11 integer image shift                     (core/image.py:70-101)
// All are HBM-bound stencils / reductions: LDS tiles with halo, integer-exact
// arithmetic, no MFMA.
#include "common.hpp"

// occupancy targets of the marching kernels (waves per SIMD; 0 = leave it to the register allocator)
#ifndef KM_EIGM_WAVES
#define KM_EIGM_WAVES 0
#endif
#ifndef KM_LAPM_WAVES
#define KM_LAPM_WAVES 0
#endif
#if KM_EIGM_WAVES
#define KM_EIGM_OCC __attribute__((amdgpu_waves_per_eu(KM_EIGM_WAVES, KM_EIGM_WAVES)))
#else
#define KM_EIGM_OCC
#endif
#if KM_LAPM_WAVES
#define KM_LAPM_OCC __attribute__((amdgpu_waves_per_eu(KM_LAPM_WAVES, KM_LAPM_WAVES)))
#else
#define KM_LAPM_OCC
#endif

#include <type_traits>

// ------------------------------------------------------------------ helpers
// 24-bit multiply-add: full-rate v_mad_i32_i24 (a plain int multiply is a quarter-rate v_mul_lo_u32);
// every use below has both factors within +-2^23 and a product within int32.
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

template <typename T> struct px_traits;
template <> struct px_traits<uint8_t> { using acc = int; static constexpr int code = KM_U8; };
template <> struct px_traits<uint16_t> { using acc = int; static constexpr int code = KM_U16; };
template <> struct px_traits<int16_t> { using acc = int; static constexpr int code = KM_I16; };
template <> struct px_traits<float> { using acc = float; static constexpr int code = KM_F32; };

__device__ __forceinline__ double wave_min(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// stretch one raw value to uint8 exactly like numpy does in _to_uint8 (klt.py:48):
// integer dtypes in fp64, float32 in fp32; truncating cast; NaN -> 0.
template <typename T>
__device__ __forceinline__ unsigned stretch_u8(T v, double mn, double range, bool degenerate)
{
    if constexpr (sizeof(T) == 1) {
        return (unsigned)v;
    } else if constexpr (px_traits<T>::code == KM_F32) {
        if (degenerate) return 0u;
        float t = __fmul_rn(__fdiv_rn(__fsub_rn(v, (float)mn), (float)range), 255.0f);
        return (t != t) ? 0u : (unsigned)(int)t;
    } else {
        if (degenerate) return 0u;
        double t = __dmul_rn(__ddiv_rn(__dsub_rn((double)v, mn), range), 255.0);
        return (unsigned)(int)t;
    }
}

// ------------------------------------------------------------------ K1 min/max
// One image (or one box of a larger raster: stride > W) reduced by the `nth` threads of a launch that share it: 16-byte vector loads
// over the aligned body of every contiguous span - the whole image when its rows are dense, else row by row (a box of a larger
// raster: the tiles of `KLT.match` on a resident pair; byte loads there cost 0.27 ms per 30-Mpx tile) - packed 16-bit min / max.
template <typename T>
__device__ __forceinline__ void minmax_image(const T *__restrict__ img, int H, int W, ptrdiff_t stride, unsigned blk, unsigned nblk, double *partial_out,
                                             bool deep = false)
{
    using A = typename px_traits<T>::acc;
    A mn, mx;
    if constexpr (px_traits<T>::code == KM_F32) { mn = INFINITY; mx = -INFINITY; }
    else { mn = 0x7fffffff; mx = -0x7fffffff - 1; }
    auto upd = [&](T v) {
        if constexpr (px_traits<T>::code == KM_F32) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
        else { mn = min(mn, (A)v); mx = max(mx, (A)v); }
    };
    constexpr int V = 16 / sizeof(T);
    // 16-bit pixels: packed minimum / maximum of the dwords as loaded (v_pk_min/max_u16|i16: 2 instructions per pixel pair
    // where widening each pixel took 6), folded into mn / mx after the loop
    typedef typename std::conditional<std::is_signed<T>::value, short, unsigned short>::type P16;
    typedef P16 pk2 __attribute__((ext_vector_type(2)));
    constexpr bool PACKED = sizeof(T) == 2 && px_traits<T>::code != KM_F32;
    pk2 pmn, pmx;
    pmn.x = pmn.y = std::is_signed<T>::value ? (P16)0x7fff : (P16)0xffff;
    pmx.x = pmx.y = std::is_signed<T>::value ? (P16)0x8000 : (P16)0;
    bool packed_used = false;
    auto take = [&](const uint4 &q) {
        if constexpr (PACKED) {
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                pk2 v;
                __builtin_memcpy(&v, &w[k], 4);
                pmn = __builtin_elementwise_min(pmn, v);
                pmx = __builtin_elementwise_max(pmx, v);
            }
            packed_used = true;
        } else {
            T e[V];
            __builtin_memcpy(e, &q, 16);
#pragma unroll
            for (int k = 0; k < V; k++) upd(e[k]);
        }
    };
    // span of n contiguous pixels, shared by threads tid of nth
    auto span = [&](const T *p, size_t n, size_t tid, size_t nth) {
        const uintptr_t base = (uintptr_t)p;
        size_t head = ((16 - (base & 15)) & 15) / sizeof(T);
        if (head > n) head = n;
        const size_t nvec = (n - head) / V;
        const uint4 *vp = (const uint4 *)(p + head);
        // `deep` - eight independent 16-byte loads in flight per thread: beside instruction-bound kernels (the next submission's min / max beside
        // LK .. ZNCC) the kernel runs as ONE workgroup per compute unit - it must reach its bandwidth with four waves per CU, and must
        // not occupy more: spread over every wave slot LK's retiring waves left, its long-lived workgroups kept the 1024-thread workgroups
        // of the frame stage waiting for whole CUs until it had drained (fb_compact 5 -> 189 us per submission)
        size_t i = tid;
        if (deep)
        for (; i + 7 * nth < nvec; i += 8 * nth) {
            uint4 q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) q[k] = vp[i + k * nth];
#pragma unroll
            for (int k = 0; k < 8; k++) take(q[k]);
        }
        for (; i < nvec; i += nth) take(vp[i]);
        if (tid < head) upd(p[tid]);
        const size_t tail0 = head + nvec * V;
        if (tail0 + tid < n && tid < (size_t)V) upd(p[tail0 + tid]);
    };
    if (stride == W) {
        span(img, (size_t)H * W, (size_t)blk * blockDim.x + threadIdx.x, (size_t)nblk * blockDim.x);
    } else {
        for (int y = (int)blk; y < H; y += (int)nblk) span(img + (size_t)y * stride, (size_t)W, threadIdx.x, blockDim.x);
    }
    if constexpr (PACKED) {
        if (packed_used) {      // (the packed accumulators hold real pixels or their neutral start values)
            mn = min(mn, min((A)pmn.x, (A)pmn.y));
            mx = max(mx, max((A)pmx.x, (A)pmx.y));
        }
    }
    double dmn = wave_min((double)mn), dmx = wave_max((double)mx);
    __shared__ double s[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s[0][w] = dmn; s[1][w] = dmx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial_out[0] = fmin(fmin(s[0][0], s[0][1]), fmin(s[0][2], s[0][3]));
        partial_out[1] = fmax(fmax(s[1][0], s[1][1]), fmax(s[1][2], s[1][3]));
    }
}

// blockIdx.y selects the image (kd_minmax_pair: both rasters of a pair in one launch); partials of image y start at 2 * gridDim.x * y
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial_kernel(const T *__restrict__ img0, const T *__restrict__ img1, int H, int W,
                                                             ptrdiff_t stride0, ptrdiff_t stride1, double *partial)
{
    minmax_image<T>(blockIdx.y ? img1 : img0, H, W, blockIdx.y ? stride1 : stride0, blockIdx.x, gridDim.x,
                    partial + (size_t)2 * gridDim.x * blockIdx.y + 2 * blockIdx.x);
}

// batched units: blockIdx.z = unit, blockIdx.y = raster (0 ref, 1 mon)
struct mm_units_args {
    const void *img[2][KM_UNITS_MAX];
    ptrdiff_t stride[2][KM_UNITS_MAX];
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX];
    double *out[KM_UNITS_MAX];
};
template <typename T>
__global__ __launch_bounds__(256) void minmax_partial_units_kernel(mm_units_args U, double *partial, int deep)
{
    const unsigned u = blockIdx.z, im = blockIdx.y;
    minmax_image<T>((const T *)U.img[im][u], U.H[u], U.W[u], U.stride[im][u], blockIdx.x, gridDim.x,
                    partial + (size_t)2 * gridDim.x * (2 * u + im) + 2 * blockIdx.x, deep != 0);
}

__device__ __forceinline__ void minmax_final(const double *partial, int nb, double *out)
{
    double mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        mn = fmin(mn, partial[2 * i]);
        mx = fmax(mx, partial[2 * i + 1]);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    __shared__ double s[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s[0][w] = mn; s[1][w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fmin(fmin(s[0][0], s[0][1]), fmin(s[0][2], s[0][3]));
        out[1] = fmax(fmax(s[1][0], s[1][1]), fmax(s[1][2], s[1][3]));
    }
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const double *partial, int nb, double *out)
{
    minmax_final(partial + (size_t)2 * nb * blockIdx.x, nb, out + 2 * blockIdx.x);     // (one block per image)
}
__global__ __launch_bounds__(256) void minmax_final_units_kernel(const double *partial, int nb, mm_units_args U)
{
    minmax_final(partial + (size_t)2 * nb * (2 * blockIdx.y + blockIdx.x), nb, U.out[blockIdx.y] + 2 * blockIdx.x);   // grid (2, units)
}

// min / max of one image (d_b == nullptr) or of the two rasters of a pair in one launch: d_mm[0..1] (and d_mm[2..3])
static int minmax_launch(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *d_mm, int ws_slot = WS_PARTIAL)
{
    // workgroups per image.  Beside LK (early min / max: ws_slot != WS_PARTIAL) the kernel is off the critical path and takes
    // fewer wave slots from the kernel it shares the GPU with
    static const int nb_early = [] { const char *e = km_dev_env("KARIOS_HIP_MM_EARLY_NB"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 2048 ? v : 2048; }();
    // (one pair at a time: the early min / max has only LK .. ZNCC of ONE pair, 0.27 ms, to hide behind - it needs the bandwidth of many
    // workgroups; a batched submission's runs as one workgroup per CU, kd_minmax_units)
    const int nb = ws_slot == WS_PARTIAL ? 2048 : nb_early, ni = d_b ? 2 : 1;
    double *partial = (double *)km_ws(c, ws_slot, (size_t)2 * nb * ni * sizeof(double));
    if (!partial) return KM_E_NOMEM;
    const dim3 grid(nb, ni);
    switch (dtype) {
    case KM_U8: minmax_partial_kernel<uint8_t><<<grid, 256, 0, c->stream>>>((const uint8_t *)d_a, (const uint8_t *)d_b, H, W, sa, sb, partial); break;
    case KM_U16: minmax_partial_kernel<uint16_t><<<grid, 256, 0, c->stream>>>((const uint16_t *)d_a, (const uint16_t *)d_b, H, W, sa, sb, partial); break;
    case KM_I16: minmax_partial_kernel<int16_t><<<grid, 256, 0, c->stream>>>((const int16_t *)d_a, (const int16_t *)d_b, H, W, sa, sb, partial); break;
    case KM_F32: minmax_partial_kernel<float><<<grid, 256, 0, c->stream>>>((const float *)d_a, (const float *)d_b, H, W, sa, sb, partial); break;
    default: return km_fail(c, KM_E_ARG, "minmax: bad dtype %d", dtype);
    }
    KM_LAUNCH_CHECK(c);
    minmax_final_kernel<<<ni, 256, 0, c->stream>>>(partial, nb, d_mm);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// both rasters of every unit of a batch in ONE launch (+ one final launch): out[u] = {min_ref, max_ref, min_mon, max_mon}
int kd_minmax_units(km_ctx *c, const km_units &U, double *const *d_out, int ws_slot)
{
    mm_units_args A;
    for (int u = 0; u < U.n; u++) {
        A.img[0][u] = U.ref[u]; A.img[1][u] = U.mon[u]; A.stride[0][u] = U.sref[u]; A.stride[1][u] = U.smon[u];
        A.H[u] = U.H[u]; A.W[u] = U.W[u]; A.out[u] = d_out[u];
    }
    // workgroups per raster: ~4096 - 8192 over the batch on the critical path.  Beside other kernels (ws_slot != WS_PARTIAL: the early
    // min / max of a submission behind another one) whole rasters run as ONE workgroup per compute unit over the batch with eight
    // loads in flight per thread (`deep`); boxes of larger rasters go row by row - too few vectors per thread and row for that
    int nb = U.n >= 8 ? 256 : U.n >= 4 ? 512 : 1024, deep = 0;
    if (ws_slot != WS_PARTIAL) {
        bool flat = true;
        for (int u = 0; u < U.n; u++) flat = flat && U.sref[u] == U.W[u] && U.smon[u] == U.W[u];
        if (flat) { nb = std::max(8, c->n_cu / (2 * U.n)); deep = 1; }
    }
    double *partial = (double *)km_ws(c, ws_slot, (size_t)2 * nb * 2 * U.n * sizeof(double));
    if (!partial) return KM_E_NOMEM;
    const dim3 grid(nb, 2, U.n);
    switch (U.dtype) {
    case KM_U8: minmax_partial_units_kernel<uint8_t><<<grid, 256, 0, c->stream>>>(A, partial, deep); break;
    case KM_U16: minmax_partial_units_kernel<uint16_t><<<grid, 256, 0, c->stream>>>(A, partial, deep); break;
    case KM_I16: minmax_partial_units_kernel<int16_t><<<grid, 256, 0, c->stream>>>(A, partial, deep); break;
    case KM_F32: minmax_partial_units_kernel<float><<<grid, 256, 0, c->stream>>>(A, partial, deep); break;
    default: return km_fail(c, KM_E_ARG, "minmax: bad dtype %d", U.dtype);
    }
    KM_LAUNCH_CHECK(c);
    minmax_final_units_kernel<<<dim3(2, U.n), 256, 0, c->stream>>>(partial, nb, A);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kd_minmax(km_ctx *c, const void *d_img, int dtype, int H, int W, ptrdiff_t stride, double *d_mm)
{
    return minmax_launch(c, d_img, nullptr, dtype, H, W, stride, 0, d_mm);
}

// (partials in a workspace slot of the caller's choice: the early min / max of the next unit runs beside kernels that use WS_PARTIAL)
int kd_minmax_pair_ws(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *d_mm, int ws_slot)
{
    return minmax_launch(c, d_a, d_b, dtype, H, W, sa, sb, d_mm, ws_slot);
}

// both rasters of a pair: d_mm = {min_a, max_a, min_b, max_b}
int kd_minmax_pair(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t sa, ptrdiff_t sb, double *d_mm)
{
    return minmax_launch(c, d_a, d_b, dtype, H, W, sa, sb, d_mm);
}

// ------------------------------------------------------------------ standalone stretch / mask
template <typename T>
__global__ __launch_bounds__(256) void to_uint8_kernel(const T *__restrict__ img, int H, int W, ptrdiff_t stride,
                                                       const double *mm, int invert, uint8_t *out)
{
    const double mn = mm[0], mx = mm[1], range = mx - mn;
    const bool deg = !(mx > mn);
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        unsigned r = stretch_u8<T>(img[(size_t)y * stride + x], mn, range, deg);
        out[i] = (uint8_t)(invert ? 255u - r : r);
    }
}

int kd_to_uint8(km_ctx *c, const void *d_img, int dtype, int H, int W, ptrdiff_t stride, const double *d_mm,
                int invert, uint8_t *d_out)
{
    const int nb = 4096;
    switch (dtype) {
    case KM_U8: to_uint8_kernel<uint8_t><<<nb, 256, 0, c->stream>>>((const uint8_t *)d_img, H, W, stride, d_mm, invert, d_out); break;
    case KM_U16: to_uint8_kernel<uint16_t><<<nb, 256, 0, c->stream>>>((const uint16_t *)d_img, H, W, stride, d_mm, invert, d_out); break;
    case KM_I16: to_uint8_kernel<int16_t><<<nb, 256, 0, c->stream>>>((const int16_t *)d_img, H, W, stride, d_mm, invert, d_out); break;
    case KM_F32: to_uint8_kernel<float><<<nb, 256, 0, c->stream>>>((const float *)d_img, H, W, stride, d_mm, invert, d_out); break;
    default: return km_fail(c, KM_E_ARG, "to_uint8: bad dtype %d", dtype);
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

struct nodata_t {
    double mon, ref;
    int has_mon, has_ref;
};

template <typename T>
__device__ __forceinline__ bool px_valid(T a /*mon*/, T b /*ref*/, const nodata_t &nd)
{
    bool ok = (a != (T)0) && (b != (T)0);
    if constexpr (px_traits<T>::code == KM_F32) ok = ok && isfinite(a) && isfinite(b);
    if (nd.has_mon) ok = ok && ((double)a != nd.mon);
    if (nd.has_ref) ok = ok && ((double)b != nd.ref);
    return ok;
}

template <typename T>
__global__ __launch_bounds__(256) void auto_mask_kernel(const T *__restrict__ mon, const T *__restrict__ ref, int H, int W,
                                                        ptrdiff_t smon, ptrdiff_t sref, nodata_t nd, uint8_t *mask,
                                                        unsigned long long *valid)
{
    const size_t n = (size_t)H * W;
    unsigned long long cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        bool ok = px_valid<T>(mon[(size_t)y * smon + x], ref[(size_t)y * sref + x], nd);
        mask[i] = ok ? 1 : 0;
        cnt += ok;
    }
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(valid, cnt);
}

static nodata_t make_nodata(const double *nodata_mon, const double *nodata_ref)
{
    nodata_t nd;
    nd.has_mon = nodata_mon != nullptr; nd.mon = nodata_mon ? *nodata_mon : 0.0;
    nd.has_ref = nodata_ref != nullptr; nd.ref = nodata_ref ? *nodata_ref : 0.0;
    return nd;
}

int kd_auto_mask(km_ctx *c, const void *d_mon, const void *d_ref, int dtype, int H, int W, ptrdiff_t smon,
                 ptrdiff_t sref, const double *nodata_mon, const double *nodata_ref, uint8_t *d_mask,
                 unsigned long long *d_valid)
{
    nodata_t nd = make_nodata(nodata_mon, nodata_ref);
    KM_HIP(c, hipMemsetAsync(d_valid, 0, sizeof(unsigned long long), c->stream));
    const int nb = 4096;
    switch (dtype) {
    case KM_U8: auto_mask_kernel<uint8_t><<<nb, 256, 0, c->stream>>>((const uint8_t *)d_mon, (const uint8_t *)d_ref, H, W, smon, sref, nd, d_mask, d_valid); break;
    case KM_U16: auto_mask_kernel<uint16_t><<<nb, 256, 0, c->stream>>>((const uint16_t *)d_mon, (const uint16_t *)d_ref, H, W, smon, sref, nd, d_mask, d_valid); break;
    case KM_I16: auto_mask_kernel<int16_t><<<nb, 256, 0, c->stream>>>((const int16_t *)d_mon, (const int16_t *)d_ref, H, W, smon, sref, nd, d_mask, d_valid); break;
    case KM_F32: auto_mask_kernel<float><<<nb, 256, 0, c->stream>>>((const float *)d_mon, (const float *)d_ref, H, W, smon, sref, nd, d_mask, d_valid); break;
    default: return km_fail(c, KM_E_ARG, "auto_mask: bad dtype %d", dtype);
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// non-zero bytes of a user mask (the valid-pixel count of klt.py:276): 16-byte loads over the aligned body, the non-zero bytes of a
// dword counted with three logic operations and a population count (a byte load per pixel made this 0.2 ms at 10980^2 - as long as
// the whole stretch + Laplacian kernel it precedes)
__device__ __forceinline__ unsigned nonzero_bytes(uint32_t w)
{
    const uint32_t t = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;   // bit 7 of every byte that is not 0
    return (unsigned)__popc(t);
}
__global__ __launch_bounds__(256) void count_nonzero_kernel(const uint8_t *__restrict__ m, size_t n, unsigned *__restrict__ partial)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    size_t head = (16 - ((uintptr_t)m & 15)) & 15;
    if (head > n) head = n;
    const size_t nvec = (n - head) / 16;
    const uint4 *vp = (const uint4 *)(m + head);
    unsigned cnt32 = 0;
    for (size_t i = tid; i < nvec; i += nth) {
        const uint4 q = vp[i];
        cnt32 += nonzero_bytes(q.x) + nonzero_bytes(q.y) + nonzero_bytes(q.z) + nonzero_bytes(q.w);     // (< 2^32: a lane sees < 2^28 bytes)
    }
    unsigned long long cnt = cnt32;
    if (tid < head) cnt += m[tid] != 0;
    const size_t tail0 = head + nvec * 16;
    if (tid < 16 && tail0 + tid < n) cnt += m[tail0 + tid] != 0;
    // one partial per workgroup, summed by sum_u32_kernel: 8192 device-scope atomics on ONE word serialise (~10 ns each: 80 of the
    // kernel's 115 us were that)
    cnt = wave_sum_u64(cnt);
    __shared__ unsigned sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (unsigned)cnt;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(1024) void sum_u32_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned long long *out);

int kd_count_nonzero(km_ctx *c, const uint8_t *d_mask, size_t n, unsigned long long *d_valid)
{
    const unsigned nb = 4096;
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, nb * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    count_nonzero_kernel<<<nb, 256, 0, c->stream>>>(d_mask, n, partial);
    KM_LAUNCH_CHECK(c);
    sum_u32_kernel<<<1, 1024, 0, c->stream>>>(partial, nb, d_valid);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// final reductions of per-workgroup partials (one workgroup; the partial arrays are tens of KB)
__global__ __launch_bounds__(1024) void sum_u32_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned long long *out)
{
    // ONE workgroup: writes (not accumulates) the total, so the launcher needs no memset
    unsigned long long s = 0;
    for (unsigned i = threadIdx.x; i < n; i += 1024) s += partial[i];
    s = wave_sum_u64(s);
    __shared__ unsigned long long sh[16];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 16; i++) t += sh[i];
        *out = t;
    }
}
__global__ __launch_bounds__(1024) void max_u32_kernel(const unsigned *__restrict__ partial, unsigned n, unsigned *out)
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

// ------------------------------------------------------------------ K2 stretch + Laplacian (+mask)
// Output tile 128x16 per 256-thread workgroup.  The Laplacian of odd ksize k is written as
//   sum_j ks[j] * (kd *x u8)[y+j] + kd[j] * (ks *x u8)[y+j]
// (ksize 1 and 3 fit the same form with kd=[1,-2,1] and ks=[0,1,0] / [1,2,1]); both images use
// radius R = max of the two, the shorter kernel zero-padded, so one launch serves mixed sizes.
//
// uint8 stretch of 16-bit integer images without a per-pixel fp64 division: numpy computes
// trunc(fl(fl(d/r)*255)) with d = v - min, r = max - min (integers).  When 255*d is not a multiple of r the exact
// quotient 255*d/r lies >= 1/r from any integer, far more than the fp64 rounding error, so the result is
// floor(255*d/r).  When 255*d == k*r the real quotient d/r equals k/255, the correctly rounded division gives
// fl(k/255) whatever d and r are, and fl(fl(k/255)*255) >= k holds for every k in 0..255 (256 cases, checked by
// tests/test_host_logic.py::test_stretch_exact_multiples) - again floor(255*d/r).  The kernels evaluate that floor as
//   (int) fma((double)d, 255/r, 0.5/r)
// : (255*d + 0.5)/r is never an integer, has the same floor, and stays >= 0.5/r >= 7.6e-6 away from the integers,
// eleven orders of magnitude above the fma's rounding error - three full-rate instructions, no branch, no table.
#define LAP_TW 128
#define LAP_TH 16

struct lap_coef {
    int kd[2][11];
    int ks[2][11];
    int b3[2] = {0, 0};      // marching kernel, radius 5: image i is kernel 11 = its 9-tap pass + a 3 x 3 binomial (lap_march_item)
};

template <typename T> struct stretcher {
    // generic (f32 / u8): arithmetic path
    double mn, range;
    bool deg;
    __device__ void init(const double *mm, int img, const uint8_t *) {
        if constexpr (sizeof(T) == 1) { mn = 0; range = 1; deg = false; }
        else { mn = mm[2 * img]; const double mx = mm[2 * img + 1]; range = mx - mn; deg = !(mx > mn); }
    }
    __device__ __forceinline__ unsigned operator()(T v, const uint8_t *) const { return stretch_u8<T>(v, mn, range, deg); }
};
template <typename T> struct stretcher_i16 {
    int mn_i;
    double c1, c0;
    __device__ void init(const double *mm, int img, const uint8_t *) {
        const double mn = mm[2 * img], mx = mm[2 * img + 1];
        mn_i = (int)mn;
        if (mx > mn) { const double r = mx - mn; c1 = 255.0 / r; c0 = 0.5 / r; }
        else { c1 = 0.0; c0 = 0.0; }                     // degenerate range: every pixel maps to 0
    }
    __device__ __forceinline__ unsigned operator()(T v, const uint8_t *) const {
        return (unsigned)(int)__fma_rn((double)((int)v - mn_i), c1, c0);
    }
};
template <> struct stretcher<uint16_t> : stretcher_i16<uint16_t> {};
template <> struct stretcher<int16_t> : stretcher_i16<int16_t> {};

template <int R, typename T, int NIMG, bool MASK>
__global__ __launch_bounds__(256) void lap_kernel(const T *__restrict__ img0, const T *__restrict__ img1, int H, int W,
                                                  ptrdiff_t stride0, ptrdiff_t stride1, const double *__restrict__ mm,
                                                  lap_coef cf, int invert1, nodata_t nd,
                                                  uint8_t *__restrict__ out0, uint8_t *__restrict__ out1,
                                                  uint8_t *__restrict__ mask_out, unsigned *__restrict__ valid_partial)
{
    using HT = typename std::conditional<(R <= 3), short, int>::type;  // |kd*x| <= 3060, ks*x <= 16320 for k <= 7
    constexpr int HX = (R + 3) & ~3;          // x halo rounded to 4 for packed LDS words
    constexpr int TWH = LAP_TW + 2 * HX;      // LDS tile row length (bytes)
    constexpr int THH = LAP_TH + 2 * R;
    constexpr int CPR = TWH / 4;              // 4-pixel chunks per row
    constexpr int NQ = LAP_TW / 4;            // output quads per row
    constexpr int NIT = (THH * CPR + 255) / 256;
    __shared__ uint32_t tile[NIMG][THH][CPR];
    __shared__ __attribute__((aligned(16))) HT hbuf[2][NIMG][THH][LAP_TW];  // [0] = kd pass, [1] = ks pass
    auto &hd = hbuf[0];
    auto &hs = hbuf[1];
    const int X0 = blockIdx.x * LAP_TW, Y0 = blockIdx.y * LAP_TH;
    const int tid = threadIdx.x;
    stretcher<T> st[NIMG];
#pragma unroll
    for (int i = 0; i < NIMG; i++) st[i].init(mm, i, nullptr);
    const T *imgs[2] = {img0, img1};
    const ptrdiff_t strides[2] = {stride0, stride1};

    // ---- phase 1a: issue every global load of this thread (raw tile + halo, REFLECT_101)
    T v[NIT][NIMG][4];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int ci = it * 256 + tid;
        if (ci < THH * CPR) {
            const int row = ci / CPR, cx = ci - row * CPR;
            const int gy = km_reflect101(Y0 - R + row, H);
            const int gx0 = X0 - HX + cx * 4;
            const bool inside = gx0 >= 0 && gx0 + 3 < W;
#pragma unroll
            for (int i = 0; i < NIMG; i++) {
                const T *rowp = imgs[i] + (size_t)gy * strides[i];
                if (inside && ((((uintptr_t)(rowp + gx0)) & (4 * sizeof(T) - 1)) == 0)) {
                    if constexpr (sizeof(T) == 1) { uint32_t q = *(const uint32_t *)(rowp + gx0); __builtin_memcpy(v[it][i], &q, 4); }
                    else if constexpr (sizeof(T) == 2) { uint2 q = *(const uint2 *)(rowp + gx0); __builtin_memcpy(v[it][i], &q, 8); }
                    else { uint4 q = *(const uint4 *)(rowp + gx0); __builtin_memcpy(v[it][i], &q, 16); }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) v[it][i][k] = rowp[km_reflect101(gx0 + k, W)];
                }
            }
        }
    }
    // ---- phase 1b: stretch to u8, pack into LDS, emit the auto mask
    unsigned cnt = 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int ci = it * 256 + tid;
        if (ci < THH * CPR) {
            const int row = ci / CPR, cx = ci - row * CPR;
            const int gx0 = X0 - HX + cx * 4;
#pragma unroll
            for (int i = 0; i < NIMG; i++) {
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    unsigned u = st[i](v[it][i][k], nullptr);
                    if (i == 1 && invert1) u = 255u - u;
                    packed |= u << (8 * k);
                }
                tile[i][row][cx] = packed;
            }
            if constexpr (MASK) {
                // auto mask for interior pixels: v[.][0] = ref, v[.][1] = mon
                const int oy = Y0 - R + row;
                if (row >= R && row < R + LAP_TH && oy < H && gx0 >= X0 && gx0 < X0 + LAP_TW && gx0 < W) {
                    uint32_t mp = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool ok = (gx0 + k < W) && px_valid<T>(v[it][1][k], v[it][0][k], nd);
                        mp |= (ok ? 1u : 0u) << (8 * k);
                        cnt += ok;
                    }
                    const size_t o = (size_t)oy * W + gx0;
                    if (gx0 + 3 < W && (o & 3) == 0) *(uint32_t *)(mask_out + o) = mp;
                    else {
                        for (int k = 0; k < 4 && gx0 + k < W; k++) mask_out[o + k] = (uint8_t)((mp >> (8 * k)) & 1u);
                    }
                }
            }
        }
    }
    if constexpr (MASK) {
        // one partial per workgroup (a single-address atomic per wave would serialise ~10^5 updates)
        __shared__ unsigned s_cnt[4];
        const unsigned c64 = (unsigned)wave_sum_u64((unsigned long long)cnt);
        if ((tid & 63) == 0) s_cnt[tid >> 6] = c64;
        __syncthreads();
        if (tid == 0) valid_partial[blockIdx.y * gridDim.x + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    } else {
        __syncthreads();
    }

    uint8_t *outs[2] = {out0, out1};
    if constexpr (R <= 3) {
        // ================= packed path (ksize <= 7): v_dot4 horizontally, v_dot2 vertically =================
        // pixels are stored biased (p - 128, bit 7 flipped) so that signed 8-bit dot products apply;
        // sum(kd) = 0 and sum(ks) = 4^R remove / restore the bias exactly.
        typedef short short2v __attribute__((ext_vector_type(2)));
        int *hds = (int *)&hd[0][0][0];  // [NIMG][THH][LAP_TW] packed (hd | hs << 16); hd+hs storage is contiguous
        int kdp[NIMG][2], ksp[NIMG][2], bias[NIMG], vk[NIMG][2 * R + 1];
#pragma unroll
        for (int i = 0; i < NIMG; i++) {
            int sum = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                unsigned a = 0, b = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int t = 4 * h + k;
                    const int d = t <= 2 * R ? cf.kd[i][t] : 0, sm = t <= 2 * R ? cf.ks[i][t] : 0;
                    a |= ((unsigned)d & 0xffu) << (8 * k);
                    b |= ((unsigned)sm & 0xffu) << (8 * k);
                    sum += sm;
                }
                kdp[i][h] = (int)a; ksp[i][h] = (int)b;
            }
            bias[i] = R == 4 ? 0 : 128 * sum;
#pragma unroll
            for (int t = 0; t <= 2 * R; t++) vk[i][t] = (cf.ks[i][t] & 0xffff) | (cf.kd[i][t] << 16);
        }
        // ---- phase 2: horizontal kd / ks passes, 4 outputs per item
        for (int it = tid; it < NIMG * THH * NQ; it += 256) {
            const int i = it / (THH * NQ);
            const int rem = it - i * (THH * NQ);
            const int row = rem / NQ, q = rem - row * NQ;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = ((q + k < CPR) ? tile[i][row][q + k] : 0u) ^ 0x80808080u;
            int o4[4];
#pragma unroll
            for (int o = 0; o < 4; o++) {
                constexpr int base = HX - R;
                const int sft = (o + base) & 3, wi = (o + base) >> 2;
                const int g0 = (int)__builtin_amdgcn_alignbyte(w[wi + 1], w[wi], sft);
                const int g1 = (int)__builtin_amdgcn_alignbyte(wi + 2 < 4 ? w[wi + 2] : 0u, w[wi + 1], sft);
                const int vd = __builtin_amdgcn_sdot4(g0, kdp[i][0], __builtin_amdgcn_sdot4(g1, kdp[i][1], 0, false), false);
                const int vs = __builtin_amdgcn_sdot4(g0, ksp[i][0], __builtin_amdgcn_sdot4(g1, ksp[i][1], bias[i], false), false);
                o4[o] = (vd & 0xffff) | (vs << 16);
            }
            *(int4 *)&hds[((size_t)i * THH + row) * LAP_TW + 4 * q] = make_int4(o4[0], o4[1], o4[2], o4[3]);
        }
        __syncthreads();
        // ---- phase 3: vertical combine on 4x4 micro-tiles: one v_dot2 per tap and pixel
        for (int it = tid; it < NIMG * (LAP_TH / 4) * NQ; it += 256) {
            const int i = it / ((LAP_TH / 4) * NQ);
            const int rem = it - i * ((LAP_TH / 4) * NQ);
            const int rg = rem / NQ, q = rem - rg * NQ;
            const int ox = X0 + 4 * q;
            if (ox >= W || Y0 + 4 * rg >= H) continue;
            int acc[4][4];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) acc[o][c2] = 0;
#pragma unroll
            for (int j = 0; j < 2 * R + 4; j++) {
                const int4 a = *(const int4 *)&hds[((size_t)i * THH + 4 * rg + j) * LAP_TW + 4 * q];
                const int av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int tap = j - o;
                    if (tap >= 0 && tap <= 2 * R) {
#pragma unroll
                        for (int c2 = 0; c2 < 4; c2++)
                            acc[o][c2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, av[c2]), __builtin_bit_cast(short2v, vk[i][tap]),
                                                                acc[o][c2], false);
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int oy = Y0 + 4 * rg + o;
                if (oy >= H) break;
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) packed |= (uint32_t)min(max(acc[o][k], 0), 255) << (8 * k);
                const size_t off = (size_t)oy * W + ox;
                if (ox + 3 < W && (off & 3) == 0) *(uint32_t *)(outs[i] + off) = packed;
                else {
                    for (int k = 0; k < 4 && ox + k < W; k++) outs[i][off + k] = (uint8_t)(packed >> (8 * k));
                }
            }
        }
    } else {
        // ================= generic path (ksize 9, 11): 24-bit multiply-adds on int32 planes =================
        // ---- phase 2: horizontal passes (kd and ks) for 4 consecutive outputs per item
        constexpr int NW = (2 * R + 4 + (HX - R) + 3) / 4;  // words covering [x+HX-R, x+HX+R+4)
        for (int it = tid; it < NIMG * THH * NQ; it += 256) {
            const int i = it / (THH * NQ);
            const int rem = it - i * (THH * NQ);
            const int row = rem / NQ, q = rem - row * NQ;
            uint32_t w[NW + 1];
#pragma unroll
            for (int k = 0; k < NW + 1; k++) w[k] = (q + k < CPR) ? tile[i][row][q + k] : 0u;
            int ad[4] = {0, 0, 0, 0}, as[4] = {0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < 2 * R + 4; t++) {
                const int b = t + (HX - R);
                const int pv = (int)((w[b >> 2] >> (8 * (b & 3))) & 0xffu);
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int k = t - o;  // tap index for output o
                    if (k >= 0 && k <= 2 * R) {
                        ad[o] = mad24(cf.kd[i][k], pv, ad[o]);
                        as[o] = mad24(cf.ks[i][k], pv, as[o]);
                    }
                }
            }
            HT *pd = &hd[i][row][4 * q], *ps = &hs[i][row][4 * q];
#pragma unroll
            for (int o = 0; o < 4; o++) { pd[o] = (HT)ad[o]; ps[o] = (HT)as[o]; }
        }
        __syncthreads();
        // ---- phase 3: vertical combine on 4x4 micro-tiles, clip to [0,255], packed stores
        for (int it = tid; it < NIMG * (LAP_TH / 4) * NQ; it += 256) {
            const int i = it / ((LAP_TH / 4) * NQ);
            const int rem = it - i * ((LAP_TH / 4) * NQ);
            const int rg = rem / NQ, q = rem - rg * NQ;
            const int ox = X0 + 4 * q;
            if (ox >= W || Y0 + 4 * rg >= H) continue;
            int acc[4][4];
#pragma unroll
            for (int o = 0; o < 4; o++)
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) acc[o][c2] = 0;
#pragma unroll
            for (int j = 0; j < 2 * R + 4; j++) {
                const HT *pa = &hd[i][4 * rg + j][4 * q], *pb = &hs[i][4 * rg + j][4 * q];
                int a[4], b[4];
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) { a[c2] = pa[c2]; b[c2] = pb[c2]; }
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int tap = j - o;
                    if (tap >= 0 && tap <= 2 * R) {
                        const int ksj = cf.ks[i][tap], kdj = cf.kd[i][tap];
#pragma unroll
                        for (int c2 = 0; c2 < 4; c2++) acc[o][c2] = mad24(ksj, a[c2], mad24(kdj, b[c2], acc[o][c2]));
                    }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const int oy = Y0 + 4 * rg + o;
                if (oy >= H) break;
                uint32_t packed = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) packed |= (uint32_t)min(max(acc[o][k], 0), 255) << (8 * k);
                const size_t off = (size_t)oy * W + ox;
                if (ox + 3 < W && (off & 3) == 0) *(uint32_t *)(outs[i] + off) = packed;
                else {
                    for (int k = 0; k < 4 && ox + k < W; k++) outs[i][off + k] = (uint8_t)(packed >> (8 * k));
                }
            }
        }
    }
}

// OpenCV getSobelKernels recurrence (order 0 / 2), centred into an 11-tap array of radius R
static bool fill_coef(int ksize, int R, int *kd, int *ks)
{
    int d[12] = {0}, s[12] = {0};
    auto gen = [](int k, int order, int *ker) {
        if (k == 3) {
            static const int k0[3] = {1, 2, 1}, k2[3] = {1, -2, 1};
            for (int i = 0; i < 3; i++) ker[i] = order == 0 ? k0[i] : k2[i];
            return;
        }
        ker[0] = 1;
        for (int i = 0; i < k; i++) ker[i + 1] = 0;
        for (int i = 0; i < k - order - 1; i++) {
            int oldv = ker[0];
            for (int j = 1; j <= k; j++) { int nv = ker[j] + ker[j - 1]; ker[j - 1] = oldv; oldv = nv; }
        }
        for (int i = 0; i < order; i++) {
            int oldv = -ker[0];
            for (int j = 1; j <= k; j++) { int nv = ker[j - 1] - ker[j]; ker[j - 1] = oldv; oldv = nv; }
        }
    };
    int r;
    if (ksize == 1) { d[0] = 1; d[1] = -2; d[2] = 1; s[0] = 0; s[1] = 1; s[2] = 0; r = 1; }
    else if (ksize == 3 || ksize == 5 || ksize == 7 || ksize == 9 || ksize == 11) { gen(ksize, 2, d); gen(ksize, 0, s); r = ksize / 2; }
    else return false;
    if (r > R) return false;
    for (int i = 0; i < 11; i++) { kd[i] = 0; ks[i] = 0; }
    for (int i = 0; i < 2 * r + 1; i++) { kd[i + (R - r)] = d[i]; ks[i + (R - r)] = s[i]; }
    return true;
}

template <typename T, int NIMG, bool MASK>
static int launch_lap(km_ctx *c, int R, const T *a, const T *b, int H, int W, ptrdiff_t sa, ptrdiff_t sb, const double *mm,
                      const lap_coef &cf, int invert1, const nodata_t &nd, uint8_t *oa, uint8_t *ob, uint8_t *mask,
                      unsigned long long *valid_out)
{
    dim3 grid((W + LAP_TW - 1) / LAP_TW, (H + LAP_TH - 1) / LAP_TH);
    unsigned *valid = nullptr;
    if (MASK) {
        valid = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)grid.x * grid.y * sizeof(unsigned));
        if (!valid) return KM_E_NOMEM;
    }
#define KM_LAP_CASE(RR)                                                                                         \
    case RR:                                                                                                    \
        lap_kernel<RR, T, NIMG, MASK><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, \
                                                                   mask, valid);                               \
        break;
    switch (R) {
        KM_LAP_CASE(1)
        KM_LAP_CASE(2)
        KM_LAP_CASE(3)
        KM_LAP_CASE(4)
        KM_LAP_CASE(5)
    default: return km_fail(c, KM_E_UNSUPPORTED, "laplacian radius %d", R);
    }
#undef KM_LAP_CASE
    KM_LAUNCH_CHECK(c);
    if (MASK) {
        sum_u32_kernel<<<1, 1024, 0, c->stream>>>(valid, grid.x * grid.y, valid_out);
        KM_LAUNCH_CHECK(c);
    }
    return KM_OK;
}

// ---- K2, fast path (both images, ksize <= 7): one wavefront marches down a 256-column strip, 4 columns per
// lane (62 of the 64 lanes produce output, the outer two only feed their neighbours).  Per source row: raw
// pixels -> exact uint8 stretch -> horizontal kd / ks passes with v_dot4 on bytes assembled from the two
// neighbour lanes (DPP wave shifts + v_alignbyte) -> (hd | hs) pairs pushed into a (2R+1)-row register ring;
// per output row: one v_dot2 per tap and pixel over the ring.  No LDS tiles, no barriers, no index arithmetic.
// First link of a dot-product chain in the three-address VOP3P form (accumulator = inline 0 or a VGPR): the
// two-address v_dot4c / v_dot2c the compiler prefers needs a v_mov to seed every chain.  Coefficients are
// wave-uniform (SGPR operand; gfx9 allows one scalar source per VALU instruction, so the bias sits in a VGPR).
__device__ __forceinline__ int dot4_seed0(int bytes, int coef_uniform)
{
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(bytes), "s"(coef_uniform));
    return r;
}
__device__ __forceinline__ int dot4_seed(int bytes, int coef_uniform, int acc)
{
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(bytes), "s"(coef_uniform), "v"(acc));
    return r;
}
__device__ __forceinline__ int dot2_seed0(int pair, int coef_uniform)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(pair), "s"(coef_uniform));
    return r;
}
// An empty asm makes a lane offset opaque at the point of use: the compiler then cannot fold it into a hoisted
// per-lane 64-bit pointer and addresses memory as scalar row base + 32-bit vector offset (no per-lane 64-bit arithmetic).
__device__ __forceinline__ unsigned opaque_lane_offset(unsigned x)
{
    asm volatile("" : "+v"(x));
    return x;
}

#define LAPM_VALID_OF(R) ((R) == 5 ? 240 : 248)      // output columns of a strip: 64 lanes x 4 columns less the halo lanes (two either side for radius 5)
#ifndef LAPM_SPLIT
#define LAPM_SPLIT 0   // 1: one image per wavefront (88 VGPRs, 5 waves/SIMD) - measured SLOWER (0.30 vs 0.264 ms: loop, address and mask work duplicated); 0: both images in one wave
#endif
#ifndef LAPM_PF
#define LAPM_PF 3      // source rows in flight per wave
#endif

// SPLIT: the two images of a work item go to two different wavefronts (wave parity) - half the register ring per wave
// (76 instead of 123 VGPRs: 6 instead of 4 waves per SIMD); the image-0 wave also loads image 1's raw row for the mask.
template <int R, typename T, bool MASK, bool SPLIT>
__device__ __forceinline__ void lap_march_item(const T *__restrict__ img0, const T *__restrict__ img1, int H, int W,
                                               ptrdiff_t stride0, ptrdiff_t stride1, const double *__restrict__ mm,
                                               const lap_coef &cf, int invert1, const nodata_t &nd,
                                               uint8_t *__restrict__ out0, uint8_t *__restrict__ out1,
                                               uint8_t *__restrict__ mask_out, unsigned *__restrict__ valid_partial, int nstrips,
                                               int rows_per_item, int nitems, int wave_lin /* wave-uniform: this wavefront's work item (x 2 with SPLIT) */)
{
    typedef short short2v __attribute__((ext_vector_type(2)));
    // R = 5 (kernel 11; sum ks = 1024: a horizontal smoothing sum needs 18 bits, the ring holds 16-bit pairs): the kernels of size 11 are
    // those of size 9 convolved with [1 2 1] (OpenCV's getSobelKernels recurrence), so
    //     Laplacian_11 = ([1 2 1] x [1 2 1]) * (kd9 x ks9 + ks9 x kd9)          (before the saturation)
    // - the 9-tap pass (radius RH = 4) of this kernel, its 32-bit row kept unsaturated, then a 3 x 3 binomial on those integers.  REFLECT_101
    // commutes with it: a symmetric filter maps the whole-sample-symmetric extension of the image onto the extension of its own output.
    // Two halo lanes either side (the binomial needs the 9-tap value one column outside the strip), one row more either end of an item.
    constexpr int RH = R == 5 ? 4 : R;
    constexpr int NR = 2 * RH + 1;
    constexpr int HALO = R == 5 ? 8 : 4, VALID = 256 - 2 * HALO;
    static_assert(VALID == LAPM_VALID_OF(R), "strip geometry");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = SPLIT ? wave_lin >> 1 : wave_lin;           // work item (row arithmetic, loop control and row bases stay scalar)
    const int img_sel = SPLIT ? __builtin_amdgcn_readfirstlane(wave_lin & 1) : -1;
    if (wave_id >= nitems) { if (!SPLIT && MASK && lane == 0) valid_partial[wave_id] = 0u; return; }
    const int rowblock = wave_id / nstrips, strip = wave_id - rowblock * nstrips;
    stretcher<T> st[2];
    st[0].init(mm, 0, nullptr); st[1].init(mm, 1, nullptr);
    // A width that is no multiple of 4 would leave the last strip's border lane straddling the right edge (the per-lane general path:
    // 1 strip in 23 of a 5490-column unit, and + 22 % on the launch).  That strip is SHIFTED left instead so that its lane 63 starts
    // exactly at column W: it recomputes - and rewrites, byte for byte the same - columns of its left neighbour and counts valid
    // pixels only from `cnt_from` on.
    const bool shifted = (W % 4 != 0) && strip == nstrips - 1 && W >= 256;      // (its lane 0 at column W - 252 lies inside the image)
    const int gx0 = (shifted ? W - (256 - HALO) : strip * VALID - HALO) + 4 * lane;           // first of this lane's 4 columns
    const int cnt_from = shifted ? (nstrips - 1) * VALID : 0;
    const uint32_t cnt_mask = gx0 >= cnt_from ? 0xffffffffu : gx0 + 4 <= cnt_from ? 0u : (0xffffffffu << (8 * (cnt_from - gx0)));
    const unsigned ugx = (unsigned)gx0;                           // used by output lanes only (gx0 >= 0 there)
    // FAST path (W % 4 == 0, aligned rows): a lane left of the image or right of it loads the 4 columns of its
    // in-image neighbour (clamped address) and mirrors the stretched bytes (REFLECT_101) with one byte permute:
    //   left  [c0 c1 c2 c3] -> [ . c3 c2 c1]   (columns -4..-1; -4 is never a tap for R <= 3; R = 4: below)
    //   right [c0 c1 c2 c3] -> [c2 c1 c0  . ]  (columns W..W+3)
    // R = 4 (ksize 9): column -4 / W + 3 IS a tap - the border lane loads the four columns ONE further inside (1..4 / W-5..W-2) and
    // reverses them: [c1 c2 c3 c4] -> [c4 c3 c2 c1] = columns -4..-1, [W-5 .. W-2] -> [W-2 .. W-5] = columns W..W+3
    // (RH = 4 in general: the lane at columns g .. g + 3 outside the image holds the reversed columns -g - 3 .. -g / 2W - 5 - g .. 2W - 2 - g;
    //  lanes further out than the halo hold columns nobody reads - clamped into the image)
    const unsigned ugx_load = RH == 4 ? (unsigned)min(max(gx0 < 0 ? -gx0 - 3 : gx0 >= W ? 2 * W - 5 - gx0 : gx0, 0), W - 4) : (unsigned)min(max(gx0, 0), W - 4);
    const unsigned edge_sel = RH == 4 ? ((gx0 < 0 || gx0 >= W) ? 0x00010203u : 0x03020100u)
                                      : (gx0 < 0 ? 0x01020300u : gx0 >= W ? 0x03000102u : 0x03020100u);
    const bool col_inside = gx0 >= 0 && gx0 + 3 < W;
    const bool vec0 = col_inside && (stride0 % 4 == 0) && ((uintptr_t)img0 % (4 * sizeof(T)) == 0);
    const bool vec1 = col_inside && (stride1 % 4 == 0) && ((uintptr_t)img1 % (4 * sizeof(T)) == 0);
    int rc[4];                                                    // REFLECT_101 columns for lanes on the border
#pragma unroll
    for (int k = 0; k < 4; k++) rc[k] = km_reflect101(gx0 + k, W);
    const bool out_lane = lane >= HALO / 4 && lane <= 63 - HALO / 4 && gx0 < W;
    const int y0 = rowblock * rows_per_item, y1 = min(H, y0 + rows_per_item);

    // packed coefficients
    // (R = 4: nine taps = three dwords; the smoothing sum stays SIGNED there - sum ks (u - 128) spans [-32768, 32512], exactly an int16 -
    //  and needs no bias: the vertical derivative taps sum to zero, so a constant added to every smoothed row cancels in kd * hs)
    constexpr int NH = RH == 4 ? 3 : 2;
    int kdp[2][NH], ksp[2][NH], bias[2], vk[2][NR];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int sum = 0;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            unsigned a = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = 4 * h + k;
                const int d = t < NR ? cf.kd[i][t] : 0, sm = t < NR ? cf.ks[i][t] : 0;
                a |= ((unsigned)d & 0xffu) << (8 * k);
                b |= ((unsigned)sm & 0xffu) << (8 * k);
                sum += sm;
            }
            kdp[i][h] = (int)a; ksp[i][h] = (int)b;
        }
        bias[i] = 128 * sum;
#pragma unroll
        for (int t = 0; t < NR; t++) vk[i][t] = (cf.ks[i][t] & 0xffff) | (cf.kd[i][t] << 16);
    }

    // 16-bit dtypes: nodata as a packed pixel pair (0 = "no further condition": absent, non-integral or out of range)
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    auto nodata16 = [](int has, double v) -> uint32_t {
        if (!has || v != floor(v)) return 0u;
        if constexpr (std::is_signed<T>::value) { if (v < -32768.0 || v > 32767.0) return 0u; }
        else { if (v < 0.0 || v > 65535.0) return 0u; }
        const uint32_t x = (uint32_t)(uint16_t)(int)v;
        return x | (x << 16);
    };
    const uint32_t nd16_mon = nodata16(nd.has_mon, nd.mon), nd16_ref = nodata16(nd.has_ref, nd.ref);
    (void)nd16_mon; (void)nd16_ref;
    int vbias[2];   // bias as vector operands (see dot4_seed); the asm keeps them out of the scalar file
    asm volatile("v_mov_b32 %0, %1" : "=v"(vbias[0]) : "s"(bias[0]));
    asm volatile("v_mov_b32 %0, %1" : "=v"(vbias[1]) : "s"(bias[1]));
    unsigned cnt = 0;
    uint8_t *outs[2] = {out0, out1};
    // FAST: every lane of the strip is an interior, aligned lane (wave-uniform) -> no per-lane fallbacks in the loop
    auto march = [&](auto fast_tag, auto img_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr int IMG = decltype(img_tag)::value;                   // -1: both images in this wave, 0 / 1: only that one
    constexpr int I0 = IMG < 0 ? 0 : IMG, I1 = IMG < 0 ? 2 : IMG + 1;
    constexpr bool LOAD0 = IMG != 1, LOAD1 = IMG != 0 || MASK;      // the mask needs both raw rows (image-0 wave)
    auto load_raw = [&](int m, T (&v)[2][4]) {
        const int r = km_reflect101(m, H);
        const T *r0 = img0 + (size_t)r * stride0, *r1 = img1 + (size_t)r * stride1;
        const unsigned lx = opaque_lane_offset(ugx_load * (unsigned)sizeof(T));   // byte offset of the lane's first column
        if (!LOAD0) {
        } else if (FAST) {   // uniform row base + unsigned 32-bit lane offset: no per-lane 64-bit address arithmetic; any row alignment
            __builtin_memcpy(v[0], (const char *)r0 + lx, 4 * sizeof(T));
        } else if (vec0) {
            if constexpr (sizeof(T) == 1) { uint32_t q = *(const uint32_t *)(r0 + gx0); __builtin_memcpy(v[0], &q, 4); }
            else if constexpr (sizeof(T) == 2) { uint2 q = *(const uint2 *)(r0 + gx0); __builtin_memcpy(v[0], &q, 8); }
            else { uint4 q = *(const uint4 *)(r0 + gx0); __builtin_memcpy(v[0], &q, 16); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[0][k] = r0[rc[k]];
        }
        if (!LOAD1) {
        } else if (FAST) {
            __builtin_memcpy(v[1], (const char *)r1 + lx, 4 * sizeof(T));
        } else if (vec1) {
            if constexpr (sizeof(T) == 1) { uint32_t q = *(const uint32_t *)(r1 + gx0); __builtin_memcpy(v[1], &q, 4); }
            else if constexpr (sizeof(T) == 2) { uint2 q = *(const uint2 *)(r1 + gx0); __builtin_memcpy(v[1], &q, 8); }
            else { uint4 q = *(const uint4 *)(r1 + gx0); __builtin_memcpy(v[1], &q, 16); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[1][k] = r1[rc[k]];
        }
    };

    int hp[2][2][4];                                  // (R = 5) the binomial's two previous rows of horizontal sums
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) hp[i][r][j] = 0;
    (void)hp;
    int ring[2][NR][4];
#pragma unroll
    for (int i = I0; i < I1; i++)
#pragma unroll
        for (int k = 0; k < NR; k++)
#pragma unroll
            for (int j = 0; j < 4; j++) ring[i][k][j] = 0;
    // raw rows travel LAPM_PF rows ahead of their use (a short register FIFO; the copies disappear in the unrolled body)
    T nxt[LAPM_PF][2][4];
#pragma unroll
    for (int f = 0; f < LAPM_PF; f++) load_raw(min(y0 - R + f, y1 + R - 1), nxt[f]);
    for (int mbase = y0 - R; mbase < y1 + R; mbase += NR) {
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int m = mbase + k;                 // source row of this step (may lie outside: mirrored)
            if (m >= y1 + R) continue;               // (no break: ring indices must stay compile-time constants)
            T v[2][4];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    v[i][j] = nxt[0][i][j];
#pragma unroll
                    for (int f = 0; f + 1 < LAPM_PF; f++) nxt[f][i][j] = nxt[f + 1][i][j];
                }
            if (m + LAPM_PF < y1 + R) load_raw(m + LAPM_PF, nxt[LAPM_PF - 1]);
            // ---- auto mask of source row m (it is an output row when y0 <= m < y1)
            if constexpr (MASK && IMG != 1 && FAST && sizeof(T) == 2) {
                // packed form: a pixel pair is valid iff min(mon, ref, mon ^ nodata_mon, ref ^ nodata_ref) != 0 (unsigned)
                if (m >= y0 && m < y1 && out_lane) {
                    uint2 qm, qr;
                    __builtin_memcpy(&qm, v[1], 8); __builtin_memcpy(&qr, v[0], 8);
                    auto nz2 = [&](uint32_t a, uint32_t b) {
                        ushort2v mn2 = __builtin_elementwise_min(
                            __builtin_elementwise_min(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, b)),
                            __builtin_elementwise_min(__builtin_bit_cast(ushort2v, a ^ nd16_mon), __builtin_bit_cast(ushort2v, b ^ nd16_ref)));
                        mn2 = __builtin_elementwise_min(mn2, (ushort2v)(1));
                        return __builtin_bit_cast(uint32_t, mn2);
                    };
                    const uint32_t mp = __builtin_amdgcn_perm(nz2(qm.y, qr.y), nz2(qm.x, qr.x), 0x06040200u);
                    cnt += (unsigned)__popc(mp & cnt_mask);
                    __builtin_memcpy((mask_out + (size_t)m * W) + opaque_lane_offset(ugx), &mp, 4);
                }
            } else if constexpr (MASK && IMG != 1) {
                if (m >= y0 && m < y1 && out_lane) {
                    uint32_t mp = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool ok = (gx0 + j < W) && px_valid<T>(v[1][j], v[0][j], nd);
                        mp |= (ok ? 1u : 0u) << (8 * j);
                        cnt += ok && gx0 + j >= cnt_from;
                    }
                    const size_t o = (size_t)m * W + gx0;
                    if (FAST) __builtin_memcpy(mask_out + o, &mp, 4);
                    else if (gx0 + 3 < W && (o & 3) == 0) *(uint32_t *)(mask_out + o) = mp;
                    else {
                        for (int j = 0; j < 4 && gx0 + j < W; j++) mask_out[o + j] = (uint8_t)((mp >> (8 * j)) & 1u);
                    }
                }
            }
#pragma unroll
            for (int i = I0; i < I1; i++) {
                // ---- stretch to uint8 (biased by -128 for the signed dot products)
                uint32_t cw = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) cw |= st[i](v[i][j], nullptr) << (8 * j);
                cw ^= (i == 1 && invert1) ? 0x7f7f7f7fu : 0x80808080u;   // (255 - u) - 128 == u ^ 0x7f
                if constexpr (FAST) cw = __builtin_amdgcn_perm(cw, cw, edge_sel);                    // border lanes: mirrored columns
                const uint32_t lw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cw, 0x138, 0xf, 0xf, false);   // lane-1
                const uint32_t rw = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cw, 0x130, 0xf, 0xf, false);   // lane+1
                // ---- horizontal kd / ks passes: bytes [4+o-R, 4+o-R+8) of (lw | cw | rw)
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    constexpr int dummy = 0; (void)dummy;
                    if constexpr (RH == 4) {
                        // nine taps: bytes [o, o + 9) of (lw | cw | rw)
                        const int g0 = (int)__builtin_amdgcn_alignbyte(cw, lw, o), g1 = (int)__builtin_amdgcn_alignbyte(rw, cw, o),
                                  g2 = (int)__builtin_amdgcn_alignbyte(0u, rw, o);
                        const int vd = __builtin_amdgcn_sdot4(g0, kdp[i][0], __builtin_amdgcn_sdot4(g1, kdp[i][1], dot4_seed0(g2, kdp[i][2]), false), false);
                        const int vs = __builtin_amdgcn_sdot4(g0, ksp[i][0], __builtin_amdgcn_sdot4(g1, ksp[i][1], dot4_seed0(g2, ksp[i][2]), false), false);
                        ring[i][k][o] = (int)__builtin_amdgcn_perm((uint32_t)vs, (uint32_t)vd, 0x05040100u);
                        continue;
                    }
                    const int sft = 4 + o - R;                 // 1..4 for R = 3, 3..6 for R = 1
                    int g0, g1;
                    if (sft < 4) {
                        g0 = (int)__builtin_amdgcn_alignbyte(cw, lw, sft & 3);
                        g1 = (int)__builtin_amdgcn_alignbyte(rw, cw, sft & 3);
                    } else if (sft == 4) {
                        g0 = (int)cw; g1 = (int)rw;
                    } else {
                        g0 = (int)__builtin_amdgcn_alignbyte(rw, cw, (sft - 4) & 3);
                        g1 = (int)__builtin_amdgcn_alignbyte(0u, rw, (sft - 4) & 3);   // taps beyond 2R are zero
                    }
                    const int vd = __builtin_amdgcn_sdot4(g0, kdp[i][0], dot4_seed0(g1, kdp[i][1]), false);
                    const int vs = __builtin_amdgcn_sdot4(g0, ksp[i][0], dot4_seed(g1, ksp[i][1], vbias[i]), false);
                    ring[i][k][o] = (int)__builtin_amdgcn_perm((uint32_t)vs, (uint32_t)vd, 0x05040100u);   // (vd & 0xffff) | (vs << 16)
                }
            }
            // ---- vertical combine for output row y = m - R (ring slot of source row y - R + j is (k + 1 + j) mod NR)
            if constexpr (R == 5) {
                // kernel 11: the 9-tap row yy = m - 4 unsaturated (every lane: the binomial reads the neighbour lanes' values through DPP), its
                // horizontal [1 2 1], then the vertical [1 2 1] over the rows yy - 2 .. yy: output row yo = yy - 1.  An image whose kernel is
                // smaller (b3 = 0) passes its 9-tap-padded row through unchanged, one row late like the other.
                const int yy = m - RH, yo = yy - 1;
                if (yy >= y0 - 1) {
#pragma unroll
                    for (int i = I0; i < I1; i++) {
                        int a[4], h[4];
#pragma unroll
                        for (int o = 0; o < 4; o++) {
                            int acc = dot2_seed0(ring[i][(k + 1) % NR][o], vk[i][0]);
#pragma unroll
                            for (int j = 1; j < NR; j++)
                                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ring[i][(k + 1 + j) % NR][o]),
                                                             __builtin_bit_cast(short2v, vk[i][j]), acc, false);
                            a[o] = acc;
                        }
                        uint32_t packed = 0;
                        if (cf.b3[i]) {
                            const int al = __builtin_amdgcn_update_dpp(0, a[3], 0x138, 0xf, 0xf, false);      // lane - 1: the column left of this lane's
                            const int ar = __builtin_amdgcn_update_dpp(0, a[0], 0x130, 0xf, 0xf, false);      // lane + 1
#pragma unroll
                            for (int o = 0; o < 4; o++) h[o] = (o == 0 ? al : a[o - 1]) + 2 * a[o] + (o == 3 ? ar : a[o + 1]);
#pragma unroll
                            for (int o = 0; o < 4; o++) packed |= (uint32_t)min(max(hp[i][1][o] + 2 * hp[i][0][o] + h[o], 0), 255) << (8 * o);
                        } else {
#pragma unroll
                            for (int o = 0; o < 4; o++) { h[o] = a[o]; packed |= (uint32_t)min(max(hp[i][0][o], 0), 255) << (8 * o); }
                        }
#pragma unroll
                        for (int o = 0; o < 4; o++) { hp[i][1][o] = hp[i][0][o]; hp[i][0][o] = h[o]; }
                        if (yo >= y0 && out_lane) {
                            const size_t off = (size_t)yo * W + gx0;
                            if (FAST) __builtin_memcpy((outs[i] + (size_t)yo * W) + opaque_lane_offset(ugx), &packed, 4);
                            else if (gx0 + 3 < W && (off & 3) == 0) *(uint32_t *)(outs[i] + off) = packed;
                            else {
                                for (int j = 0; j < 4 && gx0 + j < W; j++) outs[i][off + j] = (uint8_t)(packed >> (8 * j));
                            }
                        }
                    }
                }
            } else {
            const int y = m - R;
            if (y >= y0 && out_lane) {
#pragma unroll
                for (int i = I0; i < I1; i++) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        int acc = dot2_seed0(ring[i][(k + 1) % NR][o], vk[i][0]);
#pragma unroll
                        for (int j = 1; j < NR; j++)
                            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, ring[i][(k + 1 + j) % NR][o]),
                                                         __builtin_bit_cast(short2v, vk[i][j]), acc, false);
                        packed |= (uint32_t)min(max(acc, 0), 255) << (8 * o);
                    }
                    const size_t off = (size_t)y * W + gx0;
                    if (FAST) __builtin_memcpy((outs[i] + (size_t)y * W) + opaque_lane_offset(ugx), &packed, 4);
                    else if (gx0 + 3 < W && (off & 3) == 0) *(uint32_t *)(outs[i] + off) = packed;
                    else {
                        for (int j = 0; j < 4 && gx0 + j < W; j++) outs[i][off + j] = (uint8_t)(packed >> (8 * j));
                    }
                }
            }
            }
        }
    }
    };  // march
    // FAST: every lane of the item either lies inside the image with its 4 columns or is a whole-lane mirror of its in-image neighbour
    // (wave-uniform).  Rows need no alignment - loads and stores are 4-column accesses at whatever address the row has (a 5490-column
    // tile: every other row sits off the dword grid; the per-pixel path there cost 1.75x).  The last strip of a width that is no multiple
    // of 4 is shifted (above); only a last strip of fewer than 4 columns (its left neighbour's border lane straddles the edge) and images of
    // a single strip still take the general path.
    const bool fast = (W % 4 == 0) || shifted || (strip * VALID - HALO + 4 * 64 <= W);
    if constexpr (SPLIT) {
        if (img_sel == 0) { if (fast) march(std::true_type{}, std::integral_constant<int, 0>{}); else march(std::false_type{}, std::integral_constant<int, 0>{}); }
        else { if (fast) march(std::true_type{}, std::integral_constant<int, 1>{}); else march(std::false_type{}, std::integral_constant<int, 1>{}); }
    } else {
        if (fast) march(std::true_type{}, std::integral_constant<int, -1>{});
        else march(std::false_type{}, std::integral_constant<int, -1>{});
    }
    if constexpr (MASK) {
        if (!SPLIT || img_sel == 0) {
            const unsigned c64 = (unsigned)wave_sum_u64((unsigned long long)cnt);
            if (lane == 0) valid_partial[wave_id] = c64;
        }
    }
}

// work item = (row block, column strip), strips fastest: the 4 waves of a workgroup take 4 consecutive items
template <int R, typename T, bool MASK, bool SPLIT>
__global__ __launch_bounds__(256) KM_LAPM_OCC void lap_march_kernel(const T *__restrict__ img0, const T *__restrict__ img1, int H, int W,
                                                        ptrdiff_t stride0, ptrdiff_t stride1, const double *__restrict__ mm,
                                                        lap_coef cf, int invert1, nodata_t nd,
                                                        uint8_t *__restrict__ out0, uint8_t *__restrict__ out1,
                                                        uint8_t *__restrict__ mask_out, unsigned *__restrict__ valid_partial, int nstrips,
                                                        int rows_per_item, int nitems)
{
    unsigned tile;
    if (!km_xcd_tile((unsigned)((SPLIT ? 2 : 1) * nitems + 3) / 4u, tile)) return;
    const int wave_lin = (int)tile * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // readfirstlane: the compiler must know it is wave-uniform
    lap_march_item<R, T, MASK, SPLIT>(img0, img1, H, W, stride0, stride1, mm, cf, invert1, nd, out0, out1, mask_out, valid_partial, nstrips, rows_per_item,
                                      nitems, wave_lin);
}

// batched units: the units' work items form ONE linear item space (unit u owns [item0[u], item0[u + 1])): a wavefront finds its unit
// with a scalar scan of <= 16 bounds and then runs exactly the item of the single-unit kernel on that unit's rasters
struct lapm_units_args {
    const void *img0[KM_UNITS_MAX], *img1[KM_UNITS_MAX];
    ptrdiff_t s0[KM_UNITS_MAX], s1[KM_UNITS_MAX];
    const double *mm[KM_UNITS_MAX];
    uint8_t *out0[KM_UNITS_MAX], *out1[KM_UNITS_MAX], *mask[KM_UNITS_MAX];
    unsigned *valid[KM_UNITS_MAX];
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX], nstrips[KM_UNITS_MAX];
    int item0[KM_UNITS_MAX + 1];
    int n, rows;
};
template <int R, typename T, bool MASK>
__global__ __launch_bounds__(256) KM_LAPM_OCC void lap_march_units_kernel(lapm_units_args U, lap_coef cf, int invert1, nodata_t nd)
{
    const int total = U.item0[U.n];
    unsigned tile;
    if (!km_xcd_tile((unsigned)(total + 3) / 4u, tile)) return;
    const int wave_lin = (int)tile * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave_lin >= total) return;
    int u = 0;
    while (u + 1 < U.n && wave_lin >= U.item0[u + 1]) u++;
    lap_march_item<R, T, MASK, false>((const T *)U.img0[u], (const T *)U.img1[u], U.H[u], U.W[u], U.s0[u], U.s1[u], U.mm[u], cf, invert1, nd, U.out0[u], U.out1[u],
                                      U.mask[u], U.valid[u], U.nstrips[u], U.rows, U.item0[u + 1] - U.item0[u], wave_lin - U.item0[u]);
}

template <typename T, bool MASK>
static int launch_lap_march(km_ctx *c, int R, const T *a, const T *b, int H, int W, ptrdiff_t sa, ptrdiff_t sb, const double *mm,
                            const lap_coef &cf, int invert1, const nodata_t &nd, uint8_t *oa, uint8_t *ob,
                            uint8_t *mask, unsigned long long *valid_out)
{
    const int nstrips = (W + LAPM_VALID_OF(R) - 1) / LAPM_VALID_OF(R);
    // resident waves: 4 SIMDs per CU x the waves per SIMD the register budget of this instantiation allows
    auto slots_of = [&](const void *fn) -> long {
        int wg_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, fn, 256, 0) != hipSuccess || wg_per_cu < 1) wg_per_cu = 4;
        return (long)c->n_cu * 4 * wg_per_cu;
    };
    constexpr bool SPLIT = LAPM_SPLIT != 0;
    const void *fn = R == 1 ? (const void *)lap_march_kernel<1, T, MASK, SPLIT> : R == 2 ? (const void *)lap_march_kernel<2, T, MASK, SPLIT>
                   : R == 3 ? (const void *)lap_march_kernel<3, T, MASK, SPLIT> : R == 4 ? (const void *)lap_march_kernel<4, T, MASK, SPLIT>
                   : (const void *)lap_march_kernel<5, T, MASK, SPLIT>;
    int rows = km_pick_rows(H, (SPLIT ? 2 : 1) * nstrips, 2 * R, slots_of(fn), 32, 160);
    if (const char *e = km_dev_env("KARIOS_HIP_LAP_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 4096) rows = v; }   // tuning override
    const int nitems = nstrips * ((H + rows - 1) / rows);
    const unsigned ntiles = (unsigned)((SPLIT ? 2 : 1) * nitems + 3) / 4u;
    dim3 grid(km_xcd_grid(ntiles));
    const size_t nwaves = SPLIT ? (size_t)nitems : (size_t)ntiles * 4;     // entries of the per-item valid counts
    unsigned *valid = nullptr;
    if (MASK) {
        // (a slot of its own: with the sum deferred to the second stream - below - the eigenvalue pass, which owns WS_PARTIAL, runs first)
        valid = (unsigned *)km_ws(c, WS_LAP_VALID, ((size_t)ntiles * 4 + 4) * sizeof(unsigned));
        if (!valid) return KM_E_NOMEM;
    }
    switch (R) {
    case 1: lap_march_kernel<1, T, MASK, SPLIT><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, mask, valid, nstrips, rows, nitems); break;
    case 2: lap_march_kernel<2, T, MASK, SPLIT><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, mask, valid, nstrips, rows, nitems); break;
    case 3: lap_march_kernel<3, T, MASK, SPLIT><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, mask, valid, nstrips, rows, nitems); break;
    case 4: lap_march_kernel<4, T, MASK, SPLIT><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, mask, valid, nstrips, rows, nitems); break;
    case 5: lap_march_kernel<5, T, MASK, SPLIT><<<grid, 256, 0, c->stream>>>(a, b, H, W, sa, sb, mm, cf, invert1, nd, oa, ob, mask, valid, nstrips, rows, nitems); break;
    default: return km_fail(c, KM_E_INTERNAL, "lap_march radius %d", R);
    }
    KM_LAUNCH_CHECK(c);
    if (MASK) {
        if (c->defer_valid_sum) {
            // the count of valid pixels is only read at the end of the unit (frame header, statistics): its one-workgroup sum leaves
            // the critical path - klt_track_dev launches it on the second stream in front of the pyramids (kd_run_valid_sum)
            c->valid_job_partial = valid; c->valid_job_n = (unsigned)nwaves; c->valid_job_out = valid_out; c->valid_job_pending = true;
        } else {
            sum_u32_kernel<<<1, 1024, 0, c->stream>>>(valid, (unsigned)nwaves, valid_out);
            KM_LAUNCH_CHECK(c);
        }
    }
    return KM_OK;
}

// the deferred sum of launch_lap_march, on whatever stream c->stream is at the moment
int kd_run_valid_sum(km_ctx *c)
{
    if (!c->valid_job_pending) return KM_OK;
    c->valid_job_pending = false;
    sum_u32_kernel<<<1, 1024, 0, c->stream>>>(c->valid_job_partial, c->valid_job_n, c->valid_job_out);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

static int lap_radius(int ksize) { return ksize == 1 ? 1 : ksize / 2; }

// ---- batched units: stretch + Laplacians + automatic mask of every unit in ONE launch; the per-item counts of valid pixels are summed
// per unit by kd_valid_sum_units (one workgroup per unit, on whatever stream c->stream is: the caller puts it beside the pyramids)
// (ONE wavefront per unit: the kernel runs beside the previous submission's LK, whose single-wave workgroups refill every slot a
// retiring wave leaves - a 1024-thread workgroup waited there for the whole launch, 2.6 ms with the pyramids queued behind it, and a
// 256-thread one still 2.4 ms: profiles/timeline_r06_c4.txt)
__global__ __launch_bounds__(64) void valid_sum_units_kernel(km_valid_units J)
{
    const unsigned *partial = J.partial[blockIdx.x];
    const unsigned n = J.n_partial[blockIdx.x];
    unsigned long long s = 0;
    for (unsigned i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum_u64(s);
    if (threadIdx.x == 0) *J.out[blockIdx.x] = s;
}

int kd_valid_sum_units(km_ctx *c, const km_valid_units &J)
{
    if (J.n <= 0) return KM_OK;
    valid_sum_units_kernel<<<J.n, 64, 0, c->stream>>>(J);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// User masks of a batch (klt.py:258-266: the caller's raster instead of the automatic mask): every unit's box of its mask raster is
// packed into the unit's dense mask plane (the eigenvalue pass indexes masks with the image width) and its non-zero pixels are counted
// - what hipMemcpy2DAsync + kd_count_nonzero do for a single unit, for all units in one launch.  blockIdx.y = unit; a workgroup takes
// every gridDim.x-th group of rows; 16 bytes per thread at whatever address a row has.
struct mask_units_args {
    const uint8_t *src[KM_UNITS_MAX];
    uint8_t *dst[KM_UNITS_MAX];
    ptrdiff_t stride[KM_UNITS_MAX];
    unsigned *partial[KM_UNITS_MAX];
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX];
};
__global__ __launch_bounds__(256) void mask_pack_units_kernel(mask_units_args A)
{
    const int u = blockIdx.y, H = A.H[u], W = A.W[u];
    const uint8_t *__restrict__ src = A.src[u];
    uint8_t *__restrict__ dst = A.dst[u];
    const ptrdiff_t stride = A.stride[u];
    unsigned cnt = 0;
    for (int y = blockIdx.x; y < H; y += gridDim.x) {
        const uint8_t *r = src + (size_t)y * stride;
        uint8_t *w = dst + (size_t)y * W;
        for (int x = 16 * threadIdx.x; x < W; x += 16 * 256) {
            if (x + 16 <= W) {
                uint4 q;
                __builtin_memcpy(&q, r + x, 16);
                __builtin_memcpy(w + x, &q, 16);
                const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; k++) cnt += (unsigned)__popc(((d[k] | ((d[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u));   // bytes != 0
            } else {
                for (int k = x; k < W; k++) { const uint8_t b = r[k]; w[k] = b; cnt += b != 0; }
            }
        }
    }
    const unsigned long long s = wave_sum_u64((unsigned long long)cnt);
    __shared__ unsigned sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (unsigned)s;
    __syncthreads();
    if (threadIdx.x == 0) A.partial[u][blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename T>
static int launch_lap_march_units(km_ctx *c, int R, const km_units &U, const lap_coef &cf, int invert1, const nodata_t &nd, km_valid_units *job)
{
    lapm_units_args A;
    A.n = U.n;
    for (int u = 0; u < U.n; u++) {
        A.img0[u] = U.ref[u]; A.img1[u] = U.mon[u]; A.s0[u] = U.sref[u]; A.s1[u] = U.smon[u]; A.mm[u] = U.mm[u];
        A.out0[u] = U.lap_ref[u]; A.out1[u] = U.lap_mon[u]; A.mask[u] = U.mask[u];
        A.H[u] = U.H[u]; A.W[u] = U.W[u]; A.nstrips[u] = (U.W[u] + LAPM_VALID_OF(R) - 1) / LAPM_VALID_OF(R);
    }
    int wg_per_cu = 0;
    const void *fn = R == 1 ? (const void *)lap_march_units_kernel<1, T, true> : R == 2 ? (const void *)lap_march_units_kernel<2, T, true>
                   : R == 3 ? (const void *)lap_march_units_kernel<3, T, true> : R == 4 ? (const void *)lap_march_units_kernel<4, T, true>
                   : (const void *)lap_march_units_kernel<5, T, true>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, fn, 256, 0) != hipSuccess || wg_per_cu < 1) wg_per_cu = 4;
    const long slots = (long)c->n_cu * 4 * wg_per_cu;
    // rows per item: the value in [32, 160] that minimises whole rounds of resident waves x the work of one item, over ALL units' strips
    int rows = 32;
    {
        double best = 1e300;
        for (int r = 32; r <= 160; r++) {
            long items = 0;
            for (int u = 0; u < U.n; u++) items += (long)A.nstrips[u] * ((U.H[u] + r - 1) / r);
            const long rounds = (items + slots - 1) / slots;
            const double last = (double)(items - (rounds - 1) * slots) / (double)slots;
            const double cost = ((double)(rounds - 1) + 0.5 + 0.5 * last) * (double)(r + 2 * R);
            if (cost < best) { best = cost; rows = r; }
        }
    }
    if (const char *e = km_dev_env("KARIOS_HIP_LAP_ROWS")) { const int v = atoi(e); if (v >= 8 && v <= 4096) rows = v; }   // tuning override
    A.rows = rows;
    A.item0[0] = 0;
    for (int u = 0; u < U.n; u++) A.item0[u + 1] = A.item0[u] + A.nstrips[u] * ((U.H[u] + rows - 1) / rows);
    const int total = A.item0[U.n];
    const int mask_wgs = 256;                         // workgroups per unit of the user-mask pack
    const size_t n_partial = U.has_user_mask ? (size_t)mask_wgs * U.n : (size_t)total + 4 * KM_UNITS_MAX;
    unsigned *valid = (unsigned *)km_ws(c, WS_LAP_VALID, n_partial * sizeof(unsigned));
    if (!valid) return KM_E_NOMEM;
    job->n = U.n;
    const dim3 grid(km_xcd_grid((unsigned)(total + 3) / 4u));
    if (U.has_user_mask) {
        // the caller's mask: packed + counted here, the Laplacian pass derives none (MASK = false: it neither reads nor writes a mask)
        mask_units_args M;
        for (int u = 0; u < U.n; u++) {
            M.src[u] = U.user_mask[u]; M.dst[u] = U.mask[u]; M.stride[u] = U.user_smask[u]; M.H[u] = U.H[u]; M.W[u] = U.W[u];
            M.partial[u] = valid + (size_t)mask_wgs * u;
            A.valid[u] = nullptr;
            job->partial[u] = M.partial[u]; job->n_partial[u] = (unsigned)mask_wgs; job->out[u] = &U.sc[u]->valid;
        }
        mask_pack_units_kernel<<<dim3(mask_wgs, U.n), 256, 0, c->stream>>>(M);
        KM_LAUNCH_CHECK(c);
        switch (R) {
        case 1: lap_march_units_kernel<1, T, false><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
        case 2: lap_march_units_kernel<2, T, false><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
        case 3: lap_march_units_kernel<3, T, false><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
        case 4: lap_march_units_kernel<4, T, false><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
        case 5: lap_march_units_kernel<5, T, false><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
        default: return km_fail(c, KM_E_INTERNAL, "lap_march radius %d", R);
        }
        KM_LAUNCH_CHECK(c);
        return KM_OK;
    }
    for (int u = 0; u < U.n; u++) {
        A.valid[u] = valid + A.item0[u];
        job->partial[u] = A.valid[u]; job->n_partial[u] = (unsigned)(A.item0[u + 1] - A.item0[u]); job->out[u] = &U.sc[u]->valid;
    }
    switch (R) {
    case 1: lap_march_units_kernel<1, T, true><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
    case 2: lap_march_units_kernel<2, T, true><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
    case 3: lap_march_units_kernel<3, T, true><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
    case 4: lap_march_units_kernel<4, T, true><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
    case 5: lap_march_units_kernel<5, T, true><<<grid, 256, 0, c->stream>>>(A, cf, invert1, nd); break;
    default: return km_fail(c, KM_E_INTERNAL, "lap_march radius %d", R);
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// The marching kernel at radius 5: every image's kernel as a 9-tap pass, kernel 11 as the 9-tap pass of kernel 9 + the 3 x 3 binomial
// (lap_march_item).  Coefficients centred at radius 4.
static bool fill_coef_march5(int ksize, int *kd, int *ks, int *b3)
{
    *b3 = ksize == 11 ? 1 : 0;
    return fill_coef(ksize == 11 ? 9 : ksize, 4, kd, ks);
}

// KM_E_UNSUPPORTED (no message) when the batch form does not cover the case (tiny units): the caller submits the units one by one instead
int kd_stretch_laplacian_units(km_ctx *c, const km_units &U, int ksize_ref, int ksize_mon, int invert_mon, const double *nodata_ref,
                               const double *nodata_mon, km_valid_units *job)
{
    lap_coef cf;
    const int R = lap_radius(ksize_ref) > lap_radius(ksize_mon) ? lap_radius(ksize_ref) : lap_radius(ksize_mon);
    auto okk = [](int k) { return k >= 1 && k <= 11 && (k & 1); };
    if (!okk(ksize_ref) || !okk(ksize_mon) || !fill_coef(ksize_ref, R, cf.kd[0], cf.ks[0]) || !fill_coef(ksize_mon, R, cf.kd[1], cf.ks[1]))
        return km_fail(c, KM_E_UNSUPPORTED, "Laplacian ksize ref=%d mon=%d (supported: 1,3,5,7,9,11)", ksize_ref, ksize_mon);
    for (int u = 0; u < U.n; u++)
        if (U.W[u] < 8 || U.H[u] < 8) return KM_E_UNSUPPORTED;
    const nodata_t nd = make_nodata(nodata_mon, nodata_ref);
    if (R == 5) {
        for (int u = 0; u < U.n; u++)
            if (U.W[u] < 16 || U.H[u] < 16) return KM_E_UNSUPPORTED;
        if (!fill_coef_march5(ksize_ref, cf.kd[0], cf.ks[0], &cf.b3[0]) || !fill_coef_march5(ksize_mon, cf.kd[1], cf.ks[1], &cf.b3[1]))
            return km_fail(c, KM_E_INTERNAL, "laplacian coefficients");
    }
    switch (U.dtype) {
    case KM_U8: return launch_lap_march_units<uint8_t>(c, R, U, cf, invert_mon, nd, job);
    case KM_U16: return launch_lap_march_units<uint16_t>(c, R, U, cf, invert_mon, nd, job);
    case KM_I16: return launch_lap_march_units<int16_t>(c, R, U, cf, invert_mon, nd, job);
    case KM_F32: return launch_lap_march_units<float>(c, R, U, cf, invert_mon, nd, job);
    default: return km_fail(c, KM_E_ARG, "stretch_laplacian: bad dtype %d", U.dtype);
    }
}

int kd_laplacian_u8(km_ctx *c, const uint8_t *d_src, int H, int W, int ksize, uint8_t *d_dst)
{
    lap_coef cf;
    const int R = lap_radius(ksize);
    if (ksize < 1 || ksize > 11 || !(ksize & 1) || !fill_coef(ksize, R, cf.kd[0], cf.ks[0]))
        return km_fail(c, KM_E_UNSUPPORTED, "Laplacian ksize %d (supported: 1,3,5,7,9,11)", ksize);
    for (int i = 0; i < 11; i++) { cf.kd[1][i] = 0; cf.ks[1][i] = 0; }
    nodata_t nd = make_nodata(nullptr, nullptr);
    return launch_lap<uint8_t, 1, false>(c, R, d_src, d_src, H, W, W, W, nullptr, cf, 0, nd, d_dst, nullptr, nullptr, nullptr);
}

int kd_stretch_laplacian_pair(km_ctx *c, const void *d_ref, const void *d_mon, int dtype, int H, int W, ptrdiff_t sref,
                              ptrdiff_t smon, const double *d_mm, int ksize_ref, int ksize_mon, int invert_mon,
                              const double *nodata_ref, const double *nodata_mon, uint8_t *d_lap_ref, uint8_t *d_lap_mon,
                              uint8_t *d_mask_out, unsigned long long *d_valid)
{
    lap_coef cf;
    const int R = lap_radius(ksize_ref) > lap_radius(ksize_mon) ? lap_radius(ksize_ref) : lap_radius(ksize_mon);
    auto okk = [](int k) { return k >= 1 && k <= 11 && (k & 1); };
    if (!okk(ksize_ref) || !okk(ksize_mon) || !fill_coef(ksize_ref, R, cf.kd[0], cf.ks[0]) ||
        !fill_coef(ksize_mon, R, cf.kd[1], cf.ks[1]))
        return km_fail(c, KM_E_UNSUPPORTED, "Laplacian ksize ref=%d mon=%d (supported: 1,3,5,7,9,11)", ksize_ref, ksize_mon);
    nodata_t nd = make_nodata(nodata_mon, nodata_ref);
    lap_coef cf5;
    const bool march5 = R == 5 && W >= 16 && H >= 16 && fill_coef_march5(ksize_ref, cf5.kd[0], cf5.ks[0], &cf5.b3[0]) &&
                        fill_coef_march5(ksize_mon, cf5.kd[1], cf5.ks[1], &cf5.b3[1]);
    if (march5) cf = cf5;
    if ((R <= 4 && W >= 8 && H >= 8) || march5) {
#define KM_PAIRM(T)                                                                                                              \
    (d_mask_out ? launch_lap_march<T, true>(c, R, (const T *)d_ref, (const T *)d_mon, H, W, sref, smon, d_mm, cf, invert_mon, nd, \
                                            d_lap_ref, d_lap_mon, d_mask_out, d_valid)                                           \
                : launch_lap_march<T, false>(c, R, (const T *)d_ref, (const T *)d_mon, H, W, sref, smon, d_mm, cf, invert_mon, nd, \
                                             d_lap_ref, d_lap_mon, nullptr, nullptr))
        switch (dtype) {
        case KM_U8: return KM_PAIRM(uint8_t);
        case KM_U16: return KM_PAIRM(uint16_t);
        case KM_I16: return KM_PAIRM(int16_t);
        case KM_F32: return KM_PAIRM(float);
        default: return km_fail(c, KM_E_ARG, "stretch_laplacian: bad dtype %d", dtype);
        }
#undef KM_PAIRM
    }
#define KM_PAIR(T)                                                                                                          \
    (d_mask_out ? launch_lap<T, 2, true>(c, R, (const T *)d_ref, (const T *)d_mon, H, W, sref, smon, d_mm, cf, invert_mon, nd, \
                                         d_lap_ref, d_lap_mon, d_mask_out, d_valid)                                        \
                : launch_lap<T, 2, false>(c, R, (const T *)d_ref, (const T *)d_mon, H, W, sref, smon, d_mm, cf, invert_mon, nd, \
                                          d_lap_ref, d_lap_mon, nullptr, nullptr))
    switch (dtype) {
    case KM_U8: return KM_PAIR(uint8_t);
    case KM_U16: return KM_PAIR(uint16_t);
    case KM_I16: return KM_PAIR(int16_t);
    case KM_F32: return KM_PAIR(float);
    default: return km_fail(c, KM_E_ARG, "stretch_laplacian: bad dtype %d", dtype);
    }
#undef KM_PAIR
}

// ------------------------------------------------------------------ K3 min-eigenvalue map
// Output tile 64x32.  Exact-integer Sobel products and box sums (<= 31x31 window fits int32),
// one conversion to f32, then OpenCV's calcMinEigenVal formula with every f32 op rounded
// separately.  cov's own REFLECT_101 border (boxFilter) is honoured by evaluating the Sobel
// at the reflected position, NOT by reflecting the image under the window.
#define EIG_TW 64
#define EIG_TH 32
#define EIG_SEG 16

__device__ __forceinline__ unsigned eig_key(float f)
{
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float eig_unkey(unsigned k)
{
    unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
#endif
}

__global__ __launch_bounds__(256) void eig_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H,
                                                  int W, int block, double scale2, float *__restrict__ eig,
                                                  unsigned int *__restrict__ max_partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = block / 2, Rr = block - 1 - L;
    const int PW = EIG_TW + L + Rr, PH = EIG_TH + L + Rr;  // product region
    const int LW = (PW + 2 + 3) & ~3, LH = PH + 2;          // lap tile (1-px Sobel halo), row padded to 4
    uint8_t *lap = smem;                                    // [LH][LW]
    int *dxy = (int *)(smem + (((size_t)LH * LW + 15) & ~(size_t)15));  // [PH][PW] packed (dx | dy<<16)
    int *hsum = dxy + (size_t)PH * PW;                      // [3][PH][EIG_TW]
    const int X0 = blockIdx.x * EIG_TW, Y0 = blockIdx.y * EIG_TH;
    const int tid = threadIdx.x;
    const int lx0 = X0 - L - 1, ly0 = Y0 - L - 1;           // global coords of lap[0][0]

    for (int i = tid; i < LH * LW; i += 256) {
        const int r = i / LW, cx = i - r * LW;
        lap[i] = src[(size_t)km_reflect101(ly0 + r, H) * W + km_reflect101(lx0 + cx, W)];
    }
    __syncthreads();

    const int xlim = min(X0 + EIG_TW, W) - 1 + Rr, ylim = min(Y0 + EIG_TH, H) - 1 + Rr;
    for (int i = tid; i < PH * PW; i += 256) {
        const int r = i / PW, cx = i - r * PW;
        const int gx = X0 - L + cx, gy = Y0 - L + r;
        int packed = 0;
        if (gx <= xlim && gy <= ylim) {
            const int qx = km_reflect101(gx, W) - lx0, qy = km_reflect101(gy, H) - ly0;
            const uint8_t *p = lap + (size_t)qy * LW + qx;
            const int a00 = p[-LW - 1], a01 = p[-LW], a02 = p[-LW + 1];
            const int a10 = p[-1], a12 = p[1];
            const int a20 = p[LW - 1], a21 = p[LW], a22 = p[LW + 1];
            const int dx = (a02 + 2 * a12 + a22) - (a00 + 2 * a10 + a20);
            const int dy = (a20 + 2 * a21 + a22) - (a00 + 2 * a01 + a02);
            packed = (dx & 0xffff) | (dy << 16);
        }
        dxy[i] = packed;
    }
    __syncthreads();

    // horizontal box sums: one row segment of EIG_SEG outputs per work item (sliding window)
    const int nseg = EIG_TW / EIG_SEG;
    for (int it = tid; it < PH * nseg; it += 256) {
        const int r = it / nseg, sg = it - r * nseg;
        const int *row = dxy + (size_t)r * PW + sg * EIG_SEG;
        int sxx = 0, sxy = 0, syy = 0;
        for (int k = 0; k < block; k++) {
            const int v = row[k];
            const int dx = (int)(short)(v & 0xffff), dy = v >> 16;
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
        int *o0 = hsum + (size_t)r * EIG_TW + sg * EIG_SEG;
        int *o1 = o0 + (size_t)PH * EIG_TW, *o2 = o1 + (size_t)PH * EIG_TW;
        o0[0] = sxx; o1[0] = sxy; o2[0] = syy;
        for (int x = 1; x < EIG_SEG; x++) {
            const int vo = row[x - 1], vn = row[x - 1 + block];
            const int dxo = (int)(short)(vo & 0xffff), dyo = vo >> 16;
            const int dxn = (int)(short)(vn & 0xffff), dyn = vn >> 16;
            sxx += dxn * dxn - dxo * dxo; sxy += dxn * dyn - dxo * dyo; syy += dyn * dyn - dyo * dyo;
            o0[x] = sxx; o1[x] = sxy; o2[x] = syy;
        }
    }
    __syncthreads();

    // vertical box sums + eigenvalue: thread = column x, 8 consecutive rows
    const int x = tid & 63, yc = (tid >> 6) * (EIG_TH / 4);
    const int gx = X0 + x;
    float best = -INFINITY;
    bool have = false;
    if (gx < W) {
        const int *h0 = hsum + x, *h1 = h0 + (size_t)PH * EIG_TW, *h2 = h1 + (size_t)PH * EIG_TW;
        int sa = 0, sb = 0, sc = 0;
        for (int k = 0; k < block; k++) {
            sa += h0[(size_t)(yc + k) * EIG_TW]; sb += h1[(size_t)(yc + k) * EIG_TW]; sc += h2[(size_t)(yc + k) * EIG_TW];
        }
        for (int r = 0; r < EIG_TH / 4; r++) {
            const int gy = Y0 + yc + r;
            if (gy >= H) break;
            if (r > 0) {
                const size_t o = (size_t)(yc + r - 1) * EIG_TW, n = (size_t)(yc + r - 1 + block) * EIG_TW;
                sa += h0[n] - h0[o]; sb += h1[n] - h1[o]; sc += h2[n] - h2[o];
            }
            const float cxx = (float)__dmul_rn((double)sa, scale2);
            const float cxy = (float)__dmul_rn((double)sb, scale2);
            const float cyy = (float)__dmul_rn((double)sc, scale2);
            const float a = __fmul_rn(cxx, 0.5f), b = cxy, cc = __fmul_rn(cyy, 0.5f);
            const float t = __fsub_rn(a, cc);
            const float s = __fadd_rn(__fmul_rn(t, t), __fmul_rn(b, b));
            const float e = __fsub_rn(__fadd_rn(a, cc), sqrtf(s));
            const size_t o = (size_t)gy * W + gx;
            eig[o] = e;
            if (!mask || mask[o]) { best = have ? fmaxf(best, e) : e; have = true; }
        }
    }
    unsigned key = have ? eig_key(best) : 0u;
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    __shared__ unsigned s_key[4];
    if ((tid & 63) == 0) s_key[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) max_partial[blockIdx.y * gridDim.x + blockIdx.x] = max(max(s_key[0], s_key[1]), max(s_key[2], s_key[3]));
}

// ---- K3, fast path: one wavefront marches down a 64-column strip (64 - block - 1 output columns).
// Each lane owns an image column.  Per row: 3-row register window -> Sobel with the two neighbour lanes
// (DPP wave shifts) -> integer products -> horizontal box sum = difference of a wave prefix sum (DPP scan +
// two ds_bpermute) -> vertical box sum = running sum over a `block`-deep register ring -> eigenvalue.
// No LDS tiles, no barriers; exact integer arithmetic identical to the tiled kernel above.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_get0(int v)
{
    // full row mask: bound_ctrl makes the lanes without a source read 0, no register has to be cleared first
    if constexpr (ROW_MASK == 0xf) return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
    else return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}

// Correctly rounded float32 square root for x == 0 or x >= 2^-96 (the structure-tensor discriminant is 0 or >= 1e-18):
// the hardware estimate (<= 1 ulp) corrected with two fused residuals - the sequence the compiler emits for sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt, without its rescaling of tiny arguments and its inf/zero special case
// (for x == 0 the estimate is 0, both residual tests fail on NaN / 0 and 0 is returned).
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    const float r = __builtin_amdgcn_sqrtf(x);
    const float r_dn = __int_as_float(__float_as_int(r) - 1), r_up = __int_as_float(__float_as_int(r) + 1);
    const float e_dn = __builtin_fmaf(-r_dn, r, x), e_up = __builtin_fmaf(-r_up, r, x);
    float res = e_dn <= 0.f ? r_dn : r;
    res = e_up > 0.f ? r_up : res;
    return res;
}

__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += dpp_get0<0x111, 0xf>(v);  // row_shr:1
    v += dpp_get0<0x112, 0xf>(v);  // row_shr:2
    v += dpp_get0<0x114, 0xf>(v);  // row_shr:4
    v += dpp_get0<0x118, 0xf>(v);  // row_shr:8
    v += dpp_get0<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_get0<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

#define EIG_RS 128  // output rows per wave segment
#ifndef EIGM_CH
#define EIGM_CH 3     // rows fetched ahead per batch (3: 78 VGPRs = 6 waves per SIMD at blockSize 15; 15: 104 VGPRs = 4 waves)
#endif

__device__ __forceinline__ float dpp_shr1(float v)  // value of lane-1 (0 for lane 0)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float v)  // value of lane+1 (0 for lane 63)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// Writes the eig map + per-wave masked maxima (fallback of k_eig2.hip for images too small for its 128-column strips).
template <int BLOCK>
__global__ __launch_bounds__(256) KM_EIGM_OCC void eig_march_kernel(const uint8_t *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W,
                                                        double scale2, float *__restrict__ eig, unsigned int *__restrict__ max_partial,
                                                        int nstrips, int gyw)
{
    constexpr int L = BLOCK / 2, Rr = BLOCK - 1 - L, VALID = 64 - BLOCK - 1;
    constexpr int STRIDE = VALID;
    const int lane = threadIdx.x & 63;
    const int gxw = (nstrips + 3) / 4;                  // workgroups per row block (logical grid gxw x gyw, XCD-swizzled)
    unsigned tile;
    if (!km_xcd_tile((unsigned)(gxw * gyw), tile)) return;
    const int bx = (int)tile % gxw, by = (int)tile / gxw;
    const int strip = bx * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, known to the compiler
    const int wave_id = by * (gxw * 4) + strip;
    if (strip >= nstrips) { if (lane == 0) max_partial[wave_id] = 0u; return; }
    const int xs = strip * STRIDE;
    const int gx = xs - (L + 1) + lane;                  // image column of this lane (may be outside)
    const int cx = km_reflect101(gx, W);                 // column the lane reads (REFLECT_101 of the image)
    const bool xborder = (xs - (L + 1) < 0) || (xs - (L + 1) + 63 >= W);
    // cov's own REFLECT_101 border: an outside column takes the products of the lane holding its mirror column
    const int psrc = (cx - (xs - (L + 1))) * 4;
    const bool out_lane = lane >= L + 1 && lane < L + 1 + VALID && gx < W;
    const int hi_addr = min(lane + Rr, 63) * 4, lo_addr = max(lane - L - 1, 0) * 4;
    const bool lo_zero = lane - L - 1 < 0;
    const int y0 = by * EIG_RS, y1 = min(H, y0 + EIG_RS);
    const uint8_t *col = src + cx;

    int ring[BLOCK][3];
#pragma unroll
    for (int k = 0; k < BLOCK; k++) { ring[k][0] = 0; ring[k][1] = 0; ring[k][2] = 0; }
    int V0 = 0, V1 = 0, V2 = 0;
    int c1 = -1, c2 = -1;            // image rows cached in a1, a2
    int a0 = 0, a1 = 0, a2 = 0;
    float best = 0.f;
    bool have = false;

    for (int mbase = y0 - L; mbase < y1 + Rr; mbase += BLOCK) {
        // steady state (no row mirrored in this group of BLOCK steps): issue all BLOCK row loads (and the mask
        // bytes of the rows completed here) up front so their latency overlaps the arithmetic
        const bool steady = mbase - 1 >= 0 && mbase + BLOCK <= H - 1;
        // rows are fetched EIGM_CH at a time (source byte + mask byte of the row completed then): a deeper prefetch costs
        // registers, i.e. resident waves, and the resident waves are what hides the latency of this kernel
        constexpr int CH = BLOCK < EIGM_CH ? BLOCK : EIGM_CH;
        int pre[CH], pmask[CH];
        if (steady) {
            if (!(c1 == mbase - 1 && c2 == mbase)) {
                a1 = col[(size_t)(mbase - 1) * W]; a2 = col[(size_t)mbase * W];
                c1 = mbase - 1; c2 = mbase;
            }
        }
#pragma unroll
        for (int k = 0; k < BLOCK; k++) {
            if (k % CH == 0) {
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    if (k + j < BLOCK) {
                        if (steady) pre[j] = col[(size_t)(mbase + k + j + 1) * W];
                        pmask[j] = 1;
                        const int yq = mbase + k + j - Rr;
                        if (mask && out_lane && yq >= y0 && yq < y1) pmask[j] = mask[(size_t)yq * W + gx];
                    }
                }
            }
            const int m = mbase + k;                      // marching row (product row index, may be outside)
            if (m >= y1 + Rr) continue;                   // (no break: the ring index must stay a compile-time constant)
            if (steady) {
                a0 = a1; a1 = a2; a2 = pre[k % CH];
                c1 = m; c2 = m + 1;
            } else {
                const int r = km_reflect101(m, H);        // product row actually evaluated
                const int n0 = km_reflect101(r - 1, H), n2 = km_reflect101(r + 1, H);
                if (n0 == c1 && r == c2) {               // slide the window, one new row
                    a0 = a1; a1 = a2; a2 = col[(size_t)n2 * W];
                } else {
                    a0 = col[(size_t)n0 * W]; a1 = col[(size_t)r * W]; a2 = col[(size_t)n2 * W];
                }
                c1 = r; c2 = n2;
            }
            // Sobel: vertical parts in-lane, horizontal parts from the neighbour lanes
            const int t0 = a0 + 2 * a1 + a2, t1 = a2 - a0;
            const int t0m = dpp_get0<0x138, 0xf>(t0), t0p = dpp_get0<0x130, 0xf>(t0);   // lane-1, lane+1
            const int t1m = dpp_get0<0x138, 0xf>(t1), t1p = dpp_get0<0x130, 0xf>(t1);
            const int dx = t0p - t0m, dy = t1m + 2 * t1 + t1p;
            int pxx = __mul24(dx, dx), pxy = __mul24(dx, dy), pyy = __mul24(dy, dy);
            if (xborder) {
                pxx = __builtin_amdgcn_ds_bpermute(psrc, pxx);
                pxy = __builtin_amdgcn_ds_bpermute(psrc, pxy);
                pyy = __builtin_amdgcn_ds_bpermute(psrc, pyy);
            }
            // horizontal window [lane-L, lane+Rr] = S[lane+Rr] - S[lane-L-1]
            const int s0 = wave_incl_scan(pxx), s1 = wave_incl_scan(pxy), s2 = wave_incl_scan(pyy);
            int h0 = __builtin_amdgcn_ds_bpermute(hi_addr, s0), l0 = __builtin_amdgcn_ds_bpermute(lo_addr, s0);
            int h1 = __builtin_amdgcn_ds_bpermute(hi_addr, s1), l1 = __builtin_amdgcn_ds_bpermute(lo_addr, s1);
            int h2 = __builtin_amdgcn_ds_bpermute(hi_addr, s2), l2 = __builtin_amdgcn_ds_bpermute(lo_addr, s2);
            if (lo_zero) { l0 = 0; l1 = 0; l2 = 0; }
            h0 -= l0; h1 -= l1; h2 -= l2;
            // vertical box sum: running sum over the last BLOCK rows
            V0 += h0 - ring[k][0]; V1 += h1 - ring[k][1]; V2 += h2 - ring[k][2];
            ring[k][0] = h0; ring[k][1] = h1; ring[k][2] = h2;
            const int y = m - Rr;                         // output row completed by this step
            if (y >= y0 && out_lane) {
                const float cxx = (float)__dmul_rn((double)V0, scale2);
                const float cxy = (float)__dmul_rn((double)V1, scale2);
                const float cyy = (float)__dmul_rn((double)V2, scale2);
                const float a = __fmul_rn(cxx, 0.5f), b = cxy, cc = __fmul_rn(cyy, 0.5f);
                const float t = __fsub_rn(a, cc);
                const float sq = __fadd_rn(__fmul_rn(t, t), __fmul_rn(b, b));
                const float e = __fsub_rn(__fadd_rn(a, cc), sqrt_rn_normal(sq));
                eig[(size_t)y * W + gx] = e;
                if (pmask[k % CH]) { best = have ? fmaxf(best, e) : e; have = true; }
            }
        }
    }
    unsigned key = have ? eig_key(best) : 0u;
    for (int o = 32; o > 0; o >>= 1) key = max(key, (unsigned)__shfl_xor((int)key, o));
    if (lane == 0) {
        max_partial[wave_id] = key;
    }
}

template <int BLOCK>
static int launch_eig_march(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, double scale2, float *d_eig,
                            unsigned int *d_max_key)
{
    constexpr int VALID = 64 - BLOCK - 1;
    const int nstrips = (W + VALID - 1) / VALID;
    dim3 grid((nstrips + 3) / 4, (H + EIG_RS - 1) / EIG_RS);
    const size_t nwaves = (size_t)grid.x * 4 * grid.y;
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, nwaves * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    eig_march_kernel<BLOCK><<<km_xcd_grid(grid.x * grid.y), 256, 0, c->stream>>>(d_src, d_mask, H, W, scale2, d_eig, partial, nstrips, (int)grid.y);
    KM_LAUNCH_CHECK(c);
    max_u32_kernel<<<1, 1024, 0, c->stream>>>(partial, (unsigned)nwaves, d_max_key);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kd_min_eigen(km_ctx *c, const uint8_t *d_src, const uint8_t *d_mask, int H, int W, int block, float *d_eig,
                 unsigned int *d_max_key)
{
    if (block < 1 || block > 31) return km_fail(c, KM_E_UNSUPPORTED, "blockSize %d (supported 1..31)", block);
    static const int eig_mode = km_dev_env("KARIOS_HIP_EIG_KERNEL") ? atoi(km_dev_env("KARIOS_HIP_EIG_KERNEL")) : 2;   // 2: two pixels per lane, 1: one
    if (eig_mode == 2) {
        const int rc2 = k2_min_eigen(c, d_src, d_mask, H, W, block, d_eig, d_max_key);
        if (rc2 != KM_E_UNSUPPORTED) return rc2;
    }
    const double scale = 1.0 / (4.0 * (double)block * 255.0);
    if (W >= 2 * block + 4 && H >= 2 * block + 4) {   // mirror columns / rows stay inside one strip
        switch (block) {
        case 3: return launch_eig_march<3>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        case 5: return launch_eig_march<5>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        case 7: return launch_eig_march<7>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        case 9: return launch_eig_march<9>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        case 11: return launch_eig_march<11>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        case 15: return launch_eig_march<15>(c, d_src, d_mask, H, W, scale * scale, d_eig, d_max_key);
        default: break;
        }
    }
    // generic LDS-tiled kernel: any block size 1..31, any image size
    const int L = block / 2, Rr = block - 1 - L;
    const int PW = EIG_TW + L + Rr, PH = EIG_TH + L + Rr, LW = (PW + 2 + 3) & ~3, LH = PH + 2;
    const size_t sm = (((size_t)LH * LW + 15) & ~(size_t)15) + (size_t)PH * PW * 4 + (size_t)3 * PH * EIG_TW * 4;
    if (sm > 160 * 1024 - 256) return km_fail(c, KM_E_UNSUPPORTED, "blockSize %d needs %zu B LDS", block, sm);
    static unsigned long long opted = 0;  // dynamic-LDS opt-in, per DEVICE (hipFuncSetAttribute applies to the current one); the kernel also holds a few static words
    if (sm > 48 * 1024 && !(opted & (1ull << (c->device & 63)))) {
        KM_HIP(c, hipFuncSetAttribute((const void *)eig_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        opted |= 1ull << (c->device & 63);
    }
    dim3 grid((W + EIG_TW - 1) / EIG_TW, (H + EIG_TH - 1) / EIG_TH);
    unsigned *partial = (unsigned *)km_ws(c, WS_PARTIAL, (size_t)grid.x * grid.y * sizeof(unsigned));
    if (!partial) return KM_E_NOMEM;
    eig_kernel<<<grid, 256, sm, c->stream>>>(d_src, d_mask, H, W, block, scale * scale, d_eig, partial);
    KM_LAUNCH_CHECK(c);
    max_u32_kernel<<<1, 1024, 0, c->stream>>>(partial, grid.x * grid.y, d_max_key);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// ------------------------------------------------------------------ K4 candidates
// goodFeaturesToTrack steps 4-5: thr = (float)(maxVal*q); TOZERO threshold;
// pixel is a candidate iff it is non-zero, equals the 3x3 max of the thresholded map, lies off
// the 1-px border and passes the mask.  Key = (f32 bits << 32) | raster index, so a single
// descending u64 sort reproduces greaterThanPtr (value desc, address desc).
// One wavefront marches down a 256-column strip (one float4 per lane and row), keeping the thresholded
// rows y-1, y, y+1 in registers; the 3x3 max uses the two neighbour lanes through DPP wave shifts.
// Candidates are compacted into a per-wave LDS stage and flushed with ONE global atomic per flush.
#define CAND_RS 32     // output rows per wave
#define CAND_STAGE 512 // keys per wave stage (a row step adds at most 256)

struct cand_row {
    float v[4];
    float lft, rgt;  // thresholded neighbours x-1 (lane 0 only) and x+4 (lane 63 only) from the adjacent strips
};


__global__ __launch_bounds__(256) void cand_kernel(const float *__restrict__ eig, const uint8_t *__restrict__ mask, int H, int W,
                                                   double quality, km_scalars *sc, unsigned long long *__restrict__ keys,
                                                   size_t cap, int nstrips, int gyw)
{
    __shared__ unsigned long long stage[4][CAND_STAGE];
    const unsigned mk = sc->max_eig_key;
    const float maxv = mk ? eig_unkey(mk) : 0.f;
    const float thr = (float)__dmul_rn((double)maxv, quality);
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc->thr = thr; sc->max_eig = maxv; }
    const int gxw = (nstrips + 3) / 4;                  // logical grid gxw x gyw, XCD-swizzled
    unsigned tile;
    if (!km_xcd_tile((unsigned)(gxw * gyw), tile)) return;
    const int bx = (int)tile % gxw, by = (int)tile / gxw;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = bx * 4 + wv;
    if (strip >= nstrips) return;
    unsigned long long *st = stage[wv];
    const int x0 = strip * 256 + lane * 4;               // first of this lane's 4 columns
    const int y0 = by * CAND_RS, y1 = min(H, y0 + CAND_RS);
    const bool vec = (W % 4 == 0) && x0 + 3 < W;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // raw loads are issued a group of rows ahead (load_raw), thresholding happens when the row is consumed
    auto load_raw = [&](int y, cand_row &r) {
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f; r.lft = 0.f; r.rgt = 0.f;
        if (y < 0 || y >= H) return;                      // outside rows never matter (border rows are excluded)
        const float *row = eig + (size_t)y * W;
        if (vec) {
            const float4 q = *(const float4 *)(row + x0);
            r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) if (x0 + j < W) r.v[j] = row[x0 + j];
        }
        if (lane == 0 && x0 - 1 >= 0) r.lft = row[x0 - 1];
        if (lane == 63 && x0 + 4 < W) r.rgt = row[x0 + 4];
    };
    auto threshold = [&](cand_row &r) {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[j] = r.v[j] > thr ? r.v[j] : 0.f;   // THRESH_TOZERO
        r.lft = r.lft > thr ? r.lft : 0.f;
        r.rgt = r.rgt > thr ? r.rgt : 0.f;
    };

    unsigned cnt = 0;  // keys in the stage (wave-uniform)
    const int wave_id = by * (gxw * 4) + strip;
    const unsigned shard = (unsigned)wave_id % KM_NSHARD;
    const size_t cap_s = cap / KM_NSHARD;
    auto flush = [&]() {
        if (cnt == 0) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&sc->shard_cnt[shard], cnt);
        base = __shfl(base, 0);
        for (unsigned i = lane; i < cnt; i += 64)
            if ((size_t)base + i < cap_s) keys[shard * cap_s + base + i] = st[i];
        cnt = 0;
    };

    constexpr int PF = 4;  // rows in flight
    cand_row up, mid, dn, pre[PF];
    uint32_t pmask[PF];    // mask bytes of the 4 pixels of row yb+k (all-ones without a mask)
    auto load_mask = [&](int y) -> uint32_t {
        if (!mask || y < 0 || y >= H) return 0x01010101u;
        const uint8_t *row = mask + (size_t)y * W;
        if (vec) return *(const uint32_t *)(row + x0);
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (x0 + j < W) m |= (uint32_t)row[x0 + j] << (8 * j);
        return m;
    };
    load_raw(y0 - 1, up); threshold(up);
    load_raw(y0, mid); threshold(mid);
    for (int yb = y0; yb < y1; yb += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++) { load_raw(yb + k + 1, pre[k]); pmask[k] = load_mask(yb + k); }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int y = yb + k;
            if (y >= y1) continue;
            dn = pre[k]; threshold(dn);
            if (y >= 1 && y < H - 1) {
                bool any = false;
#pragma unroll
                for (int j = 0; j < 4; j++) any = any || (mid.v[j] != 0.f);
                if (__ballot(any)) {
                    // column-wise max of the three rows, then the horizontal neighbours
                    float m3[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) m3[j] = fmaxf(fmaxf(up.v[j], mid.v[j]), dn.v[j]);
                    float mL = dpp_shr1(m3[3]), mR = dpp_shl1(m3[0]);
                    if (lane == 0) mL = fmaxf(fmaxf(up.lft, mid.lft), dn.lft);
                    if (lane == 63) mR = fmaxf(fmaxf(up.rgt, mid.rgt), dn.rgt);
                    if (cnt + 256 > CAND_STAGE) flush();
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float left = j == 0 ? mL : m3[j - 1], right = j == 3 ? mR : m3[j + 1];
                        const float nb = fmaxf(fmaxf(left, right), fmaxf(up.v[j], dn.v[j]));
                        const float v = mid.v[j];
                        const int x = x0 + j;
                        const bool is = v != 0.f && v >= nb && x >= 1 && x < W - 1 && ((pmask[k] >> (8 * j)) & 0xffu) != 0;
                        const unsigned long long bal = __ballot(is);
                        if (is) st[cnt + __popcll(bal & lt_mask)] =
                            ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)((unsigned)y * (unsigned)W + (unsigned)x);
                        cnt += (unsigned)__popcll(bal);
                    }
                }
            }
            up = mid; mid = dn;
        }
    }
    flush();
}

int kd_candidates(km_ctx *c, const float *d_eig, const uint8_t *d_mask, int H, int W, double quality, km_scalars *d_sc,
                  unsigned long long *d_keys, size_t cap, bool rezero)
{
    if (rezero) KM_HIP(c, hipMemsetAsync(d_sc->shard_cnt, 0, KM_NSHARD * sizeof(unsigned int), c->stream));
    if (H < 3 || W < 3) {
        return KM_OK;
    }
    const int nstrips = (W + 255) / 256;
    dim3 grid((nstrips + 3) / 4, (H + CAND_RS - 1) / CAND_RS);
    cand_kernel<<<km_xcd_grid(grid.x * grid.y), 256, 0, c->stream>>>(d_eig, d_mask, H, W, quality, d_sc, d_keys, cap, nstrips, (int)grid.y);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// ------------------------------------------------------------------ K6 pyrDown
// cv::pyrDown u8: separable [1 4 6 4 1], (sum + 128) >> 8, REFLECT_101, dst = ((W+1)/2, (H+1)/2).
// Each thread owns 4 adjacent output columns and marches down PYR_RS output rows with a 5-deep register ring
// of horizontal sums (two new source rows per output row, loaded one step ahead as 4 aligned dwords each).
// Both images of a pair are processed by one launch (blockIdx.z).
#ifndef PYR_RS
#define PYR_RS 8    // output rows per thread: short items = more waves in different phases (0.14 ms at 32 rows, 0.098 at 8, 0.18 at 64; 10980^2 pair)
#endif

struct pyr_pair {
    const uint8_t *src[2];
    uint8_t *dst[2];
};

// horizontal [1 4 6 4 1] sums of 4 outputs from 16 source bytes starting at source column 8q-4: output j covers the
// bytes 2j+2 .. 2j+6 - four of them through one v_dot4_u32_u8 with the coefficients (1,4,6,4), the fifth added on top
__device__ __forceinline__ void pyr_hsum(const uint32_t (&w)[4], int (&h)[4])
{
    const unsigned coef = 0x04060401u;                                   // bytes (1, 4, 6, 4)
    const uint32_t q0 = __builtin_amdgcn_alignbyte(w[1], w[0], 2);       // bytes 2..5
    const uint32_t q2 = __builtin_amdgcn_alignbyte(w[2], w[1], 2);       // bytes 6..9
    h[0] = (int)__builtin_amdgcn_udot4(q0, coef, (w[1] >> 16) & 0xffu, false);    // + byte 6
    h[1] = (int)__builtin_amdgcn_udot4(w[1], coef, w[2] & 0xffu, false);          // bytes 4..7 + byte 8
    h[2] = (int)__builtin_amdgcn_udot4(q2, coef, (w[2] >> 16) & 0xffu, false);    // + byte 10
    h[3] = (int)__builtin_amdgcn_udot4(w[2], coef, w[3] & 0xffu, false);          // bytes 8..11 + byte 12
}

// One thread: output columns 4q .. 4q+3 of rows [y0, y1).  FAST (block-uniform): one 16-byte load per source row at whatever alignment
// the row has (a level of odd width - 5490 -> 2745 - puts three rows in four off the dword grid; the hardware reads unaligned just as
// well); otherwise byte by byte with REFLECT_101 columns.
template <bool FAST>
__device__ __forceinline__ void pyrdown_quad(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H, int W, int dw, int q, int y0, int y1)
{
    const int sx0 = 8 * q - 4;                                  // first source byte loaded
    auto load_row = [&](int sy, uint32_t (&w)[4]) {
        const uint8_t *row = src + (size_t)km_reflect101(sy, H) * W;
        if (FAST) {
            __builtin_memcpy(w, row + sx0, 16);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t v = 0;
#pragma unroll 1
                for (int b = 0; b < 4; b++) v |= (uint32_t)row[km_reflect101(sx0 + 4 * k + b, W)] << (8 * b);     // (one byte at a time: these
                w[k] = v;                                                                                              // few lanes must not set the kernel's register count)
            }
        }
    };
    // ring of horizontal sums for source rows 2y-2 .. 2y+2, two outputs per dword (a sum is at most 16 x 255 = 4080, the vertical
    // [1 4 6 4 1] of five of them at most 65 280, + 128 for the rounding: everything stays inside 16 bits - packed arithmetic)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto hsum2 = [&](const uint32_t (&w)[4], u16x2 (&h)[2]) {
        int t[4];
        pyr_hsum(w, t);
        h[0] = __builtin_bit_cast(u16x2, (uint32_t)t[0] | ((uint32_t)t[1] << 16));
        h[1] = __builtin_bit_cast(u16x2, (uint32_t)t[2] | ((uint32_t)t[3] << 16));
    };
    u16x2 h0[2], h1[2], h2[2], h3[2], h4[2];
    uint32_t wa[4], wb[4];
    load_row(2 * y0 - 2, wa); hsum2(wa, h0);
    load_row(2 * y0 - 1, wa); hsum2(wa, h1);
    load_row(2 * y0, wa); hsum2(wa, h2);
    load_row(2 * y0 + 1, wa);
    load_row(2 * y0 + 2, wb);
    auto step = [&](int y) {
        hsum2(wa, h3);
        hsum2(wb, h4);
        if (y + 1 < y1) { load_row(2 * y + 3, wa); load_row(2 * y + 4, wb); }   // next step's rows, in flight during the math
        uint32_t r[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u16x2 s = (h0[j] + h4[j]) + (h1[j] + h3[j]) * (u16x2)(4) + h2[j] * (u16x2)(6) + (u16x2)(128);
            r[j] = __builtin_bit_cast(uint32_t, s >> (u16x2)(8));
        }
        const uint32_t packed = __builtin_amdgcn_perm(r[1], r[0], 0x06040200u);
        const int ox = 4 * q;
        const size_t o = (size_t)y * dw + ox;
        if (FAST || ox + 3 < dw) __builtin_memcpy(dst + o, &packed, 4);  // (one dword store, aligned or not)
        else {
            for (int j = 0; j < 4 && ox + j < dw; j++) dst[o + j] = (uint8_t)(packed >> (8 * j));
        }
#pragma unroll
        for (int j = 0; j < 2; j++) { h0[j] = h2[j]; h1[j] = h3[j]; h2[j] = h4[j]; }
    };
    if constexpr (FAST) {
#pragma unroll
        for (int t = 0; t < PYR_RS; t++) {                       // (unrolled: the ring rotates by renaming)
            if (y0 + t >= y1) break;
            step(y0 + t);
        }
    } else {
#pragma unroll 1
        for (int y = y0; y < y1; y++) step(y);
    }
}

// Work split of one image (blockIdx.x, blockIdx.y): the x-blocks [0, gx_fast) hold the INTERIOR quads 1 .. q_hi - every lane on the
// 16-byte path, no per-lane border code in those waves; the quads that touch the left / right border (quad 0 and the one or two behind
// q_hi) of ALL rows are gathered into the x-block gx_fast, 256 (output row, border quad) items per workgroup.  (With the border
// code behind a per-lane test, the first and the last wave of every row of workgroups ran the byte-by-byte path for one or two live
// lanes - 2 waves in 22 of a 5490-column level, and 40 % of the launch's vector instructions.)
__device__ __forceinline__ void pyrdown_item(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int H, int W, int dh, int dw, int nquads)
{
    const int q_hi = W >= 28 ? (W - 12) / 8 : 0;                // interior quads: 1 <= q <= q_hi  (8q - 4 >= 0, 8q + 12 <= W; 4q + 3 < dw follows)
    const int gx_fast = (q_hi + 255) / 256;
    if ((int)blockIdx.x < gx_fast) {
        const int q = 1 + blockIdx.x * 256 + threadIdx.x;
        const int y0 = blockIdx.y * PYR_RS, y1 = min(dh, y0 + PYR_RS);
        if (q > q_hi || y0 >= dh) return;
        pyrdown_quad<true>(src, dst, H, W, dw, q, y0, y1);
        return;
    }
    if ((int)blockIdx.x > gx_fast) return;
    // (ONE output row per border thread: its byte loads are issued one at a time - a thread marching 8 rows that way was the launch's
    //  tail: 124 us alone where the interior needs 80)
    const int nb = nquads - q_hi;                               // border quads: 0, q_hi + 1 .. nquads - 1
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= nb * dh) return;
    const int y = i / nb, b = i - y * nb;
    const int q = b == 0 ? 0 : q_hi + b;
    pyrdown_quad<false>(src, dst, H, W, dw, q, y, y + 1);
}

__global__ __launch_bounds__(256) void pyrdown_kernel(pyr_pair pp, int H, int W, int dh, int dw, int nquads)
{
    pyrdown_item(pp.src[blockIdx.z], pp.dst[blockIdx.z], H, W, dh, dw, nquads);
}

// batched units: blockIdx.z = 2 * unit + image; the grid covers the largest unit, the others leave their surplus workgroups at once
struct pyr_units_args {
    const uint8_t *src[2 * KM_UNITS_MAX];
    uint8_t *dst[2 * KM_UNITS_MAX];
    int H[KM_UNITS_MAX], W[KM_UNITS_MAX];
};
__global__ __launch_bounds__(256) void pyrdown_units_kernel(pyr_units_args P)
{
    const int u = blockIdx.z >> 1, H = P.H[u], W = P.W[u], dh = (H + 1) / 2, dw = (W + 1) / 2;
    pyrdown_item(P.src[blockIdx.z], P.dst[blockIdx.z], H, W, dh, dw, (dw + 3) / 4);
}

// grid.x of pyrdown_item's work split for a level of width W: the interior x-blocks + the one that gathers the border quads
static inline int pyr_grid_x(int W)
{
    const int q_hi = W >= 28 ? (W - 12) / 8 : 0;
    return (q_hi + 255) / 256 + 1;
}

// level l of both pyramids of every unit from level l - 1 (units whose pyramid ends below l are skipped by the caller: H = 0)
int kd_pyrdown_units(km_ctx *c, const km_units &U, int level)
{
    pyr_units_args P;
    int max_dh = 0, max_q = 0, n = 0;
    for (int u = 0; u < U.n; u++) {
        if (U.A[u].levels < level) continue;
        P.src[2 * n] = U.A[u].img[level - 1]; P.src[2 * n + 1] = U.B[u].img[level - 1];
        P.dst[2 * n] = (uint8_t *)U.A[u].img[level]; P.dst[2 * n + 1] = (uint8_t *)U.B[u].img[level];
        P.H[n] = U.A[u].H[level - 1]; P.W[n] = U.A[u].W[level - 1];
        const int dh = (P.H[n] + 1) / 2;
        max_dh = dh > max_dh ? dh : max_dh;
        max_q = pyr_grid_x(P.W[n]) > max_q ? pyr_grid_x(P.W[n]) : max_q;       // (x-blocks, not quads)
        n++;
    }
    if (n == 0) return KM_OK;
    const dim3 grid(max_q, (max_dh + PYR_RS - 1) / PYR_RS, 2 * n);
    pyrdown_units_kernel<<<grid, 256, 0, c->stream>>>(P);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

static int launch_pyrdown(km_ctx *c, const pyr_pair &pp, int nimg, int H, int W)
{
    const int dh = (H + 1) / 2, dw = (W + 1) / 2;
    const int nquads = (dw + 3) / 4;
    dim3 grid(pyr_grid_x(W), (dh + PYR_RS - 1) / PYR_RS, nimg);
    pyrdown_kernel<<<grid, 256, 0, c->stream>>>(pp, H, W, dh, dw, nquads);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

int kd_pyrdown_u8(km_ctx *c, const uint8_t *d_src, int H, int W, uint8_t *d_dst)
{
    pyr_pair pp;
    pp.src[0] = pp.src[1] = d_src; pp.dst[0] = pp.dst[1] = d_dst;
    return launch_pyrdown(c, pp, 1, H, W);
}

int kd_pyrdown_u8_pair(km_ctx *c, const uint8_t *d_src_a, const uint8_t *d_src_b, int H, int W, uint8_t *d_dst_a, uint8_t *d_dst_b)
{
    pyr_pair pp;
    pp.src[0] = d_src_a; pp.src[1] = d_src_b; pp.dst[0] = d_dst_a; pp.dst[1] = d_dst_b;
    return launch_pyrdown(c, pp, 2, H, W);
}

// ------------------------------------------------------------------ K11 integer shift
// out(y, x) = img(y + y_off, x + x_off), zero outside (reference large_offset.py `_shift_image`): a row is a byte copy at an offset -
// 16 bytes per lane with unaligned loads and stores, whatever the element size (one element per lane with a 64-bit division each
// ran at 2.5 TB/s; sub-dword global accesses pass the address unit a lane at a time).  Chunks that touch an edge go byte by byte.
__global__ __launch_bounds__(256) void shift_rows_kernel(const uint8_t *__restrict__ img, int H, long long rowbytes, long long stride_bytes, int y_off,
                                                         long long xoff_bytes, uint8_t *__restrict__ out)
{
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const long long sy = (long long)y + y_off;
        const bool row_in = sy >= 0 && sy < H;
        const uint8_t *src = img + (size_t)(row_in ? sy : 0) * (size_t)stride_bytes;
        uint8_t *o = out + (size_t)y * (size_t)rowbytes;
        for (long long b = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; b < rowbytes; b += (long long)gridDim.x * 256 * 16) {
            const long long sb = b + xoff_bytes;
            if (row_in && sb >= 0 && sb + 16 <= rowbytes && b + 16 <= rowbytes) {
                uint4 v;
                __builtin_memcpy(&v, src + sb, 16);
                __builtin_memcpy(o + b, &v, 16);
            } else {
                for (int k = 0; k < 16 && b + k < rowbytes; k++) {
                    const long long sx = sb + k;
                    o[b + k] = row_in && sx >= 0 && sx < rowbytes ? src[sx] : (uint8_t)0;
                }
            }
        }
    }
}

int kd_shift_image(km_ctx *c, const void *d_img, int elem_size, int H, int W, ptrdiff_t stride, int y_off, int x_off, void *d_out)
{
    if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8) return km_fail(c, KM_E_ARG, "shift_image: elem_size %d", elem_size);
    if (H <= 0 || W <= 0) return KM_OK;
    const long long rowbytes = (long long)W * elem_size;
    const dim3 grid((unsigned)std::min<long long>((rowbytes + 16 * 256 - 1) / (16 * 256), 64), (unsigned)std::min(H, 65535));
    shift_rows_kernel<<<grid, 256, 0, c->stream>>>((const uint8_t *)d_img, H, rowbytes, (long long)stride * elem_size, y_off, (long long)x_off * elem_size, (uint8_t *)d_out);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}
