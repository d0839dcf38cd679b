// K10: large-offset pre-aligner = FFT phase correlation
// (skimage.registration.phase_cross_correlation with 0.24 defaults, reference
// matcher/large_offset.py:39; algorithm SURVEY App. B):
//   F = rfft2(reference_image), G = rfft2(moving_image)  (fp64, as the reference)
//   P = F * conj(G);  P /= max(|P|, 100*eps);  cc = irfft2(P);  (r,c) = first argmax |cc|
//   shift = (r,c), minus N where > fix(N/2).
// The two plain 2-D FFTs are library calls (rocFFT, double precision R2C / C2R -- the image
// side 10980 = 2^2*3^2*5*61 needs a radix-61 stage); conversion, cross-power normalisation and
// the arg-max reduction are hand-written kernels.
#include "common.hpp"

#include <mutex>
#include <rocfft/rocfft.h>

#define KM_FFT(ctx, call)                                                                         \
    do {                                                                                          \
        rocfft_status s_ = (call);                                                                \
        if (s_ != rocfft_status_success)                                                          \
            return km_fail((ctx), KM_E_HIP, "%s:%d %s -> rocfft status %d", __FILE__, __LINE__, #call, (int)s_); \
    } while (0)

template <typename T>
__global__ __launch_bounds__(256) void to_f64_kernel(const T *__restrict__ img, int H, int W, ptrdiff_t stride, double *__restrict__ out)
{
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        out[i] = (double)img[(size_t)y * stride + x];
    }
}

__global__ __launch_bounds__(256) void cross_power_kernel(double2 *__restrict__ F, const double2 *__restrict__ G, size_t n)
{
    const double floor_ = 100.0 * 2.220446049250313e-16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 f = F[i], g = G[i];
        // f * conj(g)
        const double re = f.x * g.x + f.y * g.y, im = f.y * g.x - f.x * g.y;
        const double mag = fmax(hypot(re, im), floor_);
        F[i] = make_double2(re / mag, im / mag);
    }
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}

// pass 1: max |cc| as ordered bits (|v| >= 0 so the raw f64 bit pattern is monotone)
__global__ __launch_bounds__(256) void absmax_kernel(const double *__restrict__ cc, size_t n, unsigned long long *out)
{
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double a = fabs(cc[i]);
        if (a == a) { const unsigned long long b = (unsigned long long)__double_as_longlong(a); best = b > best ? b : best; }
    }
    best = wave_max_u64(best);
    if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}
// pass 2: first (lowest) flat index attaining it = np.argmax tie rule
__global__ __launch_bounds__(256) void first_index_kernel(const double *__restrict__ cc, size_t n, const unsigned long long *maxbits,
                                                          unsigned long long *out)
{
    const unsigned long long mb = *maxbits;
    unsigned long long best = ~0ull;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if ((unsigned long long)__double_as_longlong(fabs(cc[i])) == mb) best = i < best ? i : best;
    }
    best = wave_min_u64(best);
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

static std::once_flag g_fft_once;

static int ensure_plans(km_ctx *c, int H, int W)
{
    std::call_once(g_fft_once, [] { (void)rocfft_setup(); });
    if (c->fft_plan_fwd && c->fft_h == H && c->fft_w == W) return KM_OK;
    kp_destroy(c);
    const size_t lengths[2] = {(size_t)W, (size_t)H};
    rocfft_plan fwd = nullptr, inv = nullptr;
    KM_FFT(c, rocfft_plan_create(&fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, rocfft_precision_double, 2,
                                 lengths, 1, nullptr));
    KM_FFT(c, rocfft_plan_create(&inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, rocfft_precision_double, 2,
                                 lengths, 1, nullptr));
    size_t wf = 0, wi = 0;
    KM_FFT(c, rocfft_plan_get_work_buffer_size(fwd, &wf));
    KM_FFT(c, rocfft_plan_get_work_buffer_size(inv, &wi));
    c->fft_plan_fwd = fwd; c->fft_plan_inv = inv; c->fft_h = H; c->fft_w = W;
    c->fft_work_bytes = wf > wi ? wf : wi;
    return KM_OK;
}

void kp_destroy(km_ctx *c)
{
    if (c->fft_plan_fwd) rocfft_plan_destroy((rocfft_plan)c->fft_plan_fwd);
    if (c->fft_plan_inv) rocfft_plan_destroy((rocfft_plan)c->fft_plan_inv);
    c->fft_plan_fwd = c->fft_plan_inv = nullptr;
    c->fft_h = c->fft_w = 0;
}

static int run_fft(km_ctx *c, void *plan, void *in, void *out, void *work, size_t work_bytes)
{
    rocfft_execution_info info = nullptr;
    KM_FFT(c, rocfft_execution_info_create(&info));
    rocfft_status s = rocfft_execution_info_set_stream(info, c->stream);
    if (s == rocfft_status_success && work_bytes) s = rocfft_execution_info_set_work_buffer(info, work, work_bytes);
    if (s == rocfft_status_success) {
        void *ib[1] = {in}, *ob[1] = {out};
        s = rocfft_execute((rocfft_plan)plan, ib, ob, info);
    }
    rocfft_execution_info_destroy(info);
    if (s != rocfft_status_success) return km_fail(c, KM_E_HIP, "rocfft execute status %d", (int)s);
    return KM_OK;
}

template <typename T>
static void launch_to_f64(km_ctx *c, const void *img, int H, int W, ptrdiff_t stride, double *out)
{
    to_f64_kernel<T><<<4096, 256, 0, c->stream>>>((const T *)img, H, W, stride, out);
}

int kp_phase_shift(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                   double out_rc[2])
{
    if (H < 1 || W < 1) return km_fail(c, KM_E_ARG, "phase_shift: empty image");
    int rc;
    c->phase_path = 2; c->phase_margin = 0.0;
    if (!c->opt_phase_fp64 && kp_fast_supported(H, W)) {
        // float32, hand-written FFT (k_fft.hip).  An integer arg-max needs no more precision than that as long as the peak
        // stands clear: when the two largest samples are within 1 % of each other (a shift of exactly x.5 pixels splits the
        // peak evenly, a flat image has none) the double-precision evaluation below decides, as in the reference.
        double margin = 0.0;
        rc = kp_phase_shift_fast(c, d_a, d_b, dtype, H, W, stride_a, stride_b, out_rc, &margin);
        c->phase_margin = margin;
        if (rc == KM_OK && (margin >= 0.01 || c->opt_fft_dbg)) { c->phase_path = 1; return KM_OK; }   // (fft_dbg: timing experiments, results are wrong anyway)
        if (rc != KM_OK && rc != KM_E_UNSUPPORTED) return rc;
    }
    rc = ensure_plans(c, H, W);
    if (rc) return rc;
    const size_t n = (size_t)H * W, nc = (size_t)H * (W / 2 + 1);
    // WS_FFT_A: real A, later the correlation surface; WS_FFT_B: real B
    double *ra = (double *)km_ws(c, WS_FFT_A, n * sizeof(double));
    double *rb = (double *)km_ws(c, WS_FFT_B, n * sizeof(double));
    double2 *fa = (double2 *)km_ws(c, WS_MISC0, nc * sizeof(double2));
    double2 *fb = (double2 *)km_ws(c, WS_MISC1, nc * sizeof(double2));
    void *work = c->fft_work_bytes ? km_ws(c, WS_FFT_WORK, c->fft_work_bytes) : nullptr;
    km_scalars *sc = (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars));
    if (!ra || !rb || !fa || !fb || !sc || (c->fft_work_bytes && !work)) return KM_E_NOMEM;
    const void *src[2] = {d_a, d_b};
    const ptrdiff_t st[2] = {stride_a, stride_b};
    double *dst[2] = {ra, rb};
    for (int i = 0; i < 2; i++) {
        switch (dtype) {
        case KM_U8: launch_to_f64<uint8_t>(c, src[i], H, W, st[i], dst[i]); break;
        case KM_U16: launch_to_f64<uint16_t>(c, src[i], H, W, st[i], dst[i]); break;
        case KM_I16: launch_to_f64<int16_t>(c, src[i], H, W, st[i], dst[i]); break;
        case KM_F32: launch_to_f64<float>(c, src[i], H, W, st[i], dst[i]); break;
        default: return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
        }
        KM_LAUNCH_CHECK(c);
    }
    if ((rc = run_fft(c, c->fft_plan_fwd, ra, fa, work, c->fft_work_bytes))) return rc;
    if ((rc = run_fft(c, c->fft_plan_fwd, rb, fb, work, c->fft_work_bytes))) return rc;
    cross_power_kernel<<<4096, 256, 0, c->stream>>>(fa, fb, nc);
    KM_LAUNCH_CHECK(c);
    if ((rc = run_fft(c, c->fft_plan_inv, fa, ra, work, c->fft_work_bytes))) return rc;
    unsigned long long *keys = &sc->argmax_key;             // max bits
    unsigned long long *idx = (unsigned long long *)&sc->valid;  // reused as first-index slot
    KM_HIP(c, hipMemsetAsync(keys, 0, sizeof(unsigned long long), c->stream));
    KM_HIP(c, hipMemsetAsync(idx, 0xff, sizeof(unsigned long long), c->stream));
    absmax_kernel<<<2048, 256, 0, c->stream>>>(ra, n, keys);
    KM_LAUNCH_CHECK(c);
    first_index_kernel<<<2048, 256, 0, c->stream>>>(ra, n, keys, idx);
    KM_LAUNCH_CHECK(c);
    unsigned long long flat = 0;
    { int rq = km_d2h_queue(c, &flat, idx, sizeof(flat)); if (!rq) rq = km_d2h_flush(c); if (rq) return rq; }
    if (flat == ~0ull) flat = 0;  // all-NaN surface: np.argmax would return the first NaN; 0 by convention
    double r = (double)(flat / (unsigned long long)W), col = (double)(flat % (unsigned long long)W);
    // np.fix(N/2) thresholds; axes of length 1 -> 0
    if (r > (double)(H / 2)) r -= H;
    if (col > (double)(W / 2)) col -= W;
    if (H == 1) r = 0;
    if (W == 1) col = 0;
    out_rc[0] = r; out_rc[1] = col;
    return KM_OK;
}
