// K10 fast path: hand-written float32 phase correlation for image sides whose prime factors lie in {2, 3, 5, 7, 61}
// (Sentinel-2: 10980 = 2^2 * 3^2 * 5 * 61) and fit one workgroup's LDS.  Reference call site: matcher/large_offset.py:39
// (skimage.registration.phase_cross_correlation, algorithm SURVEY App. B).
//
//   z = mon + i * ref                      ONE complex 2-D FFT yields both spectra:
//   Z = FFT2(z)                              F(k) = (Z(k) + conj Z(-k)) / 2,  G(k) = (Z(k) - conj Z(-k)) / 2i
//   P = F conj(G) / max(|F conj(G)|, 100 eps)
//   cc = IFFT2(P);  shift = first arg-max |cc|
//
// 2-D transform = row FFTs, tiled transpose, row FFTs (so every 1-D transform reads and writes contiguous memory).
// 1-D transform: one 256-thread workgroup per row, the row lives in LDS (N <= 12288 complex float32 = 96 KB), Stockham
// autosort stages executed in place: every thread first pulls the inputs of all its butterflies into registers, the workgroup
// synchronises, then the outputs are written.  Radix 61 is a direct DFT that pairs x[j] with x[61-j]: 30 x 30 real-coefficient
// products per half instead of 61 x 61 complex ones (3 600 FMAs per butterfly), fully unrolled so that the coefficient
// indices (j k mod 61) are compile-time constants.  Twiddles come from one table exp(-2 pi i n / N) computed in double on the
// host.  float32 is enough for an integer arg-max only when the peak is unambiguous: the caller checks the margin between
// the two largest values and falls back to the double-precision path (k_phase.hip) when it is thin.
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <string.h>
#include <vector>

#define FFT_NMAX 12288
#define FFT_T 768            // 12 wavefronts: 3 per SIMD (one workgroup per CU: the row fills LDS)

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// DFT of R points in registers, forward sign (exp(-2 pi i j k / R)); `root` = exp(-2 pi i n / R) for n < R
template <int R> __device__ __forceinline__ void dft_small(float2 (&x)[R], const float2 *root)
{
    if constexpr (R == 2) {
        const float2 a = x[0], b = x[1];
        x[0] = cadd(a, b); x[1] = csub(a, b);
    } else if constexpr (R == 4) {
        const float2 a = cadd(x[0], x[2]), b = csub(x[0], x[2]), c = cadd(x[1], x[3]), d = csub(x[1], x[3]);
        const float2 md = make_float2(d.y, -d.x);          // -i * d
        x[0] = cadd(a, c); x[2] = csub(a, c); x[1] = cadd(b, md); x[3] = csub(b, md);
    } else {
        float2 y[R];
#pragma unroll
        for (int k = 0; k < R; k++) {
            float2 s = x[0];
#pragma unroll
            for (int j = 1; j < R; j++) s = cadd(s, cmul(x[j], root[(j * k) % R]));
            y[k] = s;
        }
#pragma unroll
        for (int k = 0; k < R; k++) x[k] = y[k];
    }
}

// radix 61: X[k] = x0 + C_k - i S_k,  X[61-k] = x0 + C_k + i S_k  with  C_k = sum_j (x_j + x_{61-j}) cos(2 pi j k / 61),
// S_k = sum_j (x_j - x_{61-j}) sin(2 pi j k / 61), j, k = 1 .. 30: 3 600 FMAs instead of the 14 884 of a plain 61 x 61 DFT.
// The loop over k stays rolled and the coefficients come from a constant table laid out [k][j] = (cos, sin)(2 pi j k / 61): the
// 30 pairs of one k are 240 contiguous bytes with a wave-uniform address, i.e. four wide scalar loads (a first version indexed a
// 61-entry table with j k mod 61: 900 scattered scalar loads per butterfly, each waited for separately - 225 000 of the 295 000
// cycles a transform took).  The vector registers hold only the 60 sums / differences and four accumulators (fully unrolled
// with the coefficients in VGPRs it spilled 1 400 registers).  Outputs leave through `put(k, value)` as soon as they are
// complete (the caller stores them to LDS).
__constant__ float2 c_t61[30 * 30];

// `part` (0..3, wave-uniform) selects the outputs this call produces: k = 8 part + 1 .. min(8 part + 8, 30) and their mirrors
// 61 - k (+ X[0] from part 0) - four wavefronts share the 61-point transform of the same 64 butterflies, each paying a quarter
// of the 3 600 FMAs (with one thread per butterfly only 180 of a workgroup's threads were busy, for 3 600 FMAs each).
template <typename Put> __device__ __forceinline__ void dft61(const float2 (&x)[61], int part, Put put)
{
    float2 a[31], b[31];
    float2 sum = x[0];
#pragma unroll
    for (int j = 1; j <= 30; j++) { a[j] = cadd(x[j], x[61 - j]); b[j] = csub(x[j], x[61 - j]); sum = cadd(sum, a[j]); }
    const float2 x0 = x[0];
    if (part == 0) put(0, sum);
    const int k_lo = 8 * part + 1, k_hi = min(8 * part + 8, 30);
#pragma unroll 1
    for (int k = k_lo; k <= k_hi; k++) {
        float cr = 0.f, ci = 0.f, sr = 0.f, si = 0.f;
#pragma unroll
        for (int j = 1; j <= 30; j++) {
            const float2 w = c_t61[(k - 1) * 30 + (j - 1)];
            cr = __builtin_fmaf(a[j].x, w.x, cr); ci = __builtin_fmaf(a[j].y, w.x, ci);
            sr = __builtin_fmaf(b[j].x, w.y, sr); si = __builtin_fmaf(b[j].y, w.y, si);
        }
        // -i * (sr + i si) = si - i sr
        put(k, make_float2(x0.x + cr + si, x0.y + ci - sr));
        put(61 - k, make_float2(x0.x + cr - si, x0.y + ci + sr));
    }
}

// one Stockham stage of radix R on the row in LDS; Ns = product of the radices already done.  The loops that fetch are
// branch-free (surplus butterflies re-read the last one, only their stores are predicated): every twiddle gather of the stage
// is in flight before the first one is used - with a branch per butterfly each gather's latency (~1 us under load) was paid
// separately, which made a 10980-point transform take 125 us.
template <int R> __device__ __forceinline__ void stage(float2 *row, int N, int Ns, const float2 *__restrict__ tw, int tid)
{
    constexpr int MAXB = (FFT_NMAX / R + FFT_T - 1) / FFT_T;
    const int nb = N / R;
    const int stride = N / (Ns * R);                 // twiddle exponent step: W_N^(t k stride), t < R, k < Ns
    constexpr bool NEEDS_ROOTS = R != 2 && R != 4;      // radix 2 / 4 butterflies are additions only
    float2 root[NEEDS_ROOTS ? R : 1];
    if constexpr (NEEDS_ROOTS) {
#pragma unroll
        for (int n = 0; n < R; n++) root[n] = tw[n * (N / R)];
    }
    float2 v[MAXB][R], w[MAXB][R];
    const float inv_ns = 1.0f / (float)Ns;
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int j = min(tid + FFT_T * b, nb - 1);
        int q = (int)((float)j * inv_ns);              // j / Ns: the float estimate is off by at most one for j < 2^23
        int k = j - q * Ns;
        if (k < 0) k += Ns; else if (k >= Ns) k -= Ns;
#pragma unroll
        for (int t = 1; t < R; t++) w[b][t] = Ns > 1 ? tw[t * k * stride] : make_float2(1.f, 0.f);
#pragma unroll
        for (int t = 0; t < R; t++) v[b][t] = row[j + t * nb];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < MAXB; b++) {
        const int j = tid + FFT_T * b;
        if (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; t++) v[b][t] = cmul(v[b][t], w[b][t]);
        }
        dft_small<R>(v[b], root);
        if (j < nb) {
            int q = (int)((float)j * inv_ns);
            int k = j - q * Ns;
            if (k < 0) { k += Ns; q--; } else if (k >= Ns) { k -= Ns; q++; }
            const int j0 = q * Ns * R + k;
#pragma unroll
            for (int t = 0; t < R; t++) row[j0 + t * Ns] = v[b][t];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void stage61(float2 *row, int N, int Ns, const float2 *__restrict__ tw, int tid)
{
    const int nb = N / 61, stride = N / (Ns * 61);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int part = wave & 3;                              // which quarter of the outputs (wave-uniform: coefficient loads stay scalar)
    constexpr int PER = (FFT_T / 64 / 4) * 64;              // butterflies per trip: 192 (nb = 180 for 10980)
    if (Ns > 1) {
        // a second radix-61 stage (N a multiple of 3721): its twiddles are applied in place first, so that the butterfly below
        // never holds 60 twiddles next to its 61 inputs (61 is the first radix of every plan: normally Ns = 1 and this is skipped)
        for (int idx = nb + tid; idx < N; idx += FFT_T) {
            const int t = idx / nb, j = idx - t * nb;
            row[idx] = cmul(row[idx], tw[t * (j % Ns) * stride]);
        }
        __syncthreads();
    }
    for (int base = 0; base < nb; base += PER) {
        const int j = base + (wave >> 2) * 64 + lane;
        float2 v[61];
        if (j < nb) {
#pragma unroll
            for (int t = 0; t < 61; t++) v[t] = row[j + t * nb];
        }
        __syncthreads();
        if (j < nb) {
            const int k = j % Ns, j0 = (j / Ns) * Ns * 61 + k;
            dft61(v, part, [&](int t, float2 val) { row[j0 + t * Ns] = val; });   // every thread's inputs are in registers: in-place stores are safe
        }
        __syncthreads();
    }
}

struct fft_plan {
    int n_stages;
    int radix[16];
};

// MODE 0: rows of two real images -> z = a + i b;  1: complex rows in place;  2: complex rows, inverse (conjugate in, conjugate out);
// 3: like 2, but the output is |cc| as float32 (the last pass of the inverse transform)
template <typename T>
__global__ __launch_bounds__(FFT_T) void fft_rows_kernel(const T *__restrict__ img_a, const T *__restrict__ img_b, ptrdiff_t sa, ptrdiff_t sb,
                                                         float2 *__restrict__ data, float *__restrict__ mag, int N, int nrows, fft_plan plan,
                                                         const float2 *__restrict__ tw, int mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *row = (float2 *)smem;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        if (mode == 0) {
            const T *pa = img_a + (size_t)r * sa, *pb = img_b + (size_t)r * sb;
            for (int base = 0; base < N; base += 8 * FFT_T) {
                T xa[8], xb[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = min(base + u * FFT_T + (int)threadIdx.x, N - 1); xa[u] = pa[i]; xb[u] = pb[i]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = base + u * FFT_T + (int)threadIdx.x;
                    if (i < N) row[i] = make_float2((float)xa[u], (float)xb[u]);
                }
            }
        } else {
            // eight independent loads per thread and trip (one trip covers 4096 elements): a load-store loop pays the memory
            // latency once per trip, and a row is only three trips long
            const float2 *src = data + (size_t)r * N;
            for (int base = 0; base < N; base += 8 * FFT_T) {
                float2 x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = src[min(base + u * FFT_T + (int)threadIdx.x, N - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = base + u * FFT_T + (int)threadIdx.x;
                    if (mode >= 2) x[u].y = -x[u].y;
                    if (i < N) row[i] = x[u];
                }
            }
        }
        __syncthreads();
        int Ns = 1;
        for (int s = 0; s < plan.n_stages; s++) {
            const int R = plan.radix[s];
            // the thread index is made opaque per stage: otherwise every stage's row-invariant index arithmetic is hoisted out of
            // the row loop and held in registers through all the other stages (hundreds of values: the kernel spilled)
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
            switch (R) {
            case 61: stage61(row, N, Ns, tw, tid); break;
            case 7: stage<7>(row, N, Ns, tw, tid); break;
            case 5: stage<5>(row, N, Ns, tw, tid); break;
            case 4: stage<4>(row, N, Ns, tw, tid); break;
            case 3: stage<3>(row, N, Ns, tw, tid); break;
            default: stage<2>(row, N, Ns, tw, tid); break;
            }
            Ns *= R;
        }
        if (mode == 3) {
            float *dst = mag + (size_t)r * N;
            for (int i = threadIdx.x; i < N; i += FFT_T) { const float2 x = row[i]; dst[i] = sqrtf(x.x * x.x + x.y * x.y); }
        } else {
            float2 *dst = data + (size_t)r * N;
            for (int i = threadIdx.x; i < N; i += FFT_T) { float2 x = row[i]; if (mode == 2) x.y = -x.y; dst[i] = x; }
        }
        __syncthreads();
    }
}

// out[c][r] = in[r][c], 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, int rows, int cols)
{
    __shared__ float2 tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (by + i < rows && bx + tx < cols) tile[i][tx] = in[(size_t)(by + i) * cols + bx + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (bx + i < cols && by + tx < rows) out[(size_t)(bx + i) * rows + by + tx] = tile[tx][i];
}

// Zt = transposed spectrum of z (W rows kx, H columns ky).  P(k) = F conj(G) normalised, written in the same layout.
__global__ __launch_bounds__(256) void cross_power_f32_kernel(const float2 *__restrict__ Z, float2 *__restrict__ P, int W, int H)
{
    const size_t n = (size_t)W * H;
    const float floor_ = 100.0f * 2.220446049250313e-16f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int kx = (int)(i / H), ky = (int)(i - (size_t)kx * H);
        const int mx = kx ? W - kx : 0, my = ky ? H - ky : 0;
        const float2 z = Z[i], zm = Z[(size_t)mx * H + my];
        // F = (z + conj zm) / 2,  G = (z - conj zm) / (2 i)   (the common factor 1/4 cancels in the normalisation)
        const float2 f = make_float2(z.x + zm.x, z.y - zm.y);
        const float2 d = make_float2(z.x - zm.x, z.y + zm.y);
        const float2 g = make_float2(d.y, -d.x);             // d / i
        const float re = f.x * g.x + f.y * g.y, im = f.y * g.x - f.x * g.y;   // f * conj(g)
        const float mag = fmaxf(hypotf(re, im) * 0.25f, floor_);
        P[i] = make_float2(re * 0.25f / mag, im * 0.25f / mag);
    }
}

__device__ __forceinline__ unsigned long long wmax_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// key = (float bits of |cc| << 32) | ~index: the maximum key is the largest value and, among equals, the smallest index
__global__ __launch_bounds__(256) void argmax_f32_kernel(const float *__restrict__ cc, size_t n, unsigned long long skip, unsigned long long *out)
{
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float a = cc[i];
        if (a == a && i != skip) {
            const unsigned long long k = ((unsigned long long)__float_as_uint(a) << 32) | (0xffffffffull - (unsigned long long)i);
            best = k > best ? k : best;
        }
    }
    best = wmax_u64(best);
    if ((threadIdx.x & 63) == 0 && best) atomicMax(out, best);
}

bool factorize(int N, fft_plan *p)
{
    if (N < 2 || N > FFT_NMAX) return false;
    // radix 61 works in place with every butterfly's inputs in registers: all N / 61 butterflies must fit ONE trip of the
    // workgroup (192 butterflies); longer rows (61 x 193 .. 61 x 201) take the double-precision path
    if (N % 61 == 0 && N / 61 > (FFT_T / 64 / 4) * 64) return false;
    p->n_stages = 0;
    int n = N;
    // the big radix first: its inputs then need no twiddles (Ns = 1)
    static const int radices[] = {61, 7, 5, 4, 3, 2};
    for (int r : radices)
        while (n % r == 0) {
            if (p->n_stages == 16) return false;
            p->radix[p->n_stages++] = r;
            n /= r;
        }
    return n == 1;
}

int upload_twiddles(km_ctx *c, int slot, int N, float2 **out)
{
    static bool t61_ready = false;            // (cos, sin)(2 pi j k / 61), j, k = 1 .. 30, for the radix-61 butterfly
    if (!t61_ready) {
        std::vector<float2> h(900);
        for (int k = 1; k <= 30; k++)
            for (int j = 1; j <= 30; j++) {
                const double ang = 2.0 * M_PI * (double)((j * k) % 61) / 61.0;
                h[(size_t)(k - 1) * 30 + (j - 1)] = make_float2((float)cos(ang), (float)sin(ang));
            }
        KM_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(c_t61), h.data(), 900 * sizeof(float2)));
        t61_ready = true;
    }
    float2 *d = (float2 *)km_ws(c, slot, (size_t)N * sizeof(float2));
    if (!d) return KM_E_NOMEM;
    std::vector<float2> h((size_t)N);
    for (int n = 0; n < N; n++) {
        const double a = -2.0 * M_PI * (double)n / (double)N;
        h[(size_t)n] = make_float2((float)cos(a), (float)sin(a));
    }
    KM_HIP(c, hipMemcpyAsync(d, h.data(), (size_t)N * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    KM_HIP(c, hipStreamSynchronize(c->stream));          // `h` goes out of scope
    *out = d;
    return KM_OK;
}

template <typename T>
int launch_rows(km_ctx *c, const T *a, const T *b, ptrdiff_t sa, ptrdiff_t sb, float2 *data, float *mag, int N, int nrows, const fft_plan &plan,
                const float2 *tw, int mode)
{
    const size_t lds = (size_t)N * sizeof(float2);
    static size_t opted = 0;
    if (lds > 48 * 1024 && lds > opted) {
        KM_HIP(c, hipFuncSetAttribute((const void *)fft_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FFT_NMAX * sizeof(float2))));
        opted = lds;   // per instantiation (static local of the template)
    }
    const int grid = nrows < c->n_cu * 8 ? nrows : c->n_cu * 8;
    fft_rows_kernel<T><<<grid, FFT_T, lds, c->stream>>>(a, b, sa, sb, data, mag, N, nrows, plan, tw, mode);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

}  // namespace

// True when the float32 fast path covers this shape.
bool kp_fast_supported(int H, int W)
{
    fft_plan p;
    return H >= 2 && W >= 2 && factorize(H, &p) && factorize(W, &p) && (size_t)H * W <= 0x7fffffffull;
}

// Phase correlation in float32.  out_rc = integer shift as skimage reports it; *margin = (max - second largest) / max of |cc|:
// the caller trusts the result only when the peak stands clear of every other sample.
int kp_phase_shift_fast(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                        double out_rc[2], double *margin)
{
    fft_plan pw, ph;
    if (!factorize(W, &pw) || !factorize(H, &ph)) return KM_E_UNSUPPORTED;
    const size_t n = (size_t)H * W;
    float2 *A = (float2 *)km_ws(c, WS_FFT_A, n * sizeof(float2)), *B = (float2 *)km_ws(c, WS_FFT_B, n * sizeof(float2));
    km_scalars *sc = (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars));
    if (!A || !B || !sc) return KM_E_NOMEM;
    float2 *tw_w = nullptr, *tw_h = nullptr;
    int rc;
    if ((rc = upload_twiddles(c, WS_MISC0, W, &tw_w))) return rc;
    if (H == W) tw_h = tw_w;
    else if ((rc = upload_twiddles(c, WS_MISC1, H, &tw_h))) return rc;
    // forward: rows (length W) of z = a + i b -> A; transpose -> B (W x H); rows (length H) in place
    switch (dtype) {
#define KM_ROWS(CODE, T) case CODE: rc = launch_rows<T>(c, (const T *)d_a, (const T *)d_b, stride_a, stride_b, A, nullptr, W, H, pw, tw_w, 0); break;
        KM_ROWS(KM_U8, uint8_t) KM_ROWS(KM_U16, uint16_t) KM_ROWS(KM_I16, int16_t) KM_ROWS(KM_F32, float)
#undef KM_ROWS
    default: return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
    }
    if (rc) return rc;
    const dim3 tg((W + 63) / 64, (H + 63) / 64), tg2((H + 63) / 64, (W + 63) / 64);
    transpose_kernel<<<tg, 256, 0, c->stream>>>(A, B, H, W);
    KM_LAUNCH_CHECK(c);
    if ((rc = launch_rows<float>(c, nullptr, nullptr, 0, 0, B, nullptr, H, W, ph, tw_h, 1))) return rc;
    cross_power_f32_kernel<<<c->n_cu * 16, 256, 0, c->stream>>>(B, A, W, H);
    KM_LAUNCH_CHECK(c);
    // inverse: rows (length H) of the transposed spectrum in place; transpose -> B (H x W); rows (length W) -> |cc|
    if ((rc = launch_rows<float>(c, nullptr, nullptr, 0, 0, A, nullptr, H, W, ph, tw_h, 2))) return rc;
    transpose_kernel<<<tg2, 256, 0, c->stream>>>(A, B, W, H);
    KM_LAUNCH_CHECK(c);
    float *cc = (float *)A;
    if ((rc = launch_rows<float>(c, nullptr, nullptr, 0, 0, B, cc, W, H, pw, tw_w, 3))) return rc;
    // largest and second-largest |cc|
    unsigned long long *k1 = &sc->argmax_key, *k2 = (unsigned long long *)&sc->valid;
    KM_HIP(c, hipMemsetAsync(k1, 0, sizeof *k1, c->stream));
    KM_HIP(c, hipMemsetAsync(k2, 0, sizeof *k2, c->stream));
    argmax_f32_kernel<<<c->n_cu * 8, 256, 0, c->stream>>>(cc, n, ~0ull, k1);
    KM_LAUNCH_CHECK(c);
    unsigned long long h1 = 0, h2 = 0;
    KM_HIP(c, hipMemcpyAsync(&h1, k1, sizeof h1, hipMemcpyDeviceToHost, c->stream));
    KM_HIP(c, hipStreamSynchronize(c->stream));
    const unsigned long long flat = h1 ? 0xffffffffull - (h1 & 0xffffffffull) : 0ull;
    argmax_f32_kernel<<<c->n_cu * 8, 256, 0, c->stream>>>(cc, n, flat, k2);
    KM_LAUNCH_CHECK(c);
    KM_HIP(c, hipMemcpyAsync(&h2, k2, sizeof h2, hipMemcpyDeviceToHost, c->stream));
    KM_HIP(c, hipStreamSynchronize(c->stream));
    float v1, v2;
    const unsigned b1 = (unsigned)(h1 >> 32), b2 = (unsigned)(h2 >> 32);
    __builtin_memcpy(&v1, &b1, 4); __builtin_memcpy(&v2, &b2, 4);
    *margin = (h1 && v1 > 0.f) ? (double)(v1 - v2) / (double)v1 : 0.0;
    double r = (double)(flat / (unsigned long long)W), col = (double)(flat % (unsigned long long)W);
    if (r > (double)(H / 2)) r -= H;
    if (col > (double)(W / 2)) col -= W;
    if (H == 1) r = 0;
    if (W == 1) col = 0;
    out_rc[0] = r; out_rc[1] = col;
    return KM_OK;
}
