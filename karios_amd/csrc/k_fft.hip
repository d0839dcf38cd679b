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
#include <type_traits>
#include <utility>
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
    } else if constexpr (R == 3) {
        // W = exp(-2 pi i / 3) = -1/2 - i sqrt(3)/2:  X1,2 = x0 - (x1 + x2) / 2 -+ i (sqrt(3)/2) (x1 - x2)
        const float c = 0.86602540378443864676f;
        const float2 t = cadd(x[1], x[2]), d = csub(x[1], x[2]);
        const float2 m = make_float2(x[0].x - 0.5f * t.x, x[0].y - 0.5f * t.y), r = make_float2(c * d.y, -c * d.x);   // r = -i c d
        x[0] = cadd(x[0], t); x[1] = cadd(m, r); x[2] = csub(m, r);
    } else if constexpr (R == 5) {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f, s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const float2 a1 = cadd(x[1], x[4]), a2 = cadd(x[2], x[3]), b1 = csub(x[1], x[4]), b2 = csub(x[2], x[3]);
        const float2 m1 = make_float2(x[0].x + c1 * a1.x + c2 * a2.x, x[0].y + c1 * a1.y + c2 * a2.y);
        const float2 m2 = make_float2(x[0].x + c2 * a1.x + c1 * a2.x, x[0].y + c2 * a1.y + c1 * a2.y);
        const float2 n1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y), n2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
        // X1,4 = m1 -+ i n1, X2,3 = m2 -+ i n2   (-i n = (n.y, -n.x))
        x[0] = make_float2(x[0].x + a1.x + a2.x, x[0].y + a1.y + a2.y);
        x[1] = make_float2(m1.x + n1.y, m1.y - n1.x); x[4] = make_float2(m1.x - n1.y, m1.y + n1.x);
        x[2] = make_float2(m2.x + n2.y, m2.y - n2.x); x[3] = make_float2(m2.x - n2.y, m2.y + n2.x);
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
    constexpr bool NEEDS_ROOTS = R == 7;                 // radix 2 .. 5 butterflies carry their constants
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
            constexpr int PPD = std::is_same<T, float>::value ? 1 : 4 / (int)sizeof(T);
            if (PPD > 1 && N % PPD == 0 && sa % PPD == 0 && sb % PPD == 0 && (((unsigned long long)img_a | (unsigned long long)img_b) & 3ull) == 0) {
                // (whole dwords: see fft61_rows_kernel)
                for (int base = 0; base < N / PPD; base += 8 * FFT_T) {
                    uint32_t da[8], db[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int j = min(base + u * FFT_T + (int)threadIdx.x, N / PPD - 1);
                        da[u] = ((const uint32_t *)pa)[j]; db[u] = ((const uint32_t *)pb)[j];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int j = base + u * FFT_T + (int)threadIdx.x;
                        if (j < N / PPD) {
#pragma unroll
                            for (int k = 0; k < PPD; k++)
                                row[j * PPD + k] = make_float2((float)(T)(da[u] >> (8 * (int)sizeof(T) * k)), (float)(T)(db[u] >> (8 * (int)sizeof(T) * k)));
                        }
                    }
                }
            } else
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

// P(k) = F conj(G) / max(|F conj(G)|, floor) from z = Z(k) and zm = Z(-k) of the packed transform Z = FFT2(a + i b):
// F = (z + conj zm) / 2,  G = (z - conj zm) / (2 i)   (the common factor 1/4 cancels in the normalisation)
__device__ __forceinline__ float2 cross_power_of(float2 z, float2 zm)
{
    const float floor_ = 100.0f * 2.220446049250313e-16f;
    const float2 f = make_float2(z.x + zm.x, z.y - zm.y);
    const float2 d = make_float2(z.x - zm.x, z.y + zm.y);
    const float2 g = make_float2(d.y, -d.x);             // d / i
    const float re = f.x * g.x + f.y * g.y, im = f.y * g.x - f.x * g.y;   // f * conj(g)
    const float mag = fmaxf(hypotf(re, im) * 0.25f, floor_);
    return make_float2(re * 0.25f / mag, im * 0.25f / mag);
}

// Zt = transposed spectrum of z (W rows kx, H columns ky).  P(k) written in the same layout.
__global__ __launch_bounds__(256) void cross_power_f32_kernel(const float2 *__restrict__ Z, float2 *__restrict__ P, int W, int H)
{
    const size_t n = (size_t)W * H;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int kx = (int)(i / H), ky = (int)(i - (size_t)kx * H);
        const int mx = kx ? W - kx : 0, my = ky ? H - ky : 0;
        P[i] = cross_power_of(Z[i], Z[(size_t)mx * H + my]);
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

// ================================================================================================================
// Rows of length N = 61 * M (M <= 192 with prime factors in {2, 3, 5, 7}; Sentinel-2: 10980 = 61 * 180, 5490 = 61 * 90,
// 1830 = 61 * 30), second form.
//
// What the Stockham kernel above pays per row besides its arithmetic (44 us per 10980-point row, one 88-KB workgroup per CU):
// a dozen workgroup barriers with a global twiddle gather in front of most of them, a radix-61 butterfly whose 3 600
// coefficients arrive through scalar loads that are waited for at every k, and a row load that nothing overlaps.  Here
//   * the 61-point transforms (stride M) run first, with the coefficients as INSTRUCTION LITERALS: v_fmamk_f32 d, a, K, d -
//     no loads, no waits, four independent accumulator chains per output pair (fft61_coef.inc, tools/gen_fft61.py);
//   * what remains are 61 INDEPENDENT transforms of length M (X[t + 61 m] = sum_j Y[j][t] W_N^(j t) W_M^(j m)): each is done
//     by ONE wavefront on its own 61-strided elements of the row - wave-synchronous Stockham stages, no workgroup barrier,
//     the length-M twiddles from a 1.5-KB LDS table, the inter-step twiddles W_N^(j t) from a table laid out [t][j] so that
//     the lanes of a wavefront read consecutive words; the twelve wavefronts drift apart and hide each other's latencies;
//   * the next row travels from HBM into registers while the wavefronts work on the current one;
//   * the last inverse pass never writes |cc|: every row reports its largest and second-largest sample (fft61 mode 3), a
//     one-workgroup kernel finds the arg-max and the margin - two passes over a 482-MB plane and its store are gone.
#include "fft61_coef.inc"

template <class F, int... I> __device__ __forceinline__ void f61_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void e3_style_for(F &&f) { f61_for_impl(f, std::make_integer_sequence<int, N>{}); }   // f(0_c) .. f((N-1)_c)

#define F61_T 768
#define F61_MMAX 192

template <int K, int J> __device__ __forceinline__ void f61_mac(float &cr, float &ci, float &sr, float &si, const float2 &a, const float2 &b)
{
    asm("v_fmamk_f32 %0, %1, %2, %0" : "+v"(cr) : "v"(a.x), "n"(F61_C[K][J]));
    asm("v_fmamk_f32 %0, %1, %2, %0" : "+v"(ci) : "v"(a.y), "n"(F61_C[K][J]));
    asm("v_fmamk_f32 %0, %1, %2, %0" : "+v"(sr) : "v"(b.x), "n"(F61_S[K][J]));
    asm("v_fmamk_f32 %0, %1, %2, %0" : "+v"(si) : "v"(b.y), "n"(F61_S[K][J]));
}

// accumulators of one wavefront's share of the 61-point transform: up to 8 consecutive k (k = K0 + 1 .. K0 + NK)
struct f61_acc {
    float cr[8], ci[8], sr[8], si[8];
};

template <int K0, int J, int... I>
__device__ __forceinline__ void f61_step(f61_acc &s, const float2 &a, const float2 &b, std::integer_sequence<int, I...>)
{
    (f61_mac<K0 + I, J>(s.cr[I], s.ci[I], s.sr[I], s.si[I], a, b), ...);
}

// One wavefront's share of the 61-point transforms of 64 butterflies (lane = butterfly j): outputs k = K0 + 1 .. K0 + NK and their
// mirrors 61 - k (+ X[0] when K0 == 0).  The inputs stream from LDS in j order - x[j] and x[61 - j] are read, folded into their sum
// and difference, and fed to the NK x 4 accumulator chains at once (32 independent FMAs per j: no chain ever waits) - so a lane
// holds 4 NK accumulators and a few inputs instead of all 61 inputs: ~90 registers instead of ~250 (which spilled, next-row
// prefetch included).  Nothing is written here: the outputs overwrite other butterflies' inputs, so they wait in the accumulators
// for the workgroup barrier (`f61_emit`).
template <int K0, int NK>
__device__ __forceinline__ void f61_accumulate(const float2 *__restrict__ in /* row + j */, int M, f61_acc &s, float2 &x0, float2 &sum)
{
#pragma unroll
    for (int i = 0; i < NK; i++) s.cr[i] = s.ci[i] = s.sr[i] = s.si[i] = 0.f;
    x0 = in[0];
    sum = x0;
    e3_style_for<30>([&](auto jc) {
        constexpr int J = decltype(jc)::value;              // coefficient column J <-> input pair (J + 1, 60 - J)
        const float2 p = in[(J + 1) * M], q = in[(60 - J) * M];
        const float2 a = cadd(p, q), b = csub(p, q);
        sum = cadd(sum, a);
        f61_step<K0, J>(s, a, b, std::make_integer_sequence<int, NK>{});
    });
}

template <int K0, int NK, typename Put> __device__ __forceinline__ void f61_emit(const f61_acc &s, float2 x0, float2 sum, Put put)
{
    if (K0 == 0) put(0, sum);
#pragma unroll
    for (int i = 0; i < NK; i++) {
        // X[k] = x0 + C_k - i S_k,  X[61 - k] = x0 + C_k + i S_k   (k = K0 + i + 1;  -i (sr + i si) = si - i sr)
        put(K0 + i + 1, make_float2(x0.x + s.cr[i] + s.si[i], x0.y + s.ci[i] - s.sr[i]));
        put(60 - K0 - i, make_float2(x0.x + s.cr[i] - s.si[i], x0.y + s.ci[i] + s.sr[i]));
    }
}

// every wavefront works on its own elements of the row: only the compiler must be kept from moving LDS traffic across the phases
#define F61_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// One Stockham stage (radix R, Ns = product of the radices done) of the length-M transform whose element i lives at row[i * 61 + tp];
// one wavefront, in place: every input is in registers before the first output is written.  `big` (first stage only): the
// inter-step twiddles W_N^(i tp) of this sub-transform, consecutive in i.
template <int R, int MC = 0, int NSC = 0 /* M and Ns at compile time (0: run time): every index of the stage folds into constants */>
__device__ __forceinline__ void f61_wave_stage(float2 *row, int tp, int M_rt, int Ns_rt, const float2 *__restrict__ twM, const float2 *__restrict__ big, int lane)
{
    constexpr int TRIPS = R == 2 ? 2 : 1;                 // M / R butterflies, <= 64 per trip (M <= 192)
    const int M = MC ? MC : M_rt, Ns = NSC ? NSC : Ns_rt;
    const int nb = M / R, step = M / (Ns * R);
    const unsigned rcp_ns = (1u << 20) / (unsigned)Ns + 1u;      // b / Ns == (b * rcp_ns) >> 20 for b < 256, Ns <= 192 (checked exhaustively)
    constexpr bool NEEDS_ROOTS = R == 7;
    float2 root[NEEDS_ROOTS ? R : 1];
    if constexpr (NEEDS_ROOTS) {
#pragma unroll
        for (int n = 0; n < R; n++) root[n] = twM[n * (M / R)];
    }
    float2 v[TRIPS][R];
#pragma unroll
    for (int u = 0; u < TRIPS; u++) {
        const int b = min(lane + 64 * u, nb - 1);
        const int k = b - (int)(((unsigned)b * rcp_ns) >> 20) * Ns;
#pragma unroll
        for (int t = 0; t < R; t++) v[u][t] = row[(b + t * nb) * 61 + tp];
        if (big) {
#pragma unroll
            for (int t = 0; t < R; t++) v[u][t] = cmul(v[u][t], big[b + t * nb]);
        }
        if (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; t++) v[u][t] = cmul(v[u][t], twM[t * k * step]);
        }
    }
    F61_WAVE_SYNC();
#pragma unroll
    for (int u = 0; u < TRIPS; u++) {
        const int b = lane + 64 * u;
        dft_small<R>(v[u], root);
        if (b < nb) {
            const int q = (int)(((unsigned)b * rcp_ns) >> 20), k = b - q * Ns, j0 = q * Ns * R + k;
#pragma unroll
            for (int t = 0; t < R; t++) row[(j0 + t * Ns) * 61 + tp] = v[u][t];
        }
    }
    F61_WAVE_SYNC();
}

// the stages of a length-MC transform with everything a constant: radices in `factorize61`'s order (7s, 5s, 4s, 3s, 2s)
template <int MC, int NS>
__device__ __forceinline__ void f61_stages_c(float2 *row, int tp, const float2 *__restrict__ twM, const float2 *__restrict__ big, int lane)
{
    if constexpr (NS < MC) {
        constexpr int REM = MC / NS;
        constexpr int R = REM % 7 == 0 ? 7 : REM % 5 == 0 ? 5 : REM % 4 == 0 ? 4 : REM % 3 == 0 ? 3 : 2;
        static_assert(REM % R == 0, "M must factor into 2, 3, 5, 7");
        int ln = lane;
        asm volatile("" : "+v"(ln));                    // (keeps each stage's index arithmetic local)
        f61_wave_stage<R, MC, NS>(row, tp, MC, NS, twM, NS == 1 ? big : nullptr, ln);
        f61_stages_c<MC, NS * R>(row, tp, twM, big, lane);
    }
}

struct f61_plan {
    int M, n_stages;
    int radix[8];
};

// per-row result of mode 3: key of the largest |cc| (value bits << 32 | ~flat index) and the second-largest value's bits
struct f61_top2 {
    unsigned long long best;
    unsigned second, pad;
};

// MODE 0: rows of two real images -> z = a + i b;  1: complex rows in place;  2: complex rows, inverse (conjugate in, conjugate out);
// 3: like 2 without an output plane: top2[r] = largest / second-largest |cc| of row r;  4: like 2 with the cross-power step fused
// into the row load: row r of `data` (= Z, transposed spectrum) and its mirror row are read, P(k) is formed on the way into LDS,
// the inverse transform of the row goes to `out_plane` (a kernel of its own moved 2.9 GB for that);  5: the LAST inverse pass on a
// Hermitian half plane (the cross-power spectrum of two real images: P(-k) = conj P(k), so the first inverse pass only ran the rows
// kx <= N / 2): a workgroup step takes TWO image rows y1 = 2 r, y2 = 2 r + 1 of the half plane (pitch N / 2 + 1), completes
// S(kx) = Q_y1(kx) + i Q_y2(kx) by symmetry, and ONE complex transform returns cc_y1 in the real and cc_y2 in the imaginary part -
// half the rows, half the bytes; both rows report their own (largest, second-largest) |cc| like mode 3
template <typename T, int MC /* M at compile time (0: run time): the 61 strided reads of a butterfly then carry immediate offsets */, int mode>
__global__ __launch_bounds__(F61_T) void fft61_rows_kernel(const T *__restrict__ img_a, const T *__restrict__ img_b, ptrdiff_t sa, ptrdiff_t sb,
                                                           float2 *__restrict__ data, f61_top2 *__restrict__ top2, int N, int nrows, f61_plan plan,
                                                           const float2 *__restrict__ twM_g, const float2 *__restrict__ big_g,
                                                           float2 *__restrict__ out_plane, int run_rows /* 0: all `nrows`; mode 4: only rows [0, run_rows); mode 5: row PAIRS */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *row = (float2 *)smem;
    float2 *twM = row + N;                                  // exp(-2 pi i n / M), n < M
    constexpr bool CROSS = mode == 4, PAIR = mode == 5;
    const int nrun = run_rows ? run_rows : nrows;
    __shared__ unsigned long long s_best[F61_T / 64];
    __shared__ unsigned s_second[F61_T / 64];
    const int M = MC ? MC : plan.M;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PF = (FFT_NMAX + F61_T - 1) / F61_T;      // elements of a row per thread (16)
    for (int i = tid; i < M; i += F61_T) twM[i] = twM_g[i];

    // the row a thread holds in flight: elements tid + F61_T * u, exactly as loaded (conversions and the conjugation wait for
    // `commit`: an instruction that consumes a load would stall the wavefront until the data has arrived)
    float2 pf[PF], pg[(CROSS || PAIR) ? PF : 1];
    constexpr int PFH = PF / 2 + 1;                         // mode 5 loads half rows (N / 2 + 1 elements)
    constexpr int PPD = std::is_same<T, float>::value ? 1 : 4 / (int)sizeof(T), PFD = (PF + PPD - 1) / PPD;   // mode 0: pixels per dword
    const bool px_dwords = mode == 0 && PPD > 1 && N % PPD == 0 && sa % PPD == 0 && sb % PPD == 0 &&
                           (((unsigned long long)img_a | (unsigned long long)img_b) & 3ull) == 0;
    auto fetch = [&](int r) {
        if constexpr (PAIR) {
            const int Wh = N / 2 + 1, y2 = min(2 * r + 1, nrows - 1);
            const float2 *s1 = data + (size_t)(2 * r) * Wh, *s2 = data + (size_t)y2 * Wh;
#pragma unroll
            for (int u = 0; u < PFH; u++) {
                const int i = min(tid + F61_T * u, Wh - 1);
                pf[u] = s1[i];
                pg[u] = s2[i];
            }
        } else if constexpr (CROSS) {
            // Z(kx, ky) and Z(-kx, -ky): row r and its mirror row read backwards (element 0 pairs with element 0)
            const int mr = r ? nrows - r : 0;
            const float2 *src = data + (size_t)r * N, *msrc = data + (size_t)mr * N;
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int i = min(tid + F61_T * u, N - 1);
                pf[u] = src[i];
                pg[u] = msrc[i ? N - i : 0];
            }
        } else if (mode == 0) {
            const T *pa = img_a + (size_t)r * sa, *pb = img_b + (size_t)r * sb;
            if (px_dwords) {
                // 8- / 16-bit pixels as whole dwords, 4 / 2 pixels per lane (sub-dword global loads pass the address unit one lane at a
                // time: the 2-byte form of this loop cost ~0.35 of the pass's 0.79 ms)
#pragma unroll
                for (int u = 0; u < PFD; u++) {
                    const int j = tid + F61_T * u;
                    if (j < N / PPD) {
                        pf[u].x = __uint_as_float(((const uint32_t *)pa)[j]);
                        pf[u].y = __uint_as_float(((const uint32_t *)pb)[j]);
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int i = min(tid + F61_T * u, N - 1);
                if constexpr (std::is_same<T, float>::value) { pf[u].x = pa[i]; pf[u].y = pb[i]; }
                else { pf[u].x = __uint_as_float((unsigned)(int)pa[i]); pf[u].y = __uint_as_float((unsigned)(int)pb[i]); }   // (the load itself extends)
            }
            }
        } else {
            const float2 *src = data + (size_t)r * N;
#pragma unroll
            for (int u = 0; u < PF; u++) pf[u] = src[min(tid + F61_T * u, N - 1)];
        }
    };
    auto commit = [&](int r) {
        if constexpr (PAIR) {
            const int Wh = N / 2 + 1;
            const bool has2 = 2 * r + 1 < nrows;
#pragma unroll
            for (int u = 0; u < PFH; u++) {
                const int i = tid + F61_T * u;
                if (i < Wh) {
                    const float2 q1 = pf[u], q2 = has2 ? pg[u] : make_float2(0.f, 0.f);
                    // S(i) = q1 + i q2, S(N - i) = conj q1 + i conj q2; stored conjugated (inverse = conj . forward . conj)
                    if (i == 0 || 2 * i == N) row[i] = make_float2(q1.x, -q2.x);            // self-conjugate frequencies are real
                    else {
                        row[i] = make_float2(q1.x - q2.y, -(q1.y + q2.x));
                        row[N - i] = make_float2(q1.x + q2.y, q1.y - q2.x);
                    }
                }
            }
            return;
        }
        if (mode == 0 && px_dwords) {
            if constexpr (PPD > 1) {
#pragma unroll
                for (int u = 0; u < PFD; u++) {
                    const int j = tid + F61_T * u;
                    if (j < N / PPD) {
                        const uint32_t da = __float_as_uint(pf[u].x), db = __float_as_uint(pf[u].y);
#pragma unroll
                        for (int k = 0; k < PPD; k++)
                            row[j * PPD + k] = make_float2((float)(T)(da >> (8 * (int)sizeof(T) * k)), (float)(T)(db >> (8 * (int)sizeof(T) * k)));
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int i = tid + F61_T * u;
            float2 x = pf[u];
            if (mode == 0) {
                if constexpr (!std::is_same<T, float>::value)
                    x = make_float2((float)(T)(int)__float_as_uint(pf[u].x), (float)(T)(int)__float_as_uint(pf[u].y));
            } else if constexpr (CROSS) {
                x = cross_power_of(pf[u], pg[u]);
                x.y = -x.y;
            } else if (mode >= 2) {
                x.y = -x.y;
            }
            if (i < N) row[i] = x;
        }
    };
    int r = blockIdx.x;
    if (r < nrun) { fetch(r); commit(r); }
    __syncthreads();
    for (; r < nrun; r += gridDim.x) {
        // ---- phase A: 61-point transforms of the M butterflies (inputs at stride M, outputs contiguous: Y[j][t] -> row[61 j + t])
        {
            const int part = wave & 3, j = (wave >> 2) * 64 + lane;
            f61_acc acc;                                    // (part 3 uses 6 of the 8)
            float2 x0 = make_float2(0.f, 0.f), sum = x0;
            if (j < M) {
                switch (part) {
                case 0: f61_accumulate<0, 8>(row + j, M, acc, x0, sum); break;
                case 1: f61_accumulate<8, 8>(row + j, M, acc, x0, sum); break;
                case 2: f61_accumulate<16, 8>(row + j, M, acc, x0, sum); break;
                default: f61_accumulate<24, 6>(row + j, M, acc, x0, sum); break;
                }
            }
            __syncthreads();
            if (j < M) {
                auto put = [&](int t, float2 val) { row[j * 61 + t] = val; };
                switch (part) {
                case 0: f61_emit<0, 8>(acc, x0, sum, put); break;
                case 1: f61_emit<8, 8>(acc, x0, sum, put); break;
                case 2: f61_emit<16, 8>(acc, x0, sum, put); break;
                default: f61_emit<24, 6>(acc, x0, sum, put); break;
                }
            }
        }
        __syncthreads();
        // ---- the next row starts its way from HBM (consumed after the store below)
        const int rn = r + gridDim.x;
        if (!CROSS && !PAIR && rn < nrun) fetch(rn);     // (mode 4 holds two rows: fetched behind the transforms, where the registers are free)
        // ---- phase B: 61 independent length-M transforms, one wavefront each
        for (int tp = wave; tp < 61; tp += F61_T / 64) {
            if constexpr (MC != 0) {
                // radix, Ns and M as constants: the stage sequence `factorize61` produces, unrolled at compile time - with run-time
                // sizes the index arithmetic of a stage (quotients, strides, twiddle steps) cost more than its butterflies
                f61_stages_c<MC, 1>(row, tp, twM, big_g + (size_t)tp * M, lane);
                continue;
            }
            int Ns = 1;
            for (int s = 0; s < plan.n_stages; s++) {
                const int R = plan.radix[s];
                const float2 *big = s == 0 ? big_g + (size_t)tp * M : nullptr;
                int ln = lane;
                asm volatile("" : "+v"(ln));                // (keeps each stage's index arithmetic local)
                switch (R) {
                case 7: f61_wave_stage<7>(row, tp, M, Ns, twM, big, ln); break;
                case 5: f61_wave_stage<5>(row, tp, M, Ns, twM, big, ln); break;
                case 4: f61_wave_stage<4>(row, tp, M, Ns, twM, big, ln); break;
                case 3: f61_wave_stage<3>(row, tp, M, Ns, twM, big, ln); break;
                default: f61_wave_stage<2>(row, tp, M, Ns, twM, big, ln); break;
                }
                Ns *= R;
            }
        }
        __syncthreads();
        if ((CROSS || PAIR) && rn < nrun) fetch(rn);
        // ---- the finished row leaves (natural order: X[t + 61 m] at row[t + 61 m])
        if (mode == 3 || PAIR) {
          for (int half = 0; half < (PAIR ? 2 : 1); half++) {
            const int ro = PAIR ? 2 * r + half : r;         // the image row these samples belong to
            if (ro >= nrows) break;                          // (odd number of rows: the last pair has no second one; uniform)
            if (half) __syncthreads();                       // (s_best / s_second are read by thread 0 below)
            unsigned long long best = 0;
            unsigned second = 0;
            for (int i = tid; i < N; i += F61_T) {
                const float2 x = row[i];
                const float a = PAIR ? fabsf(half ? x.y : x.x) : sqrtf(x.x * x.x + x.y * x.y);
                if (a == a) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(a) << 32) | (0xffffffffull - ((unsigned long long)ro * (unsigned)N + (unsigned)i));
                    if (key > best) { second = max(second, (unsigned)(best >> 32)); best = key; }
                    else second = max(second, __float_as_uint(a));
                }
            }
            // wave: largest key; second = largest value among everything that is not that sample
            unsigned long long wb = wmax_u64(best);
            unsigned cand = best == wb ? second : (unsigned)(best >> 32);
            cand = max(cand, second);
            for (int o = 32; o > 0; o >>= 1) cand = max(cand, (unsigned)__shfl_xor((int)cand, o));
            if (lane == 0) { s_best[wave] = wb; s_second[wave] = cand; }
            __syncthreads();
            if (tid == 0) {
                unsigned long long rb = 0;
                for (int w = 0; w < F61_T / 64; w++) rb = s_best[w] > rb ? s_best[w] : rb;
                unsigned rs = 0;
                for (int w = 0; w < F61_T / 64; w++) rs = max(rs, s_best[w] == rb ? s_second[w] : max(s_second[w], (unsigned)(s_best[w] >> 32)));
                top2[ro].best = rb; top2[ro].second = rs; top2[ro].pad = 0;
            }
          }
        } else {
            float2 *dst = (CROSS ? out_plane : data) + (size_t)r * N;
            for (int i = tid; i < N; i += F61_T) { float2 x = row[i]; if (mode == 2 || CROSS) x.y = -x.y; dst[i] = x; }
        }
        __syncthreads();
        if (rn < nrun) commit(rn);
        __syncthreads();
    }
}

// rows' (largest key, second-largest value) -> overall: out[0] = key of the arg-max, out[1] = bits of the second-largest sample
__global__ __launch_bounds__(1024) void f61_top2_reduce_kernel(const f61_top2 *__restrict__ top2, int nrows, unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long s_b[16];
    __shared__ unsigned s_s[16];
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < nrows; i += 1024) best = top2[i].best > best ? top2[i].best : best;
    unsigned long long wb = wmax_u64(best);
    if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = wb;
    __syncthreads();
    unsigned long long gb = 0;
    for (int w = 0; w < 16; w++) gb = s_b[w] > gb ? s_b[w] : gb;
    unsigned sec = 0;
    for (int i = threadIdx.x; i < nrows; i += 1024) sec = max(sec, top2[i].best == gb ? top2[i].second : max(top2[i].second, (unsigned)(top2[i].best >> 32)));
    for (int o = 32; o > 0; o >>= 1) sec = max(sec, (unsigned)__shfl_xor((int)sec, o));
    if ((threadIdx.x & 63) == 0) s_s[threadIdx.x >> 6] = sec;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned gs = 0;
        for (int w = 0; w < 16; w++) gs = max(gs, s_s[w]);
        out[0] = gb; out[1] = (unsigned long long)gs;
    }
}

bool factorize61(int N, f61_plan *p)
{
    if (N < 122 || N > FFT_NMAX || N % 61) return false;
    int m = N / 61;
    if (m > F61_MMAX || m % 61 == 0) return false;
    p->M = m; p->n_stages = 0;
    static const int radices[] = {7, 5, 4, 3, 2};
    for (int r : radices)
        while (m % r == 0) {
            if (p->n_stages == 8) return false;
            p->radix[p->n_stages++] = r;
            m /= r;
        }
    return m == 1 && p->n_stages > 0;
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

// Twiddle tables of one row length, kept in a workspace slot of their own between calls (a Sentinel-2 product is a stack of bands
// of the same size): tw[n] = exp(-2 pi i n / N), n < N; for N = 61 M also twM[n] = exp(-2 pi i n / M), n < M, and
// big[t][j] = exp(-2 pi i j t / N), t < 61, j < M (the inter-step twiddles, consecutive in j).  Computed in double on the host.
struct fft_tables {
    const float2 *tw = nullptr, *twM = nullptr, *big = nullptr;
};

int upload_twiddles(km_ctx *c, int which, int N, const f61_plan *p61, fft_tables *out)
{
    static unsigned long long t61_ready = 0;  // per DEVICE (a constant-memory symbol lives in each device's code object): bit = device ordinal
    const unsigned long long dev_bit = 1ull << (c->device & 63);
    if (!(t61_ready & dev_bit)) {             // (cos, sin)(2 pi j k / 61), j, k = 1 .. 30, for the first form's radix-61 butterfly
        std::vector<float2> h(900);
        for (int k = 1; k <= 30; k++)
            for (int j = 1; j <= 30; j++) {
                const double ang = 2.0 * M_PI * (double)((j * k) % 61) / 61.0;
                h[(size_t)(k - 1) * 30 + (j - 1)] = make_float2((float)cos(ang), (float)sin(ang));
            }
        KM_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(c_t61), h.data(), 900 * sizeof(float2)));
        t61_ready |= dev_bit;
    }
    const int M = p61 ? p61->M : 0;
    const size_t total = (size_t)N + (size_t)M + (p61 ? (size_t)N : 0);
    float2 *d = (float2 *)km_ws(c, which ? WS_FFT_TW1 : WS_FFT_TW0, (size_t)(2 * FFT_NMAX + F61_MMAX) * sizeof(float2));
    if (!d) return KM_E_NOMEM;
    out->tw = d; out->twM = p61 ? d + N : nullptr; out->big = p61 ? d + N + M : nullptr;
    if (c->fft_tw_n[which] == N && c->fft_tw_m[which] == M) return KM_OK;   // (the table's layout depends on the plan: N alone is not a key)
    std::vector<float2> h(total);
    for (int n = 0; n < N; n++) {
        const double a = -2.0 * M_PI * (double)n / (double)N;
        h[(size_t)n] = make_float2((float)cos(a), (float)sin(a));
    }
    if (p61) {
        for (int n = 0; n < M; n++) {
            const double a = -2.0 * M_PI * (double)n / (double)M;
            h[(size_t)N + n] = make_float2((float)cos(a), (float)sin(a));
        }
        for (int t = 0; t < 61; t++)
            for (int j = 0; j < M; j++) {
                const double a = -2.0 * M_PI * (double)((long long)j * t) / (double)N;
                h[(size_t)N + M + (size_t)t * M + j] = make_float2((float)cos(a), (float)sin(a));
            }
    }
    { const int rcs = km_h2d_staged(c, c->stream, d, total * sizeof(float2), h.data(), total * sizeof(float2), total * sizeof(float2), 1); if (rcs) return rcs; }   // (`h` has been read on return)
    c->fft_tw_n[which] = N; c->fft_tw_m[which] = M;
    return KM_OK;
}

template <typename T>
int launch_rows(km_ctx *c, const T *a, const T *b, ptrdiff_t sa, ptrdiff_t sb, float2 *data, float *mag, int N, int nrows, const fft_plan &plan,
                const float2 *tw, int mode)
{
    const size_t lds = (size_t)N * sizeof(float2);
    static unsigned long long opted = 0;   // per instantiation (static local of the template) and per DEVICE (hipFuncSetAttribute applies to the current one)
    if (lds > 48 * 1024 && !(opted & (1ull << (c->device & 63)))) {
        KM_HIP(c, hipFuncSetAttribute((const void *)fft_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(FFT_NMAX * sizeof(float2))));
        opted |= 1ull << (c->device & 63);
    }
    const int grid = nrows < c->n_cu * 8 ? nrows : c->n_cu * 8;
    fft_rows_kernel<T><<<grid, FFT_T, lds, c->stream>>>(a, b, sa, sb, data, mag, N, nrows, plan, tw, mode);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

template <typename T, int MC, int MODE>
int launch_rows61_as(km_ctx *c, const T *a, const T *b, ptrdiff_t sa, ptrdiff_t sb, float2 *data, f61_top2 *top2, int N, int nrows, const f61_plan &plan,
                     const fft_tables &tb, float2 *out_plane, int run_rows)
{
    const size_t lds = ((size_t)N + (size_t)plan.M) * sizeof(float2);
    static unsigned long long opted = 0;   // per instantiation and per DEVICE: hipFuncSetAttribute applies to the current device only
    if (!(opted & (1ull << (c->device & 63)))) {
        KM_HIP(c, hipFuncSetAttribute((const void *)fft61_rows_kernel<T, MC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)((FFT_NMAX + F61_MMAX) * sizeof(float2))));
        opted |= 1ull << (c->device & 63);
    }
    const int nrun = run_rows ? run_rows : nrows;
    const int grid = nrun < c->n_cu ? nrun : c->n_cu;        // one 88-KB workgroup per CU: each walks its rows with the next one in flight
    fft61_rows_kernel<T, MC, MODE><<<grid, F61_T, lds, c->stream>>>(a, b, sa, sb, data, top2, N, nrows, plan, tb.twM, tb.big, out_plane, run_rows);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// mode 0 reads two images of pixel type T; modes 1 .. 3 work on the complex plane (T = float).  Sentinel-2's band sizes run with M
// as a compile-time constant: 10980 = 61 * 180 (10 m), 5490 = 61 * 90 (20 m), 1830 = 61 * 30 (60 m).
template <typename T, int MODE>
int launch_rows61_m(km_ctx *c, const T *a, const T *b, ptrdiff_t sa, ptrdiff_t sb, float2 *data, f61_top2 *top2, int N, int nrows, const f61_plan &plan,
                    const fft_tables &tb, float2 *out_plane, int run_rows = 0)
{
    switch (plan.M) {
    case 180: return launch_rows61_as<T, 180, MODE>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, out_plane, run_rows);
    case 90: return launch_rows61_as<T, 90, MODE>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, out_plane, run_rows);
    case 30: return launch_rows61_as<T, 30, MODE>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, out_plane, run_rows);
    default: return launch_rows61_as<T, 0, MODE>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, out_plane, run_rows);
    }
}

template <typename T>
int launch_rows61(km_ctx *c, const T *a, const T *b, ptrdiff_t sa, ptrdiff_t sb, float2 *data, f61_top2 *top2, int N, int nrows, const f61_plan &plan,
                  const fft_tables &tb, int mode, float2 *out_plane = nullptr, int run_rows = 0)
{
    if (mode == 0) return launch_rows61_m<T, 0>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, nullptr);
    if constexpr (std::is_same<T, float>::value) {
        switch (mode) {
        case 1: return launch_rows61_m<float, 1>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, nullptr);
        case 2: return launch_rows61_m<float, 2>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, nullptr);
        case 4: return launch_rows61_m<float, 4>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, out_plane, run_rows);
        case 5: return launch_rows61_m<float, 5>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, nullptr, run_rows);
        default: return launch_rows61_m<float, 3>(c, a, b, sa, sb, data, top2, N, nrows, plan, tb, nullptr);
        }
    }
    return km_fail(c, KM_E_INTERNAL, "fft61: complex passes run on float planes");
}

}  // namespace

// True when the float32 fast path covers this shape.
bool kp_fast_supported(int H, int W)
{
    fft_plan p;
    f61_plan q;
    return H >= 2 && W >= 2 && (factorize(H, &p) || factorize61(H, &q)) && (factorize(W, &p) || factorize61(W, &q)) && (size_t)H * W <= 0x7fffffffull;
}

// Phase correlation in float32.  out_rc = integer shift as skimage reports it; *margin = (max - second largest) / max of |cc|:
// the caller trusts the result only when the peak stands clear of every other sample.
int kp_phase_shift_fast(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b,
                        double out_rc[2], double *margin)
{
    fft_plan pw, ph;
    f61_plan qw, qh;
    const bool w61 = c->opt_fft61 && factorize61(W, &qw), h61 = c->opt_fft61 && factorize61(H, &qh);
    if ((!w61 && !factorize(W, &pw)) || (!h61 && !factorize(H, &ph))) return KM_E_UNSUPPORTED;
    const size_t n = (size_t)H * W;
    float2 *A = (float2 *)km_ws(c, WS_FFT_A, n * sizeof(float2)), *B = (float2 *)km_ws(c, WS_FFT_B, n * sizeof(float2));
    km_scalars *sc = (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars));
    f61_top2 *top2 = (f61_top2 *)km_ws(c, WS_FFT_TOP2, (size_t)H * sizeof(f61_top2));
    if (!A || !B || !sc || !top2) return KM_E_NOMEM;
    fft_tables tw_w, tw_h;
    int rc;
    if ((rc = upload_twiddles(c, 0, W, w61 ? &qw : nullptr, &tw_w))) return rc;
    if (H == W) tw_h = tw_w;
    else if ((rc = upload_twiddles(c, 1, H, h61 ? &qh : nullptr, &tw_h))) return rc;
    // complex rows of length N (in place; mode 3: no output plane when the 61 M form runs, |cc| into `mag` otherwise)
    auto rows = [&](float2 *data, float *mag, int N, int nrows, bool is_w, int mode) -> int {
        if (is_w ? w61 : h61) return launch_rows61<float>(c, nullptr, nullptr, 0, 0, data, top2, N, nrows, is_w ? qw : qh, is_w ? tw_w : tw_h, mode);
        return launch_rows<float>(c, nullptr, nullptr, 0, 0, data, mag, N, nrows, is_w ? pw : ph, (is_w ? tw_w : tw_h).tw, mode);
    };
    // (transpose kernels between the passes: folded into the row passes' stores they measured slower - CHANGELOG.md, round 4)
    // forward: rows (length W) of z = a + i b -> (transposed) -> B (W x H); rows (length H) in place
    switch (dtype) {
#define KM_ROWS(CODE, T)                                                                                                                    \
    case CODE:                                                                                                                              \
        rc = w61 ? launch_rows61<T>(c, (const T *)d_a, (const T *)d_b, stride_a, stride_b, A, nullptr, W, H, qw, tw_w, 0) \
                 : launch_rows<T>(c, (const T *)d_a, (const T *)d_b, stride_a, stride_b, A, nullptr, W, H, pw, tw_w.tw, 0);                 \
        break;
        KM_ROWS(KM_U8, uint8_t) KM_ROWS(KM_U16, uint16_t) KM_ROWS(KM_I16, int16_t) KM_ROWS(KM_F32, float)
#undef KM_ROWS
    default: return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
    }
    if (rc) return rc;
    const dim3 tg((W + 63) / 64, (H + 63) / 64), tg2((H + 63) / 64, (W + 63) / 64);
    transpose_kernel<<<tg, 256, 0, c->stream>>>(A, B, H, W);
    KM_LAUNCH_CHECK(c);
    if ((rc = rows(B, nullptr, H, W, false, 1))) return rc;
    // inverse: rows (length H) of the transposed cross-power spectrum; (transposed) -> H x W; rows (length W) -> |cc|
    float2 *last = B;          // the plane the last pass reads
    // Hermitian inverse ("fft_herm", both sides 61 M): the cross-power spectrum of two real images satisfies P(-k) = conj P(k), so the
    // first inverse pass only runs the rows kx <= W / 2, the transpose moves half a plane and the last pass transforms two image
    // rows per complex transform (fft61_rows_kernel mode 5) - the correlation surface comes out exactly real
    const bool herm = w61 && h61 && c->opt_fft_cross_fused && c->opt_fft_herm;
    if (herm) {
        const int Wh = W / 2 + 1;
        if ((rc = launch_rows61<float>(c, nullptr, nullptr, 0, 0, B, top2, H, W, qh, tw_h, 4, A, Wh))) return rc;       // A: Wh x H
        const dim3 tgh((H + 63) / 64, (Wh + 63) / 64);
        transpose_kernel<<<tgh, 256, 0, c->stream>>>(A, B, Wh, H);                                                           // B: H x Wh
        KM_LAUNCH_CHECK(c);
        if ((rc = launch_rows61<float>(c, nullptr, nullptr, 0, 0, B, top2, W, H, qw, tw_w, 5, nullptr, (H + 1) / 2))) return rc;
    } else if (h61 && c->opt_fft_cross_fused) {
        // the cross-power step rides on the row load of the first inverse pass: B (= Z) -> A
        if ((rc = launch_rows61<float>(c, nullptr, nullptr, 0, 0, B, top2, H, W, qh, tw_h, 4, A))) return rc;
    } else {
        cross_power_f32_kernel<<<c->n_cu * 16, 256, 0, c->stream>>>(B, A, W, H);
        KM_LAUNCH_CHECK(c);
        if ((rc = rows(A, nullptr, H, W, false, 2))) return rc;
    }
    if (!herm) {
        transpose_kernel<<<tg2, 256, 0, c->stream>>>(A, B, W, H);
        KM_LAUNCH_CHECK(c);
    }
    float *cc = (float *)A;
    if (!herm && (rc = rows(last, cc, W, H, true, 3))) return rc;
    // largest and second-largest |cc|
    unsigned long long *k1 = &sc->argmax_key, *k2 = (unsigned long long *)&sc->valid;
    unsigned long long h1 = 0, h2 = 0;
    unsigned long long flat;
    unsigned b1, b2;
    if (w61) {
        // the rows reported their two largest samples: one small launch, one read-back
        unsigned long long *d_res = (unsigned long long *)&sc->hist[0];
        f61_top2_reduce_kernel<<<1, 1024, 0, c->stream>>>(top2, H, d_res);
        KM_LAUNCH_CHECK(c);
        unsigned long long res[2] = {0, 0};
        { int rq = km_d2h_queue(c, res, d_res, sizeof res); if (!rq) rq = km_d2h_flush(c); if (rq) return rq; }
        h1 = res[0];
        flat = h1 ? 0xffffffffull - (h1 & 0xffffffffull) : 0ull;
        b1 = (unsigned)(h1 >> 32); b2 = (unsigned)res[1];
    } else {
        KM_HIP(c, hipMemsetAsync(k1, 0, sizeof *k1, c->stream));
        KM_HIP(c, hipMemsetAsync(k2, 0, sizeof *k2, c->stream));
        argmax_f32_kernel<<<c->n_cu * 8, 256, 0, c->stream>>>(cc, n, ~0ull, k1);
        KM_LAUNCH_CHECK(c);
        { int rq = km_d2h_queue(c, &h1, k1, sizeof h1); if (!rq) rq = km_d2h_flush(c); if (rq) return rq; }
        flat = h1 ? 0xffffffffull - (h1 & 0xffffffffull) : 0ull;
        argmax_f32_kernel<<<c->n_cu * 8, 256, 0, c->stream>>>(cc, n, flat, k2);
        KM_LAUNCH_CHECK(c);
        { int rq = km_d2h_queue(c, &h2, k2, sizeof h2); if (!rq) rq = km_d2h_flush(c); if (rq) return rq; }
        b1 = (unsigned)(h1 >> 32); b2 = (unsigned)(h2 >> 32);
    }
    float v1, v2;
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
