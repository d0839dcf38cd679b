// K10, double precision: phase correlation in the reference's own arithmetic, hand-written
// (skimage.registration.phase_cross_correlation, reference matcher/large_offset.py:39: complex128 fftn / ifftn).
// Takes over from the float32 transform of k_fft.hip when its peak is not clear, under `phase_fp64`, and for every side length the
// float32 kernels do not factor.  No FFT library is linked.
//
// A 10980-point complex128 row is 175 KB - more than a CU's LDS - so the transform is NOT a row-in-LDS kernel retyped.  Each
// dimension of length N = n_1 n_2 .. n_L is done as L *levels* (the "four-step" decomposition, applied recursively), every level
// one kernel that runs IN PLACE over the whole plane:
//     forward level l:  for every block of N_l = n_l R_l consecutive elements (R_l = n_{l+1} .. n_L) and every r < R_l:
//                       the n_l elements at stride R_l are replaced by their DFT, output k multiplied by W_{N_l}^(r k)
//     inverse level l:  the same elements are multiplied by conj W_{N_l}^(r k), then replaced by their inverse DFT
// forward runs l = 1 .. L and leaves frequency k = k_1 + n_1 k_2 + n_1 n_2 k_3 .. at POSITION k_1 R_1 + k_2 R_2 + .. (a digit
// permutation); the inverse runs l = L .. 1 on that order and ends in natural order.  Nothing in between needs natural order: the
// cross-power spectrum is point-wise apart from the pairing k <-> -k, which a position table per dimension provides.  So the plane
// (z = a + i b, both images in ONE complex transform; 16 B per pixel, 1.93 GB at 10980 x 10980) is the only large buffer, there is
// no transpose, and a level's tile (T transforms of length n_l, T chosen so that a tile row is a whole 128-byte line in either
// dimension) needs n_l T 16 B of LDS: 46 KB for 10980 = 61 x 180, several workgroups per CU, loads of one under the arithmetic of
// another.  Two level kernels:
//     * smooth  n (prime factors 2, 3, 5, 7; <= 2048): Stockham stages in LDS, radices 7 / 5 / 4 / 3 / 2;
//     * prime   p (11 <= p <= 127, e.g. the 61 of Sentinel-2's 10980): lane = transform, the p inputs stream from LDS once per
//       group of 8 output pairs (X_k, X_{p-k} share the folded inputs x_j +- x_{p-j}), coefficients through the scalar path.
// A side with a prime factor above 127 goes through Bluestein's chirp-z on top of the same kernels (power-of-two length >= 2 N - 1).
// The unnormalised inverse is used (the arg-max does not depend on a factor).
#include "common.hpp"
#include "fft64_plan.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {

using namespace f64plan;
typedef double2 cd;

__device__ __forceinline__ cd c_add(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd c_sub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cd c_mul(cd a, cd b) { return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)); }

#define F64_KB 8              // output pairs per wavefront pass of the prime kernel

struct lvl_args {
    cd *data;
    const cd *tw;                    // exp(-2 pi i j / N), j < N  (N = length of the dimension)
    long long se, sb, s_blk, s_r;    // strides in elements: between the elements of a transform, between the transforms of a tile, of `a`
    int n;                           // transform length
    int T, logT;                     // transforms per workgroup (a power of two unless `contiguous`)
    int AR;                          // a -> (a / AR) * s_blk + (a % AR) * s_r
    int B, tiles_b;                  // transforms along b, tiles of T
    int tw_mode;                     // inter-level twiddle W_{n R}^(q k):  0 none, 1 q = b, 2 q = a % AR
    int tw_scale;                    // N / (n R)
    int N;
    int inverse;
    int contiguous;                  // smooth kernel: every transform is contiguous in memory (se == 1): the lanes walk i, not t
    int nst;
    int radix[F64_MAX_STAGES];
};

// ---------------------------------------------------------------------------------------------------------------- small DFTs
template <int P> struct odd_tab;
template <> struct odd_tab<3> {
    static constexpr double c[3] = {1.0, -0.5, -0.5};
    static constexpr double s[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
};
template <> struct odd_tab<5> {
    static constexpr double c[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410, 0.30901699437494742410};
    static constexpr double s[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212};
};
template <> struct odd_tab<7> {
    static constexpr double c[7] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624,
                                    -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053};
    static constexpr double s[7] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048,
                                    -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871};
};

// forward DFT (e^-) of R values in registers
template <int R> __device__ __forceinline__ void dft_r(cd *v)
{
    if constexpr (R == 2) {
        const cd a = v[0], b = v[1];
        v[0] = c_add(a, b); v[1] = c_sub(a, b);
    } else if constexpr (R == 4) {
        const cd a = c_add(v[0], v[2]), b = c_sub(v[0], v[2]), c = c_add(v[1], v[3]), d = c_sub(v[1], v[3]);
        v[0] = c_add(a, c); v[2] = c_sub(a, c);
        v[1] = make_double2(b.x + d.y, b.y - d.x);          // b - i d
        v[3] = make_double2(b.x - d.y, b.y + d.x);          // b + i d
    } else {
        // X_k = x_0 + sum_j (x_j + x_{R-j}) cos(2 pi j k / R) - i sum_j (x_j - x_{R-j}) sin(2 pi j k / R),  X_{R-k}: + i
        constexpr int Hh = (R - 1) / 2;
        cd a[Hh], b[Hh];
        cd sum = v[0];
#pragma unroll
        for (int j = 0; j < Hh; j++) { a[j] = c_add(v[j + 1], v[R - 1 - j]); b[j] = c_sub(v[j + 1], v[R - 1 - j]); sum = c_add(sum, a[j]); }
        const cd x0 = v[0];
        v[0] = sum;
#pragma unroll
        for (int k = 1; k <= Hh; k++) {
            cd C = x0, S = make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 1; j <= Hh; j++) {
                const double cc = odd_tab<R>::c[(j * k) % R], ss = odd_tab<R>::s[(j * k) % R];
                C.x = fma(a[j - 1].x, cc, C.x); C.y = fma(a[j - 1].y, cc, C.y);
                S.x = fma(b[j - 1].x, ss, S.x); S.y = fma(b[j - 1].y, ss, S.y);
            }
            v[k] = make_double2(C.x + S.y, C.y - S.x);       // C - i S
            v[R - k] = make_double2(C.x - S.y, C.y + S.x);   // C + i S
        }
    }
}

// one Stockham butterfly: radix R, Ns = product of the radices done; element i of transform t at buf[i * lsi + toff]
template <int R>
__device__ __forceinline__ void bfly(const cd *__restrict__ src, cd *__restrict__ dst, int b, int nb, int Ns, unsigned magic, int lsi, int toff,
                                     const cd *__restrict__ twl, int twstep)
{
    const int q = Ns == 1 ? b : (int)__umulhi((unsigned)b, magic), k = b - q * Ns;     // b / Ns, b % Ns (exact: b Ns < 2^32)
    cd v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = src[(b + r * nb) * lsi + toff];
    if (Ns > 1) {
#pragma unroll
        for (int r = 1; r < R; r++) v[r] = c_mul(v[r], twl[r * k * twstep]);
    }
    dft_r<R>(v);
    const int j0 = q * Ns * R + k;
#pragma unroll
    for (int r = 0; r < R; r++) dst[(j0 + r * Ns) * lsi + toff] = v[r];
}

__device__ __forceinline__ long long tile_base(const lvl_args &A, int a, int b0)
{
    return (long long)(a / A.AR) * A.s_blk + (long long)(a % A.AR) * A.s_r + (long long)b0 * A.sb;
}

// ---------------------------------------------------------------------------------------------------------------- smooth level
__global__ __launch_bounds__(256) void f64_smooth_kernel(const lvl_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int n = A.n, T = A.T, logT = A.logT;
    cd *buf0 = (cd *)smem64, *buf1 = buf0 + n * T, *twl = buf1 + n * T;       // twl[j] = exp(-2 pi i j / n)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, a = wg / A.tiles_b, tb = wg - a * A.tiles_b, b0 = tb * T, nt = min(T, A.B - b0);
    const long long base = tile_base(A, a, b0);
    const int qa = a % A.AR;
    cd *__restrict__ data = A.data;
    const cd *__restrict__ tw = A.tw;
    for (int j = tid; j < n; j += 256) twl[j] = tw[(size_t)j * (size_t)(A.N / n)];
    const bool contig = A.contiguous != 0;
    const int lsi = contig ? 1 : T, lst = contig ? n : 1;

    auto ld = [&](int i, int t) {
        cd v = data[base + (long long)i * A.se + (long long)t * A.sb];
        if (A.inverse) {
            v.y = -v.y;                                                        // inverse DFT = conj . DFT . conj
            if (A.tw_mode) v = c_mul(v, tw[(size_t)(A.tw_mode == 1 ? b0 + t : qa) * (size_t)i * (size_t)A.tw_scale]);   // conj(x conj w) = conj(x) w
        }
        buf0[i * lsi + t * lst] = v;
    };
    if (contig) {
        for (int t = wave; t < nt; t += 4)
            for (int i = lane; i < n; i += 64) ld(i, t);
    } else {
        for (int idx = tid; idx < n * T; idx += 256) {
            const int t = idx & (T - 1), i = idx >> logT;
            if (t < nt) ld(i, t); else buf0[idx] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    cd *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int s = 0; s < A.nst; s++) {
        const int R = A.radix[s], nb = n / R, twstep = n / (Ns * R);
        const unsigned magic = Ns > 1 ? (unsigned)((0x100000000ull + (unsigned)Ns - 1) / (unsigned)Ns) : 0u;
        auto run = [&](auto rc) {
            constexpr int RR = decltype(rc)::value;
            if (contig) {
                for (int t = wave; t < nt; t += 4)
                    for (int b = lane; b < nb; b += 64) bfly<RR>(src, dst, b, nb, Ns, magic, 1, t * n, twl, twstep);
            } else {
                for (int idx = tid; idx < nb * T; idx += 256) bfly<RR>(src, dst, idx >> logT, nb, Ns, magic, T, idx & (T - 1), twl, twstep);
            }
        };
        switch (R) {
        case 7: run(std::integral_constant<int, 7>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        default: run(std::integral_constant<int, 2>{}); break;
        }
        __syncthreads();
        cd *tmp = src; src = dst; dst = tmp;
        Ns *= R;
    }

    auto st = [&](int i, int t) {
        cd v = src[i * lsi + t * lst];
        if (A.inverse) v.y = -v.y;
        else if (A.tw_mode) v = c_mul(v, tw[(size_t)(A.tw_mode == 1 ? b0 + t : qa) * (size_t)i * (size_t)A.tw_scale]);
        data[base + (long long)i * A.se + (long long)t * A.sb] = v;
    };
    if (contig) {
        for (int t = wave; t < nt; t += 4)
            for (int i = lane; i < n; i += 64) st(i, t);
    } else {
        for (int idx = tid; idx < n * T; idx += 256) {
            const int t = idx & (T - 1), i = idx >> logT;
            if (t < nt) st(i, t);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- prime level
__global__ __launch_bounds__(256) void f64_prime_kernel(const lvl_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    cd *sm = (cd *)smem64;                                                     // [i][t], t < T <= 64
    const int p = A.n, T = A.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, a = wg / A.tiles_b, tb = wg - a * A.tiles_b, b0 = tb * T, nt = min(T, A.B - b0);
    const long long base = tile_base(A, a, b0);
    cd *__restrict__ data = A.data;
    const cd *__restrict__ tw = A.tw;
    const int q = A.tw_mode == 1 ? b0 + lane : a % A.AR;
    const long long loff = (long long)lane * A.sb;
    for (int i = wave; i < p; i += 4) {
        cd v = make_double2(0.0, 0.0);
        if (lane < nt) {
            v = data[base + (long long)i * A.se + loff];
            if (A.inverse) {
                v.y = -v.y;
                if (A.tw_mode) v = c_mul(v, tw[(size_t)q * (size_t)i * (size_t)A.tw_scale]);
            }
        }
        if (lane < T) sm[i * T + lane] = v;
    }
    __syncthreads();
    const int h = (p - 1) / 2;
    const size_t tws = (size_t)(A.N / p);
    const int l = min(lane, T - 1);
    auto put = [&](int k, cd v) {
        if (lane >= nt) return;                                                // (q of an idle lane would index past the table)
        if (A.inverse) v.y = -v.y;
        else if (A.tw_mode) v = c_mul(v, tw[(size_t)q * (size_t)k * (size_t)A.tw_scale]);
        data[base + (long long)k * A.se + loff] = v;
    };
    // the coefficient addresses are the same for every lane: through the constant address space they become scalar loads
    typedef const __attribute__((address_space(4))) double *scalar_f64_ptr;
    const scalar_f64_ptr twc = (scalar_f64_ptr)(unsigned long long)tw;
    auto coef = [&](int i) { const size_t o = 2 * (size_t)i * tws; return make_double2(twc[o], twc[o + 1]); };
    for (int g0 = wave * F64_KB; g0 < h; g0 += 4 * F64_KB) {                   // (uniform per wavefront)
        cd C[F64_KB], S[F64_KB], w[F64_KB];
        int idx[F64_KB];
#pragma unroll
        for (int kk = 0; kk < F64_KB; kk++) {
            C[kk] = S[kk] = make_double2(0.0, 0.0);
            idx[kk] = g0 + kk + 1;                                             // (j k) mod p for j = 1, k = g0 + kk + 1 < p
            w[kk] = coef(idx[kk]);                                             // (cos, -sin)(2 pi j k / p)
        }
        const cd x0 = sm[l];
        cd sum = x0;
        cd xa = sm[T + l], xb = sm[(p - 1) * T + l];
        for (int j = 1; j <= h; j++) {
            // inputs and coefficients of j + 1 travel while j is accumulated (j = h fetches a pair nobody uses)
            const cd nxa = sm[(j + 1) * T + l], nxb = sm[(p - j - 1) * T + l];
            cd nw[F64_KB];
#pragma unroll
            for (int kk = 0; kk < F64_KB; kk++) {
                idx[kk] += g0 + kk + 1;
                if (idx[kk] >= p) idx[kk] -= p;
                nw[kk] = coef(idx[kk]);
            }
            const cd fa = c_add(xa, xb), fb = c_sub(xa, xb);
            sum = c_add(sum, fa);
#pragma unroll
            for (int kk = 0; kk < F64_KB; kk++) {
                C[kk].x = fma(fa.x, w[kk].x, C[kk].x); C[kk].y = fma(fa.y, w[kk].x, C[kk].y);
                S[kk].x = fma(fb.x, w[kk].y, S[kk].x); S[kk].y = fma(fb.y, w[kk].y, S[kk].y);      // S = - sum (x_j - x_{p-j}) sin
                w[kk] = nw[kk];
            }
            xa = nxa; xb = nxb;
        }
        const int nk = min(F64_KB, h - g0);
#pragma unroll
        for (int kk = 0; kk < F64_KB; kk++) {
            if (kk < nk) {
                const int k = g0 + kk + 1;
                const double cr = x0.x + C[kk].x, ci = x0.y + C[kk].y;
                put(k, make_double2(cr - S[kk].y, ci + S[kk].x));              // x0 + C + i S
                put(p - k, make_double2(cr + S[kk].y, ci - S[kk].x));          // x0 + C - i S
            }
        }
        if (g0 == 0) put(0, sum);
    }
}

// ---------------------------------------------------------------------------------------------------------------- around the transform
template <typename T>
__global__ __launch_bounds__(256) void f64_pack_kernel(const T *__restrict__ a, const T *__restrict__ b, ptrdiff_t sa, ptrdiff_t sb, int H, int W,
                                                        cd *__restrict__ z)
{
    for (int y = blockIdx.y; y < H; y += gridDim.y) {
        const T *pa = a + (size_t)y * sa, *pb = b + (size_t)y * sb;
        cd *pz = z + (size_t)y * W;
        for (int x = blockIdx.x * 256 + threadIdx.x; x < W; x += gridDim.x * 256) pz[x] = make_double2((double)pa[x], (double)pb[x]);
    }
}

// Z = spectrum of a + i b at permuted positions; neg*[pos] = position of the negated frequency.  In place, a pair (k, -k) per thread:
// A = (Z(k) + conj Z(-k)) / 2, B = (Z(k) - conj Z(-k)) / 2i, P = A conj(B) / max(|A conj(B)|, 100 eps); P(-k) = conj P(k) (real images)
__global__ __launch_bounds__(256) void f64_cross_kernel(cd *__restrict__ Z, const int *__restrict__ negx, const int *__restrict__ negy, int H, int W)
{
    const double floor_ = 100.0 * 2.220446049250313e-16;
    for (int py = blockIdx.y; py < H; py += gridDim.y) {
        const int ny = negy[py];
        if (ny < py) continue;
        for (int px = blockIdx.x * 256 + threadIdx.x; px < W; px += gridDim.x * 256) {
            const int nx = negx[px];
            if (ny == py && nx < px) continue;
            const size_t i0 = (size_t)py * W + px, i1 = (size_t)ny * W + nx;
            const cd zk = Z[i0], zn = Z[i1];
            const cd zm = make_double2(zn.x, -zn.y);
            const cd fa = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y));
            const cd d = c_sub(zk, zm);
            const cd fb = make_double2(0.5 * d.y, -0.5 * d.x);
            const double re = fa.x * fb.x + fa.y * fb.y, im = fa.y * fb.x - fa.x * fb.y;       // fa conj(fb)
            const double mag = fmax(hypot(re, im), floor_);
            const cd P = make_double2(re / mag, im / mag);
            Z[i0] = P;
            if (i1 != i0) Z[i1] = make_double2(P.x, -P.y);
        }
    }
}

__device__ __forceinline__ unsigned long long wmax64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ unsigned long long wmin64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
// np.argmax(np.abs(cc)): largest |cc| (bit pattern of a non-negative double is monotone), then the first flat index that attains it
__global__ __launch_bounds__(256) void f64_absmax_kernel(const cd *__restrict__ cc, size_t n, unsigned long long *out)
{
    unsigned long long best = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const cd v = cc[i];
        const double m = hypot(v.x, v.y);
        if (m == m) { const unsigned long long b = (unsigned long long)__double_as_longlong(m); best = b > best ? b : best; }
    }
    best = wmax64(best);
    if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}
__global__ __launch_bounds__(256) void f64_first_index_kernel(const cd *__restrict__ cc, size_t n, const unsigned long long *maxbits, unsigned long long *out)
{
    const unsigned long long mb = *maxbits;
    unsigned long long best = ~0ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const cd v = cc[i];
        if ((unsigned long long)__double_as_longlong(hypot(v.x, v.y)) == mb) best = i < best ? i : best;
    }
    best = wmin64(best);
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

// ---- Bluestein's pieces (rows of a contiguous [rows][N] array <-> rows of a [rows][L] scratch)
__global__ __launch_bounds__(256) void blue_in_kernel(const cd *__restrict__ x, cd *__restrict__ s, const cd *__restrict__ chirp, int rows, int N, int L, int inverse)
{
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        for (int m = blockIdx.x * 256 + threadIdx.x; m < L; m += gridDim.x * 256) {
            cd v = make_double2(0.0, 0.0);
            if (m < N) {
                v = x[(size_t)r * N + m];
                if (inverse) v.y = -v.y;
                v = c_mul(v, chirp[m]);
            }
            s[(size_t)r * L + m] = v;
        }
}
__global__ __launch_bounds__(256) void blue_mul_kernel(cd *__restrict__ s, const cd *__restrict__ bhat, int rows, int L)
{
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        for (int m = blockIdx.x * 256 + threadIdx.x; m < L; m += gridDim.x * 256) s[(size_t)r * L + m] = c_mul(s[(size_t)r * L + m], bhat[m]);
}
__global__ __launch_bounds__(256) void blue_out_kernel(cd *__restrict__ x, const cd *__restrict__ s, const cd *__restrict__ chirp, int rows, int N, int L, int inverse)
{
    const double scale = 1.0 / (double)L;
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        for (int k = blockIdx.x * 256 + threadIdx.x; k < N; k += gridDim.x * 256) {
            cd v = c_mul(s[(size_t)r * L + k], chirp[k]);
            v.x *= scale; v.y *= scale;
            if (inverse) v.y = -v.y;
            x[(size_t)r * N + k] = v;
        }
}
__global__ __launch_bounds__(256) void f64_transpose_kernel(const cd *__restrict__ in, cd *__restrict__ out, int rows, int cols)
{
    __shared__ cd tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * 16, y0 = blockIdx.y * 16;
    if (y0 + ty < rows && x0 + tx < cols) tile[ty][tx] = in[(size_t)(y0 + ty) * cols + x0 + tx];
    __syncthreads();
    if (x0 + ty < cols && y0 + tx < rows) out[(size_t)(x0 + ty) * rows + y0 + tx] = tile[tx][ty];
}

// ================================================================================================================ host
// exp(-2 pi i j / N) in extended precision, rounded once
void host_twiddles(int N, std::vector<cd> &tw)
{
    tw.resize((size_t)N);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int j = 0; j < N; j++) {
        // reduce to the first octant-pair by symmetry about pi: angle of j and of N - j are conjugates
        const int jj = j <= N - j ? j : N - j;
        const long double ang = two_pi * (long double)jj / (long double)N;
        double c = (double)cosl(ang), s = (double)sinl(ang);
        if (jj == 0) { c = 1.0; s = 0.0; }
        else if (2ll * jj == N) { c = -1.0; s = 0.0; }
        else if (4ll * jj == N) { c = 0.0; s = 1.0; }
        tw[(size_t)j] = make_double2(c, j == jj ? -s : s);
    }
}

int opt_in_lds(km_ctx *c)
{
    static unsigned long long opted = 0;     // per DEVICE
    const unsigned long long bit = 1ull << (c->device & 63);
    if (!(opted & bit)) {
        KM_HIP(c, hipFuncSetAttribute((const void *)f64_smooth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KM_HIP(c, hipFuncSetAttribute((const void *)f64_prime_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        opted |= bit;
    }
    return KM_OK;
}

// one level over a plane of `rows` rows of `width` elements (contiguous): along the rows (`cols` false, dimension length = width) or
// along the columns (dimension length = rows)
int run_level(km_ctx *c, cd *data, const cd *tw, int N, const lvl &L, bool inverse, bool cols, int rows, int width)
{
    lvl_args A;
    A.data = data; A.tw = tw; A.n = L.n; A.N = N; A.inverse = inverse ? 1 : 0;
    const long long Nl = (long long)L.n * L.R;
    A.tw_scale = (int)(N / Nl);
    long long nA;
    if (!cols) {
        if (L.R > 1) {
            A.AR = 1; nA = (long long)rows * (width / Nl); A.s_blk = Nl; A.s_r = 0;
            A.B = L.R; A.sb = 1; A.se = L.R; A.tw_mode = 1; A.contiguous = 0;
        } else {
            A.AR = 1; nA = 1; A.s_blk = 0; A.s_r = 0;
            const long long nb = (long long)rows * (width / L.n);
            if (nb > 0x7fffffffll) return km_fail(c, KM_E_ARG, "phase correlation: plane too large");
            A.B = (int)nb; A.sb = L.n; A.se = 1; A.tw_mode = 0; A.contiguous = 1;
        }
    } else {
        A.AR = L.R; nA = (long long)(rows / Nl) * L.R; A.s_blk = Nl * width; A.s_r = width;
        A.B = width; A.sb = 1; A.se = (long long)L.R * width; A.tw_mode = L.R > 1 ? 2 : 0; A.contiguous = 0;
    }
    A.nst = L.nst;
    for (int i = 0; i < F64_MAX_STAGES; i++) A.radix[i] = L.radix[i];
    size_t lds;
    if (L.kind == 1) {
        const int tmax = std::min(64, std::max(1, 4096 / L.n));
        const int tiles = (A.B + tmax - 1) / tmax;
        A.T = (A.B + tiles - 1) / tiles; A.logT = 0;
        A.tiles_b = (A.B + A.T - 1) / A.T;
        lds = (size_t)L.n * A.T * sizeof(cd);
    } else if (A.contiguous) {
        int T = std::max(1, std::min(8, F64_SMOOTH_MAX / L.n));
        T = std::min(T, A.B);
        A.T = T; A.logT = 0;
        A.tiles_b = (A.B + T - 1) / T;
        lds = ((size_t)2 * L.n * T + L.n) * sizeof(cd);
    } else {
        int T = 8, lg = 3;
        while (T > 1 && L.n * T > F64_SMOOTH_MAX) { T >>= 1; lg--; }
        while (T > 1 && (T >> 1) >= A.B) { T >>= 1; lg--; }
        A.T = T; A.logT = lg;
        A.tiles_b = (A.B + T - 1) / T;
        lds = ((size_t)2 * L.n * T + L.n) * sizeof(cd);
    }
    const long long grid = nA * A.tiles_b;
    if (grid <= 0 || grid > 0x7fffffffll) return km_fail(c, KM_E_ARG, "phase correlation: plane too large");
    if (L.kind == 1) f64_prime_kernel<<<(unsigned)grid, 256, lds, c->stream>>>(A);
    else f64_smooth_kernel<<<(unsigned)grid, 256, lds, c->stream>>>(A);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

struct blue_tabs {
    dimplan inner;                 // plan of the power-of-two length L (rows of the scratch)
    const cd *tw = nullptr;        // exp(-2 pi i j / L)
    const cd *chirp = nullptr;     // exp(-i pi m^2 / N), m < N
    cd *bhat = nullptr;            // transform of the chirp filter, in the permuted order the forward levels leave
};

int fft_rows_levels(km_ctx *c, cd *data, const cd *tw, const dimplan &P, bool inverse, int rows, int width)
{
    const int nl = (int)P.lv.size();
    for (int i = 0; i < nl; i++) {
        const int rc = run_level(c, data, tw, P.N, P.lv[(size_t)(inverse ? nl - 1 - i : i)], inverse, false, rows, width);
        if (rc) return rc;
    }
    return KM_OK;
}
int fft_cols_levels(km_ctx *c, cd *data, const cd *tw, const dimplan &P, bool inverse, int rows, int width)
{
    const int nl = (int)P.lv.size();
    for (int i = 0; i < nl; i++) {
        const int rc = run_level(c, data, tw, P.N, P.lv[(size_t)(inverse ? nl - 1 - i : i)], inverse, true, rows, width);
        if (rc) return rc;
    }
    return KM_OK;
}

// tables of a Bluestein dimension in workspace slot `slot`: [tw_L (L)] [chirp (N)] [bhat (L)]
int blue_prepare(km_ctx *c, int slot, const dimplan &P, blue_tabs *B)
{
    const int N = P.N, L = P.L;
    if (!plan_dim(L, F64_SMOOTH_MAX, &B->inner) || B->inner.blue) return km_fail(c, KM_E_UNSUPPORTED, "phase correlation: side %d too long", N);
    cd *d = (cd *)km_ws(c, slot, ((size_t)2 * L + N) * sizeof(cd));
    if (!d) return KM_E_NOMEM;
    std::vector<cd> h((size_t)2 * L + N);
    std::vector<cd> twL;
    host_twiddles(L, twL);
    std::copy(twL.begin(), twL.end(), h.begin());
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int m = 0; m < N; m++) {
        const long long e = ((long long)m * m) % (2ll * N);                    // exp(-i pi m^2 / N) = exp(-2 pi i e / 2N)
        const long double ang = pi * (long double)e / (long double)N;
        h[(size_t)L + m] = make_double2((double)cosl(ang), -(double)sinl(ang));
    }
    cd *filt = &h[(size_t)L + N];
    for (int m = 0; m < L; m++) filt[m] = make_double2(0.0, 0.0);
    for (int m = 0; m < N; m++) {
        const cd ch = h[(size_t)L + m];
        const cd cj = make_double2(ch.x, -ch.y);
        filt[m] = cj;
        if (m) filt[L - m] = cj;
    }
    int rc = km_h2d_small(c, d, h.data(), h.size() * sizeof(cd));
    if (rc) return rc;
    B->tw = d; B->chirp = d + L; B->bhat = d + L + N;
    return fft_rows_levels(c, B->bhat, B->tw, B->inner, false, 1, L);
}

// DFT (or unnormalised inverse x N) of every row of a contiguous [rows][N] array, natural order in and out
int blue_rows(km_ctx *c, cd *x, int rows, const dimplan &P, const blue_tabs &B, bool inverse)
{
    const int N = P.N, L = P.L;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)rows, ((size_t)1 << 30) / ((size_t)L * sizeof(cd))));
    cd *s = (cd *)km_ws(c, WS_FFT_WORK, (size_t)chunk * L * sizeof(cd));
    if (!s) return KM_E_NOMEM;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = std::min(chunk, rows - r0);
        cd *xr = x + (size_t)r0 * N;
        const dim3 g((unsigned)std::min((L + 255) / 256, 64), (unsigned)std::min(nr, 4096));
        blue_in_kernel<<<g, 256, 0, c->stream>>>(xr, s, B.chirp, nr, N, L, inverse ? 1 : 0);
        KM_LAUNCH_CHECK(c);
        int rc = fft_rows_levels(c, s, B.tw, B.inner, false, nr, L);
        if (rc) return rc;
        blue_mul_kernel<<<g, 256, 0, c->stream>>>(s, B.bhat, nr, L);
        KM_LAUNCH_CHECK(c);
        rc = fft_rows_levels(c, s, B.tw, B.inner, true, nr, L);
        if (rc) return rc;
        blue_out_kernel<<<g, 256, 0, c->stream>>>(xr, s, B.chirp, nr, N, L, inverse ? 1 : 0);
        KM_LAUNCH_CHECK(c);
    }
    return KM_OK;
}

int transpose(km_ctx *c, const cd *in, cd *out, int rows, int cols)
{
    const dim3 g((unsigned)((cols + 15) / 16), (unsigned)((rows + 15) / 16));
    if (g.y > 65535u) return km_fail(c, KM_E_ARG, "phase correlation: side too long");
    f64_transpose_kernel<<<g, 256, 0, c->stream>>>(in, out, rows, cols);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

template <typename T>
int launch_pack(km_ctx *c, const void *a, const void *b, ptrdiff_t sa, ptrdiff_t sb, int H, int W, cd *z)
{
    const dim3 g((unsigned)std::min((W + 255) / 256, 64), (unsigned)std::min(H, 8192));
    f64_pack_kernel<T><<<g, 256, 0, c->stream>>>((const T *)a, (const T *)b, sa, sb, H, W, z);
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

}  // namespace

// The double-precision evaluation.  out_rc = (row, col) shift as skimage reports it.
int kp_phase_shift_f64(km_ctx *c, const void *d_a, const void *d_b, int dtype, int H, int W, ptrdiff_t stride_a, ptrdiff_t stride_b, double out_rc[2])
{
    if ((size_t)H * (size_t)W > 0x7fffffffull) return km_fail(c, KM_E_ARG, "phase correlation: image of %d x %d is too large", H, W);
    int rc = opt_in_lds(c);
    if (rc) return rc;
    dimplan PX, PY;
    if (!plan_dim(W, F64_SMOOTH_MAX, &PX) || !plan_dim(H, 256, &PY)) return km_fail(c, KM_E_UNSUPPORTED, "phase correlation: no plan for %d x %d", H, W);
    const size_t n = (size_t)H * W;
    cd *z = (cd *)km_ws(c, WS_FFT_A, n * sizeof(cd));
    km_scalars *sc = (km_scalars *)km_ws(c, WS_SCALARS, sizeof(km_scalars));
    if (!z || !sc) return KM_E_NOMEM;

    // tables: twiddles and negated-frequency positions of both dimensions (kept between calls for the same shape)
    cd *twx = nullptr, *twy = nullptr;
    int *negx = nullptr, *negy = nullptr;
    {
        const bool same = c->f64_h == H && c->f64_w == W;
        twx = (cd *)km_ws(c, WS_F64_TWX, (size_t)std::max(W, 1) * sizeof(cd));
        twy = (cd *)km_ws(c, WS_F64_TWY, (size_t)std::max(H, 1) * sizeof(cd));
        negx = (int *)km_ws(c, WS_F64_NEGX, (size_t)std::max(W, 1) * sizeof(int));
        negy = (int *)km_ws(c, WS_F64_NEGY, (size_t)std::max(H, 1) * sizeof(int));
        if (!twx || !twy || !negx || !negy) return KM_E_NOMEM;
        if (!same) {
            c->f64_h = c->f64_w = 0;
            std::vector<cd> t;
            std::vector<int> ng;
            host_twiddles(std::max(W, 1), t);
            if ((rc = km_h2d_small(c, twx, t.data(), t.size() * sizeof(cd)))) return rc;
            host_twiddles(std::max(H, 1), t);
            if ((rc = km_h2d_small(c, twy, t.data(), t.size() * sizeof(cd)))) return rc;
            host_negpos(PX, ng);
            if ((rc = km_h2d_small(c, negx, ng.data(), ng.size() * sizeof(int)))) return rc;
            host_negpos(PY, ng);
            if ((rc = km_h2d_small(c, negy, ng.data(), ng.size() * sizeof(int)))) return rc;
            c->f64_h = H; c->f64_w = W;
        }
    }
    blue_tabs BX, BY;
    if (PX.blue && (rc = blue_prepare(c, WS_F64_BLUEX, PX, &BX))) return rc;
    if (PY.blue && (rc = blue_prepare(c, WS_F64_BLUEY, PY, &BY))) return rc;
    cd *zt = nullptr;                                            // the transposed plane of a Bluestein column pass
    if (PY.blue) {
        zt = (cd *)km_ws(c, WS_FFT_B, n * sizeof(cd));
        if (!zt) return KM_E_NOMEM;
    }

    switch (dtype) {
    case KM_U8: rc = launch_pack<uint8_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
    case KM_U16: rc = launch_pack<uint16_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
    case KM_I16: rc = launch_pack<int16_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
    case KM_F32: rc = launch_pack<float>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
    default: return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
    }
    if (rc) return rc;

    auto along_rows = [&](bool inverse) -> int {
        if (W <= 1) return KM_OK;
        if (PX.blue) return blue_rows(c, z, H, PX, BX, inverse);
        return fft_rows_levels(c, z, twx, PX, inverse, H, W);
    };
    auto along_cols = [&](bool inverse) -> int {
        if (H <= 1) return KM_OK;
        if (!PY.blue) return fft_cols_levels(c, z, twy, PY, inverse, H, W);
        int r = transpose(c, z, zt, H, W);
        if (!r) r = blue_rows(c, zt, W, PY, BY, inverse);
        if (!r) r = transpose(c, zt, z, W, H);
        return r;
    };
    if ((rc = along_rows(false))) return rc;
    if ((rc = along_cols(false))) return rc;
    {
        const dim3 g((unsigned)std::min((W + 255) / 256, 64), (unsigned)std::min(H, 8192));
        f64_cross_kernel<<<g, 256, 0, c->stream>>>(z, negx, negy, H, W);
        KM_LAUNCH_CHECK(c);
    }
    if ((rc = along_cols(true))) return rc;
    if ((rc = along_rows(true))) return rc;

    unsigned long long *keys = &sc->argmax_key;                  // max bits
    unsigned long long *idx = (unsigned long long *)&sc->valid;  // reused as the first-index slot
    KM_HIP(c, hipMemsetAsync(keys, 0, sizeof(unsigned long long), c->stream));
    KM_HIP(c, hipMemsetAsync(idx, 0xff, sizeof(unsigned long long), c->stream));
    f64_absmax_kernel<<<2048, 256, 0, c->stream>>>(z, n, keys);
    KM_LAUNCH_CHECK(c);
    f64_first_index_kernel<<<2048, 256, 0, c->stream>>>(z, n, keys, idx);
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
