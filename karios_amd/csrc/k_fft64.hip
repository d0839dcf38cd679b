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
//       group of F64_KB = 4 output pairs per wavefront (X_k, X_{p-k} share the folded inputs x_j +- x_{p-j}; 8 wavefronts per tile), coefficients through the scalar path.
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

#define F64_KB 4              // output pairs per wavefront pass of the prime kernel
#define F64_PR_NW 8           // wavefronts per workgroup of the prime kernel: 8 x 4 pairs cover h <= 32 in one pass (p = 61: h = 30), and a
                              // wavefront fills its 8 of the 61 tile rows in ONE batch of loads (4 wavefronts x 8 pairs before: column level 0.97 -> 0.84 ms, inverse 0.70 -> 0.61; 16 x 2: slower)

struct lvl_args {
    cd *data;
    const cd *tw;                    // exp(-2 pi i j / N), j < N  (N = length of the dimension)
    const cd *ltw;                   // inter-level twiddles of this level, laid out like a block of the data: ltw[k R + r] = W_{n R}^(r k)
    const cd *ptab;                  // prime level: (cos, -sin)(2 pi j k / p) at [(j - 1) (h + 7) + k - 1], j = 1 .. h + 2, k = 1 .. h + 7 (h = (p - 1) / 2)
    long long se, sb, s_blk, s_r;    // strides in elements: between the elements of a transform, between the transforms of a tile, of `a`
    int n;                           // transform length
    int T, logT;                     // transforms per workgroup (a power of two in the smooth kernel unless `contiguous`)
    int AR;                          // a -> (a / AR) * s_blk + (a % AR) * s_r
    int B, tiles_b;                  // transforms along b, tiles of T
    int tw_mode;                     // inter-level twiddle W_{n R}^(q k):  0 none, 1 q = b, 2 q = a % AR
    int ltw_R;                       // R of this level
    int N;
    int inverse;
    int contiguous;                  // the transforms of a tile are contiguous in memory one after the other (se == 1, sb == n)
    int nst;
    int radix[F64_MAX_STAGES];
    // first forward level along the rows: the plane z = a + i b is read from the two images instead (pixel (y, x); no pack pass)
    const void *img_a, *img_b;
    long long img_sa, img_sb;
    // last inverse level: the outputs are the correlation surface - every workgroup reports the largest |cc| it wrote (bit pattern)
    // and the first flat index that attains it: best[2 wg], best[2 wg + 1]  (two passes over the plane are gone)
    unsigned long long *best;
    // the inverse along the rows runs on PAIRS of image rows (the correlation surface of two real images is real): row 2 j of the
    // plane holds S = Q_(2j) + i Q_(2j+1) - formed by the first of these levels while it loads (pair_w = distance of the partner row,
    // pair_h = number of image rows; 0: off) - and the last level reports |re| for row 2 j and |im| for row 2 j + 1 (best_pair_w)
    int pair_w, pair_h, best_pair_w;
    // ... and on the Hermitian HALF of the columns (Q(y, -kx) = conj Q(y, kx)): the inverse column levels skip the tiles that hold
    // no column with kx <= W / 2 (tile_mask, one byte per tile of T columns), the first row level reads the pairs from the plane
    // `pair_src` (pitch pair_w), completing the columns kx > W / 2 from their mirror positions (srcx), and writes PACKED rows
    // (pitch = row length) into `data`; the last level then adds best_row_add per packed row to get the image row's flat index
    const unsigned char *tile_mask;
    const cd *pair_src;
    const int *srcx;                 // column px of a pair comes from position srcx[px] (>= 0: as it is) or ~srcx[px] (< 0: conjugated)
    int pair_hstart;                 // elements >= pair_hstart of every transform are mirrored ones: visited in reverse, so that the reads ascend (-1: none)
    int best_row_add;
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
    } else if constexpr (R == 6) {
        // 2 x 3: X[2 k] = DFT3(v_j + v_{j+3})[k],  X[2 k + 1] = DFT3((v_j - v_{j+3}) W_6^j)[k]
        cd e[3], o[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { e[j] = c_add(v[j], v[j + 3]); o[j] = c_sub(v[j], v[j + 3]); }
        o[1] = c_mul(o[1], make_double2(0.5, -0.86602540378443864676));
        o[2] = c_mul(o[2], make_double2(-0.5, -0.86602540378443864676));
        dft_r<3>(e);
        dft_r<3>(o);
#pragma unroll
        for (int k = 0; k < 3; k++) { v[2 * k] = e[k]; v[2 * k + 1] = o[k]; }
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

// one Stockham butterfly: radix R, Ns = product of the radices done; element i of transform t at buf[i * lsi + toff].  In two halves
// with the workgroup's barrier between them - every butterfly of a stage is read before any is written, so the stage runs in ONE buffer
// (half the LDS of a source / destination pair: twice the resident workgroups to hide a tile's loads and stores behind).
template <int R>
__device__ __forceinline__ int bfly_read(const cd *__restrict__ buf, cd (&v)[R], int b, int nb, int Ns, unsigned magic, int lsi, int toff,
                                         const cd *__restrict__ twl, int twstep)
{
    const int q = Ns == 1 ? b : (int)__umulhi((unsigned)b, magic), k = b - q * Ns;     // b / Ns, b % Ns (exact: b Ns < 2^32)
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = buf[(b + r * nb) * lsi + toff];
    if (Ns > 1) {
#pragma unroll
        for (int r = 1; r < R; r++) v[r] = c_mul(v[r], twl[r * k * twstep]);
    }
    dft_r<R>(v);
    return (q * Ns * R + k) * lsi + toff;
}
template <int R> __device__ __forceinline__ void bfly_write(cd *__restrict__ buf, const cd (&v)[R], int at, int Ns, int lsi)
{
#pragma unroll
    for (int r = 0; r < R; r++) buf[at + r * Ns * lsi] = v[r];
}

__device__ __forceinline__ long long tile_base(const lvl_args &A, int a, int b0)
{
    return (long long)(a / A.AR) * A.s_blk + (long long)(a % A.AR) * A.s_r + (long long)b0 * A.sb;
}

// IMG: pixel type of the images the first forward level reads (KM_U8 / KM_U16 / KM_I16 / KM_F32), or -1: the plane (a kernel per
// type: a run-time switch per pixel put every load of the unrolled batch into a basic block of its own)
template <int IMG> __device__ __forceinline__ double px_f64(const void *p, size_t off)
{
    if constexpr (IMG == KM_U8) return (double)((const uint8_t *)p)[off];
    else if constexpr (IMG == KM_U16) return (double)((const uint16_t *)p)[off];
    else if constexpr (IMG == KM_I16) return (double)((const int16_t *)p)[off];
    else return (double)((const float *)p)[off];
}
template <int IMG> __device__ __forceinline__ float px_f32(const void *p, size_t off)
{
    if constexpr (IMG == KM_U8) return (float)((const uint8_t *)p)[off];
    else if constexpr (IMG == KM_U16) return (float)((const uint16_t *)p)[off];
    else if constexpr (IMG == KM_I16) return (float)((const int16_t *)p)[off];
    else return ((const float *)p)[off];
}
template <int IMG> __device__ __forceinline__ cd px_pair(const lvl_args &A, int y, int x)
{
    return make_double2(px_f64<IMG>(A.img_a, (size_t)y * (size_t)A.img_sa + (size_t)x), px_f64<IMG>(A.img_b, (size_t)y * (size_t)A.img_sb + (size_t)x));
}

// (largest |cc| as bits, first flat index): a thread's running best, the wavefront's, the workgroup's
struct best_t {
    unsigned long long bits = 0, idx = ~0ull;
    __device__ __forceinline__ void see(cd v, unsigned long long flat)
    {
        const double m = hypot(v.x, v.y);
        if (m == m) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(m);
            if (b > bits || (b == bits && flat < idx)) { bits = b; idx = flat; }
        }
    }
    __device__ __forceinline__ void see_abs(double v, unsigned long long flat)
    {
        const double m = fabs(v);
        if (m == m) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(m);
            if (b > bits || (b == bits && flat < idx)) { bits = b; idx = flat; }
        }
    }
    __device__ __forceinline__ void merge(unsigned long long ob, unsigned long long oi)
    {
        if (ob > bits || (ob == bits && oi < idx)) { bits = ob; idx = oi; }
    }
};
__device__ __forceinline__ void best_publish(best_t bt, unsigned long long *out, unsigned long long *s_b /* 2 words of LDS per wavefront */, int nwaves = 4)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(bt.bits, o), oi = __shfl_xor(bt.idx, o);
        bt.merge(ob, oi);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_b[2 * wave] = bt.bits; s_b[2 * wave + 1] = bt.idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        best_t r;
        for (int w = 0; w < nwaves; w++) r.merge(s_b[2 * w], s_b[2 * w + 1]);
        out[2 * (size_t)blockIdx.x] = r.bits; out[2 * (size_t)blockIdx.x + 1] = r.idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------- smooth level
// n T <= 2048 elements per tile: at most 8 per thread - all of a thread's loads are in flight before the first one is consumed
#define F64_SM_PER_THREAD 8
template <int IMG> __global__ __launch_bounds__(256) void f64_smooth_kernel(const lvl_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    __shared__ unsigned long long s_b[8];
    const int n = A.n, T = A.T, logT = A.logT;
    cd *buf0 = (cd *)smem64, *twl = buf0 + n * T;                              // twl[j] = exp(-2 pi i j / n)
    const int tid = threadIdx.x;
    const int wg = blockIdx.x, a = wg / A.tiles_b, tb = wg - a * A.tiles_b, b0 = tb * T, nt = min(T, A.B - b0);
    if (A.tile_mask && !A.tile_mask[tb]) return;                               // (a tile of columns nobody reads: uniform, before any barrier)
    const long long base = tile_base(A, a, b0);
    const int qa = a % A.AR;
    cd *__restrict__ data = A.data;
    const cd *__restrict__ tw = A.tw;
    const cd *__restrict__ ltw = A.ltw;
    const bool contig = A.contiguous != 0;
    const int lsi = contig ? 1 : T;
    const unsigned magic_n = (unsigned)((0x100000000ull + (unsigned)n - 1) / (unsigned)n);     // e / n for e < 2048 (n >= 2)
    // element e of the tile (= its LDS slot): strided tiles are [i][t] (t = e % T), contiguous tiles [t][i] (the memory order)
    auto split = [&](int e, int &i, int &t) {
        if (contig) { t = (int)__umulhi((unsigned)e, magic_n); i = e - t * n; }
        else { t = e & (T - 1); i = e >> logT; }
    };
    auto addr = [&](int e, int i, int t) { return contig ? base + e : base + (long long)i * A.se + t; };
    const bool has2 = (A.pair_w || A.best_pair_w) && 2 * (a / A.AR) + 1 < A.pair_h;   // (pairs of image rows: does this one have a second member?)
    {
        cd v[F64_SM_PER_THREAD], w[F64_SM_PER_THREAD];
        int es[F64_SM_PER_THREAD];                                             // LDS slot of each loaded element
        const bool pre = A.inverse && A.tw_mode;
#pragma unroll
        for (int u = 0; u < F64_SM_PER_THREAD; u++) {
            int e = tid + 256 * u;
            int i, t;
            split(e, i, t);
            if (A.pair_src && A.pair_hstart >= 0 && i >= A.pair_hstart) { i = A.pair_hstart + n - 1 - i; e = t * n + i; }   // (contiguous tiles: a bijection inside the transform)
            es[u] = e;
            v[u] = make_double2(0.0, 0.0);
            w[u] = make_double2(1.0, 0.0);
            if (e < n * T && t < nt) {
                if constexpr (IMG >= 0) v[u] = contig ? px_pair<IMG>(A, b0 + t, i) : px_pair<IMG>(A, a, b0 + t + i * (int)A.se);
                else {
                    if (A.pair_src) {
                        // packed pairs from the half plane (contiguous tiles only): column px of image rows 2 j, 2 j + 1
                        const int sx = A.srcx[b0 * n + e];
                        const bool low = sx >= 0;
                        const long long sp = (long long)(a / A.AR) * 2 * A.pair_w + (low ? sx : ~sx);
                        const cd q1 = A.pair_src[sp], q2 = has2 ? A.pair_src[sp + A.pair_w] : make_double2(0.0, 0.0);
                        v[u] = low ? make_double2(q1.x - q2.y, q1.y + q2.x)          // Q_y1 + i Q_y2
                                   : make_double2(q1.x + q2.y, q2.x - q1.y);         // conj Q_y1(-kx) + i conj Q_y2(-kx)
                    } else {
                        v[u] = data[addr(e, i, t)];
                        if (A.pair_w) {
                            const cd q2 = has2 ? data[addr(e, i, t) + A.pair_w] : make_double2(0.0, 0.0);
                            v[u] = make_double2(v[u].x - q2.y, v[u].y + q2.x);        // Q_y1 + i Q_y2
                        }
                    }
                }
                if (pre) w[u] = ltw[(size_t)i * (size_t)A.ltw_R + (size_t)(A.tw_mode == 1 ? b0 + t : qa)];
            }
        }
        for (int j = tid; j < n; j += 256) twl[j] = tw[(size_t)j * (size_t)(A.N / n)];
#pragma unroll
        for (int u = 0; u < F64_SM_PER_THREAD; u++) {
            const int e = es[u];
            if (tid + 256 * u < n * T) {
                cd x = v[u];
                if (A.inverse) {
                    x.y = -x.y;                                                // inverse DFT = conj . DFT . conj
                    if (pre) x = c_mul(x, w[u]);                               // conj(x conj w) = conj(x) w
                }
                buf0[e] = x;
            }
        }
    }
    __syncthreads();

    int Ns = 1;
    for (int s = 0; s < A.nst; s++) {
        const int R = A.radix[s], nb = n / R, twstep = n / (Ns * R);
        const unsigned magic = Ns > 1 ? (unsigned)((0x100000000ull + (unsigned)Ns - 1) / (unsigned)Ns) : 0u;
        auto run = [&](auto rc) {
            constexpr int RR = decltype(rc)::value;
            constexpr int MAXB = (F64_SMOOTH_MAX / RR + 255) / 256;            // butterflies of a stage per thread
            cd v[MAXB][RR];
            int at[MAXB];
            // (contiguous tiles: flat over the tile like the strided form - one wavefront per transform left 64 - n / R lanes idle)
            const unsigned magic_nb = (unsigned)((0x100000000ull + (unsigned)nb - 1) / (unsigned)nb);
            const int count = contig ? nb * nt : nb * T;
#pragma unroll
            for (int k = 0; k < MAXB; k++) {
                const int idx = tid + 256 * k;
                at[k] = -1;
                if (idx < count) {
                    if (contig) {
                        const int t = nb > 1 ? (int)__umulhi((unsigned)idx, magic_nb) : idx;
                        at[k] = bfly_read<RR>(buf0, v[k], idx - t * nb, nb, Ns, magic, 1, t * n, twl, twstep);
                    } else at[k] = bfly_read<RR>(buf0, v[k], idx >> logT, nb, Ns, magic, lsi, idx & (T - 1), twl, twstep);
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MAXB; k++)
                if (at[k] >= 0) bfly_write<RR>(buf0, v[k], at[k], Ns, lsi);
        };
        switch (R) {
        case 7: run(std::integral_constant<int, 7>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        default: run(std::integral_constant<int, 2>{}); break;
        }
        __syncthreads();
        Ns *= R;
    }
    cd *src = buf0;

    best_t bt;
    {
        const bool post = !A.inverse && A.tw_mode;
        cd w[F64_SM_PER_THREAD];
        if (post) {
#pragma unroll
            for (int u = 0; u < F64_SM_PER_THREAD; u++) {
                const int e = tid + 256 * u;
                int i, t;
                split(e, i, t);
                w[u] = make_double2(1.0, 0.0);
                if (e < n * T && t < nt) w[u] = ltw[(size_t)i * (size_t)A.ltw_R + (size_t)(A.tw_mode == 1 ? b0 + t : qa)];
            }
        }
#pragma unroll
        for (int u = 0; u < F64_SM_PER_THREAD; u++) {
            const int e = tid + 256 * u;
            int i, t;
            split(e, i, t);
            if (e < n * T && t < nt) {
                cd x = src[e];
                if (A.inverse) x.y = -x.y;
                else if (post) x = c_mul(x, w[u]);
                const long long o = addr(e, i, t);
                data[o] = x;
                if (A.best) {
                    if (A.best_pair_w) {
                        const unsigned long long f1 = (unsigned long long)o + (unsigned long long)(a / A.AR) * (unsigned long long)A.best_row_add;
                        bt.see_abs(x.x, f1);
                        if (has2) bt.see_abs(x.y, f1 + A.best_pair_w);
                    } else bt.see(x, (unsigned long long)o);
                }
            }
        }
    }
    if (A.best) best_publish(bt, A.best, s_b);
}

// ---------------------------------------------------------------------------------------------------------------- prime level
#define F64_PR_BATCH 8     // rows of the tile a wavefront has in flight at once while it fills the LDS (16 measured slower on the plane: 1.06 against 0.97 ms)
// The level that reads the IMAGES keeps its tile as float pairs (every pixel type is exact in float32): 8 bytes per element instead of
// 16, five workgroups per CU instead of two - that level was bound by the latency of its loads (ablation, round 5: 0.80 of its 1.48 ms
// went when the loads were removed, 0.22 when the arithmetic was).
template <int IMG> struct pr_lds { typedef cd type; };
template <> struct pr_lds<KM_U8> { typedef float2 type; };
template <> struct pr_lds<KM_U16> { typedef float2 type; };
template <> struct pr_lds<KM_I16> { typedef float2 type; };
template <> struct pr_lds<KM_F32> { typedef float2 type; };
__device__ __forceinline__ cd pr_value(cd v) { return v; }
__device__ __forceinline__ cd pr_value(float2 v) { return make_double2((double)v.x, (double)v.y); }

template <int IMG> __global__ __launch_bounds__(64 * F64_PR_NW) void f64_prime_kernel(const lvl_args A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    __shared__ unsigned long long s_b[2 * F64_PR_NW];
    typedef typename pr_lds<IMG>::type sm_t;
    sm_t *sm = (sm_t *)smem64;                                                 // [i][t], t < T <= 64
    const int p = A.n, T = A.T;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x, a = wg / A.tiles_b, tb = wg - a * A.tiles_b, b0 = tb * T, nt = min(T, A.B - b0);
    if (A.tile_mask && !A.tile_mask[tb]) return;
    const long long base = tile_base(A, a, b0);
    cd *__restrict__ data = A.data;
    const cd *__restrict__ ltw = A.ltw;
    const int q = A.tw_mode == 1 ? b0 + lane : a % A.AR;                       // (tw_mode 2: the same for the whole workgroup)
    const long long loff = (long long)lane * A.sb;
    const bool pre = A.inverse && A.tw_mode, contig = A.contiguous != 0;
    const bool has2 = (A.pair_w || A.best_pair_w) && 2 * (a / A.AR) + 1 < A.pair_h;
    // 8- and 16-bit pixels: a row of the tile is nt consecutive pixels of an image row - read as whole DWORDS, a lane taking 2 or 4
    // neighbouring transforms' pixels of both images and writing them as float pairs (sub-dword global loads go through the address
    // unit one lane at a time: the 2-byte gathers of this level cost 0.8 of its 1.49 ms).  Needs every row start 4-byte aligned.
    bool img_dwords = false;
    if constexpr (IMG == KM_U8 || IMG == KM_U16 || IMG == KM_I16) {
        typedef typename std::conditional<IMG == KM_U8, uint8_t, typename std::conditional<IMG == KM_U16, uint16_t, int16_t>::type>::type px_t;
        constexpr int ES = (int)sizeof(px_t), PPD = 4 / ES;
        img_dwords = !contig && nt % PPD == 0 && nt >= 2 * PPD && T % PPD == 0 && (int)A.se % PPD == 0 && A.img_sa % PPD == 0 && A.img_sb % PPD == 0 &&
                     (((unsigned long long)A.img_a | (unsigned long long)A.img_b) & 3ull) == 0;
        if (img_dwords) {
            const int dpr = nt / PPD, nrows = (p - wave + F64_PR_NW - 1) / F64_PR_NW, total = nrows * dpr;      // this wavefront's rows: i = wave + NW u
            const unsigned magic = (unsigned)((0x100000000ull + (unsigned)dpr - 1) / (unsigned)dpr);   // f / dpr for f < 4096
            const px_t *ia = (const px_t *)A.img_a + (size_t)a * (size_t)A.img_sa + (size_t)b0;
            const px_t *ib = (const px_t *)A.img_b + (size_t)a * (size_t)A.img_sb + (size_t)b0;
            for (int f0 = 0; f0 < total; f0 += 64 * F64_PR_BATCH) {
                uint32_t da[F64_PR_BATCH], db[F64_PR_BATCH];
                int slot[F64_PR_BATCH];
#pragma unroll
                for (int u = 0; u < F64_PR_BATCH; u++) {
                    const int f = f0 + 64 * u + lane;
                    const int r = (int)__umulhi((unsigned)f, magic), d = f - r * dpr, i = wave + F64_PR_NW * r;
                    slot[u] = -1; da[u] = db[u] = 0u;
                    if (f < total) {
                        const size_t off = (size_t)i * (size_t)A.se + (size_t)(d * PPD);
                        da[u] = *(const uint32_t *)(ia + off);
                        db[u] = *(const uint32_t *)(ib + off);
                        slot[u] = i * T + d * PPD;
                    }
                }
#pragma unroll
                for (int u = 0; u < F64_PR_BATCH; u++) {
                    if (slot[u] < 0) continue;
#pragma unroll
                    for (int k = 0; k < PPD; k++) {
                        const px_t pa = (px_t)(da[u] >> (8 * ES * k)), pb = (px_t)(db[u] >> (8 * ES * k));
                        sm[slot[u] + k] = make_float2((float)pa, (float)pb);
                    }
                }
            }
            if (nt < T)                                                         // (the lanes behind a short last tile transform zeros)
                for (int e = tid; e < p * (T - nt); e += 64 * F64_PR_NW) { const int i = e / (T - nt); sm[i * T + nt + (e - i * (T - nt))] = make_float2(0.f, 0.f); }
        }
    }
    for (int u0 = 0; wave + F64_PR_NW * u0 < p; u0 += F64_PR_BATCH) {
        if constexpr (IMG >= 0) {
            // (the first forward level: no conjugation, no twiddle in front)
            if (img_dwords) break;
            float2 v[F64_PR_BATCH];
#pragma unroll
            for (int u = 0; u < F64_PR_BATCH; u++) {
                const int i = wave + F64_PR_NW * (u0 + u);
                v[u] = make_float2(0.f, 0.f);
                if (i < p && lane < nt) {
                    const size_t y = contig ? (size_t)(b0 + lane) : (size_t)a, x = contig ? (size_t)i : (size_t)(b0 + lane + i * (int)A.se);
                    v[u] = make_float2(px_f32<IMG>(A.img_a, y * (size_t)A.img_sa + x), px_f32<IMG>(A.img_b, y * (size_t)A.img_sb + x));
                }
            }
#pragma unroll
            for (int u = 0; u < F64_PR_BATCH; u++) {
                const int i = wave + F64_PR_NW * (u0 + u);
                if (i < p && lane < T) sm[i * T + lane] = v[u];
            }
        } else {
        cd v[F64_PR_BATCH], w[F64_PR_BATCH];
#pragma unroll
        for (int u = 0; u < F64_PR_BATCH; u++) {
            const int i = wave + F64_PR_NW * (u0 + u);
            v[u] = make_double2(0.0, 0.0);
            w[u] = make_double2(1.0, 0.0);
            if (i < p && lane < nt) {
                if (A.pair_src) {
                    const int sx = A.srcx[(b0 + lane) * p + i];                    // (contiguous tiles: transform t = elements t p .. t p + p - 1)
                    const bool low = sx >= 0;
                    const long long sp = (long long)(a / A.AR) * 2 * A.pair_w + (low ? sx : ~sx);
                    const cd q1 = A.pair_src[sp], q2 = has2 ? A.pair_src[sp + A.pair_w] : make_double2(0.0, 0.0);
                    v[u] = low ? make_double2(q1.x - q2.y, q1.y + q2.x) : make_double2(q1.x + q2.y, q2.x - q1.y);
                } else {
                    v[u] = data[base + (long long)i * A.se + loff];
                    if (A.pair_w) {
                        const cd q2 = has2 ? data[base + (long long)i * A.se + loff + A.pair_w] : make_double2(0.0, 0.0);
                        v[u] = make_double2(v[u].x - q2.y, v[u].y + q2.x);
                    }
                }
                if (pre) w[u] = ltw[(size_t)i * (size_t)A.ltw_R + (size_t)q];
            }
        }
#pragma unroll
        for (int u = 0; u < F64_PR_BATCH; u++) {
            const int i = wave + F64_PR_NW * (u0 + u);
            if (i < p && lane < T) {
                cd x = v[u];
                if (A.inverse) {
                    x.y = -x.y;
                    if (pre) x = c_mul(x, w[u]);
                }
                sm[i * T + lane] = x;
            }
        }
        }
    }
    __syncthreads();
    const int h = (p - 1) / 2;
    const int l = min(lane, T - 1);
    best_t bt;
    const bool post = !A.inverse && A.tw_mode;
    auto put = [&](int k, cd v, cd w) {
        if (lane >= nt) return;
        if (A.inverse) v.y = -v.y;
        else if (post) v = c_mul(v, w);
        const long long o = base + (long long)k * A.se + loff;
        data[o] = v;
        if (A.best) {
            if (A.best_pair_w) {
                const unsigned long long f1 = (unsigned long long)o + (unsigned long long)(a / A.AR) * (unsigned long long)A.best_row_add;
                bt.see_abs(v.x, f1);
                if (has2) bt.see_abs(v.y, f1 + A.best_pair_w);
            } else bt.see(v, (unsigned long long)o);
        }
    };
    auto ltw_of = [&](int k) {                                                 // W_{n R}^(q k): one coalesced row of the table per k
        return post && lane < nt ? ltw[(size_t)k * (size_t)A.ltw_R + (size_t)q] : make_double2(1.0, 0.0);
    };
    // the coefficient addresses are the same for every lane: through the constant address space they become scalar loads; the
    // table holds one row per j with k running along it, so that the F64_KB coefficient pairs of a step are 64 consecutive bytes
    // (one wide scalar load and one pointer step; an index (j k) mod p stepped and wrapped per coefficient cost 7 scalar
    // instructions each - more issue slots than the FMAs they feed)
    typedef const __attribute__((address_space(4))) double *scalar_f64_ptr;
    const scalar_f64_ptr ptab = (scalar_f64_ptr)(unsigned long long)A.ptab;
    const size_t prow = 2 * (size_t)(h + 7);                                   // doubles per table row
    for (int g0 = wave * F64_KB; g0 < h; g0 += F64_PR_NW * F64_KB) {                   // (uniform per wavefront)
        cd C[F64_KB], S[F64_KB], w[F64_KB];
        const scalar_f64_ptr col = ptab + 2 * (size_t)g0;                      // k = g0 + 1 .. g0 + 8
#pragma unroll
        for (int kk = 0; kk < F64_KB; kk++) {
            C[kk] = S[kk] = make_double2(0.0, 0.0);
            w[kk] = make_double2(col[2 * kk], col[2 * kk + 1]);                // j = 1
        }
        const cd x0 = pr_value(sm[l]);
        cd sum = x0;
        cd xa = pr_value(sm[T + l]), xb = pr_value(sm[(p - 1) * T + l]);
        // two steps of j per trip, the inputs and coefficients of the next step travelling while the current one is accumulated
        // (registers alternate: no copies; a wavefront issues one instruction per four cycles whatever its kind, so every
        // instruction that is not one of the 4 F64_KB FMAs of a step counts)
        auto fetch = [&](int j, cd &na, cd &nb, cd *nw) {                      // step j (its table column is j - 1)
            na = pr_value(sm[j * T + l]); nb = pr_value(sm[(p - j) * T + l]);
#pragma unroll
            for (int kk = 0; kk < F64_KB; kk++) nw[kk] = make_double2(col[(size_t)(j - 1) * prow + 2 * kk], col[(size_t)(j - 1) * prow + 2 * kk + 1]);
        };
        auto accumulate = [&](const cd &pa, const cd &pb, const cd *cw) {
            const cd fa = c_add(pa, pb), fb = c_sub(pa, pb);
            sum = c_add(sum, fa);
#pragma unroll
            for (int kk = 0; kk < F64_KB; kk++) {
                C[kk].x = fma(fa.x, cw[kk].x, C[kk].x); C[kk].y = fma(fa.y, cw[kk].x, C[kk].y);
                S[kk].x = fma(fb.x, cw[kk].y, S[kk].x); S[kk].y = fma(fb.y, cw[kk].y, S[kk].y);    // S = - sum (x_j - x_{p-j}) sin
            }
        };
        const int hh = h;
        for (int j = 1; j <= hh; j += 2) {
            cd ya, yb, w1[F64_KB];
            fetch(j + 1, ya, yb, w1);                                          // (j + 1 = h + 1: fetched, never used)
            __builtin_amdgcn_sched_barrier(0);                                 // the loads stay up here: their latency is the FMAs below
            accumulate(xa, xb, w);
            __builtin_amdgcn_sched_barrier(0);                                 // ... and nothing that waits for them moves in front of the FMAs
            __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0) HERE: LDS and scalar loads share the counter and return
            __builtin_amdgcn_sched_barrier(0);                                 // out of order between them - a wait for the LDS data placed behind the next scalar loads would wait for those too
            if (j + 1 > hh) break;
            fetch(j + 2, xa, xb, w);
            __builtin_amdgcn_sched_barrier(0);
            accumulate(ya, yb, w1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int nk = min(F64_KB, h - g0);
        // the inter-level twiddles of this group's outputs: all loads first
        cd wk[F64_KB], wm[F64_KB];
#pragma unroll
        for (int kk = 0; kk < F64_KB; kk++) {
            const int k = min(g0 + kk + 1, h);
            wk[kk] = ltw_of(k);
            wm[kk] = ltw_of(p - k);
        }
#pragma unroll
        for (int kk = 0; kk < F64_KB; kk++) {
            if (kk < nk) {
                const int k = g0 + kk + 1;
                const double cr = x0.x + C[kk].x, ci = x0.y + C[kk].y;
                put(k, make_double2(cr - S[kk].y, ci + S[kk].x), wk[kk]);      // x0 + C + i S
                put(p - k, make_double2(cr + S[kk].y, ci - S[kk].x), wm[kk]);  // x0 + C - i S
            }
        }
        if (g0 == 0) put(0, sum, make_double2(1.0, 0.0));                      // (k = 0: W^0)
    }
    if (A.best) best_publish(bt, A.best, s_b, F64_PR_NW);
}

// every workgroup's (bits, first index) -> the plane's: out[0] = first flat index of the largest |cc| (~0: none, e.g. all NaN)
__global__ __launch_bounds__(1024) void f64_best_reduce_kernel(const unsigned long long *__restrict__ best, size_t count, unsigned long long *out)
{
    __shared__ unsigned long long s_r[32];
    best_t bt;
    for (size_t i = threadIdx.x; i < count; i += 1024) bt.merge(best[2 * i], best[2 * i + 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(bt.bits, o), oi = __shfl_xor(bt.idx, o);
        bt.merge(ob, oi);
    }
    if ((threadIdx.x & 63) == 0) { s_r[2 * (threadIdx.x >> 6)] = bt.bits; s_r[2 * (threadIdx.x >> 6) + 1] = bt.idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        best_t r;
        for (int w = 0; w < 16; w++) r.merge(s_r[2 * w], s_r[2 * w + 1]);
        out[0] = r.idx;
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
// `keepx` (Hermitian half of the inverse): only the columns it marks are read afterwards - the others are not written (0.97 of 3.9 GB).
__global__ __launch_bounds__(256) void f64_cross_kernel(cd *__restrict__ Z, const int *__restrict__ negx, const int *__restrict__ negy, int H, int W,
                                                        const unsigned char *__restrict__ keepx)
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
            if (!keepx || keepx[px]) Z[i0] = P;
            if (i1 != i0 && (!keepx || keepx[nx])) Z[i1] = make_double2(P.x, -P.y);
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
        const void *kernels[] = {(const void *)f64_smooth_kernel<-1>, (const void *)f64_smooth_kernel<KM_U8>, (const void *)f64_smooth_kernel<KM_U16>,
                                 (const void *)f64_smooth_kernel<KM_I16>, (const void *)f64_smooth_kernel<KM_F32>,
                                 (const void *)f64_prime_kernel<-1>, (const void *)f64_prime_kernel<KM_U8>, (const void *)f64_prime_kernel<KM_U16>,
                                 (const void *)f64_prime_kernel<KM_I16>, (const void *)f64_prime_kernel<KM_F32>};
        for (const void *k : kernels) KM_HIP(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
        opted |= bit;
    }
    return KM_OK;
}

// device tables of one dimension: [exp(-2 pi i j / N), j < N] [inter-level twiddles of every level with R > 1, each laid out
// like a block of the data: k R + r -> W_{n R}^(r k)]
struct dim_tabs {
    const cd *tw = nullptr;
    const cd *ltw[16] = {nullptr};
    const cd *ptab[16] = {nullptr};      // prime levels: the coefficient rows of f64_prime_kernel
};
inline size_t ptab_len(int p) { const size_t h = (size_t)(p - 1) / 2; return (h + 7) * (h + 2); }
size_t host_dim_tables(const dimplan &P, std::vector<cd> &h, size_t ltw_off[16], size_t ptab_off[16])
{
    const int N = std::max(P.N, 1);
    host_twiddles(N, h);
    for (size_t l = 0; l < P.lv.size() && l < 16; l++) {
        const lvl &L = P.lv[l];
        ltw_off[l] = ptab_off[l] = 0;
        if (L.kind == 1) {
            const int p = L.n, hh = (p - 1) / 2;
            ptab_off[l] = h.size();
            h.resize(h.size() + ptab_len(p));
            cd *t = &h[ptab_off[l]];
            for (int k = 1; k <= hh + 7; k++)
                for (int j = 1; j <= hh + 2; j++) t[(size_t)(j - 1) * (hh + 7) + (k - 1)] = h[(size_t)(((long long)j * k) % p) * (size_t)(N / p)];
        }
        if (L.R <= 1) continue;
        ltw_off[l] = h.size();
        const size_t Nl = (size_t)L.n * L.R, scale = (size_t)N / Nl;
        h.resize(h.size() + Nl);
        cd *t = &h[ltw_off[l]];
        for (int k = 0; k < L.n; k++)
            for (int r = 0; r < L.R; r++) t[(size_t)k * L.R + r] = h[(size_t)r * k * scale];
    }
    return h.size();
}
void bind_dim_tables(const dimplan &P, const cd *d, const size_t ltw_off[16], const size_t ptab_off[16], dim_tabs *T)
{
    T->tw = d;
    for (size_t l = 0; l < P.lv.size() && l < 16; l++) {
        T->ltw[l] = ltw_off[l] ? d + ltw_off[l] : nullptr;
        T->ptab[l] = ptab_off[l] ? d + ptab_off[l] : nullptr;
    }
}
// the offsets alone (tables already on the device)
void dim_table_offsets(const dimplan &P, size_t ltw_off[16], size_t ptab_off[16], size_t *total)
{
    size_t n = (size_t)std::max(P.N, 1);
    for (size_t l = 0; l < P.lv.size() && l < 16; l++) {
        ltw_off[l] = ptab_off[l] = 0;
        if (P.lv[l].kind == 1) { ptab_off[l] = n; n += ptab_len(P.lv[l].n); }
        if (P.lv[l].R <= 1) continue;
        ltw_off[l] = n;
        n += (size_t)P.lv[l].n * P.lv[l].R;
    }
    *total = n;
}

struct lvl_extra {
    const void *img_a = nullptr, *img_b = nullptr;     // first forward level along the rows: read the images
    long long img_sa = 0, img_sb = 0;
    int img_dtype = 0;
    int pair_rows = 0;                                  // > 0: the plane's rows are PAIRS of image rows (inverse along the rows); = image rows
    bool pair_load = false;                             //   ... and this level forms them while it loads
    // Hermitian half of the columns (inverse only): column levels skip tiles without a needed column; row levels work on PACKED pairs
    const unsigned char *tile_mask = nullptr;           //   column level: one byte per tile of `tile_T` columns
    int tile_T = 0;
    bool packed = false;                                //   row level: `data` holds packed pairs (pitch = width) ...
    const cd *pair_src = nullptr;                       //   ... which this level (pair_load) reads from here, completing kx > W / 2 by symmetry
    const int *srcx = nullptr;
    int pair_hstart = -1;
    bool want_best = false;                             // last inverse level: report (largest |cc|, first index) per workgroup
    unsigned long long *best = nullptr;                 //   -> WS_FFT_TOP2, `best_count` entries of two words
    size_t best_count = 0;
};

// one level over a plane of `rows` rows of `width` elements (contiguous): along the rows (`cols` false, dimension length = width) or
// along the columns (dimension length = rows)
int run_level(km_ctx *c, cd *data, const dim_tabs &tabs, int N, const lvl &L, int level_index, bool inverse, bool cols, int rows, int width,
              lvl_extra *extra = nullptr)
{
    lvl_args A;
    A.data = data; A.tw = tabs.tw; A.ltw = tabs.ltw[level_index]; A.ptab = tabs.ptab[level_index]; A.ltw_R = L.R; A.n = L.n; A.N = N; A.inverse = inverse ? 1 : 0;
    A.img_a = A.img_b = nullptr; A.img_sa = A.img_sb = 0; A.best = nullptr;
    A.pair_w = A.pair_h = A.best_pair_w = A.best_row_add = 0;
    A.tile_mask = nullptr; A.pair_src = nullptr; A.srcx = nullptr; A.pair_hstart = -1;
    int img = -1;
    const bool paired = !cols && extra && extra->pair_rows > 0;     // rows of the plane = pairs of image rows at a pitch of two rows
    const long long Nl = (long long)L.n * L.R;
    long long nA;
    if (paired) {
        const int hp = (extra->pair_rows + 1) / 2;
        const long long pitch = extra->packed ? (long long)width : 2ll * width;     // in place in the rows 2 j, or packed rows
        A.pair_h = extra->pair_rows;
        if (L.R > 1) {
            A.AR = (int)(width / Nl); nA = (long long)hp * A.AR; A.s_blk = pitch; A.s_r = Nl;
            A.B = L.R; A.sb = 1; A.se = L.R; A.tw_mode = 1; A.contiguous = 0;
        } else {
            A.AR = 1; nA = hp; A.s_blk = pitch; A.s_r = 0;
            A.B = width / L.n; A.sb = L.n; A.se = 1; A.tw_mode = 0; A.contiguous = 1;
        }
        if (extra->pair_load) {
            A.pair_w = width;
            if (extra->packed) { A.pair_src = extra->pair_src; A.srcx = extra->srcx; A.pair_hstart = L.kind == 0 ? extra->pair_hstart : -1; }
        }
        if (extra->packed) A.best_row_add = width;
    } else if (!cols) {
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
    if (A.tw_mode && !A.ltw) return km_fail(c, KM_E_ARG, "phase correlation: level twiddles missing");
    A.nst = L.nst;
    for (int i = 0; i < F64_MAX_STAGES; i++) A.radix[i] = L.radix[i];
    size_t lds;
    if (L.kind == 1) {
        int tmax = std::min(64, std::max(1, 4096 / L.n));
        if (c->opt_f64_prime_t > 0) tmax = std::min(tmax, c->opt_f64_prime_t);
        const int tiles = (A.B + tmax - 1) / tmax;
        A.T = (A.B + tiles - 1) / tiles; A.logT = 0;
        if (cols && extra && extra->tile_mask && extra->tile_T > 0) A.T = extra->tile_T;   // (chosen with the mask: half_plane_tiles)
        A.tiles_b = (A.B + A.T - 1) / A.T;
        lds = (size_t)L.n * A.T * (extra && extra->img_a ? sizeof(float2) : sizeof(cd));   // (the level that reads the images: float pairs)
    } else if (A.contiguous) {
        int T = std::max(1, std::min(8, F64_SMOOTH_MAX / L.n));
        if (c->opt_f64_smooth_t > 0) T = std::min(T, c->opt_f64_smooth_t);
        T = std::min(T, A.B);
        A.T = T; A.logT = 0;
        A.tiles_b = (A.B + T - 1) / T;
        lds = ((size_t)L.n * T + L.n) * sizeof(cd);
    } else {
        int T = 8, lg = 3;
        while (T > 1 && (L.n * T > F64_SMOOTH_MAX || (c->opt_f64_smooth_t > 0 && T > c->opt_f64_smooth_t))) { T >>= 1; lg--; }
        while (T > 1 && (T >> 1) >= A.B) { T >>= 1; lg--; }
        A.T = T; A.logT = lg;
        A.tiles_b = (A.B + T - 1) / T;
        lds = ((size_t)L.n * T + L.n) * sizeof(cd);
    }
    const long long grid = nA * A.tiles_b;
    if (grid <= 0 || grid > 0x7fffffffll) return km_fail(c, KM_E_ARG, "phase correlation: plane too large");
    if (cols && extra && extra->tile_mask) {
        if (extra->tile_T != A.T) return km_fail(c, KM_E_INTERNAL, "phase correlation: tile mask of %d columns, tiles of %d", extra->tile_T, A.T);
        A.tile_mask = extra->tile_mask;
    }
    if (extra) {
        A.img_a = extra->img_a; A.img_b = extra->img_b; A.img_sa = extra->img_sa; A.img_sb = extra->img_sb;
        if (extra->img_a) img = extra->img_dtype;
        if (extra->want_best) {
            extra->best = (unsigned long long *)km_ws(c, WS_FFT_TOP2, (size_t)grid * 2 * sizeof(unsigned long long));
            if (!extra->best) return KM_E_NOMEM;
            extra->best_count = (size_t)grid;
            A.best = extra->best;
            if (paired) A.best_pair_w = width;
        }
    }
    auto launch = [&](auto ic) {
        constexpr int IMG = decltype(ic)::value;
        if (L.kind == 1) f64_prime_kernel<IMG><<<(unsigned)grid, 64 * F64_PR_NW, lds, c->stream>>>(A);
        else f64_smooth_kernel<IMG><<<(unsigned)grid, 256, lds, c->stream>>>(A);
    };
    switch (img) {
    case KM_U8: launch(std::integral_constant<int, KM_U8>{}); break;
    case KM_U16: launch(std::integral_constant<int, KM_U16>{}); break;
    case KM_I16: launch(std::integral_constant<int, KM_I16>{}); break;
    case KM_F32: launch(std::integral_constant<int, KM_F32>{}); break;
    default: launch(std::integral_constant<int, -1>{}); break;
    }
    KM_LAUNCH_CHECK(c);
    return KM_OK;
}

// tiles of a column level (what run_level chooses for it) and, on the Hermitian half, which of them hold a needed column
int cols_level_T(km_ctx *c, const lvl &L, int width, const std::vector<unsigned char> *lowx)
{
    if (L.kind == 1) {
        int tmax = std::min(64, std::max(1, 4096 / L.n));
        if (c->opt_f64_prime_t > 0) tmax = std::min(tmax, c->opt_f64_prime_t);
        const int tiles = (width + tmax - 1) / tmax;
        int best_T = (width + tiles - 1) / tiles;
        if (lowx) {
            // the tile width that leaves the fewest tiles to run (a tile costs the same whatever its width)
            auto needed = [&](int T) {
                int cnt = 0;
                for (int t0 = 0; t0 < width; t0 += T) {
                    bool any = false;
                    for (int x = t0; x < std::min(width, t0 + T) && !any; x++) any = (*lowx)[(size_t)x] != 0;
                    cnt += any;
                }
                return cnt;
            };
            int best_n = needed(best_T);
            for (int T = tmax; T >= std::max(16, tmax / 2); T--) {
                const int nn = needed(T);
                if (nn < best_n) { best_n = nn; best_T = T; }
            }
        }
        return best_T;
    }
    int T = 8;
    while (T > 1 && (L.n * T > F64_SMOOTH_MAX || (c->opt_f64_smooth_t > 0 && T > c->opt_f64_smooth_t))) T >>= 1;
    while (T > 1 && (T >> 1) >= width) T >>= 1;
    return T;
}

struct blue_tabs {
    dimplan inner;                 // plan of the power-of-two length L (rows of the scratch)
    dim_tabs tabs;                 // its tables
    const cd *chirp = nullptr;     // exp(-i pi m^2 / N), m < N
    cd *bhat = nullptr;            // transform of the chirp filter, in the permuted order the forward levels leave
};

// all levels of one dimension; `first` / `last`: extras of the first forward level / the last inverse level (level 0 either way)
// the Hermitian half of the columns in the inverse transform (host side of lvl_args::tile_mask / pair_src)
struct half_plane {
    bool on = false;
    cd *packed = nullptr;                               // the plane of packed row pairs the row levels work on
    const cd *src = nullptr;                            // the plane the column levels left (rows 2 j, 2 j + 1 of every pair)
    const int *srcx = nullptr;
    int hstart = -1;
    const unsigned char *mask[16] = {nullptr};          // per column level
    const unsigned char *keepx = nullptr;               // per column position: kx <= W / 2 (the cross-power pass writes only these)
    int T[16] = {0};
};

// pair_rows > 0 (inverse along the rows only): the levels run on pairs of image rows, the first one forming them while it loads
int fft_levels(km_ctx *c, cd *data, const dim_tabs &tabs, const dimplan &P, bool inverse, bool cols, int rows, int width, lvl_extra *level0 = nullptr,
               int pair_rows = 0, const half_plane *hp = nullptr)
{
    const int nl = (int)P.lv.size();
    const bool half = hp && hp->on && inverse;
    if (half && !cols) data = hp->packed;
    for (int i = 0; i < nl; i++) {
        const int l = inverse ? nl - 1 - i : i;
        lvl_extra ex;
        lvl_extra *pe = nullptr;
        if (l == 0 && level0) { ex = *level0; pe = &ex; }
        if (half && cols) { ex.tile_mask = hp->mask[l]; ex.tile_T = hp->T[l]; pe = &ex; }
        if (half && !cols) { ex.packed = true; ex.pair_src = hp->src; ex.srcx = hp->srcx; ex.pair_hstart = hp->hstart; }
        if (pair_rows > 0) { ex.pair_rows = pair_rows; ex.pair_load = i == 0; pe = &ex; }
        const int rc = run_level(c, data, tabs, P.N, P.lv[(size_t)l], l, inverse, cols, rows, width, pe);
        if (rc) return rc;
        if (l == 0 && level0) { level0->best = ex.best; level0->best_count = ex.best_count; }
    }
    return KM_OK;
}

// tables of a Bluestein dimension in workspace slot `slot`: [tables of the inner length L] [chirp (N)] [bhat (L)]
int blue_prepare(km_ctx *c, int slot, const dimplan &P, blue_tabs *B)
{
    const int N = P.N, L = P.L;
    if (!plan_dim(L, F64_SMOOTH_MAX, &B->inner) || B->inner.blue || B->inner.lv.size() > 16)
        return km_fail(c, KM_E_UNSUPPORTED, "phase correlation: side %d too long", N);
    std::vector<cd> h;
    size_t off[16], poff[16];
    const size_t nt = host_dim_tables(B->inner, h, off, poff);
    h.resize(nt + (size_t)N + (size_t)L);
    const long double pi = 3.141592653589793238462643383279502884L;
    for (int m = 0; m < N; m++) {
        const long long e = ((long long)m * m) % (2ll * N);                    // exp(-i pi m^2 / N) = exp(-2 pi i e / 2N)
        const long double ang = pi * (long double)e / (long double)N;
        h[nt + (size_t)m] = make_double2((double)cosl(ang), -(double)sinl(ang));
    }
    cd *filt = &h[nt + (size_t)N];
    for (int m = 0; m < L; m++) filt[m] = make_double2(0.0, 0.0);
    for (int m = 0; m < N; m++) {
        const cd ch = h[nt + (size_t)m];
        const cd cj = make_double2(ch.x, -ch.y);
        filt[m] = cj;
        if (m) filt[L - m] = cj;
    }
    cd *d = (cd *)km_ws(c, slot, h.size() * sizeof(cd));
    if (!d) return KM_E_NOMEM;
    int rc = km_h2d_small(c, d, h.data(), h.size() * sizeof(cd));
    if (rc) return rc;
    bind_dim_tables(B->inner, d, off, poff, &B->tabs);
    B->chirp = d + nt; B->bhat = d + nt + N;
    return fft_levels(c, B->bhat, B->tabs, B->inner, false, false, 1, L);
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
        int rc = fft_levels(c, s, B.tabs, B.inner, false, false, nr, L);
        if (rc) return rc;
        blue_mul_kernel<<<g, 256, 0, c->stream>>>(s, B.bhat, nr, L);
        KM_LAUNCH_CHECK(c);
        rc = fft_levels(c, s, B.tabs, B.inner, true, false, nr, L);
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

    if (PX.lv.size() > 16 || PY.lv.size() > 16) return km_fail(c, KM_E_UNSUPPORTED, "phase correlation: no plan for %d x %d", H, W);
    // tables: twiddles (+ the inter-level twiddles) and negated-frequency positions of both dimensions, kept between calls for the same shape
    dim_tabs TX, TY;
    int *negx = nullptr, *negy = nullptr;
    {
        const bool same = c->f64_h == H && c->f64_w == W;
        size_t offx[16], offy[16], poffx[16], poffy[16], nx = 0, ny = 0;
        dim_table_offsets(PX, offx, poffx, &nx);
        dim_table_offsets(PY, offy, poffy, &ny);
        cd *twx = (cd *)km_ws(c, WS_F64_TWX, nx * sizeof(cd));
        cd *twy = (cd *)km_ws(c, WS_F64_TWY, ny * sizeof(cd));
        negx = (int *)km_ws(c, WS_F64_NEGX, (size_t)std::max(W, 1) * sizeof(int));
        negy = (int *)km_ws(c, WS_F64_NEGY, (size_t)std::max(H, 1) * sizeof(int));
        if (!twx || !twy || !negx || !negy) return KM_E_NOMEM;
        if (!same) {
            c->f64_h = c->f64_w = 0;
            std::vector<cd> t;
            std::vector<int> ng;
            size_t off[16], poff[16];
            if (host_dim_tables(PX, t, off, poff) != nx) return km_fail(c, KM_E_ARG, "phase correlation: table size");
            if ((rc = km_h2d_small(c, twx, t.data(), t.size() * sizeof(cd)))) return rc;
            if (host_dim_tables(PY, t, off, poff) != ny) return km_fail(c, KM_E_ARG, "phase correlation: table size");
            if ((rc = km_h2d_small(c, twy, t.data(), t.size() * sizeof(cd)))) return rc;
            host_negpos(PX, ng);
            if ((rc = km_h2d_small(c, negx, ng.data(), ng.size() * sizeof(int)))) return rc;
            host_negpos(PY, ng);
            if ((rc = km_h2d_small(c, negy, ng.data(), ng.size() * sizeof(int)))) return rc;
            c->f64_h = H; c->f64_w = W;
        }
        bind_dim_tables(PX, twx, offx, poffx, &TX);
        bind_dim_tables(PY, twy, offy, poffy, &TY);
    }
    blue_tabs BX, BY;
    if (PX.blue && (rc = blue_prepare(c, WS_F64_BLUEX, PX, &BX))) return rc;
    if (PY.blue && (rc = blue_prepare(c, WS_F64_BLUEY, PY, &BY))) return rc;
    cd *zt = nullptr;                                            // the transposed plane of a Bluestein column pass
    if (PY.blue) {
        zt = (cd *)km_ws(c, WS_FFT_B, n * sizeof(cd));
        if (!zt) return KM_E_NOMEM;
    }

    // with level kernels along the rows, the first of them reads the images itself and the last inverse one reports the arg-max;
    // otherwise (one-pixel-wide image, Bluestein along the rows) a pack pass in front and two reduction passes behind
    const bool fused_ends = W > 1 && !PX.blue && !PX.lv.empty() && !c->opt_f64_plain;
    lvl_extra first, last;
    if (fused_ends) {
        first.img_a = d_a; first.img_b = d_b; first.img_sa = stride_a; first.img_sb = stride_b; first.img_dtype = dtype;
        last.want_best = true;
        if (dtype != KM_U8 && dtype != KM_U16 && dtype != KM_I16 && dtype != KM_F32) return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
    } else {
        switch (dtype) {
        case KM_U8: rc = launch_pack<uint8_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
        case KM_U16: rc = launch_pack<uint16_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
        case KM_I16: rc = launch_pack<int16_t>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
        case KM_F32: rc = launch_pack<float>(c, d_a, d_b, stride_a, stride_b, H, W, z); break;
        default: return km_fail(c, KM_E_ARG, "phase_shift: bad dtype %d", dtype);
        }
        if (rc) return rc;
    }

    // inverse: the surface is real - two image rows per complex transform, and (level kernels along the columns too) only the
    // columns kx <= W / 2 through the inverse column levels
    const bool pairs = fused_ends && H >= 2 && c->opt_f64_pair;
    half_plane HP;
    if (pairs && !PY.blue && !PY.lv.empty() && c->opt_f64_half && W >= 32) {
        std::vector<unsigned char> lowx((size_t)W);
        for (int pos = 0; pos < W; pos++) {
            long long fr = 0, mult = 1;
            for (const lvl &L : PX.lv) { fr += (long long)((pos / L.R) % L.n) * mult; mult *= L.n; }
            lowx[(size_t)pos] = fr <= W / 2;
        }
        const size_t nl = PY.lv.size();
        // [srcx: W ints] [one mask of W bytes per column level] [lowx: W bytes]
        std::vector<unsigned char> h((size_t)W * (4 + nl + 1), 0);
        {
            std::vector<int> ng;
            host_negpos(PX, ng);
            int *sx = (int *)h.data();
            for (int pos = 0; pos < W; pos++) sx[pos] = lowx[(size_t)pos] ? pos : ~ng[(size_t)pos];
        }
        for (size_t l = 0; l < nl; l++) {
            const int T = cols_level_T(c, PY.lv[l], W, &lowx);
            HP.T[l] = T;
            unsigned char *m = &h[(size_t)W * (4 + l)];
            for (int t0 = 0, tb = 0; t0 < W; t0 += T, tb++)
                for (int x = t0; x < std::min(W, t0 + T); x++) m[tb] |= lowx[(size_t)x];
        }
        // from which element on every transform of the first row level (the last level of the plan: contiguous) is mirrored
        {
            const int n0 = PX.lv.back().n;
            int hs = 0;
            for (int pos = 0; pos < W; pos++) if (lowx[(size_t)pos]) hs = std::max(hs, pos % n0 + 1);
            HP.hstart = hs < n0 ? hs : -1;
        }
        std::copy(lowx.begin(), lowx.end(), h.begin() + (size_t)W * (4 + nl));
        unsigned char *d = (unsigned char *)km_ws(c, WS_F64_MASK, h.size());
        cd *z2 = (cd *)km_ws(c, WS_FFT_B, (size_t)((H + 1) / 2) * W * sizeof(cd));
        if (!d || !z2) return KM_E_NOMEM;
        if ((rc = km_h2d_small(c, d, h.data(), h.size()))) return rc;
        HP.on = true; HP.packed = z2; HP.src = z; HP.srcx = (const int *)d;
        for (size_t l = 0; l < nl; l++) HP.mask[l] = d + (size_t)W * (4 + l);
        HP.keepx = d + (size_t)W * (4 + nl);
    }
    auto along_rows = [&](bool inverse) -> int {
        if (W <= 1) return KM_OK;
        if (PX.blue) return blue_rows(c, z, H, PX, BX, inverse);
        return fft_levels(c, z, TX, PX, inverse, false, H, W, fused_ends ? (inverse ? &last : &first) : nullptr, inverse && pairs ? H : 0, &HP);
    };
    auto along_cols = [&](bool inverse) -> int {
        if (H <= 1) return KM_OK;
        if (!PY.blue) return fft_levels(c, z, TY, PY, inverse, true, H, W, nullptr, 0, &HP);
        int r = transpose(c, z, zt, H, W);
        if (!r) r = blue_rows(c, zt, W, PY, BY, inverse);
        if (!r) r = transpose(c, zt, z, W, H);
        return r;
    };
    if ((rc = along_rows(false))) return rc;
    if ((rc = along_cols(false))) return rc;
    {
        const dim3 g((unsigned)std::min((W + 255) / 256, 64), (unsigned)std::min(H, 8192));
        f64_cross_kernel<<<g, 256, 0, c->stream>>>(z, negx, negy, H, W, HP.on ? HP.keepx : nullptr);
        KM_LAUNCH_CHECK(c);
    }
    if ((rc = along_cols(true))) return rc;
    if ((rc = along_rows(true))) return rc;

    unsigned long long *keys = &sc->argmax_key;                  // max bits
    unsigned long long *idx = (unsigned long long *)&sc->valid;  // reused as the first-index slot
    if (fused_ends) {
        f64_best_reduce_kernel<<<1, 1024, 0, c->stream>>>(last.best, last.best_count, idx);
        KM_LAUNCH_CHECK(c);
    } else {
        KM_HIP(c, hipMemsetAsync(keys, 0, sizeof(unsigned long long), c->stream));
        KM_HIP(c, hipMemsetAsync(idx, 0xff, sizeof(unsigned long long), c->stream));
        f64_absmax_kernel<<<2048, 256, 0, c->stream>>>(z, n, keys);
        KM_LAUNCH_CHECK(c);
        f64_first_index_kernel<<<2048, 256, 0, c->stream>>>(z, n, keys, idx);
        KM_LAUNCH_CHECK(c);
    }
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
