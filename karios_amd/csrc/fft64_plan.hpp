// Host-side planning of the double-precision transform of k_fft64.hip (no device code: also compiled into the sanitizer build of
// the library's host half and reachable without a GPU through km_phase_plan).
#pragma once
#include <algorithm>
#include <vector>

#define F64_PMAX 127          // largest prime factor with a level kernel of its own
#define F64_SMOOTH_MAX 2048   // longest smooth level (contiguous transforms); strided levels stay <= 256 so that T = 8 fits
#define F64_MAX_STAGES 12

namespace f64plan {

struct lvl {
    int n = 0, R = 1, kind = 0;            // kind 0: smooth, 1: prime
    int nst = 0;
    int radix[F64_MAX_STAGES] = {0};
};
struct dimplan {
    int N = 0;
    std::vector<lvl> lv;
    bool blue = false;
    int L = 0;                             // Bluestein: length of the convolution
};

inline void factor_primes(int N, std::vector<int> &out)
{
    for (int p = 2; (long long)p * p <= N; p++)
        while (N % p == 0) { out.push_back(p); N /= p; }
    if (N > 1) out.push_back(N);
}

// radices of a smooth level, in stage order: 7s and 5s, then 6s (a 2 and a 3 in ONE pass over the LDS: 180 = 5 x 6 x 6 is three
// stages instead of the four of 5 x 4 x 3 x 3), 4s, and what is left of the 3s and 2s
inline bool smooth_radices(int n, lvl *L)
{
    L->nst = 0;
    auto push = [&](int r) { if (L->nst == F64_MAX_STAGES) return false; L->radix[L->nst++] = r; return true; };
    int c[8] = {0};
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) { c[p]++; n /= p; }
    if (n != 1) return false;
    for (int i = 0; i < c[7]; i++) if (!push(7)) return false;
    for (int i = 0; i < c[5]; i++) if (!push(5)) return false;
    int sixes = std::min(c[2], c[3]);
    for (int i = 0; i < sixes; i++) if (!push(6)) return false;
    int twos = c[2] - sixes, threes = c[3] - sixes;
    for (; twos >= 2; twos -= 2) if (!push(4)) return false;
    for (; threes > 0; threes--) if (!push(3)) return false;
    if (twos && !push(2)) return false;
    return L->nst > 0;
}

// levels of one dimension.  `last_cap`: longest smooth level allowed at the end (rows: its transforms are contiguous, 2048; columns:
// every level is strided and wants T = 8 transforms per tile, 256)
inline bool plan_dim(int N, int last_cap, dimplan *P)
{
    P->N = N; P->lv.clear(); P->blue = false; P->L = 0;
    if (N <= 1) return true;
    std::vector<int> pr;
    factor_primes(N, pr);
    long long S = 1;
    std::vector<int> big;
    for (int p : pr) {
        if (p > F64_PMAX) { P->blue = true; }
        if (p >= 11) big.push_back(p); else S *= p;
    }
    if (P->blue) {
        int L = 1;
        while (L < 2 * N - 1) L <<= 1;
        P->L = L;
        return true;
    }
    std::sort(big.begin(), big.end(), [](int x, int y) { return x > y; });
    std::vector<int> ns;
    std::vector<int> kinds;
    for (int p : big) { ns.push_back(p); kinds.push_back(1); }
    while (S > last_cap) {
        int d = 1;
        for (int c = 256; c >= 2; c--) if (S % c == 0) { d = c; break; }
        if (d == 1) return false;
        ns.push_back(d); kinds.push_back(0);
        S /= d;
    }
    if (S > 1) { ns.push_back((int)S); kinds.push_back(0); }
    long long R = N;
    for (size_t i = 0; i < ns.size(); i++) {
        lvl L;
        L.n = ns[i]; L.kind = kinds[i];
        R /= ns[i];
        L.R = (int)R;
        if (L.kind == 0 && !smooth_radices(L.n, &L)) return false;
        P->lv.push_back(L);
    }
    return true;
}

// position of frequency -k for every position (k = frequency stored there)
inline void host_negpos(const dimplan &P, std::vector<int> &neg)
{
    const int N = P.N;
    neg.resize((size_t)std::max(N, 1));
    if (N <= 1) { neg[0] = 0; return; }
    if (P.blue || P.lv.empty()) {
        for (int i = 0; i < N; i++) neg[(size_t)i] = (N - i) % N;
        return;
    }
    std::vector<int> f((size_t)N), posof((size_t)N);
    for (int pos = 0; pos < N; pos++) {
        long long fr = 0, mult = 1;
        for (const lvl &L : P.lv) { fr += (long long)((pos / L.R) % L.n) * mult; mult *= L.n; }
        f[(size_t)pos] = (int)fr;
        posof[(size_t)fr] = pos;
    }
    for (int pos = 0; pos < N; pos++) neg[(size_t)pos] = posof[(size_t)((N - f[(size_t)pos]) % N)];
}

}  // namespace f64plan
