// Is the gfx90a+ "DOT result read too early" hazard (LLVM GCNHazardRecognizer::checkMAIVALUHazards: 3 wait states before a
// DIFFERENT opcode may read a v_dot* result; only the same opcode's src C is forwarded) observable on gfx950, and does it
// bite the pattern the Laplacian kernel used in round 1: a VOP3P v_dot4_i32_i8 / v_dot2_i32_i16 written as inline asm
// (invisible to the compiler's hazard recognizer) whose result is accumulated into by the compiler's VOP2 v_dot4c / v_dot2c
// one or two wait states later?
//
// (The result of every sequence is followed by s_nop 4: the harness's own compare must not read it too early either.)
// Every variant computes  r = dot(x, k0) then r = dotc(y, k1) + r  (or an add) with N wait states between the two
// instructions, N = 0, 1, 2, 3 (+ a reference with s_nop 7), over random data, with 1..8 waves per SIMD resident and a
// spinning neighbour wave mix; mismatches against the reference are counted.
// Build: hipcc --offload-arch=gfx950 -O3 -o dot_hazard dot_hazard.hip ; run: ./dot_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SEQ_DOT4(NOPS)                                                                                             \
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0\n" NOPS "v_dot4c_i32_i8_e32 %0, %3, %4\n s_nop 4" : "=&v"(r) : "v"(x), "v"(k0), "v"(y), "v"(k1))
#define SEQ_DOT2(NOPS)                                                                                             \
    asm volatile("v_dot2_i32_i16 %0, %1, %2, 0\n" NOPS "v_dot2c_i32_i16_e32 %0, %3, %4\n s_nop 4" : "=&v"(r) : "v"(x), "v"(k0), "v"(y), "v"(k1))
#define SEQ_ADD(NOPS)                                                                                              \
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0\n" NOPS "v_add_u32 %0, %0, %3\n s_nop 4" : "=&v"(r) : "v"(x), "v"(k0), "v"(y))
#define SEQ_ADD2(NOPS)                                                                                             \
    asm volatile("v_dot2_i32_i16 %0, %1, %2, 0\n" NOPS "v_add_u32 %0, %0, %3\n s_nop 4" : "=&v"(r) : "v"(x), "v"(k0), "v"(y))
#define SEQ_SAME(NOPS)                                                                                             \
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0\n" NOPS "v_dot4_i32_i8 %0, %3, %4, %0\n s_nop 4" : "=&v"(r) : "v"(x), "v"(k0), "v"(y), "v"(k1))

template <int KIND, int WS>
__device__ __forceinline__ int seq(int x, int k0, int y, int k1)
{
    int r;
    if (KIND == 0) { if (WS == 0) SEQ_DOT4(""); else if (WS == 1) SEQ_DOT4("s_nop 0\n"); else if (WS == 2) SEQ_DOT4("s_nop 1\n"); else if (WS == 3) SEQ_DOT4("s_nop 2\n"); else SEQ_DOT4("s_nop 7\n"); }
    if (KIND == 1) { if (WS == 0) SEQ_DOT2(""); else if (WS == 1) SEQ_DOT2("s_nop 0\n"); else if (WS == 2) SEQ_DOT2("s_nop 1\n"); else if (WS == 3) SEQ_DOT2("s_nop 2\n"); else SEQ_DOT2("s_nop 7\n"); }
    if (KIND == 2) { if (WS == 0) SEQ_ADD(""); else if (WS == 1) SEQ_ADD("s_nop 0\n"); else if (WS == 2) SEQ_ADD("s_nop 1\n"); else if (WS == 3) SEQ_ADD("s_nop 2\n"); else SEQ_ADD("s_nop 7\n"); }
    if (KIND == 3) { if (WS == 0) SEQ_SAME(""); else if (WS == 1) SEQ_SAME("s_nop 0\n"); else if (WS == 2) SEQ_SAME("s_nop 1\n"); else if (WS == 3) SEQ_SAME("s_nop 2\n"); else SEQ_SAME("s_nop 7\n"); }
    if (KIND == 4) { if (WS == 0) SEQ_ADD2(""); else if (WS == 1) SEQ_ADD2("s_nop 0\n"); else if (WS == 2) SEQ_ADD2("s_nop 1\n"); else if (WS == 3) SEQ_ADD2("s_nop 2\n"); else SEQ_ADD2("s_nop 7\n"); }
    return r;
}

template <int KIND, int WS>
__global__ __launch_bounds__(256) void k(int iters, unsigned seed, unsigned long long *bad, int lds_pad)
{
    extern __shared__ int pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0;   // never: keeps the dynamic LDS (occupancy limiter) alive
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    unsigned long long nbad = 0;
    int poison = (int)s;
    for (int i = 0; i < iters; i++) {
        s = s * 1664525u + 1013904223u; const int x = (int)s;
        s = s * 1664525u + 1013904223u; const int y = (int)s;
        s = s * 1664525u + 1013904223u; const int k0 = (int)s;
        s = s * 1664525u + 1013904223u; const int k1 = (int)s;
        const int want = seq<KIND, 9>(x, k0, y, k1);
        // the destination register holds an unrelated value before the sequence: a stale read then shows up as a wrong sum
        int got;
        {
            int r = poison;
            asm volatile("" : "+v"(r));
            got = seq<KIND, WS>(x, k0, y, k1);
            poison ^= got + r;
        }
        nbad += got != want;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (poison == 0x7fffffff && iters < 0) bad[1] = 1;
}

template <int KIND, int WS>
unsigned long long run(int waves_per_simd, int iters, unsigned long long *d_bad)
{
    hipMemset(d_bad, 0, 16);
    // occupancy through dynamic LDS: 160 KB per CU, 256-thread workgroups = 1 wave per SIMD each
    const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 1024;
    hipFuncSetAttribute((const void *)k<KIND, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<KIND, WS><<<256 * waves_per_simd, 256, lds>>>(iters, 12345u + waves_per_simd, d_bad, 0);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost);
    return h;
}

template <int KIND> void sweep(const char *name, unsigned long long *d_bad, int iters)
{
    for (int w : {1, 2, 4, 8}) {
        const unsigned long long total = 256ull * w * 256 * iters;
        printf("%-44s waves/SIMD %d  instances %.3g  mismatches: ws0 %llu  ws1 %llu  ws2 %llu  ws3 %llu\n", name, w, (double)total,
               run<KIND, 0>(w, iters, d_bad), run<KIND, 1>(w, iters, d_bad), run<KIND, 2>(w, iters, d_bad), run<KIND, 3>(w, iters, d_bad));
        fflush(stdout);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long *d_bad;
    hipMalloc(&d_bad, 16);
    sweep<2>("dot4 (VOP3P) -> v_add_u32 [different VALU]", d_bad, iters);
    sweep<4>("dot2 (VOP3P) -> v_add_u32 [different VALU]", d_bad, iters);
    sweep<0>("dot4 (VOP3P) -> v_dot4c acc [round-1 K2]", d_bad, iters);
    sweep<1>("dot2 (VOP3P) -> v_dot2c acc [round-1 K2]", d_bad, iters);
    sweep<3>("dot4 (VOP3P) -> dot4 (VOP3P) src C", d_bad, iters);
    return 0;
}
