// Unit check of the inline-assembly helpers of k_eig3.hip against plain arithmetic (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o asm_check asm_check.hip && ./asm_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ int mad_lo(uint32_t a, uint32_t b, int c) { int d; asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int mad_hi(uint32_t a, uint32_t b, int c) { int d; asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ uint32_t pk_mad2(uint32_t a, uint32_t c) { uint32_t d; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(0x00020002u), "v"(c)); return d; }
__device__ __forceinline__ int add_next(int w, int v) { int d; asm("v_add_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w)); return d; }
__device__ __forceinline__ int add_prev(int w, int v) { int d; asm("v_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w)); return d; }
// v[lane + 1] - w / v[lane - 1] - w  (v_subrev_u32_dpp written in assembly computed dpp(src1) - src0 on this toolchain: not used)
__device__ __forceinline__ int sub_next(int w, int v) { int d; asm("v_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w)); return d; }
__device__ __forceinline__ int sub_prev(int w, int v) { int d; asm("v_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(v), "v"(w)); return d; }

__global__ void k(const uint32_t *a, const uint32_t *b, const int *c, int *out)
{
    const int l = threadIdx.x;
    uint32_t A = a[l], B = b[l];
    int C = c[l];
    asm volatile("s_nop 4" : "+v"(A), "+v"(B), "+v"(C));
    out[0 * 64 + l] = mad_lo(A, B, C);
    out[1 * 64 + l] = mad_hi(A, B, C);
    out[2 * 64 + l] = (int)pk_mad2(A, B);
    out[3 * 64 + l] = add_next(C, (int)A);
    out[4 * 64 + l] = add_prev(C, (int)A);
    out[5 * 64 + l] = sub_next(C, (int)A);
    out[6 * 64 + l] = sub_prev(C, (int)A);
}

int main()
{
    std::vector<uint32_t> a(64), b(64);
    std::vector<int> c(64), out(7 * 64);
    for (int l = 0; l < 64; l++) {
        a[l] = ((uint32_t)(uint16_t)(int16_t)(-1020 + 37 * l) << 16) | (uint16_t)(int16_t)(900 - 29 * l);
        b[l] = ((uint32_t)(uint16_t)(int16_t)(500 - 17 * l) << 16) | (uint16_t)(int16_t)(-333 + 11 * l);
        c[l] = 1000000 + 12345 * l;
    }
    uint32_t *da, *db; int *dc, *dout;
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dc, 256); (void)hipMalloc(&dout, 7 * 256);
    (void)hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dc, c.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dc, dout);
    (void)hipMemcpy(out.data(), dout, 7 * 256, hipMemcpyDeviceToHost);
    const char *names[7] = {"mad_lo", "mad_hi", "pk_mad2", "add_next", "add_prev", "sub_next", "sub_prev"};
    int bad[7] = {0};
    for (int l = 0; l < 64; l++) {
        const int alo = (int16_t)(a[l] & 0xffff), ahi = (int16_t)(a[l] >> 16), blo = (int16_t)(b[l] & 0xffff), bhi = (int16_t)(b[l] >> 16);
        int exp[7];
        exp[0] = alo * blo + c[l];
        exp[1] = ahi * bhi + c[l];
        exp[2] = (int)((((uint32_t)(uint16_t)(2 * ahi + bhi)) << 16) | (uint16_t)(2 * alo + blo));
        const int nxt = l < 63 ? (int)a[l + 1] : 0, prv = l > 0 ? (int)a[l - 1] : 0;
        exp[3] = c[l] + nxt; exp[4] = c[l] + prv; exp[5] = nxt - c[l]; exp[6] = prv - c[l];
        for (int t = 0; t < 7; t++)
            if (out[t * 64 + l] != exp[t]) { if (bad[t]++ < 3) printf("%s lane %d: got %d expected %d\n", names[t], l, out[t * 64 + l], exp[t]); }
    }
    int total = 0;
    for (int t = 0; t < 7; t++) { printf("%-9s %s\n", names[t], bad[t] ? "MISMATCH" : "ok"); total += bad[t]; }
    return total ? 1 : 0;
}
