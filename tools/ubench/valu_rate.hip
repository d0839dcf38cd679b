// Micro-benchmark: sustained issue rate of a few VALU instruction classes on gfx950, per SIMD, as a function of the
// number of resident waves.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, int *out)
{
    int a = threadIdx.x, b = blockIdx.x + 3, c = 7, d = 11, e = 13, f = 17, g = (threadIdx.x & 63) * 8, h = 23;
    double da = a, db = b, dc = 1.000001, dd = 0.5;
    float fa = a, fb = b;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // independent full-rate int ops (4 chains)
            REP16(asm volatile("v_mad_i32_i24 %0, %0, %4, %1\n v_mad_i32_i24 %1, %1, %4, %2\n v_mad_i32_i24 %2, %2, %4, %3\n v_mad_i32_i24 %3, %3, %4, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (KIND == 1) {   // one dependent chain
            REP16(asm volatile("v_mad_i32_i24 %0, %0, %1, %0\n v_mad_i32_i24 %0, %0, %1, %0\n v_mad_i32_i24 %0, %0, %1, %0\n v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a) : "v"(e));)
        } else if (KIND == 2) {   // dot4
            REP16(asm volatile("v_dot4_i32_i8 %0, %4, %5, %0\n v_dot4_i32_i8 %1, %4, %5, %1\n v_dot4_i32_i8 %2, %4, %5, %2\n v_dot4_i32_i8 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 3) {   // fma f64
            REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(da), "+v"(db) : "v"(dc), "v"(dd));)
        } else if (KIND == 4) {   // cvt f64<-i32, cvt i32<-f64
            REP16(asm volatile("v_cvt_f64_i32 %2, %0\n v_cvt_i32_f64 %0, %2\n v_cvt_f64_i32 %3, %1\n v_cvt_i32_f64 %1, %3" : "+v"(a), "+v"(b), "+v"(da), "+v"(db));)
        } else if (KIND == 5) {   // DPP wave shift adds (dependent)
            REP16(asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n v_add_u32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
        } else if (KIND == 6) {   // sqrt f32
            REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1" : "+v"(fa), "+v"(fb));)
        } else if (KIND == 7) {   // packed 16-bit add
            REP16(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (KIND == 8) {   // mul_lo u32 (quarter rate?)
            REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (KIND == 9) {   // ds_bpermute + wait + use (latency bound per wave)
            REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(e));)
        } else if (KIND == 10) {  // dot2c two-address form
            REP16(asm volatile("v_dot2c_i32_i16 %0, %4, %5\n v_dot2c_i32_i16 %1, %4, %5\n v_dot2c_i32_i16 %2, %4, %5\n v_dot2c_i32_i16 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 11) {  // f32 fma full rate reference
            REP16(asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(fa), "+v"(fb) : "v"(fa), "v"(fb));)
        } else if (KIND == 12) {  // cvt f32<-f64
            REP16(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3\n v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3" : "+v"(fa), "+v"(fb) : "v"(da), "v"(db));)
        } else if (KIND == 13) {  // v_perm / alignbyte
            REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, 1\n v_perm_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, 2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 14) {  // mul_f64
            REP16(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2\n v_mul_f64 %0, %0, %3\n v_mul_f64 %1, %1, %3" : "+v"(da), "+v"(db) : "v"(dc), "v"(dd));)
        } else if (KIND == 15) {  // mad_u64_u32
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(da), "+v"(db) : "v"(e), "v"(f) : "vcc");)
        } else if (KIND == 16) {  // cmp (sgpr dst) + cndmask
            REP16(asm volatile("v_cmp_gt_f32_e64 s[20:21], %2, %3\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cmp_lt_f32_e64 s[22:23], %2, %3\n v_cndmask_b32_e64 %1, %1, %0, s[22:23]" : "+v"(a), "+v"(b) : "v"(fa), "v"(fb) : "s20", "s21", "s22", "s23");)
        } else if (KIND == 17) {  // bfe_i32 / ashr
            REP16(asm volatile("v_bfe_i32 %0, %1, 0, 16\n v_ashrrev_i32 %1, 16, %0\n v_bfe_i32 %2, %3, 0, 16\n v_ashrrev_i32 %3, 16, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (KIND == 18) {  // pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(da), "+v"(db) : "v"(dc));)
        } else if (KIND == 19) {  // max3_f32 / add3
            REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_add3_u32 %3, %3, %4, %5\n v_max3_f32 %1, %0, %1, %2\n v_add3_u32 %4, %3, %4, %5" : "+v"(fa), "+v"(fb), "+v"(fa), "+v"(a), "+v"(b), "+v"(c));)
        } else if (KIND == 20) {  // pk_sub_i16 + pk_lshlrev_b16
            REP16(asm volatile("v_pk_sub_i16 %0, %0, %4\n v_pk_lshlrev_b16 %1, 1, %1\n v_pk_sub_i16 %2, %2, %4\n v_pk_lshlrev_b16 %3, 1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (KIND == 21) {  // ds_write_b64 + ds_read (no wait)
            REP16(asm volatile("ds_write_b64 %2, %0\n ds_write_b64 %2, %1\n ds_write_b64 %2, %0\n ds_write_b64 %2, %1" :: "v"(da), "v"(db), "v"(g)); ) asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == 22) {  // v_cmp to vcc + v_cndmask e32
            REP16(asm volatile("v_cmp_gt_u32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");)
        } else if (KIND == 23) {  // mbcnt + bcnt
            REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %2, 0\n v_mbcnt_hi_u32_b32 %0, %3, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %1, %3, %1" : "+v"(a), "+v"(b) : "v"(c), "v"(d));)
        } else if (KIND == 24) {  // mad_i64_i32
            REP16(asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1" : "+v"(da), "+v"(db) : "v"(e), "v"(f) : "vcc");)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (int)da + (int)db + (int)fa + (int)fb + g + h;
}

template <int KIND> void run(const char *name, int ops_per_iter)
{
    int *out;
    hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    printf("%-28s", name);
    for (int wg_per_cu : {1, 2, 4, 8}) {   // 256-thread WGs: 1 wave per SIMD each
        const int grid = 256 * wg_per_cu;
        k<KIND><<<grid, 256>>>(10, out);
        hipEventRecord(e0);
        k<KIND><<<grid, 256>>>(iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: wg_per_cu waves, each issuing iters*ops instructions
        const double instr_per_simd = (double)wg_per_cu * iters * ops_per_iter;
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("  w/SIMD=%d: %.2f cyc/instr", wg_per_cu, cycles / instr_per_simd);
    }
    printf("\n");
    hipFree(out);
}

int main()
{
    run<0>("mad_i32_i24 x4 indep", 64);
    run<1>("mad_i32_i24 dependent", 64);
    run<2>("dot4_i32_i8", 64);
    run<10>("dot2c_i32_i16", 64);
    run<3>("fma_f64", 64);
    run<4>("cvt f64<->i32", 64);
    run<12>("cvt f32<-f64", 64);
    run<5>("dpp add (dep, s_nop 1)", 64);
    run<6>("sqrt_f32", 64);
    run<7>("pk_add_u16", 64);
    run<8>("mul_lo_u32", 64);
    run<11>("fma_f32", 64);
    run<13>("perm/alignbyte", 64);
    run<9>("bpermute+wait+add (x2)", 32);
    run<14>("mul_f64", 64);
    run<15>("mad_u64_u32", 64);
    run<24>("mad_i64_i32", 64);
    run<16>("cmp_e64 + cndmask_e64", 64);
    run<22>("cmp vcc + cndmask_e32", 64);
    run<17>("bfe_i32 / ashrrev", 64);
    run<18>("pk_mul_f32", 64);
    run<19>("max3_f32 / add3_u32", 64);
    run<20>("pk_sub_i16/pk_lshlrev_b16", 64);
    run<23>("mbcnt / bcnt", 64);
    run<21>("ds_write_b64", 64);
    return 0;
}
