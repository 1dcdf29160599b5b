// Microbenchmark: issue cost per wave64 instruction of the VALU instructions the path kernels are made of, with 4 waves
// per SIMD (the occupancy k_paths runs at; `valu_mix N` runs a short list with N = 1..8 waves per SIMD instead), eight
// independent chains per wave. Also reports the shader clock from the
// ratio of s_memtime to the constant 100 MHz s_memrealtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(64) void k(float* out, int iters, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x + i; a[i].y = threadIdx.x * 0.5f + i; }
    f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#define ONE(j) \
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)); \
            if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); \
            if (OP == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[j].x)); \
            if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j].x)); \
            if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 9) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(a[j]) : "v"(b.x), "v"(c.x) : "s10", "s11"); \
            if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[j].x), "v"(b.x) : "vcc"); \
            if (OP == 11) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b)); \
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 14) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 15) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[10:11]" : "=v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 16) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j].x) : "v"(b.x), "v"(c.x) : "vcc"); \
            if (OP == 17) asm volatile("v_floor_f32 %0, %0" : "+v"(a[j].x)); \
            if (OP == 18) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[j].x)); \
            if (OP == 19) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[j].x) : "v"(b.x) : "vcc"); \
            if (OP == 20) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 21) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 22) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 23) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 24) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 25) asm volatile("v_readlane_b32 s10, %0, 3" :: "v"(a[j].x) : "s10"); \
            if (OP == 26) asm volatile("v_cmp_lt_f32 s[10:11], %0, %1" :: "v"(a[j].x), "v"(b.x) : "s10", "s11"); \
            if (OP == 27) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j].x) : "v"(b.x)); \
            if (OP == 28) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 29) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[j].x) : "v"(b.x), "v"(c.x)); \
            if (OP == 30) asm volatile("s_add_u32 s10, s10, 1\n s_and_b64 s[12:13], s[12:13], exec" ::: "s10", "s12", "s13", "scc");
            REP8(ONE)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
static int g_wps = 4;
template <int OP> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000, wps = g_wps, blocks = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<OP><<<blocks, 64>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double inst = (double)iters * 64, ghz = (double)c[0] / ((double)c[1] * 10.0);
    printf("%-28s waves/SIMD %d  %.3f ms  clock %.2f GHz  SIMD cycles per instruction %.2f  (one wave's own cycles per instruction %.2f)\n", name, wps, ms, ghz,
           ms * 1e-3 * ghz * 1e9 / (inst * wps), (double)c[0] / inst);
}
int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 1024 * 64 * 4); hipMalloc(&cyc, 16);
    if (argc > 1) {   // short list at another occupancy
        g_wps = atoi(argv[1]);
        if (g_wps < 1 || g_wps > 8) return 1;
        run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<27>("v_mul_f32", out, cyc); run<3>("v_min_f32", out, cyc);
        run<4>("v_max3_f32", out, cyc); run<15>("v_cndmask_e64 sgpr (indep)", out, cyc); run<18>("v_cvt_f32_u32", out, cyc);
        run<13>("v_mov_b32", out, cyc); run<7>("v_rcp_f32", out, cyc); run<30>("s_add + s_and (2 salu)", out, cyc);
        return 0;
    }
    run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<2>("v_pk_mul_f32", out, cyc); run<12>("v_pk_add_f32", out, cyc);
    run<3>("v_min_f32", out, cyc); run<4>("v_max3_f32", out, cyc); run<5>("v_mul_lo_u32", out, cyc); run<11>("v_mul_u32_u24", out, cyc);
    run<9>("v_mad_u64_u32", out, cyc); run<6>("v_sqrt_f32", out, cyc); run<7>("v_rcp_f32", out, cyc); run<8>("v_cndmask_b32", out, cyc);
    run<10>("v_cmp_lt_f32 vcc", out, cyc); run<13>("v_mov_b32", out, cyc);
    run<14>("v_cndmask_e64 sgpr (chain)", out, cyc); run<15>("v_cndmask_e64 sgpr (indep)", out, cyc); run<16>("v_cmp vcc + v_cndmask (2)", out, cyc);
    run<17>("v_floor_f32", out, cyc); run<18>("v_cvt_f32_u32", out, cyc); run<19>("v_div_scale_f32", out, cyc); run<20>("v_div_fmas_f32", out, cyc);
    run<21>("v_div_fixup_f32", out, cyc); run<22>("v_add_u32", out, cyc); run<23>("v_xor_b32", out, cyc); run<24>("v_fmac_f32", out, cyc);
    run<25>("v_readlane_b32", out, cyc); run<26>("v_cmp_lt_f32 sgpr", out, cyc); run<27>("v_mul_f32", out, cyc); run<28>("v_bfi_b32", out, cyc);
    run<29>("v_med3_f32", out, cyc); run<30>("s_add + s_and (2 salu)", out, cyc);
    return 0;
}
