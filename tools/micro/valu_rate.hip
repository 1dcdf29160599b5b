// Microbenchmark: how many cycles one wave64 v_fma_f32 costs a SIMD with 1, 2, 4, 8 waves per SIMD, for a stream of
// independent instructions and for one dependent chain. Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int DEP>
__global__ __launch_bounds__(64) void k(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (DEP) {
#pragma unroll
            for (int k2 = 0; k2 < 64; ++k2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 1024 * 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    for (int dep = 0; dep < 2; ++dep)
        for (int wps = 1; wps <= 8; wps *= 2) {
            int blocks = 256 * 4 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (dep) k<1><<<blocks, 64>>>(out, iters, cyc); else k<0><<<blocks, 64>>>(out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double inst = (double)iters * 64;
            printf("dep=%d waves/SIMD=%d  %.3f ms  wave0 cycles/inst %.2f  chip TFLOP/s %.1f  SIMD cycles/inst(at 2.4GHz) %.2f\n", dep, wps, ms,
                   c / inst, blocks * inst * 128 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (inst * wps));
        }
    return 0;
}
