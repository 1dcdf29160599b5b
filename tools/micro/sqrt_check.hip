// Checks pt_sqrt_in_range against the compiler's correctly rounded sqrtf for every f32 in [2^-96, 2^64) and for 0.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I path_tracing_amd/csrc tools/micro/sqrt_check.hip -o tools/micro/sqrt_check
#include "pt_math.h"
#include <stdio.h>
__global__ void k(uint32_t first, uint32_t count, unsigned long long* bad, uint32_t* example) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float x = __uint_as_float(first + i);
        const float a = pt_sqrt_in_range(x), b = __builtin_sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b)) { atomicAdd(bad, 1ull); *example = first + i; }
    }
}
int main() {
    unsigned long long* bad; uint32_t* ex;
    hipMalloc(&bad, 8); hipMalloc(&ex, 4); hipMemset(bad, 0, 8); hipMemset(ex, 0, 4);
    const uint32_t lo = (127u - 96u) << 23, hi = (127u + 64u) << 23;   // [2^-96, 2^64)
    k<<<4096, 256>>>(lo, hi - lo, bad, ex);
    k<<<1, 64>>>(0u, 1u, bad, ex);
    unsigned long long b; uint32_t e;
    hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&e, ex, 4, hipMemcpyDeviceToHost);
    printf("checked %u values and 0: %llu differ (last 0x%08x)\n", hi - lo, b, e);
    return b != 0;
}
