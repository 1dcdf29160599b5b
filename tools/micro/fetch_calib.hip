// fetch_calib.hip — what rocprofv3's FETCH_SIZE reports for the access pattern of the tree walks (a lane gathers one
// 64-byte record, four dwordx4 loads, from a table at a random position) next to a wide streaming read of the same
// table, for which the factor is known (gfx950 reports half the bytes of a coalesced 16-byte-per-lane stream).
//   fetch_calib stream   <table MiB>            every byte of the table once, 16 B per lane, coalesced
//   fetch_calib gather64 <table MiB> <gathers per lane>   random 64-byte records, one lane each
// Prints the bytes the kernel asks for; tools/fetch_calib.sh runs it under `rocprofv3 --pmc FETCH_SIZE` and divides.
// A table far larger than the L2s (32 MiB in all) makes nearly every gather a miss, so asked-for bytes = fetched bytes;
// an 11 MiB table (the size of the 83,334-object scene's records) shows what the L2s absorb.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_fill(uint4* t, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        t[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void k_stream(const uint4* __restrict__ t, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0xDEADBEEFu) out[0] = acc;   // (keeps the loads)
}
__device__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void k_gather64(const uint4* __restrict__ t, unsigned n_records, unsigned per_lane, unsigned* out) {
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0, r = hash32(id * 2654435761u + 12345u);
    for (unsigned k = 0; k < per_lane; ++k) {
        r = hash32(r + k);
        const uint4* rec = t + (size_t)(r % n_records) * 4u;    // the next record depends on nothing loaded: gathers overlap
        const uint4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0xDEADBEEFu) out[0] = acc;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: fetch_calib stream|gather64 <table MiB> [gathers per lane]\n"); return 2; }
    const bool gather = !std::strcmp(argv[1], "gather64");
    const size_t bytes = (size_t)std::atoll(argv[2]) << 20;
    const unsigned per_lane = argc > 3 ? (unsigned)std::atoi(argv[3]) : 64u;
    uint4* t = nullptr;
    unsigned* out = nullptr;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMalloc(&out, 256));
    const size_t n16 = bytes / 16;
    k_fill<<<2048, 256>>>(t, n16);
    CHECK(hipDeviceSynchronize());
    const unsigned grid = 256 * 8, block = 256;
    double asked;
    if (gather) {
        k_gather64<<<grid, block>>>(t, (unsigned)(bytes / 64), per_lane, out);
        asked = 64.0 * grid * block * per_lane;
    } else {
        k_stream<<<grid, block>>>(t, n16, out);
        asked = (double)bytes;
    }
    CHECK(hipDeviceSynchronize());
    std::printf("{\"mode\": \"%s\", \"table_mib\": %zu, \"bytes_asked_for\": %.0f}\n", argv[1], bytes >> 20, asked);
    return 0;
}
