// Would S1 writing its records straight into the 64 k FINE buckets (one device-wide cursor per bucket, no S2) be affordable?
// (a) N returning device-scope atomicAdds on B random cursors; (b) the same followed by a 16-byte store at bucket[b][pos] -- what S1 would
// do per record; (c) the stores alone with the positions precomputed.  57 M records per sample at config 2, 64 256 buckets.
// hipcc --offload-arch=gfx950 -O3 -o scatter_rate scatter_rate.hip && ./scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; return x ^ (x >> 16); }
template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(unsigned int *cursors, uint4 *buckets, uint32_t nb, uint32_t cap, uint64_t n, unsigned int *sink)
{
    unsigned int acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * nb) >> 32);
        uint32_t pos;
        if (MODE == 2) pos = (uint32_t)(i / nb) % cap;                  // (no atomic: a precomputed slot)
        else pos = atomicAdd(&cursors[b], 1u);
        if (MODE >= 1) { if (pos < cap) buckets[(uint64_t)b * cap + pos] = make_uint4((uint32_t)i, b, pos, 7u); }
        else acc += pos;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const uint32_t nb = 64256, cap = 1400;
    const uint64_t n = 57000000;
    unsigned int *cursors, *sink; uint4 *buckets;
    hipMalloc(&cursors, nb * 4); hipMalloc(&sink, 64); hipMalloc(&buckets, (uint64_t)nb * cap * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"atomics only", "atomic + 16-byte store", "16-byte stores, slots precomputed"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int grid : {768, 3072}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(cursors, 0, nb * 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_scatter<0>, dim3(grid), dim3(256), 0, 0, cursors, buckets, nb, cap, n, sink);
                if (mode == 1) hipLaunchKernelGGL(k_scatter<1>, dim3(grid), dim3(256), 0, 0, cursors, buckets, nb, cap, n, sink);
                if (mode == 2) hipLaunchKernelGGL(k_scatter<2>, dim3(grid), dim3(256), 0, 0, cursors, buckets, nb, cap, n, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-36s grid %4d: %.3f ms for %.0f M records = %.1f G/s\n", names[mode], grid, best, n / 1e6, n / best / 1e6);
        }
    }
    return 0;
}
