// can S1 scatter its records straight into the fine buckets through global cursors?  57 M returning atomics on 130 k addresses
// + 57 M 24-byte stores: hipcc --offload-arch=gfx950 -O3 -o scatter scatter.hip && ./scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t *cur, uint64_t *seg, uint32_t nb, uint32_t cap, uint64_t n)
{
    const uint64_t tid = blockIdx.x * 512ull + threadIdx.x, nt = (uint64_t)gridDim.x * 512ull;
    uint64_t acc = 0;
    for (uint64_t i = tid; i < n; i += nt) {
        const uint32_t b = (uint32_t)(((uint64_t)mix((uint32_t)i * 2654435761u + 12345u) * nb) >> 32);
        uint32_t pos = 0;
        if (MODE & 1) pos = atomicAdd(&cur[b], 1u);
        else pos = (uint32_t)(i / nb);
        if (MODE & 2) {
            if (pos < cap) {
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
                uint64_t *dst = seg + ((uint64_t)b * cap + pos) * 3;
                *(u64x2 *)dst = u64x2{i, i ^ 0x55};
                dst[2] = i + 7;
            }
        } else acc += pos;
    }
    if (acc == 0x123456789ull) cur[0] = 1;
}
template <int MODE>
void run(const char *what, uint32_t *cur, uint64_t *seg, uint32_t nb, uint32_t cap, uint64_t n)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(cur, 0, nb * 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(768), dim3(512), 0, 0, cur, seg, nb, cap, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("%-40s %.3f ms (%.1f G/s)\n", what, ms, n / ms / 1e6);
    }
}
int main()
{
    const uint32_t nb = 255 * 512; const uint64_t n = 57000000; const uint32_t cap = (uint32_t)(n / nb * 3 / 2 + 64);
    uint32_t *cur; uint64_t *seg;
    hipMalloc(&cur, nb * 4); hipMalloc(&seg, (uint64_t)nb * cap * 24);
    run<1>("returning atomics only", cur, seg, nb, cap, n);
    run<2>("stores only (position arithmetic)", cur, seg, nb, cap, n);
    run<3>("atomics + 24-byte stores", cur, seg, nb, cap, n);
    return 0;
}
