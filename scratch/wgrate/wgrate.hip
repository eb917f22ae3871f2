// what a grid of many short workgroups costs on MI355X: hipcc --offload-arch=gfx950 -O3 -o wgrate wgrate.hip && ./wgrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int THREADS, int LDSB, int MODE>
__global__ __launch_bounds__(THREADS) void k(const uint32_t *in, uint32_t *out, uint8_t *tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDSB / 4];
    uint32_t v = 0;
    if (MODE >= 1) { if (threadIdx.x < 32) v = in[blockIdx.x * 32 + threadIdx.x]; lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 1) % THREADS]; }
    if (MODE >= 2) { for (int j = threadIdx.x; j < LDSB / 16; j += THREADS) ((uint4 *)lds)[j] = make_uint4(v, 0, 0, 0); __syncthreads(); }
    if (MODE >= 3) { for (int j = threadIdx.x; j < LDSB / 16; j += THREADS) ((uint4 *)(tab + (size_t)blockIdx.x * LDSB))[j] = ((uint4 *)lds)[j]; }
    if (v == 0xdeadbeef) out[0] = v;
}
template <int THREADS, int LDSB, int MODE>
void run(int nblocks, const uint32_t *in, uint32_t *out, uint8_t *tab)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<THREADS, LDSB, MODE>), dim3(nblocks), dim3(THREADS), 0, 0, in, out, tab);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("threads %4d lds %6d mode %d blocks %6d: %.3f ms (%.1f blocks/us, %.1f GB/s written)\n", THREADS, LDSB, MODE, nblocks, ms, nblocks / ms / 1e3,
                             MODE >= 3 ? (double)nblocks * LDSB / ms / 1e6 : 0.0);
    }
}
int main()
{
    const int nb = 30520;
    uint32_t *in, *out; uint8_t *tab;
    hipMalloc(&in, (size_t)nb * 4 * 32 * 4); hipMemset(in, 0, (size_t)nb * 4 * 32 * 4); hipMalloc(&out, 64); hipMalloc(&tab, (size_t)nb * 65536);
    run<1024, 65536, 0>(nb, in, out, tab); run<1024, 65536, 1>(nb, in, out, tab); run<1024, 65536, 2>(nb, in, out, tab); run<1024, 65536, 3>(nb, in, out, tab);
    run<512, 32768, 0>(2 * nb, in, out, tab); run<512, 32768, 1>(2 * nb, in, out, tab); run<512, 32768, 2>(2 * nb, in, out, tab); run<512, 32768, 3>(2 * nb, in, out, tab);
    run<256, 16384, 0>(4 * nb, in, out, tab); run<256, 16384, 2>(4 * nb, in, out, tab); run<256, 16384, 3>(4 * nb, in, out, tab);
    run<1024, 1024 * 4, 0>(nb, in, out, tab); run<256, 1024, 0>(4 * nb, in, out, tab); run<64, 256, 0>(16 * nb, in, out, tab);
    return 0;
}
