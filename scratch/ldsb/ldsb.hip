// LDS op throughput microbenchmark (random addresses): lanes per clock per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N_ITER 4096
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(uint32_t *out, uint32_t nbins, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < nbins; i += THREADS) lds[i] = 0;
    __syncthreads();
    uint32_t x = seed ^ (blockIdx.x * 9781u + threadIdx.x * 6271u + 1u);
    uint32_t acc = 0;
    for (int it = 0; it < N_ITER; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = (x >> 8) & (nbins - 1);
        if (MODE == 0) acc += atomicAdd(&lds[a], 1u);                    // returning add
        else if (MODE == 1) atomicAdd(&lds[a], 1u);                      // non-returning add
        else if (MODE == 2) lds[a] = x;                                  // plain write
        else if (MODE == 3) acc += lds[a];                               // plain read
        else if (MODE == 4) { uint32_t old = lds[a]; acc += atomicCAS(&lds[a], old, old + 1); }   // read + CAS
        else if (MODE == 5) acc += __builtin_amdgcn_ds_bpermute((int)((x >> 8) & 63) << 2, (int)x);
        else if (MODE == 6) { atomicAdd(&lds[a], 1u << ((x & 3) * 8)); }  // non-returning, byte lane
        else if (MODE == 7) { acc += a; }                                 // ALU only baseline
    }
    __syncthreads();
    if (acc == 0xdeadbeef || MODE == 1 || MODE == 2 || MODE == 6) out[blockIdx.x * THREADS + threadIdx.x] = acc + lds[threadIdx.x % nbins];
}
template <int MODE>
void run(const char *name, uint32_t nbins, int wgs_per_cu)
{
    constexpr int THREADS = 512;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clock_khz = 0;
    hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    uint32_t *out;
    const int grid = cus * wgs_per_cu;
    hipMalloc(&out, (size_t)grid * THREADS * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), nbins * 4, 0, out, nbins, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), nbins * 4, 0, out, nbins, 7u + r);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double ops = (double)grid * THREADS * N_ITER;
    printf("%-28s bins %6u wg/cu %d : %7.3f ms  %7.1f Gops/s  %5.2f lanes/clk/CU (at %d MHz)\n", name, nbins, wgs_per_cu, ms,
           ops / ms / 1e6, ops / (ms * 1e-3) / cus / (clock_khz * 1e3), clock_khz / 1000);
    hipFree(out);
}
int main()
{
    for (uint32_t nbins : {16384u, 128u}) {
        for (int w : {2, 4}) {
            run<7>("alu only", nbins, w);
            run<0>("atomicAdd returning", nbins, w);
            run<1>("atomicAdd non-returning", nbins, w);
            run<6>("atomicAdd nonret byte-lane", nbins, w);
            run<2>("plain write", nbins, w);
            run<3>("plain read", nbins, w);
            run<4>("read + CAS", nbins, w);
            run<5>("bpermute", nbins, w);
        }
    }
    return 0;
}
