// Issue rate of the integer instructions the count's drain is made of (gfx950): wave-instructions per cycle and SIMD for
// v_add_u32, v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32, v_fma_f64, v_lshlrev_b64, v_cmp_u64 + cndmask, 8 independent chains per lane,
// a grid that puts 8 waves on every SIMD.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(512, 8) void k_rate(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    uint64_t b[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7u + i; b[i] = ((uint64_t)a[i] << 32) | (a[i] * 3u); d[i] = (double)a[i]; }
    const uint32_t c = seed | 1u;
    const uint64_t c64 = ((uint64_t)seed << 33) | 0x9e3779b97f4a7c15ull;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b[i]) : "v"(a[i]), "v"(c) : "vcc");
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            if (OP == 5) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(b[i]));
            if (OP == 6) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(c));
            if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 9) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(b[i]) : "v"(c64));
            if (OP == 10) asm volatile("v_cmp_ne_u64 vcc, %0, %1" : : "v"(b[i]), "v"(c64) : "vcc");
            if (OP == 11) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a[i]));
            if (OP == 12) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(c));
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= a[i] ^ (uint32_t)b[i] ^ (uint32_t)(b[i] >> 32) ^ (uint32_t)d[i];
    if (acc == 0x12345u) out[0] = acc;
}
template <int OP>
double run(uint32_t *d_out, int cus, double mhz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * 4;          // 4 workgroups of 512 = 32 waves per CU = 8 per SIMD
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(512), 0, 0, d_out, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(grid), dim3(512), 0, 0, d_out, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = 8.0 * ITERS * 8.0;       // 8 waves x ITERS x 8 instructions
    const double cycles = ms * 1e-3 * mhz * 1e6;
    return cycles / insts_per_simd;
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double mhz = p.clockRate / 1000.0;
    uint32_t *d_out;
    hipMalloc(&d_out, 64);
    printf("%s, %d CUs, %.0f MHz (cycles per wave-instruction and SIMD at the nominal clock, 8 waves per SIMD)\n", p.name, p.multiProcessorCount, mhz);
    const char *names[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_fma_f64", "v_lshlrev_b64", "v_xor_b32", "v_alignbit_b32",
                           "v_add3_u32", "v_lshl_add_u64", "v_cmp_ne_u64", "v_bfe_u32", "v_mul_u32_u24", "v_mad_u32_u24"};
    double r[14];
    r[0] = run<0>(d_out, p.multiProcessorCount, mhz); r[1] = run<1>(d_out, p.multiProcessorCount, mhz); r[2] = run<2>(d_out, p.multiProcessorCount, mhz);
    r[3] = run<3>(d_out, p.multiProcessorCount, mhz); r[4] = run<4>(d_out, p.multiProcessorCount, mhz); r[5] = run<5>(d_out, p.multiProcessorCount, mhz);
    r[6] = run<6>(d_out, p.multiProcessorCount, mhz); r[7] = run<7>(d_out, p.multiProcessorCount, mhz); r[8] = run<8>(d_out, p.multiProcessorCount, mhz);
    r[9] = run<9>(d_out, p.multiProcessorCount, mhz); r[10] = run<10>(d_out, p.multiProcessorCount, mhz); r[11] = run<11>(d_out, p.multiProcessorCount, mhz);
    r[12] = run<12>(d_out, p.multiProcessorCount, mhz); r[13] = run<13>(d_out, p.multiProcessorCount, mhz);
    for (int i = 0; i < 14; ++i) printf("%-16s %.2f\n", names[i], r[i]);
    return 0;
}
