// issue cost of the integer instructions the count / scan kernels are made of, relative to v_add_u32, on one MI355X:
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every kernel runs 8 waves per SIMD (2048 workgroups of 256 threads on 256 CUs), each wave ITERS x 16 independent
// instructions of one kind; cycles per wave-instruction = time x clock / (8 waves x ITERS x 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL32(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                 \
    {                                                                                                         \
        uint32_t a[8], b = seed + threadIdx.x, c = seed * 3u + 1u; uint64_t bb = ((uint64_t)b << 32) | c;      \
        for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;                                            \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            R16(ASM)                                                                                          \
        }                                                                                                     \
        uint32_t s = 0;                                                                                       \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                                \
        if (s == 0x12345678u) out[0] = s;                                                                     \
    }
#define KERNEL64(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t seed)                                 \
    {                                                                                                         \
        uint64_t a[8]; uint32_t b = seed + threadIdx.x, c = seed * 3u + 1u; uint64_t bb = ((uint64_t)b << 32) | c;     \
        for (int i = 0; i < 8; ++i) a[i] = ((uint64_t)(seed + i) << 32) + threadIdx.x;                         \
        for (int it = 0; it < ITERS; ++it) {                                                                  \
            R16(ASM)                                                                                          \
        }                                                                                                     \
        uint64_t s = 0;                                                                                       \
        for (int i = 0; i < 8; ++i) s ^= a[i];                                                                \
        if (s == 0x12345678u) out[0] = (uint32_t)s;                                                           \
    }

#define A_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_XOR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
#define A_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(a[i]) : "v"(b));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]));
#define A_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_BFREV(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
#define A_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
#define A_CMP32(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define A_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define A_ADDDPP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define A_BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(b));
#define A_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0" : "+v"(a[i]) : : "s20");
#define A_SAD(i) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));

#define C_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_SHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
#define C_SHLV(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define C_SHR(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
#define C_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define C_NOT(i) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
#define C_MAX(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_ADDK(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
#define C_ANDK(i) asm volatile("v_and_b32 %0, 0x3ffffff, %0" : "+v"(a[i]));
#define C_CMPSEL(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define C_CMPSELS(i) asm volatile("v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s21");
#define C_CMPEQ64(i) asm volatile("v_cmp_eq_u64 vcc, %0, %1" : : "v"(bb), "v"(bb) : "vcc");
#define C_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_FMAF(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define C_FMACF(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define C_XORS(i) asm volatile("v_xor_b32 %0, s20, %0" : "+v"(a[i]) : : );
#define C_ADDE64(i) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define C_SALU(i) asm volatile("s_add_u32 s20, s20, 7" : : : "s20");
#define C_MIX(i) asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 7" : "+v"(a[i]) : "v"(b) : "s20");
#define B_SHL64(i) asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(a[i]));
#define B_SHL64V(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(c));
#define B_SHR64(i) asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(a[i]));
#define B_MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
#define B_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[i]) : "v"(bb));
#define B_CMP64(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[i]), "v"(bb) : "vcc");
#define B_ADD64(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define B_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(bb));
#define B_MUL64F(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(bb));
#define B_CVTF64(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(b));
#define B_CVTU32(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(a[i]) : "v"(bb));
#define B_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define B_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(a[i]) : "v"(bb));

KERNEL32(k_add, A_ADD) KERNEL32(k_xor, A_XOR) KERNEL32(k_xor3, A_XOR3) KERNEL32(k_add3, A_ADD3) KERNEL32(k_lshladd, A_LSHLADD)
KERNEL32(k_lshlor, A_LSHLOR) KERNEL32(k_andor, A_ANDOR) KERNEL32(k_mullo, A_MULLO) KERNEL32(k_mulhi, A_MULHI) KERNEL32(k_mul24, A_MUL24)
KERNEL32(k_mulhi24, A_MULHI24) KERNEL32(k_mad24, A_MAD24) KERNEL32(k_alignbit, A_ALIGNBIT) KERNEL32(k_perm, A_PERM) KERNEL32(k_bfe, A_BFE)
KERNEL32(k_bfi, A_BFI) KERNEL32(k_bcnt, A_BCNT) KERNEL32(k_bfrev, A_BFREV) KERNEL32(k_cndmask, A_CNDMASK) KERNEL32(k_cmp32, A_CMP32)
KERNEL32(k_minu, A_MINU) KERNEL32(k_dpp, A_DPP) KERNEL32(k_adddpp, A_ADDDPP) KERNEL32(k_bperm, A_BPERM) KERNEL32(k_readlane, A_READLANE) KERNEL32(k_sad, A_SAD)
KERNEL64(k_shl64, B_SHL64) KERNEL64(k_shl64v, B_SHL64V) KERNEL64(k_shr64, B_SHR64) KERNEL64(k_mad64, B_MAD64) KERNEL64(k_lshladd64, B_LSHLADD64)
KERNEL64(k_cmp64, B_CMP64) KERNEL32(k_add64, B_ADD64) KERNEL64(k_fma64, B_FMA64) KERNEL64(k_mul64f, B_MUL64F) KERNEL64(k_cvtf64, B_CVTF64)
KERNEL32(k_cvtu32, B_CVTU32) KERNEL32(k_pkadd, B_PKADD) KERNEL64(k_mov64, B_MOV64)

KERNEL32(k_c_and, C_AND) KERNEL32(k_c_or, C_OR) KERNEL32(k_c_sub, C_SUB) KERNEL32(k_c_shl, C_SHL) KERNEL32(k_c_shlv, C_SHLV) KERNEL32(k_c_shr, C_SHR) KERNEL32(k_c_mov, C_MOV) KERNEL32(k_c_not, C_NOT) KERNEL32(k_c_max, C_MAX) KERNEL32(k_c_addk, C_ADDK) KERNEL32(k_c_andk, C_ANDK) KERNEL32(k_c_cmpsel, C_CMPSEL) KERNEL32(k_c_cmpsels, C_CMPSELS) KERNEL32(k_c_cmpeq64, C_CMPEQ64) KERNEL32(k_c_addf, C_ADDF) KERNEL32(k_c_fmaf, C_FMAF) KERNEL32(k_c_fmacf, C_FMACF) KERNEL32(k_c_xors, C_XORS) KERNEL32(k_c_adde64, C_ADDE64) KERNEL32(k_c_salu, C_SALU) KERNEL32(k_c_mix, C_MIX)
static double g_base = 0;
static void run(const char *name, void (*kern)(uint32_t *, uint32_t), uint32_t *out, int per)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, out, (uint32_t)rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    const double ns_per = best * 1e6 / (8.0 * ITERS * 16 * per);      // per wave-instruction on one SIMD
    if (g_base == 0) g_base = ns_per;
    printf("%-24s %8.3f ms  %6.3f ns per wave-instruction  = %5.2f x v_add_u32 (%4.1f cycles if v_add_u32 is 2)\n", name, best, ns_per, ns_per / g_base, 2.0 * ns_per / g_base);
}
int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    uint32_t *out; hipMalloc(&out, 64);
    run("v_add_u32", k_add, out, 1); run("v_xor_b32", k_xor, out, 1); run("v_or3_b32", k_xor3, out, 1); run("v_add3_u32", k_add3, out, 1);
    run("v_lshl_add_u32", k_lshladd, out, 1); run("v_lshl_or_b32", k_lshlor, out, 1); run("v_and_or_b32", k_andor, out, 1);
    run("v_mul_lo_u32", k_mullo, out, 1); run("v_mul_hi_u32", k_mulhi, out, 1); run("v_mul_u32_u24", k_mul24, out, 1); run("v_mul_hi_u32_u24", k_mulhi24, out, 1);
    run("v_mad_u32_u24", k_mad24, out, 1); run("v_alignbit_b32", k_alignbit, out, 1); run("v_perm_b32", k_perm, out, 1); run("v_bfe_u32", k_bfe, out, 1);
    run("v_bfi_b32", k_bfi, out, 1); run("v_bcnt_u32_b32", k_bcnt, out, 1); run("v_bfrev_b32", k_bfrev, out, 1); run("v_cndmask_b32", k_cndmask, out, 1);
    run("v_cmp_lt_u32", k_cmp32, out, 1); run("v_min_u32", k_minu, out, 1); run("v_mov_b32_dpp", k_dpp, out, 1); run("v_add_u32_dpp", k_adddpp, out, 1);
    run("ds_bpermute_b32+wait", k_bperm, out, 1); run("v_readlane+v_add", k_readlane, out, 1); run("v_sad_u32", k_sad, out, 1);
    run("v_lshlrev_b64 imm", k_shl64, out, 1); run("v_lshlrev_b64 vgpr", k_shl64v, out, 1); run("v_lshrrev_b64", k_shr64, out, 1); run("v_mad_u64_u32", k_mad64, out, 1);
    run("v_lshl_add_u64", k_lshladd64, out, 1); run("v_cmp_lt_u64", k_cmp64, out, 1); run("v_add_co+v_addc_co (2)", k_add64, out, 1);
    run("v_fma_f64", k_fma64, out, 1); run("v_mul_f64", k_mul64f, out, 1); run("v_cvt_f64_u32", k_cvtf64, out, 1); run("v_cvt_u32_f64", k_cvtu32, out, 1);
    run("v_pk_add_u16", k_pkadd, out, 1); run("v_mov_b64", k_mov64, out, 1);
    run("and", k_c_and, out, 1); run("or", k_c_or, out, 1); run("sub", k_c_sub, out, 1); run("shl", k_c_shl, out, 1); run("shlv", k_c_shlv, out, 1); run("shr", k_c_shr, out, 1); run("mov", k_c_mov, out, 1); run("not", k_c_not, out, 1); run("max", k_c_max, out, 1); run("addk", k_c_addk, out, 1); run("andk", k_c_andk, out, 1); run("cmpsel", k_c_cmpsel, out, 1); run("cmpsels", k_c_cmpsels, out, 1); run("cmpeq64", k_c_cmpeq64, out, 1); run("addf", k_c_addf, out, 1); run("fmaf", k_c_fmaf, out, 1); run("fmacf", k_c_fmacf, out, 1); run("xors", k_c_xors, out, 1); run("adde64", k_c_adde64, out, 1); run("salu", k_c_salu, out, 1); run("mix", k_c_mix, out, 1);
    return 0;
}
