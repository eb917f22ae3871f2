// kv_fastmod.h -- h % size for the table sizes of a sketch (khmer's primes: every bin of every table is hash % size,
// khmer Storage / kevlar/sketch.py:105-131), exactly, without a division.  Pure functions that compile for the host
// too (tests/test_skm_host.py checks them against the % operator).
//
// Two forms, chosen by the size alone so that every caller agrees:
//  * 2^16 <= size < 2^32 (every table from 64 KB to 4 GB of byte counters): the quotient comes out of the FP64 pipe.
//    h is rounded to a double (relative error 2^-53), multiplied by the correctly rounded 1.0 / size and rounded to an
//    integer by adding 2^52 -- one fused multiply-add.  The three roundings move h / size by at most
//    3 x 2^-53 x 2^64 / size <= 0.07, so the integer is floor(h / size) or one more, and the remainder needs ONE
//    correction.  12 instructions against the 30 of the 64 x 64 -> 128-bit Barrett form, four times per distinct k-mer in
//    the count kernel and once per probe in the scan.
//  * otherwise: Barrett with magic = floor((2^64 - 1) / size); the quotient is at most 2 short.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define KVF_HD __host__ __device__ __forceinline__
#else
#define KVF_HD static inline
#endif

#if defined(KV_FASTMOD_NO_FP)          // A/B builds (scratch/ab_build.py NAME -DKV_FASTMOD_NO_FP): Barrett for every size
KVF_HD bool kv_fastmod_fp(uint64_t) { return false; }
#else
KVF_HD bool kv_fastmod_fp(uint64_t size) { return size - 65536ull < 0xffff0000ull; }
#endif

// what SketchDev::magic holds for a table of `size` bins (host side)
static inline uint64_t kv_fastmod_magic(uint64_t size)
{
    if (kv_fastmod_fp(size)) {
        const double inv = 1.0 / (double)size;
        uint64_t bits;
        memcpy(&bits, &inv, 8);
        return bits;
    }
    return UINT64_MAX / size;
}

KVF_HD uint64_t kv_mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

KVF_HD uint64_t fastmod(uint64_t h, uint64_t size, uint64_t magic)
{
    if (kv_fastmod_fp(size)) {
        double inv;
#if defined(__HIP_DEVICE_COMPILE__)
        inv = __longlong_as_double((long long)magic);
#else
        memcpy(&inv, &magic, 8);
#endif
        const double hd = fma((double)(uint32_t)(h >> 32), 4294967296.0, (double)(uint32_t)h);
        const double t = fma(hd, inv, 4503599627370496.0);            // 2^52 + round(h / size): the integer sits in the mantissa
        uint64_t tb;
#if defined(__HIP_DEVICE_COMPILE__)
        tb = (uint64_t)__double_as_longlong(t);
#else
        memcpy(&tb, &t, 8);
#endif
        const uint64_t q = tb & 0xfffffffffffffull;
        const uint32_t p = (uint32_t)size;
        uint64_t r = h - q * p;
        r += (int64_t)r < 0 ? (uint64_t)p : 0ull;                       // the quotient was one too many
        return r;
    }
    uint64_t q = kv_mulhi64(h, magic);
    uint64_t r = h - q * size;
    r -= r >= size ? size : 0;       // twice, branch-free: the quotient is at most 2 short
    r -= r >= size ? size : 0;
    return r;
}
