// kv_kmer2bit_device.h -- every k-mer of a batch of equal-length reads, hashed from its 2-bit form.
//
// The per-k-mer kernels of rounds 1-2 (k_consume, k_bin_hash_direct, k_novel_mark) stage a tile's reads as ASCII of both strands in
// LDS and roll byte windows over it: ~470-540 lane-instructions per k-mer (profiles/r4_final/sq_counters_cfg4_band.txt).  The
// per-distinct-k-mer kernels of kv_skm.hip hash a k-mer from its packed key through a 256-entry byte -> 4 characters table
// (skm_key_hash): ~290.  Batches with nothing to deduplicate -- config 4's 0.6x batches under banding, where seven k-mers in
// eight are hashed only to learn that they belong to another band -- are all hashing, so they get that path too: a thread takes
// CH consecutive k-mers of one read, cuts the first out of the packed words, takes its reverse complement once and rolls both
// strands through the rest (two shifts each).  Needs reads of one length (the layout is then arithmetic) and 16 <= k <= 64.
#pragma once
#include "kv_novel_device.h"

// q / d for q < 2^16 with the precomputed float reciprocal (off by at most one before the fix-up)
__device__ __forceinline__ uint32_t k2_div(uint32_t q, uint32_t d, float inv)
{
    uint32_t r = (uint32_t)((float)q * inv);
    if (r * d > q) r -= 1;
    else if ((r + 1) * d <= q) r += 1;
    return r;
}

// body(live, hash): called CH times by every lane (wave-uniform, so the body may ballot); live = this lane has a k-mer
// FK != 0: k is FK, known when compiled (shifts and masks are constants) and `lut` holds the two product tables of skm_key_hash_pl
// (512 x 64 bits: P1 then P2) instead of the 256-entry ASCII table
template <int KW, int CH, int FK = 0, typename Body>
__device__ __forceinline__ void kmer2bit_walk(const uint32_t *read_words, uint32_t j0, uint32_t cnt, int k_run, const uint32_t *lut,
                                              const HashParams &hp, Body body)
{
    const int k = FK ? FK : k_run;
    SkmKey<KW> fw;
    fw.w[0] = skm_bases32(read_words, j0);
    if (KW == 2) fw.w[KW - 1] = skm_bases32(read_words, j0 + 32u) & skm_topmask<2>(k);
    else fw.w[0] &= skm_topmask<1>(k);
    SkmKey<KW> rc = skm_revcomp<KW>(fw, k);
    const uint64_t tail = skm_bases32(read_words, j0 + (uint32_t)k);          // the bases that enter k-mers 1 .. CH - 1 (CH <= 32)
#pragma unroll
    for (uint32_t u = 0; u < (uint32_t)CH; ++u) {
        if (u) skm_roll<KW>(fw, rc, (uint32_t)(tail >> (2u * (u - 1u))) & 3u, k);
        const bool live = u < cnt;
        uint64_t h = 0;
        if (live) {
            if constexpr (FK != 0) h = skm_key_hash_pl<KW, FK>(skm_canonical<KW>(fw, rc), (const uint64_t *)lut, (const uint64_t *)lut + 256);
            else h = skm_key_hash<KW>(skm_canonical<KW>(fw, rc), lut, hp);
        }
        body(live, h);
    }
}

// survivors of a wave's filter, collected in LDS and handed on 64 at a time, all lanes busy (the filter passes one k-mer in
// eight under 8-fold banding: taking the survivors where they fall would run what follows at an eighth of the lanes)
struct WaveQueue {
    unsigned long long *q;      // 128 entries, this wave's
    uint32_t n;                 // wave-uniform; < 64 between calls
};

template <typename Drain>
__device__ __forceinline__ void wave_queue_push(WaveQueue &wq, bool pass, uint64_t value, Drain drain)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long vote = __ballot(pass);
    if (pass) __hip_atomic_store(&wq.q[wq.n + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull))], (unsigned long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    wq.n += (uint32_t)__popcll(vote);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (wq.n >= 64u) {
        wq.n -= 64u;
        drain(true, (uint64_t)__hip_atomic_load(&wq.q[wq.n + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename Drain>
__device__ __forceinline__ void wave_queue_flush(WaveQueue &wq, Drain drain)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (wq.n) {
        const bool have = lane < wq.n;
        drain(have, have ? (uint64_t)__hip_atomic_load(&wq.q[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull);
        wq.n = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// the same with a second value beside each entry (the scan queues a k-mer's hash and where it sits)
struct WaveQueue2 {
    unsigned long long *q, *q2;
    uint32_t n;
};

template <typename Drain>
__device__ __forceinline__ void wave_queue_push2(WaveQueue2 &wq, bool pass, uint64_t a, uint64_t b, Drain drain)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long vote = __ballot(pass);
    if (pass) {
        const uint32_t at = wq.n + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull));
        __hip_atomic_store(&wq.q[at], (unsigned long long)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_store(&wq.q2[at], (unsigned long long)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    wq.n += (uint32_t)__popcll(vote);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (wq.n >= 64u) {
        wq.n -= 64u;
        drain(true, (uint64_t)__hip_atomic_load(&wq.q[wq.n + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT),
              (uint64_t)__hip_atomic_load(&wq.q2[wq.n + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename Drain>
__device__ __forceinline__ void wave_queue_flush2(WaveQueue2 &wq, Drain drain)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (wq.n) {
        const bool have = lane < wq.n;
        drain(have, have ? (uint64_t)__hip_atomic_load(&wq.q[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull,
              have ? (uint64_t)__hip_atomic_load(&wq.q2[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0ull);
        wq.n = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
