// kv_novel_device.h -- the abundance test of the novel scan, kmer_is_interesting() (kevlar/novel.py:21-53), shared by
// the tile scan (kv_novel.hip) and the per-distinct-k-mer scan over super-k-mer buckets (kv_skm.hip).
#pragma once
#include "kv_device.h"
#include "kv_skm_device.h"

struct NovelParams {
    HashParams hp;
    int ncase, nctrl;
    const SketchDev *sk[KV_MAX_SAMPLES];  // cases first, then controls
    int case_min, ctrl_max, screen;
    int band_mode, nbands, band;
    uint64_t band_lo, band_hi;
    uint64_t first_read;
    uint32_t *disc_first;   // per read: offset of its first k-mer that trips the abundance screen, else ~0 (NULL: screen off)
    uint32_t *mask;         // bit (read * mask_stride + offset)
    uint64_t mask_stride;
    uint32_t *tile_count;   // hits per tile
    const uint64_t *tile_base;
    uint32_t *hit_read, *hit_off;
    uint8_t *hit_abund;
    unsigned long long *vcache;   // hashes proven rejected by a control (NULL = off)
    int vcache_shift;             // slot = h >> shift
    int vcache_sets;              // k_novel_mark: 8-entry sets indexed by the k-mer's minimizer (0 = direct-mapped by hash)
    uint32_t vcache_set_mask;     // number of sets - 1
    int vcache_window;            // m-mers per k-mer considered for the minimizer
    int vcache_2bit;              // k_novel_mark_2bit looks the hash up too (8-entry sets indexed by the hash): the batch is one of many of its
                                  // sample, so an inherited k-mer met in an earlier batch costs one 64-byte request instead of five
    // Set mode (kv_novel_scan_set; the read-sharded multi-GPU scan): the interesting k-mers are already known -- the
    // owners of the hash bands evaluated them -- and arrive as an open-addressing table of their hashes with the S
    // abundances beside each; "interesting" is then membership, and no sketch is touched (sk[] unset, ncase = S).
    const unsigned long long *set_keys;   // NULL = off; empty slots hold ~0 (no k-mer of any band hashes to it)
    const uint8_t *set_abund;             // [slots][S]
    uint64_t set_mask;                    // slots - 1
    // Abundances of the interesting k-mers, kept by the scan that found them (k_skm_novel_list) for the kernel that reports them
    // hit by hit (k_hit_abund): an open-addressing table of hashes (0 = empty) with the S abundances beside each.  A k-mer that is
    // not in there -- the table was full where it wanted to go, or the scan took another path -- is simply probed again.
    unsigned long long *ab_keys;          // NULL = off
    uint8_t *ab_vals;                     // [slots][S]
    uint64_t ab_mask;                     // slots - 1
    uint32_t *ab_list;                    // the slots claimed, in the order they were (k_ab_fill visits these, not all 8 M slots); [0] of
    unsigned int *ab_count;               // ab_count counts them, past ab_list_cap k_ab_fill walks the table after all
    uint32_t ab_list_cap;
    const void *host_ctrls;               // host side only: the control sketches (kv_sketch *const *) behind sk[ncase..], for their
    int host_nctrl;                       // abundance lists (kv_skm_novel_mark)
    const void *host_case0;               // host side only: the first case sketch (kv_sketch *), for case0_bits
    // One bit per bin of table 0 of the first case sample: counter >= case_min (k_case_bits, a streaming pass in front of the list scan).
    // The scan's first probe -- where a sequencing-error k-mer ends: 81 M of them at config 2 -- asks exactly that, and a bit map is an
    // eighth of the table (62.5 MB for 500 M bins: it stays in the 256 MB Infinity Cache where the table does not).  NULL = probe the table.
    const uint32_t *case0_bits;
};
#define KV_SET_NONE 0xffffffffffffffffull

// Table descriptors of every sample, copied to LDS once per workgroup: the probe loops then read
// sizes / reciprocals / base pointers with broadcast LDS loads instead of three dependent global
// loads per probe (which is what the compiler emits for `p.sk[c]->size[t]` inside a divergent loop).
struct ProbeDesc {
    uint64_t size, magic;
    const uint8_t *tab;
};
struct NovelShared {
    ProbeDesc d[KV_MAX_SAMPLES * KV_MAX_TABLES];
    int ntab[KV_MAX_SAMPLES];
    int storage[KV_MAX_SAMPLES];
};

namespace {

// (shared by the super-k-mer kernels of kv_skm.hip and the hit kernels of kv_novel.hip)
// the two murmur hashes of a canonical k-mer: both strands expanded to ASCII register windows through the
// 256-entry byte -> 4 characters table in LDS
template <int KW>
__device__ __forceinline__ uint64_t skm_key_hash(const SkmKey<KW> &c, const uint32_t *lut, const HashParams &hp)
{
    constexpr int NW = 8 * KW;
    const SkmKey<KW> r = skm_revcomp<KW>(c, hp.k);
    uint32_t wf[NW], wr[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        wf[q] = lut[(uint32_t)(c.w[q >> 3] >> (8 * (q & 7))) & 0xffu];
        wr[q] = lut[(uint32_t)(r.w[q >> 3] >> (8 * (q & 7))) & 0xffu];
    }
    return murmur_regs<NW>(wf, hp) ^ murmur_regs<NW>(wr, hp);
}

// ---- the same two murmurs without their first multiplications (k known when compiled) -----------------------------------------
// MurmurHash3 starts on every 8 bytes of key with a multiplication by a constant (k1 *= c1, k2 *= c2).  Eight bytes of a k-mer's
// ASCII are two BYTES of its 2-bit form, four bases each, and the product is linear in them modulo 2^64:
//   (w0 | w1 << 32) * c  =  w0 * c  +  ((w1 * c) << 32)
// so with two 256-entry tables in LDS -- P1[b] = ascii4(b) * c1, P2[b] = ascii4(b) * c2, 64 bits each (4 KB for both) -- the product
// is a 64-bit and a 32-bit look-up and one addition in place of two look-ups of the ASCII table, a 64 x 64-bit multiplication (a
// v_mad_u64_u32 and two v_mul_lo_u32: integer multiplications issue at a quarter of the rate of the other integer instructions, and
// they are what the drain of k_skm_count is made of) and the additions that put it together: 24 of a k-mer's 88 multiplications.
// A word of which the k-mer fills only v < 4 characters (its last) is looked up with the other bases zero, i.e. with 'A' in their
// place, and what those 'A's contribute is a constant taken off again (murmur3 masks those bytes to zero).
KV_HD uint64_t skm_ascii4_times(uint32_t byte, uint64_t c) { return (uint64_t)skm_ascii4(byte) * c; }

template <int K> __host__ __device__ constexpr int pl_valid(int q) { return K - 4 * q >= 4 ? 4 : (K - 4 * q > 0 ? K - 4 * q : 0); }
__host__ __device__ constexpr uint64_t pl_corr(int valid, uint64_t c)
{
    return valid >= 4 ? 0ull : (uint64_t)(0x41414141u & ~((1u << (8 * valid)) - 1u)) * c;
}

// (ASCII words QA and QA + 1 of the k-mer as one little-endian u64, masked to the k-mer's length) * C, from the table P of that C
template <int KW, int K, int QA>
__device__ __forceinline__ uint64_t pl_mul(const SkmKey<KW> &c, const uint64_t *P, uint64_t C)
{
    constexpr int v0 = pl_valid<K>(QA), v1 = pl_valid<K>(QA + 1);
    static_assert(v0 > 0, "a murmur lane without a byte");
    const uint32_t b0 = (uint32_t)(c.w[QA >> 3] >> (8 * (QA & 7))) & (v0 == 4 ? 0xffu : ((1u << (2 * v0)) - 1u));
    uint64_t x = P[b0] - pl_corr(v0, C);
    if (v1 > 0) {
        const uint32_t b1 = (uint32_t)(c.w[(QA + 1) >> 3] >> (8 * ((QA + 1) & 7))) & (v1 == 4 ? 0xffu : ((1u << (2 * v1)) - 1u));
        x += (uint64_t)((uint32_t)P[b1] - (uint32_t)pl_corr(v1, C)) << 32;
    }
    return x;
}

template <int KW, int K, int B>
__device__ __forceinline__ void pl_block(const SkmKey<KW> &c, const uint64_t *P1, const uint64_t *P2, uint64_t &h1, uint64_t &h2)
{
    uint64_t k1 = pl_mul<KW, K, 4 * B>(c, P1, MM_C1), k2 = pl_mul<KW, K, 4 * B + 2>(c, P2, MM_C2);
    k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
}

template <int KW, int K>
__device__ __forceinline__ uint64_t murmur_pl(const SkmKey<KW> &c, const uint64_t *P1, const uint64_t *P2)
{
    constexpr int nb = K / 16, rem = K % 16;
    static_assert(K >= 1 && K <= 32 * KW && nb <= 4, "k out of the key's range");
    uint64_t h1 = 0, h2 = 0;
    if constexpr (nb > 0) pl_block<KW, K, 0>(c, P1, P2, h1, h2);
    if constexpr (nb > 1) pl_block<KW, K, 1>(c, P1, P2, h1, h2);
    if constexpr (nb > 2) pl_block<KW, K, 2>(c, P1, P2, h1, h2);
    if constexpr (nb > 3) pl_block<KW, K, 3>(c, P1, P2, h1, h2);
    if constexpr (rem > 8) {
        uint64_t k2 = pl_mul<KW, K, 4 * nb + 2>(c, P2, MM_C2);
        k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
    }
    if constexpr (rem > 0) {
        uint64_t k1 = pl_mul<KW, K, 4 * nb>(c, P1, MM_C1);
        k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    }
    return mm_final(h1, h2, K);
}

// skm_key_hash for a k known when compiled, from the product tables (P1 / P2: 256 u64 each, filled with skm_ascii4_times)
template <int KW, int K>
__device__ __forceinline__ uint64_t skm_key_hash_pl(const SkmKey<KW> &c, const uint64_t *P1, const uint64_t *P2)
{
    const SkmKey<KW> r = skm_revcomp<KW>(c, K);
    return murmur_pl<KW, K>(c, P1, P2) ^ murmur_pl<KW, K>(r, P1, P2);
}

__device__ __forceinline__ bool band_pass(const NovelParams &p, uint64_t h)
{
    if (p.band_mode == KV_BAND_RANGE) return h >= p.band_lo && h < p.band_hi;
    if (p.band_mode == KV_BAND_REFQUIRK) return (h & (uint64_t)(p.nbands - 1)) == (uint64_t)(int64_t)(p.band - 1);
    return true;
}

// slot of hash h in the set, or KV_SET_NONE
__device__ __forceinline__ uint64_t set_find(const NovelParams &p, uint64_t h)
{
    for (uint64_t slot = (h ^ (h >> 32)) & p.set_mask;; slot = (slot + 1) & p.set_mask) {
        const unsigned long long key = p.set_keys[slot];
        if (key == (unsigned long long)h) return slot;
        if (key == KV_SET_NONE) return KV_SET_NONE;
    }
}

__device__ __forceinline__ void load_descs(NovelShared &ns, const NovelParams &p)
{
    if (p.set_keys) return;
    const int S = p.ncase + p.nctrl;
    for (int i = threadIdx.x; i < S * KV_MAX_TABLES; i += blockDim.x) {
        const int c = i / KV_MAX_TABLES, t = i % KV_MAX_TABLES;
        const SketchDev *s = p.sk[c];
        if (t < s->ntables) { ns.d[i].size = s->size[t]; ns.d[i].magic = s->magic[t]; ns.d[i].tab = s->tab[t]; }
        if (t == 0) { ns.ntab[c] = s->ntables; ns.storage[c] = s->storage; }
    }
}

__device__ __forceinline__ uint32_t probe(const NovelShared &ns, int c, int t, uint64_t h)
{
    const ProbeDesc &d = ns.d[c * KV_MAX_TABLES + t];
    const uint64_t bin = fastmod(h, d.size, d.magic);
    const int st = ns.storage[c];
    // the pointer comes out of LDS as a generic one: say that it is global, or every probe is a FLAT load that also
    // waits for the LDS counter
    const __attribute__((address_space(1))) uint8_t *tab = (const __attribute__((address_space(1))) uint8_t *)d.tab;
    if (st == ST_BYTE) return tab[bin];
    if (st == ST_NIBBLE) return (tab[bin >> 1] >> ((bin & 1) ? 0 : 4)) & 15u;
    return (tab[bin >> 3] >> (bin & 7)) & 1u;
}

// The abundance test (screen off).  Same predicate as kmer_is_interesting(), cheapest evidence first:
// a control passes as soon as ONE table is <= ctrl_max (its Count-Min minimum is then <= ctrl_max) and
// rejects only after all T exceed it; a case fails as soon as ONE table is < case_min.  Measured: the
// scan is bound by the rate of random 64-B requests (~55 G/s), so probes are spent one at a time --
// issuing a control's T probes together was slower.
//
// Verdict cache: whether a k-mer is rejected by the controls is a pure function of its 64-bit hash
// (every bin derives from it), and an inherited k-mer recurs once per unit of coverage.  A direct-
// mapped table of hashes already proven "rejected by a control" turns its T probes into one.  Entries
// are single 8-byte words, races only cost a re-evaluation, a wrong answer is impossible: a slot
// either holds exactly this hash (proven) or it does not (0 = empty; hash 0 itself is never cached).
__device__ __forceinline__ bool novel_test_fast(const NovelShared &ns, const NovelParams &p, uint64_t h,
                                                unsigned long long *slot, unsigned long long cached)
{
    if (p.set_keys) return h != KV_SET_NONE && set_find(p, h) != KV_SET_NONE;
    // 0 marks an empty cache entry, so the (one) k-mer hash 0 is never cached: it is always evaluated
    if (slot && h != 0 && cached == h) return false;
    // table 0 of every case first: a sequencing-error k-mer (case count 1) leaves here after one probe
    for (int c = 0; c < p.ncase; ++c)
        if ((int)probe(ns, c, 0, h) < p.case_min) return false;
    for (int c = p.ncase; c < p.ncase + p.nctrl; ++c) {
        const int T = ns.ntab[c];
        bool pass = false;
        for (int t = 0; t < T && !pass; ++t) pass = (int)probe(ns, c, t, h) <= p.ctrl_max;
        if (!pass) {
            if (slot && h != 0) __hip_atomic_store(slot, (unsigned long long)h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    for (int c = 0; c < p.ncase; ++c) {
        const int T = ns.ntab[c];
        for (int t = 1; t < T; ++t)
            if ((int)probe(ns, c, t, h) < p.case_min) return false;
    }
    return true;
}

// The same predicate for the few k-mers that survive the first probe of the list scan (k_skm_novel_list): there the chain of
// dependent probes novel_test_fast spends -- the case's other tables, then control after control -- is a round trip to HBM each in
// front of a workgroup barrier, so the probes that decide nearly every k-mer are requested together: up to eight tables of the first
// case sample and table 0 of up to eight controls.  Whatever they leave open (a control whose table 0 exceeds ctrl_max, further
// cases, more tables than eight) is probed one at a time as before.  Over-fetching is free here: ~2 candidates per bucket.
__device__ __forceinline__ bool novel_test_wide(const NovelShared &ns, const NovelParams &p, uint64_t h)
{
    constexpr int W = 8;
    const int T0 = ns.ntab[0];
    uint32_t cv[W], c0[W];
#pragma unroll
    for (int t = 0; t < W; ++t) cv[t] = t < T0 ? probe(ns, 0, t, h) : 255u;
#pragma unroll
    for (int c = 0; c < W; ++c) c0[c] = c < p.nctrl ? probe(ns, p.ncase + c, 0, h) : 0u;
    bool ok = true;
#pragma unroll
    for (int t = 0; t < W; ++t) ok = ok && (int)cv[t] >= p.case_min;
    if (!ok) return false;
    for (int t = W; t < T0; ++t)
        if ((int)probe(ns, 0, t, h) < p.case_min) return false;
    for (int c = 1; c < p.ncase; ++c) {
        const int T = ns.ntab[c];
        for (int t = 0; t < T; ++t)
            if ((int)probe(ns, c, t, h) < p.case_min) return false;
    }
#pragma unroll
    for (int c = 0; c < W; ++c) {                   // (unrolled: c0[] stays in registers)
        if (c < p.nctrl) {
            bool pass = (int)c0[c] <= p.ctrl_max;
            const int T = ns.ntab[p.ncase + c];
            for (int t = 1; t < T && !pass; ++t) pass = (int)probe(ns, p.ncase + c, t, h) <= p.ctrl_max;
            if (!pass) return false;
        }
    }
    for (int c = W; c < p.nctrl; ++c) {
        bool pass = false;
        const int T = ns.ntab[p.ncase + c];
        for (int t = 0; t < T && !pass; ++t) pass = (int)probe(ns, p.ncase + c, t, h) <= p.ctrl_max;
        if (!pass) return false;
    }
    return true;
}

// the S abundances reported with a hit: Count-Min minimum of every sample, or what the set carries
__device__ __forceinline__ void hit_abundances(const NovelShared &ns, const NovelParams &p, uint64_t h, uint8_t *out)
{
    const int S = p.ncase + p.nctrl;
    if (p.set_keys) {
        const uint64_t slot = set_find(p, h);
        for (int c = 0; c < S; ++c) out[c] = slot != KV_SET_NONE ? p.set_abund[slot * (uint64_t)S + c] : 0;
        return;
    }
    // a sample's T probes are independent loads (descriptors come from LDS): issued together, then reduced
    for (int c = 0; c < S; ++c) {
        const int T = ns.ntab[c];
        uint32_t v[KV_MAX_TABLES];
#pragma unroll
        for (int t = 0; t < KV_MAX_TABLES; ++t) v[t] = t < T ? probe(ns, c, t, h) : 255u;
        uint32_t best = 255u;
#pragma unroll
        for (int t = 0; t < KV_MAX_TABLES; ++t) best = v[t] < best ? v[t] : best;
        out[c] = (uint8_t)best;
    }
}

#define KV_AB_PROBES 32
__device__ __forceinline__ uint64_t ab_slot(const NovelParams &p, uint64_t h) { return (h ^ (h >> 29)) & p.ab_mask; }

// the S bytes the abundances of hash h go to, or nullptr (no room near its slot)
__device__ __forceinline__ uint8_t *ab_claim(const NovelParams &p, uint64_t h, int S)
{
    if (h == 0) return nullptr;
    uint64_t slot = ab_slot(p, h);
    for (int probe = 0; probe < KV_AB_PROBES; ++probe, slot = (slot + 1) & p.ab_mask) {
        const unsigned long long old = atomicCAS(&p.ab_keys[slot], 0ull, (unsigned long long)h);
        if (old == 0ull && p.ab_list) {
            const unsigned int at = atomicAdd(p.ab_count, 1u);
            if (at < p.ab_list_cap) p.ab_list[at] = (uint32_t)slot;
        }
        if (old == 0ull || old == (unsigned long long)h) return p.ab_vals + slot * (uint64_t)S;       // (two k-mers with one hash have one set of abundances)
    }
    return nullptr;
}

__device__ __forceinline__ bool ab_lookup(const NovelParams &p, uint64_t h, uint8_t *out, int S)
{
    if (h == 0) return false;
    uint64_t slot = ab_slot(p, h);
    for (int probe = 0; probe < KV_AB_PROBES; ++probe, slot = (slot + 1) & p.ab_mask) {
        const unsigned long long key = p.ab_keys[slot];
        if (key == (unsigned long long)h) {
            for (int c = 0; c < S; ++c) out[c] = p.ab_vals[slot * (uint64_t)S + c];
            return true;
        }
        if (key == 0ull) return false;
    }
    return false;
}

}  // namespace
