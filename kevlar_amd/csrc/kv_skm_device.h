// kv_skm_device.h -- bit-level building blocks of the super-k-mer front end (kv_skm.hip).  Everything here is
// a pure function of its arguments and compiles for the host as well (tests/test_skm_host.py drives it through
// a small C++ harness), so the packing / canonical-form / expansion arithmetic can be checked without a GPU.
//
// Why super-k-mers.  The reference adds every k-mer of every read to the Count-Min tables one by one
// (khmer consume_seqfile, kevlar/count.py:49-71) and evaluates every k-mer of the case sample one by one
// (kevlar/novel.py:123-169).  At 30x coverage each genomic k-mer arrives ~20 times; both loops repeat the two
// murmur hashes, the T reductions and the T table accesses for every repeat.  Saturating adds commute, and
// kmer_is_interesting() is a pure function of the k-mer, so a batch can be processed per DISTINCT k-mer:
// add min(max, count) once, evaluate once.  To find the repeats without moving 8 bytes per k-mer through HBM
// the reads are cut into super-k-mers -- maximal runs of consecutive k-mers that share a minimizer bucket -- which
// cost ~2 bits per base, and every occurrence of a k-mer (either strand) lands in the same bucket because the
// minimizer is a strand-symmetric function of the k-mer.  A bucket is small enough to be deduplicated in LDS.
//
// Layout.  Bases are 2 bits (A0 C1 G2 T3), base p of a sequence in bits 2p.. of a little-endian bit string (the
// kv_reads layout).  A super-k-mer record is 1 + NBW 64-bit words:
//   word 0   bits  0..39  position of its first k-mer: read * stride + offset (stride = k-mers of the longest read)
//            bits 40..47  n, the number of k-mers (1 <= n <= ncap)
//            bits 48..63  fine bucket (inside the coarse bucket whose stream carries the record)
//   word 1.. the n + k - 1 bases, 32 per word
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KV_HD __host__ __device__ __forceinline__
#else
#define KV_HD static inline
#endif

#define SKM_MIN_K 16
#define SKM_MAX_K 64
#define SKM_POS_BITS 40
#define SKM_EMPTY (~0ull)           // empty key word of the LDS tables (no canonical k-mer word equals it, see skm_cacheable)

KV_HD uint32_t skm_brev32(uint32_t x) { return __builtin_bitreverse32(x); }
KV_HD uint64_t skm_brev64(uint64_t x) { return __builtin_bitreverse64(x); }

// reverse the 2-bit groups of a word (no complement)
KV_HD uint32_t skm_rev2_32(uint32_t x)
{
    x = skm_brev32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
KV_HD uint64_t skm_rev2_64(uint64_t x)
{
    x = skm_brev64(x);
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// lowbias32 (public-domain integer mixer): a bijection on 32 bits
KV_HD uint32_t skm_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// the order the minimizer is the minimum of: a bijection whose high bits -- the ones a comparison looks at first -- depend on every
// bit of the code (odd multiplier), low bits folded in for the ties.  Two instructions cheaper per base than the full mixer, which S1
// is sensitive to (one value per base); how evenly minimizers fall into buckets is skm_bucket_of's business, not this function's.
KV_HD uint32_t skm_order32(uint32_t x)
{
    x *= 0x9e3779b1u;
    return x ^ (x >> 15);
}

// Order value of an m-mer with forward code f and reverse-complement code r: the order of the canonical one (the smaller), doubled,
// plus a STRAND BIT -- 1 when the canonical form is the reverse complement, i.e. the m-mer stands in the read the other way round.
// The minimum over a k-mer's window then carries the strand of the minimizer, which (round 5) decides which strand of the K-MER is
// its key: the one on which its minimizer is canonical.  Both strands of a k-mer have the same m-mers with the strand bits flipped, so
// they agree on that key without either of them computing the other strand (kv_skm.hip, "oriented records").  Where they do not
// agree -- the minimal value occurs on both strands of one window, or the minimizer is its own reverse complement -- a k-mer is kept
// under two keys, which costs a little deduplication and nothing else: adds compose, kmer_is_interesting() is a function of the hash.
KV_HD uint32_t skm_order_s(uint32_t f, uint32_t r)
{
    return (skm_order32(f < r ? f : r) << 1) | (r < f ? 1u : 0u);
}
// order value of the m-mer (m <= 16) whose forward code is f
KV_HD uint32_t skm_mmer_value(uint32_t f, int m)
{
    const uint32_t mmask = m == 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    const uint32_t r = (skm_rev2_32(f) >> (32 - 2 * m)) ^ mmask;
    return skm_order_s(f, r);
}

// minimizer value -> (coarse, fine) bucket.  The minimum of w uniform values is far from uniform, hence the second mix.
KV_HD void skm_bucket_of(uint32_t minv, uint32_t C1, uint32_t fbits, uint32_t &coarse, uint32_t &fine)
{
    const uint32_t g = skm_mix32((minv | 1u) + 0x9e3779b9u);          // (the strand bit of the value is no part of the bucket: both strands of a k-mer meet)
    const uint64_t prod = (uint64_t)g * C1;
    coarse = (uint32_t)(prod >> 32);
    fine = fbits ? (uint32_t)prod >> (32 - fbits) : 0u;
}

// (bit 47, the top of n's byte: the record's bases are the REVERSE COMPLEMENT of the read's -- an oriented record, kv_skm.hip; its k-mer j
// then stands at read position pos + n - 1 - j)
KV_HD uint64_t skm_header(uint64_t pos, uint32_t n, uint32_t fine, uint32_t rev = 0u) { return pos | ((uint64_t)(n | (rev << 7)) << SKM_POS_BITS) | ((uint64_t)fine << 48); }
KV_HD uint64_t skm_hdr_pos(uint64_t h) { return h & ((1ull << SKM_POS_BITS) - 1ull); }
KV_HD uint32_t skm_hdr_n(uint64_t h) { return (uint32_t)(h >> SKM_POS_BITS) & 0x7fu; }
KV_HD uint32_t skm_hdr_rev(uint64_t h) { return (uint32_t)(h >> (SKM_POS_BITS + 7)) & 1u; }
// read position of k-mer j of a record
KV_HD uint64_t skm_hdr_pos_of(uint64_t h, uint32_t j) { return skm_hdr_pos(h) + (skm_hdr_rev(h) ? skm_hdr_n(h) - 1u - j : j); }
KV_HD uint32_t skm_hdr_fine(uint64_t h) { return (uint32_t)(h >> 48); }

// ---- compact records (round 5): 16 bytes, for batches nobody needs positions of (a control sample's count) -----------
//   word 0   bases 0..31
//   word 1   bits  0..39  bases 32..51   bits 40..45  n (n + k - 1 <= 52 bases)   bits 46..57  fine bucket   bit 58  reversed
// One aligned 16-byte store in S1 and one 16-byte load per lane in S2 / S3 instead of a 16- and an 8-byte piece of a 24-byte
// record that straddles sectors; a third fewer bytes through S1, S2 and S3.  k <= 32 only.
#define SKM_C_BASES 52
KV_HD uint64_t skm_c_pack1(uint64_t b1, uint32_t n, uint32_t fine, uint32_t rev = 0u)
{
    return (b1 & ((1ull << 40) - 1ull)) | ((uint64_t)n << 40) | ((uint64_t)fine << 46) | ((uint64_t)rev << 58);      // (bit 58: an oriented record, reversed)
}
KV_HD uint32_t skm_c_rev(uint64_t w1) { return (uint32_t)(w1 >> 58) & 1u; }
KV_HD uint32_t skm_c_n(uint64_t w1) { return (uint32_t)(w1 >> 40) & 63u; }
KV_HD uint32_t skm_c_fine(uint64_t w1) { return (uint32_t)(w1 >> 46) & 4095u; }
KV_HD uint64_t skm_c_b1(uint64_t w1) { return w1 & ((1ull << 40) - 1ull); }

// 32 bases starting at base `b` of a packed sequence (u32 words, 16 bases each); reads three words
KV_HD uint64_t skm_bases32(const uint32_t *words, uint32_t b)
{
    const uint32_t a = b >> 4, sh = 2u * (b & 15u);
    const uint64_t lo = (uint64_t)words[a] | ((uint64_t)words[a + 1] << 32);
    uint64_t v = lo >> sh;
    if (sh) v |= (uint64_t)words[a + 2] << (64u - sh);
    return v;
}

// reverse complement of the nb bases (1 <= nb <= 32 * nbw) held in bw[0 .. nbw), in place; bits beyond 2 nb come out zero
KV_HD void skm_rc_bases(uint64_t *bw, int nbw, uint32_t nb)
{
    // reverse all 32 * nbw base slots, complement, then drop the (32 * nbw - nb) slots that came to the front
    uint64_t r[4] = {0, 0, 0, 0};
    for (int t = 0; t < 3; ++t)
        if (t < nbw) r[t] = ~skm_rev2_64(bw[nbw - 1 - t]);
    const uint32_t s = 2u * (32u * (uint32_t)nbw - nb), ws = s >> 6, bs = s & 63u;
    for (int t = 0; t < 3; ++t) {
        if (t >= nbw) break;
        const uint32_t a = (uint32_t)t + ws;
        const uint64_t lo = a == 0 ? r[0] : (a == 1 ? r[1] : (a == 2 ? r[2] : 0ull));
        const uint64_t hi = a == 0 ? r[1] : (a == 1 ? r[2] : 0ull);
        const uint64_t lim = (uint32_t)nbw;                       // words at and beyond nbw are not part of the string
        const uint64_t lo_ok = a < lim ? lo : 0ull, hi_ok = a + 1u < lim ? hi : 0ull;
        bw[t] = bs ? (lo_ok >> bs) | (hi_ok << (64u - bs)) : lo_ok;
    }
}

// ---- k-mers in registers: KW = 1 (k <= 32) or 2 (k <= 64) words, base 0 in the low bits -----------------------
template <int KW> struct SkmKey { uint64_t w[KW]; };

template <int KW>
KV_HD bool skm_key_less(const SkmKey<KW> &a, const SkmKey<KW> &b)
{
    if (KW == 2 && a.w[KW - 1] != b.w[KW - 1]) return a.w[KW - 1] < b.w[KW - 1];
    return a.w[0] < b.w[0];
}
template <int KW>
KV_HD bool skm_key_eq(const SkmKey<KW> &a, const SkmKey<KW> &b)
{
    return a.w[0] == b.w[0] && (KW == 1 || a.w[KW - 1] == b.w[KW - 1]);
}

// mask of the top word: 2 * (k - 32 * (KW - 1)) bits
template <int KW>
KV_HD uint64_t skm_topmask(int k)
{
    const int bits = 2 * (k - 32 * (KW - 1));
    return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
}

// the first k-mer of a record (bases 0..k-1 of its base words)
template <int KW>
KV_HD SkmKey<KW> skm_first_kmer(const uint64_t *bw, int k)
{
    SkmKey<KW> f;
    f.w[0] = KW == 1 ? bw[0] & skm_topmask<1>(k) : bw[0];
    if (KW == 2) f.w[KW - 1] = bw[1] & skm_topmask<2>(k);
    return f;
}

template <int KW>
KV_HD SkmKey<KW> skm_revcomp(const SkmKey<KW> &f, int k)
{
    SkmKey<KW> r;
    if (KW == 1) {
        r.w[0] = (skm_rev2_64(f.w[0]) >> (64 - 2 * k)) ^ skm_topmask<1>(k);
    } else {
        const uint64_t lo = skm_rev2_64(f.w[KW - 1]), hi = skm_rev2_64(f.w[0]);   // 128-bit reversal
        const int s = 128 - 2 * k;                                                 // 0 <= s < 64
        r.w[0] = ~((lo >> s) | (s ? hi << (64 - s) : 0ull));
        r.w[KW - 1] = (hi >> s) ^ skm_topmask<2>(k);
    }
    return r;
}

// forward / reverse-complement pair of consecutive k-mers: drop base 0, append `base` (forward) -- i.e. prepend its
// complement on the other strand
template <int KW>
KV_HD void skm_roll(SkmKey<KW> &f, SkmKey<KW> &r, uint32_t base, int k)
{
    if (KW == 1) {
        f.w[0] = (f.w[0] >> 2) | ((uint64_t)base << (2 * k - 2));
        r.w[0] = ((r.w[0] << 2) | (uint64_t)(base ^ 3u)) & skm_topmask<1>(k);
    } else {
        const int top = 2 * (k - 32) - 2;
        f.w[0] = (f.w[0] >> 2) | (f.w[KW - 1] << 62);
        f.w[KW - 1] = (f.w[KW - 1] >> 2) | ((uint64_t)base << top);
        r.w[KW - 1] = ((r.w[KW - 1] << 2) | (r.w[0] >> 62)) & skm_topmask<2>(k);
        r.w[0] = (r.w[0] << 2) | (uint64_t)(base ^ 3u);
    }
}

// base p of a record's base words
KV_HD uint32_t skm_base_at(const uint64_t *bw, uint32_t p)
{
    const uint64_t w = p < 32u ? bw[0] : (p < 64u ? bw[1] : bw[2]);     // selects, not an indexed register array
    return (uint32_t)(w >> (2u * (p & 31u))) & 3u;
}

// 64 bits starting at base p of a record's base words (zeros beyond word 2): the window every k-mer, tail and unit of the
// bucket walk is cut from
KV_HD uint64_t skm_window64(uint64_t b0, uint64_t b1, uint64_t b2, uint32_t p)
{
    const uint32_t ws = p >> 5, sh = 2u * (p & 31u);
    const uint64_t x0 = ws == 0 ? b0 : (ws == 1 ? b1 : (ws == 2 ? b2 : 0ull));
    const uint64_t x1 = ws == 0 ? b1 : (ws == 1 ? b2 : 0ull);
    return sh ? (x0 >> sh) | (x1 << (64u - sh)) : x0;
}

// the k-mer starting at base j of a record
template <int KW>
KV_HD SkmKey<KW> skm_kmer_of(uint64_t b0, uint64_t b1, uint64_t b2, uint32_t j, int k)
{
    SkmKey<KW> f;
    if (KW == 1) {
        f.w[0] = skm_window64(b0, b1, b2, j) & skm_topmask<1>(k);
    } else {
        f.w[0] = skm_window64(b0, b1, b2, j);
        f.w[KW - 1] = skm_window64(b0, b1, b2, j + 32u) & skm_topmask<2>(k);
    }
    return f;
}

template <int KW>
KV_HD SkmKey<KW> skm_canonical(const SkmKey<KW> &f, const SkmKey<KW> &r) { return skm_key_less<KW>(r, f) ? r : f; }

// a key can live in the LDS tables unless one of its words equals the empty marker (possible only for a handful
// of k = 32 / k = 64 sequences; those k-mers are handled one occurrence at a time instead)
template <int KW>
KV_HD bool skm_cacheable(const SkmKey<KW> &c) { return c.w[0] != SKM_EMPTY && (KW == 1 || c.w[KW - 1] != SKM_EMPTY); }

template <int KW>
KV_HD uint32_t skm_slot_hash(const SkmKey<KW> &c)
{
    uint64_t x = c.w[0];
    if (KW == 2) x ^= c.w[KW - 1] * 0x9e3779b97f4a7c15ull;
    uint32_t y = (uint32_t)x ^ (uint32_t)(x >> 32);
    y *= 0x9e3779b1u;
    return y ^ (y >> 15);
}

// ASCII of four bases (one byte of 2-bit codes) as a little-endian u32: the 256-entry table the kernels keep in LDS
KV_HD uint32_t skm_ascii4(uint32_t byte)
{
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) out |= ((0x54474341u >> (8u * ((byte >> (2 * i)) & 3u))) & 0xffu) << (8 * i);   // "ACGT"
    return out;
}
