// kv_inflate_device.h -- the DEFLATE pieces both inflaters share (kv_inflate.hip: one wavefront per BGZF member;
// kv_gunzip.hip: one wavefront per stretch of an ordinary gzip stream): the bit reader, canonical Huffman tables and the
// wave-uniform symbol decode.  Device code only.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace {

#define INF_FAST_LL 10          // bits of the literal/length lookup table
#define INF_FAST_D 8            // ... of the distance table
#define INF_MAX_OUT 65536u

struct BitReader {
    const uint32_t *words;      // aligned words of the payload
    uint32_t next;              // index of the next word to take
    uint64_t buf;               // bits not yet consumed, LSB first
    uint32_t cnt;               // how many
    uint32_t ahead;             // words[next], already loaded (hides the load behind the decode)
};

__device__ __forceinline__ void br_init(BitReader &br, const uint8_t *payload)
{
    const uint64_t addr = (uint64_t)payload;
    const uint32_t mis = (uint32_t)(addr & 3u);
    br.words = (const uint32_t *)(addr - mis);
    br.buf = (uint64_t)(br.words[0] >> (8u * mis));
    br.cnt = 32u - 8u * mis;
    br.next = 1;
    br.ahead = br.words[1];
}

__device__ __forceinline__ void br_need(BitReader &br, uint32_t n)   // n <= 32
{
    if (br.cnt < n) {
        br.buf |= (uint64_t)br.ahead << br.cnt;
        br.cnt += 32u;
        br.next += 1;
        br.ahead = br.words[br.next];
    }
}

__device__ __forceinline__ uint32_t br_bits(BitReader &br, uint32_t n)   // n <= 16
{
    br_need(br, n);
    const uint32_t v = (uint32_t)br.buf & ((1u << n) - 1u);
    br.buf >>= n;
    br.cnt -= n;
    return v;
}

// canonical Huffman code of `n` symbols with the given lengths: count[len], symbols sorted by (len, symbol), and the
// lookup table of the codes of at most `fast` bits (entry = symbol | len << 9; 0 = longer code).  Returns false if the
// lengths oversubscribe the code space.
__device__ bool build_code(const uint8_t *lengths, int n, uint16_t *count, uint16_t *symbol, uint16_t *table, int fast)
{
    for (int l = 0; l <= 15; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lengths[s]]++;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < n; ++s)
        if (lengths[s]) symbol[offs[lengths[s]]++] = (uint16_t)s;
    for (int i = 0; i < (1 << fast); ++i) table[i] = 0;
    // codes in canonical order: first code of each length, then consecutive
    uint32_t code = 0, index = 0;
    for (int l = 1; l <= fast; ++l) {
        for (uint32_t j = 0; j < count[l]; ++j, ++code, ++index) {
            const uint32_t rev = __brev(code) >> (32 - l);            // the stream carries a code's bits MSB first
            const uint16_t entry = (uint16_t)(symbol[index] | (l << 9));
            for (uint32_t fill = rev; fill < (1u << fast); fill += 1u << l) table[fill] = entry;
        }
        code <<= 1;
    }
    return true;
}

// One symbol.  Every lane runs this with the same (wave-uniform) reader state, so the arithmetic lives on the scalar unit;
// only the table word comes through a vector register and is made uniform again at once.
#define INF_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ int decode_sym(BitReader &br, const uint16_t *count, const uint16_t *symbol, const uint16_t *table, int fast)
{
    br_need(br, 15);
    const uint32_t e = INF_UNI(table[(uint32_t)br.buf & ((1u << fast) - 1u)]);
    if (e) {
        const uint32_t l = e >> 9;
        br.buf >>= l;
        br.cnt -= l;
        return (int)(e & 0x1ffu);
    }
    // a code longer than the table: bit by bit through the canonical counts
    int code = 0, first = 0, index = 0;
    uint64_t bits = br.buf;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int c = (int)INF_UNI(count[l]);
        if (code - c < first) {
            br.buf >>= l;
            br.cnt -= l;
            return (int)INF_UNI(symbol[index + (code - first)]);
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// The length / distance code tables of RFC 1951 section 3.2.5 as arithmetic: a table in constant memory is a vector load
// here, and the s_waitcnt vmcnt(0) behind it also waits for every text store the wave still has in flight -- a
// microsecond per match.  (tests/test_gpu_ingest.py compares the text with zlib over every length and distance code.)
__device__ __forceinline__ uint32_t inf_len_extra(uint32_t ls) { return ls < 8u || ls == 28u ? 0u : (ls - 4u) >> 2; }
__device__ __forceinline__ uint32_t inf_len_base(uint32_t ls)
{
    return ls < 8u ? 3u + ls : ls == 28u ? 258u : 3u + ((4u + (ls & 3u)) << ((ls - 4u) >> 2));
}
__device__ __forceinline__ uint32_t inf_dist_extra(uint32_t ds) { return ds < 4u ? 0u : (ds >> 1) - 1u; }
__device__ __forceinline__ uint32_t inf_dist_base(uint32_t ds) { return ds < 4u ? 1u + ds : 1u + ((2u + (ds & 1u)) << ((ds >> 1) - 1u)); }
// the order the code-length code's own lengths arrive in: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
__device__ __forceinline__ uint32_t inf_clen_order(uint32_t s)
{
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return (uint32_t)((s < 12u ? lo >> (5u * s) : hi >> (5u * (s - 12u))) & 31u);
}

}  // namespace
