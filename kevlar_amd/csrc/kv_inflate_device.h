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

__constant__ uint16_t c_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};


}  // namespace
