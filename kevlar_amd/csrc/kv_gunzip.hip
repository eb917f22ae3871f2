// kv_gunzip.hip -- an ORDINARY gzip stream inflated on the device, in parallel (ingest).
//
// What `gzip`, `pigz` and most sequencers write is one long DEFLATE stream: a block can only be decoded by whoever knows
// where it starts (blocks end on arbitrary bits) and what the 32 KB of text in front of it were (matches point back into
// them).  zlib on one host core manages ~1.7 M reads/s of FASTQ that way.  Both obstacles have known ways round them
// (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to DNA sequences", 2019;
// Knespel & Brunst, "Rapidgzip", 2023), restated here for wavefronts:
//
//   1. k_gz_find    the stream is cut into chunks of 16 KB; one workgroup per chunk tries every bit offset as the start
//                   of a dynamic-Huffman block: 3 header bits, the two code counts, a COMPLETE code-length code (all 64
//                   lanes x 4 waves, 32 offsets each from registers), then for the few survivors the full set of
//                   literal/length and distance code lengths, which must form complete codes with an end-of-block symbol.
//                   The first four hits of a chunk are kept.
//   1b. k_gz_sync   (streams with few, long blocks) symbol boundaries INSIDE a block: 64 decoders from 64 neighbouring bit
//                   offsets fall into step with the true symbols; where 24 agree a stretch of its own begins.
//   2. k_gz_decode  one wavefront per found start decodes until it lands exactly on a later start (kv_inflate.hip's
//                   wave-uniform Huffman walk).  Text goes out as 16-bit symbols: a byte, or -- for a match that reaches
//                   back beyond the wave's own start -- a MARKER naming the position in the unknown 32 KB window.
//                   The host checks the chain (start 0 is exact; a start is good iff a good decoder lands on it; a
//                   start that was skipped over was a false positive and its output is dropped; a gap is decoded again).
//   3. k_gz_tails / k_gz_scan   every stretch leaves a 32 K-symbol tail (its own last symbols, or markers passed
//                   through); substituting a tail's markers from the tail before it is associative, so log2(n) rounds of
//                   pointer doubling resolve all tails at once.
//   4. k_gz_resolve every stretch replaces its markers from the (now plain) tail in front of it and stores bytes, packed to
//                   the text offset the prefix sum of the lengths gives it.
//   5. k_gz_crc     CRC-32 of the text in slices of 16 KB, cut at the member ends the decoders saw; the host joins the slices
//                   of a member and compares with its trailer (ISIZE too).
//
// The file is taken a segment of compressed bytes at a time; the last tail, the running CRC and the exact bit position
// travel to the next segment.  Concatenated members are followed inside k_gz_decode.  Tests compare with zlib byte for
// byte.  Anything unexpected (no block found for megabytes, trailing garbage, corrupt codes, a CRC or length that does not
// match) is reported as KV_ERR_TYPE and the caller's host parser (zlib) takes the file.  Reference: the reader this replaces is
// khmer.ReadParser's gzip stream (kevlar/__init__.py:125-128 opens every *.gz through it).
#include <algorithm>
#include <functional>
#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

#include <zlib.h>

#include "kv_binned.h"
#include "kv_internal.h"
#include "kv_inflate_device.h"

namespace {

#define GZ_WIN 32768u
#define GZ_MARK 0x8000u
#define GZ_FAST_LL 9                 // bits of the literal/length lookup table (9: 32 workgroups of LDS per CU)
#define GZ_SLACK 2048u               // readable zero bytes behind the compressed buffer
#define GZ_FIND_KEEP 4u                // block starts kept per chunk
#define GZ_FIND_THREADS 256
#define GZ_FIND_LIST 1024u           // survivors of the cheap test a chunk may have (more: the later ones are not looked at)

enum { GZ_LANDED = 0, GZ_END = 1, GZ_BAD = 2, GZ_FULL = 3, GZ_SHORT = 4 };

struct GzJob {
    uint64_t start_bit;              // first symbol (or block header) to decode, relative to the compressed buffer
    uint64_t target_idx;             // starts[target_idx]: the first start behind start_bit; the stretch stops ON the first start it meets
    uint64_t out_off;                // symbols
    uint64_t header_bit;             // header of the block start_bit lies in (== start_bit for a stretch that begins with a block)
    uint32_t out_cap;
    uint32_t pad;
};
struct GzGuess {
    uint64_t header_bit;             // a found block start ...
    uint64_t guess_bit;              // ... and a place inside that block near which a symbol boundary is wanted
};
struct GzResult {
    uint64_t end_bit;
    uint32_t n_out;
    uint16_t status;
    uint16_t members;                // member trailers the stretch passed,
    uint32_t isize_sum;              // ... the sum of their ISIZE fields,
    uint32_t trailer_at;             // ... how many symbols the stretch had produced at the first one
    uint32_t trailer_crc;            // ... and the CRC-32 that one announces
    uint32_t pad;
};

__device__ __forceinline__ void br_seek(BitReader &br, const uint32_t *words, uint64_t bit)
{
    const uint64_t w = bit >> 5;
    const uint32_t s = (uint32_t)(bit & 31u);
    br.words = words;
    br.buf = (uint64_t)(words[w] >> s);
    br.cnt = 32u - s;
    br.next = (uint32_t)w + 1;
    br.ahead = words[w + 1];
}
__device__ __forceinline__ uint64_t br_pos(const BitReader &br) { return (uint64_t)br.next * 32ull - br.cnt; }

// ---------------------------------------------------------------- 1. block starts
__device__ __forceinline__ uint32_t lds_bits(const uint32_t *w, uint32_t bit, uint32_t n)     // n <= 25
{
    const uint32_t i = bit >> 5, s = bit & 31u;
    const uint64_t v = ((uint64_t)w[i + 1] << 32) | w[i];
    return (uint32_t)(v >> s) & ((1u << n) - 1u);
}

// the 17 header bits at some offset: BFINAL = 0, BTYPE = dynamic, no more than 286 / 30 codes announced
__device__ __forceinline__ bool gz_header_counts(uint32_t head)
{
    return (head & 7u) == 4u && ((head >> 3) & 31u) <= 29u && ((head >> 8) & 31u) <= 29u;
}

// ... and a COMPLETE code-length code behind them (bits [p + 17, ...) of the staged chunk)
__device__ __forceinline__ bool gz_header_precode(const uint32_t *w, uint32_t p)
{
    const uint32_t ncode = lds_bits(w, p + 13, 4) + 4u;
    uint32_t kraft = 0, q = p + 17;
    for (uint32_t s = 0; s < ncode; s += 8) {
        uint32_t v = lds_bits(w, q, 24);
        q += 24;
        const uint32_t m = min(8u, ncode - s);
        v &= (1u << (3u * m)) - 1u;
        for (uint32_t i = 0; i < 8; ++i, v >>= 3) kraft += (256u >> (v & 7u)) & 0xffu;        // length l > 0 weighs 2^(8 - l), l = 0 nothing
    }
    return kraft == 256u;
}

// ... and the code lengths it spells: no overrun, an end-of-block code, complete literal/length and distance codes
__device__ bool gz_header_full(const uint32_t *w, uint32_t p, uint32_t limit_bits)
{
    uint32_t q = p + 3;
    const uint32_t nlen = lds_bits(w, q, 5) + 257u; q += 5;
    const uint32_t ndist = lds_bits(w, q, 5) + 1u; q += 5;
    const uint32_t ncode = lds_bits(w, q, 4) + 4u; q += 4;
    uint64_t cl = 0;                                   // 19 lengths of 3 bits
    for (uint32_t s = 0; s < ncode; ++s, q += 3) cl |= (uint64_t)lds_bits(w, q, 3) << (3u * inf_clen_order(s));
    uint64_t cnt = 0;                                  // symbols per length, 8 bits each
    uint64_t sorted_lo = 0, sorted_hi = 0;             // the symbols by (length, symbol), 5 bits each, 12 per word
    uint32_t filled = 0;
    for (uint32_t l = 1; l <= 7; ++l)
        for (uint32_t s = 0; s < 19; ++s)
            if (((cl >> (3u * s)) & 7u) == l) {
                cnt += 1ull << (8u * l);
                if (filled < 12) sorted_lo |= (uint64_t)s << (5u * filled);
                else sorted_hi |= (uint64_t)s << (5u * (filled - 12));
                ++filled;
            }
    const uint32_t total = nlen + ndist;
    uint32_t kraft_ll = 0, kraft_d = 0, used_d = 0, idx = 0, prev = 0;
    bool eob = false;
    while (idx < total) {
        if (q + 16 > limit_bits) return false;
        uint32_t window = lds_bits(w, q, 7);
        int code = 0, first = 0, index = 0, sym = -1;
        for (uint32_t l = 1; l <= 7; ++l) {
            code |= (int)(window & 1u);
            window >>= 1;
            const int c = (int)((cnt >> (8u * l)) & 255u);
            if (code - c < first) {
                const uint32_t at = (uint32_t)(index + (code - first));
                sym = (int)(at < 12 ? (sorted_lo >> (5u * at)) & 31u : (sorted_hi >> (5u * (at - 12))) & 31u);
                q += l;
                break;
            }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        if (sym < 0) return false;
        uint32_t rep = 1, val = (uint32_t)sym;
        if (sym == 16) {
            if (idx == 0) return false;
            val = prev;
            rep = 3 + lds_bits(w, q, 2); q += 2;
        } else if (sym == 17) { val = 0; rep = 3 + lds_bits(w, q, 3); q += 3; }
        else if (sym == 18) { val = 0; rep = 11 + lds_bits(w, q, 7); q += 7; }
        if (idx + rep > total) return false;
        if (val) {
            const uint32_t in_ll = idx < nlen ? min(rep, nlen - idx) : 0u;
            kraft_ll += in_ll << (15u - val);
            kraft_d += (rep - in_ll) << (15u - val);
            used_d += rep - in_ll;
            if (idx <= 256u && 256u < idx + rep) eob = true;
        }
        idx += rep;
        prev = val;
    }
    return eob && kraft_ll == 32768u && (kraft_d == 32768u || used_d == 0 || (used_d == 1 && kraft_d == 16384u));
}

// cand[c * GZ_FIND_KEEP ...] = bit positions (relative to `comp`) of the first plausible block starts in chunk c, ascending, ~0
// where there are no more.  (All of them, not only the first: a false positive in front of a true start would otherwise hide
// it, and the stretch that runs over the false one would end where nobody began -- a gap the host has to have decoded again,
// one stretch's latency, ~10 ms, on its own.)
__global__ __launch_bounds__(GZ_FIND_THREADS) void k_gz_find(const uint8_t *__restrict__ comp, uint64_t n_bytes, uint32_t chunk_bytes, uint32_t n_chunks,
                                                             unsigned long long *__restrict__ cand)
{
    extern __shared__ uint32_t sh_words[];             // chunk_bytes + GZ_SLACK bytes of the stream
    __shared__ uint32_t sh_list[GZ_FIND_LIST];
    __shared__ uint32_t sh_n, sh_n_good, sh_good[16];
    const uint32_t c = blockIdx.x;
    if (c >= n_chunks) return;
    const uint64_t base = (uint64_t)c * chunk_bytes;
    const uint32_t staged = chunk_bytes + GZ_SLACK;
    const uint32_t *src = (const uint32_t *)(comp + base);          // comp is 256-aligned, chunk_bytes a multiple of 4
    for (uint32_t i = threadIdx.x; i < staged / 4; i += GZ_FIND_THREADS) sh_words[i] = src[i];     // the buffer has GZ_SLACK zero bytes behind n_bytes
    if (threadIdx.x == 0) sh_n_good = 0;
    __syncthreads();
    const uint64_t left = n_bytes > base ? n_bytes - base : 0;
    const uint32_t limit_bits = (uint32_t)(left < staged ? left : staged) * 8u;
    const uint32_t chunk_bits = chunk_bytes * 8u < limit_bits ? chunk_bytes * 8u : limit_bits;
    // Every bit offset is a candidate.  A lane takes 32 consecutive ones (one word of the stream and the next): the 13 bits of
    // block type and code counts are tested from registers and leave about one offset in nine; those go through the
    // code-length code's completeness (one in 250 passes), lane by lane as they come; what is left (~60 of a chunk's
    // 131072) is collected and put through the full test all at once: a survivor costs ~300 dependent code-length
    // decodes, so they must not queue up behind each other.
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    for (uint32_t word = threadIdx.x; word * 32u < chunk_bits; word += GZ_FIND_THREADS) {
        const uint64_t v = ((uint64_t)sh_words[word + 1] << 32) | sh_words[word];
        uint32_t some = 0;
        for (uint32_t s = 0; s < 32; ++s) some |= gz_header_counts((uint32_t)(v >> s)) ? 1u << s : 0u;
        if (word * 32u + 32u > chunk_bits) some &= (1u << (chunk_bits - word * 32u)) - 1u;
        while (some) {
            const uint32_t p = word * 32u + (uint32_t)__builtin_ctz(some);
            some &= some - 1u;
            if (gz_header_precode(sh_words, p)) {
                const uint32_t at = atomicAdd(&sh_n, 1u);
                if (at < GZ_FIND_LIST) sh_list[at] = p;
            }
        }
    }
    __syncthreads();
    const uint32_t n = min(sh_n, GZ_FIND_LIST);
    for (uint32_t i = threadIdx.x; i < n; i += GZ_FIND_THREADS)
        if (gz_header_full(sh_words, sh_list[i], limit_bits)) {
            const uint32_t at = atomicAdd(&sh_n_good, 1u);
            if (at < 16) sh_good[at] = sh_list[i];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t good = min(sh_n_good, 16u);
        for (uint32_t i = 1; i < good; ++i)               // a handful: insertion sort
            for (uint32_t j = i; j > 0 && sh_good[j] < sh_good[j - 1]; --j) { const uint32_t t = sh_good[j]; sh_good[j] = sh_good[j - 1]; sh_good[j - 1] = t; }
        for (uint32_t i = 0; i < GZ_FIND_KEEP; ++i) cand[(uint64_t)c * GZ_FIND_KEEP + i] = i < good ? base * 8ull + sh_good[i] : ~0ull;
    }
}

// ---------------------------------------------------------------- 2. decode into symbols
__device__ __forceinline__ bool gz_build_code(const uint8_t *lengths, int n, uint16_t *count, uint16_t *symbol, uint16_t *table, int fast)
{
    return build_code(lengths, n, count, symbol, table, fast);
}

template <int RBITS>                 // the decoder's LDS window: the last 2^RBITS symbols
struct GunzipShared {
    uint16_t ring[1 << RBITS];
    uint16_t ll_table[1 << GZ_FAST_LL];
    uint16_t d_table[1 << INF_FAST_D];
    uint16_t ll_count[16], d_count[16];
    uint16_t ll_symbol[288], d_symbol[32];
    uint8_t lengths[352];
    // where the stretch is to stop (kept here, not in registers: looked at once per block and when the target comes near)
    unsigned long long target, cur_header;
    uint32_t target_idx, pad;
};

// gzip member header at byte `at` of the buffer: returns the byte its deflate data start at, 0 if this is no header,
// ~0 if the buffer ends inside it
__device__ __attribute__((noinline)) uint64_t gz_member_header(const uint8_t *comp, uint64_t at, uint64_t n_bytes)
{
    if (at + 18 > n_bytes) return ~0ull;
    if (comp[at] != 0x1f || comp[at + 1] != 0x8b || comp[at + 2] != 8 || (comp[at + 3] & 0xe0)) return 0;
    const uint32_t flags = comp[at + 3];
    uint64_t p = at + 10;
    if (flags & 4) {
        const uint32_t xlen = comp[p] | (comp[p + 1] << 8);
        p += 2 + xlen;
    }
    for (int field = 0; field < 2; ++field)              // name, comment: zero-terminated
        if (flags & (field == 0 ? 8 : 16)) {
            while (p < n_bytes && comp[p]) ++p;
            ++p;
        }
    if (flags & 2) p += 2;
    return p + 8 <= n_bytes ? p : ~0ull;
}

// The Huffman codes of a block whose 3 header bits have been read (btype 1: the fixed ones, 2: the ones the block spells
// out) -> lookup tables in LDS.  All lanes take part; returns 0 for an invalid description.
template <int RBITS>
__device__ uint32_t gz_block_codes(BitReader &br, GunzipShared<RBITS> &sh, uint32_t lane, uint32_t btype)
{
    uint32_t ok = 1;
    if (btype == 1) {
        if (lane == 0) {
            for (int s = 0; s < 144; ++s) sh.lengths[s] = 8;
            for (int s = 144; s < 256; ++s) sh.lengths[s] = 9;
            for (int s = 256; s < 280; ++s) sh.lengths[s] = 7;
            for (int s = 280; s < 288; ++s) sh.lengths[s] = 8;
            ok = gz_build_code(sh.lengths, 288, sh.ll_count, sh.ll_symbol, sh.ll_table, GZ_FAST_LL);
            for (int s = 0; s < 30; ++s) sh.lengths[s] = 5;
            ok = ok && gz_build_code(sh.lengths, 30, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
        }
    } else {
        const uint32_t nlen = br_bits(br, 5) + 257, ndist = br_bits(br, 5) + 1, ncode = br_bits(br, 4) + 4;
        if (nlen > 286 || ndist > 30) return 0;
        if (lane < 19) sh.lengths[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t s = 0; s < ncode; ++s) {
            const uint32_t v = br_bits(br, 3);
            if (lane == 0) sh.lengths[inf_clen_order(s)] = (uint8_t)v;
        }
        if (lane == 0) ok = gz_build_code(sh.lengths, 19, sh.d_count, sh.d_symbol, sh.d_table, 7);
        ok = INF_UNI(ok);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t idx = 0, prev = 0;
        while (ok && idx < nlen + ndist) {
            const int sym = decode_sym(br, sh.d_count, sh.d_symbol, sh.d_table, 7);
            if (sym < 0) { ok = 0; break; }
            uint32_t rep = 1, val = (uint32_t)sym;
            if (sym == 16) {
                if (idx == 0) { ok = 0; break; }
                val = prev;
                rep = 3 + br_bits(br, 2);
            } else if (sym == 17) { val = 0; rep = 3 + br_bits(br, 3); }
            else if (sym == 18) { val = 0; rep = 11 + br_bits(br, 7); }
            if (idx + rep > nlen + ndist) { ok = 0; break; }
            if (lane < rep) sh.lengths[19 + idx + lane] = (uint8_t)val;
            if (lane + 64 < rep) sh.lengths[19 + idx + lane + 64] = (uint8_t)val;
            if (lane + 128 < rep) sh.lengths[19 + idx + lane + 128] = (uint8_t)val;
            idx += rep;
            prev = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && ok) {
            ok = sh.lengths[19 + 256] != 0;
            ok = ok && gz_build_code(sh.lengths + 19, (int)nlen, sh.ll_count, sh.ll_symbol, sh.ll_table, GZ_FAST_LL);
            ok = ok && gz_build_code(sh.lengths + 19 + nlen, (int)ndist, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
        }
    }
    ok = INF_UNI(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return ok;
}

// ---------------------------------------------------------------- 1b. symbol boundaries inside a block
// A block of 100 KB of text is one stretch, one wavefront, 10 ms.  To cut it, a symbol boundary in its middle is needed, and
// Huffman codes synchronise by themselves: decoding from a WRONG bit offset falls into step with the true sequence of
// symbols after a few codes.  One wavefront per guess: the block's codes are read from its header, then every lane decodes
// (without output) from its own bit offset guess + lane until it has passed a common horizon 4 Kbit further on; the
// lanes that have fallen into step stop on the same bit.  If 24 of them agree, that bit starts a stretch of its own
// (the decoder of the stretch before must still stop exactly ON it for it to count).
struct LaneBits {
    uint64_t buf;
    uint32_t cnt, next;
};
__device__ __forceinline__ void lane_need(LaneBits &b, const uint32_t *words)
{
    if (b.cnt <= 32) { b.buf |= (uint64_t)words[b.next++] << b.cnt; b.cnt += 32; }
}
__device__ __forceinline__ int lane_symbol(LaneBits &b, const uint32_t *words, const uint16_t *count, const uint16_t *symbol, const uint16_t *table, int fast)
{
    lane_need(b, words);                               // 33 bits or more
    const uint32_t e = table[(uint32_t)b.buf & ((1u << fast) - 1u)];
    if (e) {
        const uint32_t l = e >> 9;
        b.buf >>= l; b.cnt -= l;
        return (int)(e & 0x1ffu);
    }
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int l = 1; l <= 15; ++l) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int c = (int)count[l];
        if (code - c < first) {
            b.buf >>= l; b.cnt -= l;
            return (int)symbol[index + (code - first)];
        }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__global__ __launch_bounds__(64) void k_gz_sync(const uint8_t *__restrict__ comp, uint64_t n_bytes, const GzGuess *__restrict__ guesses, uint32_t n_guesses,
                                                unsigned long long *__restrict__ found)
{
    __shared__ GunzipShared<10> sh;
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    if (g >= n_guesses) return;
    const uint32_t *words = (const uint32_t *)comp;
    const GzGuess guess = guesses[g];
    if (lane == 0) found[g] = ~0ull;
    BitReader br;
    br_seek(br, words, guess.header_bit);
    (void)br_bits(br, 1);
    const uint32_t btype = br_bits(br, 2);
    if (btype != 1 && btype != 2) return;
    if (!gz_block_codes(br, sh, lane, btype)) return;
    const uint64_t limit = n_bytes * 8ull;
    const uint64_t horizon = guess.guess_bit + 64 + 4096;
    if (horizon + 4096 > limit) return;
    const uint64_t mine = guess.guess_bit + lane;
    LaneBits b;
    b.next = (uint32_t)(mine >> 5);
    b.buf = (uint64_t)(words[b.next++] >> (mine & 31u));
    b.cnt = 32u - (uint32_t)(mine & 31u);
    bool ok = true;
    uint64_t pos = mine;
    for (int steps = 0; ok && pos < horizon && steps < 4096; ++steps) {
        const int sym = lane_symbol(b, words, sh.ll_count, sh.ll_symbol, sh.ll_table, GZ_FAST_LL);
        if (sym < 0 || sym == 256) ok = false;          // no code, or the block would end: not a place to start from
        else if (sym > 256) {
            const uint32_t ls = (uint32_t)sym - 257u;
            if (ls >= 29u) { ok = false; break; }
            lane_need(b, words);
            const uint32_t le = inf_len_extra(ls);
            b.buf >>= le; b.cnt -= le;
            const int ds = lane_symbol(b, words, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
            if (ds < 0 || ds >= 30) { ok = false; break; }
            lane_need(b, words);
            const uint32_t de = inf_dist_extra((uint32_t)ds);
            b.buf >>= de; b.cnt -= de;
        }
        pos = (uint64_t)b.next * 32ull - b.cnt;
    }
    ok = ok && pos >= horizon;
    // the bit most lanes stopped on
    uint64_t best = ~0ull;
    uint32_t best_votes = 0;
    for (uint32_t i = 0; i < 64; ++i) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pos, i), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pos >> 32), i);
        const uint64_t candidate = ((uint64_t)hi << 32) | lo;
        const bool lane_ok = __builtin_amdgcn_readlane((int)ok, i) != 0;
        const uint32_t votes = (uint32_t)__popcll(__ballot(ok && pos == candidate));
        if (lane_ok && votes > best_votes) { best_votes = votes; best = candidate; }
    }
    if (lane == 0 && best_votes >= 24) found[g] = best;
}

// Has the stretch met the start it is to stop on?  pos: where the reader is; at_boundary: in front of a block header (else
// between two symbols).  Moves the target on over starts that were run over or are of the wrong kind.  All lanes call it.
template <int RBITS>
__device__ __forceinline__ bool gz_met_start(GunzipShared<RBITS> &sh, const unsigned long long *starts, const unsigned long long *headers, uint64_t pos,
                                                       bool at_boundary, uint32_t *target_word)
{
    uint32_t idx = INF_UNI(sh.target_idx);
    uint64_t target = starts[idx];
    bool met = false;
    for (;;) {
        while (pos > target) target = starts[++idx];
        if (pos != target) break;
        const uint64_t header = headers[idx];
        const uint64_t cur = ((uint64_t)INF_UNI((uint32_t)(sh.cur_header >> 32)) << 32) | INF_UNI((uint32_t)sh.cur_header);
        if (header == (at_boundary ? pos : cur)) { met = true; break; }
        target = starts[++idx];                        // a start of the other kind, or of another block, on this very bit: not this stretch's
    }
    if (threadIdx.x == 0) { sh.target_idx = idx; sh.target = target; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    *target_word = (uint32_t)(target >> 5 > 0xffffffffull ? 0xffffffffull : target >> 5);
    return met;
}

template <int RBITS, bool EXACT>      // EXACT: the segment has stretches that begin inside a block (the target is looked for between symbols too)
__global__ __launch_bounds__(64, (RBITS <= 10 ? 8 : RBITS == 11 ? 6 : RBITS == 12 ? 3 : RBITS == 13 ? 2 : 1)) void k_gz_decode(const uint8_t *__restrict__ comp, uint64_t n_bytes, int is_file_end, const GzJob *__restrict__ jobs,
                                                      uint32_t n_jobs, const unsigned long long *__restrict__ starts, const unsigned long long *__restrict__ headers,
                                                      uint64_t terminal_bit, uint16_t *syms,
                                                      GzResult *__restrict__ results, unsigned long long *ctr)
{
    __shared__ GunzipShared<RBITS> sh;
    constexpr uint32_t RMASK = (1u << RBITS) - 1u;
    const uint32_t lane = threadIdx.x;
    const uint32_t *words = (const uint32_t *)comp;
    const uint64_t limit_words = (n_bytes + 3) / 4 + 2;        // the reader may be this far without having left the data
    for (;;) {
        uint32_t job_id = 0;
        if (lane == 0) job_id = (uint32_t)atomicAdd(&ctr[0], 1ull);
        job_id = INF_UNI(job_id);
        if (job_id >= n_jobs) return;
        const GzJob job = jobs[job_id];
        BitReader br;
        br_seek(br, words, job.header_bit);
        bool mid_block = job.header_bit != job.start_bit;        // the codes come from header_bit, the symbols from start_bit
        // starts[]: every place a stretch begins at, ascending, ~0 behind the last; headers[]: the block header each takes its
        // codes from (itself, for a stretch that begins with a block).  The stretch ends ON the first start it meets that is
        // of its kind -- a block start at a block boundary, a start inside a block between two symbols of THAT block; one it
        // runs over, or meets in another block, was not a start (a false positive of the block search, a symbol boundary
        // of a parse with the wrong codes) and the one after it becomes the target.
        if (lane == 0) { sh.target_idx = (uint32_t)job.target_idx; sh.target = starts[job.target_idx]; sh.cur_header = job.header_bit; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t target_word = (uint32_t)std::min<uint64_t>(starts[job.target_idx] >> 5, 0xffffffffull);
        uint16_t *dst = syms + job.out_off;
        const uint32_t cap = job.out_cap;
        uint32_t o = 0, lit = 0, n_lit = 0, isize_sum = 0, members = 0, trailer_at = 0, trailer_crc = 0;
        int status = -1;
        uint64_t end_bit = job.start_bit;
        auto flush = [&]() {
            if (lane < n_lit) { sh.ring[(o + lane) & RMASK] = (uint16_t)lit; dst[o + lane] = (uint16_t)lit; }
            o += n_lit;
            n_lit = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        bool first_block = true;
        while (status < 0) {
            // ---- at a block boundary
            end_bit = br_pos(br);
            if (!first_block && (gz_met_start(sh, starts, headers, end_bit, true, &target_word) || end_bit >= terminal_bit)) { status = GZ_LANDED; break; }
            if (lane == 0) sh.cur_header = end_bit;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            first_block = false;
            if (br.next > limit_words) { status = GZ_SHORT; break; }
            const bool last_block = br_bits(br, 1) != 0;
            const uint32_t btype = br_bits(br, 2);
            if (btype == 3 || (btype == 0 && mid_block)) { status = GZ_BAD; break; }
            if (btype == 0) {
                const uint32_t drop = br.cnt & 7u;
                br.buf >>= drop; br.cnt -= drop;
                const uint32_t stored_len = br_bits(br, 16);
                const uint32_t inv = br_bits(br, 16);
                if ((stored_len ^ inv) != 0xffffu) { status = GZ_BAD; break; }
                if (br_pos(br) + 8ull * stored_len > n_bytes * 8ull) { status = GZ_SHORT; break; }
                if (o + stored_len > cap) { status = GZ_FULL; break; }
                flush();
                const uint64_t from_byte = br_pos(br) >> 3;                          // a byte boundary
                for (uint32_t j = lane; j < stored_len; j += 64) {
                    const uint16_t v = comp[from_byte + j];
                    dst[o + j] = v;
                    if (j + (1u << RBITS) >= stored_len) sh.ring[(o + j) & RMASK] = v;
                }
                o += stored_len;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                br_seek(br, words, (from_byte + stored_len) * 8);
            } else {
                const uint32_t ok = gz_block_codes(br, sh, lane, btype);
                if (!ok) { status = GZ_BAD; break; }
                if (mid_block) { br_seek(br, words, job.start_bit); mid_block = false; }
                // ---- symbols
                // (room and the end of the input are checked when text is stored -- at least every 64 symbols, i.e. every 120
                // bytes of input, which the zeroed slack behind the buffer covers -- not per literal: the loop is bound by the
                // number of instructions it issues)
                for (;;) {
                    if (EXACT && br.next >= target_word && gz_met_start(sh, starts, headers, br_pos(br), false, &target_word)) {        // the next stretch starts here, inside the block
                        if (o + n_lit > cap) { status = GZ_FULL; break; }
                        flush();
                        end_bit = br_pos(br);
                        status = GZ_LANDED;
                        break;
                    }
                    const int sym = decode_sym(br, sh.ll_count, sh.ll_symbol, sh.ll_table, GZ_FAST_LL);
                    if ((uint32_t)sym < 256u) {
                        lit = lane == n_lit ? (uint32_t)sym : lit;
                        if (++n_lit == 64) {
                            if (br.next > limit_words) { status = GZ_SHORT; break; }
                            if (o + 64 > cap) { status = GZ_FULL; break; }
                            flush();
                        }
                        continue;
                    }
                    if (sym < 0) { status = GZ_BAD; break; }
                    if (br.next > limit_words) { status = GZ_SHORT; break; }
                    if (o + n_lit > cap) { status = GZ_FULL; break; }
                    flush();
                    if (sym == 256) break;
                    const int ls = sym - 257;
                    if (ls >= 29) { status = GZ_BAD; break; }
                    const uint32_t len = inf_len_base((uint32_t)ls) + br_bits(br, inf_len_extra((uint32_t)ls));
                    const int ds = decode_sym(br, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
                    if (ds < 0 || ds >= 30) { status = GZ_BAD; break; }
                    const uint32_t extra = inf_dist_extra((uint32_t)ds);
                    br_need(br, 16);
                    const uint32_t dist = inf_dist_base((uint32_t)ds) + ((uint32_t)br.buf & ((1u << extra) - 1u));
                    br.buf >>= extra; br.cnt -= extra;
                    if (o + len > cap) { status = GZ_FULL; break; }
                    // a source in front of this wave's first symbol is text somebody else decodes: a marker for position
                    // 32768 + src of the window in front of the stretch stands in for it
                    const int32_t from = (int32_t)o - (int32_t)dist;             // (a stretch holds fewer than 2^31 symbols)
                    if (dist + 258u <= (1u << RBITS)) {
                        for (uint32_t j = lane; j < len; j += 64) {
                            const int32_t src = dist >= len ? from + (int32_t)j : from + (int32_t)(j % dist);
                            const uint16_t v = src < 0 ? (uint16_t)(GZ_MARK | (uint32_t)((int32_t)GZ_WIN + src)) : sh.ring[(uint32_t)src & RMASK];
                            sh.ring[(o + j) & RMASK] = v;
                            dst[o + j] = v;
                        }
                    } else {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        for (uint32_t j = lane; j < len; j += 64) {                    // dist > 258 >= len here: no overlap
                            const int32_t src = from + (int32_t)j;
                            const uint16_t v = src < 0 ? (uint16_t)(GZ_MARK | (uint32_t)((int32_t)GZ_WIN + src))
                                                       : __hip_atomic_load(dst + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            sh.ring[(o + j) & RMASK] = v;
                            dst[o + j] = v;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    o += len;
                }
                if (status >= 0) break;
            }
            if (last_block) {
                // the member's trailer (CRC-32, ISIZE), then the end of the file or another member
                uint64_t at = ((br_pos(br) + 7) >> 3) + 8;
                if (at > n_bytes) { status = GZ_SHORT; break; }
                isize_sum += INF_UNI(comp[at - 4] | (comp[at - 3] << 8) | (comp[at - 2] << 16) | ((uint32_t)comp[at - 1] << 24));
                if (members == 0) {
                    trailer_at = o;
                    trailer_crc = INF_UNI(comp[at - 8] | (comp[at - 7] << 8) | (comp[at - 6] << 16) | ((uint32_t)comp[at - 5] << 24));
                }
                members += 1;
                if (at == n_bytes) {
                    end_bit = at * 8;
                    status = is_file_end ? GZ_END : GZ_SHORT;
                    break;
                }
                uint64_t data = 0;
                if (lane == 0) data = gz_member_header(comp, at, n_bytes);
                data = ((uint64_t)INF_UNI((uint32_t)(data >> 32)) << 32) | INF_UNI((uint32_t)data);
                if (data == ~0ull) { status = is_file_end ? GZ_BAD : GZ_SHORT; break; }
                if (data == 0) { status = GZ_BAD; break; }
                br_seek(br, words, data * 8);
            }
        }
        if (lane == 0) {
            GzResult r;
            r.end_bit = end_bit;
            r.n_out = o;
            r.status = (uint16_t)status;
            r.members = (uint16_t)min(members, 65535u);
            r.isize_sum = isize_sum;
            r.trailer_at = trailer_at;
            r.trailer_crc = trailer_crc;
            r.pad = 0;
            results[job_id] = r;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------- 3. tails
// tail q (q = 0: the window the segment starts with, as bytes; q > 0: stretch q - 1): the 32 K symbols in front of
// stretch q.  A stretch shorter than the window passes the rest of the tail before it through as markers.
__device__ __forceinline__ uint32_t lanes_below(uint64_t ballot)       // set bits of `ballot` in lanes below this one
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0u));
}

__global__ __launch_bounds__(256) void k_gz_tails(const uint16_t *__restrict__ syms, const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ n_out,
                                                  const uint8_t *__restrict__ window_in, uint16_t *__restrict__ tails, uint16_t *__restrict__ tails_b,
                                                  uint32_t *__restrict__ list, uint64_t list_cap, unsigned long long *__restrict__ n_markers)
{
    __shared__ uint32_t sh_wave[4];
    __shared__ unsigned long long sh_base;
    const uint32_t q = blockIdx.x;
    uint16_t *t = tails + (size_t)q * GZ_WIN, *tb = tails_b + (size_t)q * GZ_WIN;
    if (q == 0) {
        for (uint32_t i = threadIdx.x; i < GZ_WIN; i += blockDim.x) { const uint16_t v = window_in[i]; t[i] = v; tb[i] = v; }
        return;
    }
    const uint32_t n = n_out[q - 1];
    const uint16_t *src = syms + out_off[q - 1];
    // a wave takes a quarter of the tail, 64 consecutive positions at a time, so that the markers' places can be listed in
    // the order of the positions: the rounds of pointer doubling then touch neighbouring symbols from neighbouring lanes
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, first = wave * (GZ_WIN / 4);
    uint64_t flags[2] = {0, 0};                        // this lane's markers: bit k for position first + 64 k + lane
    uint32_t total = 0;
    for (uint32_t k = 0; k < GZ_WIN / 256; ++k) {
        const uint32_t i = first + 64u * k + lane;
        const int64_t rel = (int64_t)n - (int64_t)GZ_WIN + i;
        const uint16_t v = rel >= 0 ? src[rel] : (uint16_t)(GZ_MARK | (uint32_t)(GZ_WIN + rel));
        const bool is_marker = (v & GZ_MARK) != 0;
        if (is_marker) flags[k >> 6] |= 1ull << (k & 63u);
        total += (uint32_t)__popcll(__ballot(is_marker));
        t[i] = v;
        tb[i] = v;
    }
    if (lane == 0) sh_wave[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t sum = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
        sh_base = sum ? atomicAdd(n_markers, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    unsigned long long at = sh_base;
    for (uint32_t w = 0; w < wave; ++w) at += sh_wave[w];
    for (uint32_t k = 0; k < GZ_WIN / 256 && total; ++k) {
        const bool is_marker = (flags[k >> 6] >> (k & 63u)) & 1ull;
        const uint64_t ballot = __ballot(is_marker);
        const unsigned long long mine = at + lanes_below(ballot);
        if (is_marker && mine < list_cap) list[mine] = (q << 15) | (first + 64u * k + lane);
        at += (uint32_t)__popcll(ballot);
    }
}

// One round of pointer doubling over the list of markers: entry (q, i) names position i of tail q, whose marker (in `src`)
// names a position of tail q - d.  The new value goes to `dst`; a byte (the end of the chain) goes to BOTH copies, so
// that whichever one a later reader looks at is right, and leaves the list; a marker stays on it, in the same order,
// for the next round.
__global__ __launch_bounds__(256) void k_gz_scan_list(uint16_t *__restrict__ src, uint16_t *__restrict__ dst, const uint32_t *__restrict__ list, uint64_t n_list,
                                                      uint32_t *__restrict__ list_out, uint32_t d, unsigned long long *__restrict__ n_out)
{
    __shared__ uint32_t sh_wave[4];
    __shared__ unsigned long long sh_base;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t first = (uint64_t)blockIdx.x * 4096u + wave * 1024u;        // 16 rows of 64 entries per wave
    uint32_t keep = 0, total = 0;
    for (uint32_t k = 0; k < 16; ++k) {
        const uint64_t at = first + 64u * k + lane;
        bool stays = false;
        if (at < n_list) {
            const uint32_t e = list[at], q = e >> 15, i = e & (GZ_WIN - 1u);
            if (q >= d) {                              // (always: the tails in front of the stride are plain)
                const uint16_t v = src[(size_t)q * GZ_WIN + i];
                const uint16_t w = (v & GZ_MARK) ? src[(size_t)(q - d) * GZ_WIN + (v & (GZ_WIN - 1u))] : v;
                dst[(size_t)q * GZ_WIN + i] = w;
                stays = (w & GZ_MARK) != 0;
                if (!stays) src[(size_t)q * GZ_WIN + i] = w;
            }
        }
        if (stays) keep |= 1u << k;
        total += (uint32_t)__popcll(__ballot(stays));
    }
    if (lane == 0) sh_wave[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t sum = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
        sh_base = sum ? atomicAdd(n_out, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    unsigned long long out = sh_base;
    for (uint32_t w = 0; w < wave; ++w) out += sh_wave[w];
    for (uint32_t k = 0; k < 16 && total; ++k) {
        const bool stays = (keep >> k) & 1u;
        const uint64_t ballot = __ballot(stays);
        if (stays) list_out[out + lanes_below(ballot)] = list[first + 64u * k + lane];
        out += (uint32_t)__popcll(ballot);
    }
}

// the same round over whole tails (when the list of markers would not fit its buffer): the markers of tail q name
// positions of tail q - d.  *n_markers: how many are left.
__global__ __launch_bounds__(256) void k_gz_scan(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, uint32_t d, unsigned long long *__restrict__ n_markers)
{
    const uint32_t q = blockIdx.x;
    const uint2 *t = (const uint2 *)(in + (size_t)q * GZ_WIN);
    uint2 *o = (uint2 *)(out + (size_t)q * GZ_WIN);
    if (q < d) {
        for (uint32_t i = threadIdx.x; i < GZ_WIN / 4; i += blockDim.x) o[i] = t[i];
        return;
    }
    const uint16_t *before = in + (size_t)(q - d) * GZ_WIN;
    uint32_t marked = 0;
    for (uint32_t i = threadIdx.x; i < GZ_WIN / 4; i += blockDim.x) {          // four symbols per thread and step
        uint2 v = t[i];
        uint32_t s0 = v.x & 0xffffu, s1 = v.x >> 16, s2 = v.y & 0xffffu, s3 = v.y >> 16;
        if ((v.x | v.y) & 0x80008000u) {
            if (s0 & GZ_MARK) s0 = before[s0 & (GZ_WIN - 1u)];
            if (s1 & GZ_MARK) s1 = before[s1 & (GZ_WIN - 1u)];
            if (s2 & GZ_MARK) s2 = before[s2 & (GZ_WIN - 1u)];
            if (s3 & GZ_MARK) s3 = before[s3 & (GZ_WIN - 1u)];
            v.x = s0 | (s1 << 16);
            v.y = s2 | (s3 << 16);
            marked += (uint32_t)__popc((v.x | 0u) & 0x80008000u) + (uint32_t)__popc(v.y & 0x80008000u);
        }
        o[i] = v;
    }
    if (marked) atomicAdd(n_markers, (unsigned long long)marked);
}

// ---------------------------------------------------------------- 4. bytes
#define GZ_RESOLVE_SPLIT 8u
__global__ void k_gz_resolve(const uint16_t *__restrict__ syms, const uint64_t *__restrict__ out_off, const uint32_t *__restrict__ n_out,
                             const uint64_t *__restrict__ text_base, const uint16_t *__restrict__ tails, uint8_t *__restrict__ text)
{
    const uint32_t q = blockIdx.x / GZ_RESOLVE_SPLIT, part = blockIdx.x % GZ_RESOLVE_SPLIT;
    const uint16_t *t = tails + (size_t)q * GZ_WIN;             // the window in front of stretch q
    const uint16_t *src = syms + out_off[q];
    uint8_t *dst = text + text_base[q];
    const uint32_t n = n_out[q];
    for (uint32_t i = part * blockDim.x + threadIdx.x; i < n; i += GZ_RESOLVE_SPLIT * blockDim.x) {
        uint16_t v = src[i];
        if (v & GZ_MARK) v = t[v & (GZ_WIN - 1u)];
        dst[i] = (uint8_t)v;
    }
}

__global__ void k_gz_window_out(const uint16_t *__restrict__ tail, uint8_t *__restrict__ window)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < GZ_WIN; i += gridDim.x * blockDim.x) window[i] = (uint8_t)tail[i];
}

// ---------------------------------------------------------------- 5. CRC-32
// out[r] = CRC-32 (the gzip one: reflected 0xEDB88320) of text[start[r], start[r] + len[r]): a thread per range, four table
// lookups per four bytes (slicing by 4, tables built in LDS); the host joins the ranges of a member (crc32_combine)
__global__ __launch_bounds__(256) void k_gz_crc(const uint8_t *__restrict__ text, const uint64_t *__restrict__ start, const uint32_t *__restrict__ len, uint32_t n,
                                                uint32_t *__restrict__ out)
{
    __shared__ uint32_t T[4][256];
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        T[0][threadIdx.x] = c;
        __syncthreads();
        uint32_t v = c;
        for (int t = 1; t < 4; ++t) {
            v = T[0][v & 0xffu] ^ (v >> 8);
            T[t][threadIdx.x] = v;
        }
        __syncthreads();
    }
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint8_t *p = text + start[r];
    uint32_t left = len[r], c = 0xffffffffu;
    while (left && ((uint64_t)p & 15u)) { c = T[0][(c ^ *p++) & 0xffu] ^ (c >> 8); --left; }
    for (; left >= 16; left -= 16, p += 16) {
        const uint4 q = *(const uint4 *)p;
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c ^= w[i];
            c = T[3][c & 0xffu] ^ T[2][(c >> 8) & 0xffu] ^ T[1][(c >> 16) & 0xffu] ^ T[0][c >> 24];
        }
    }
    for (; left; --left) c = T[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    out[r] = c ^ 0xffffffffu;
}

// start of the deflate data of the gzip member at byte `at` of a host image; 0 if there is no member header there
uint64_t host_member_header(const uint8_t *file, uint64_t size, uint64_t at)
{
    if (at + 18 > size) return 0;
    if (file[at] != 0x1f || file[at + 1] != 0x8b || file[at + 2] != 8 || (file[at + 3] & 0xe0)) return 0;
    const uint32_t flags = file[at + 3];
    uint64_t p = at + 10;
    if (flags & 4) {
        const uint32_t xlen = file[p] | (file[p + 1] << 8);
        p += 2 + xlen;
    }
    for (int field = 0; field < 2; ++field)
        if (flags & (field == 0 ? 8 : 16)) {
            while (p < size && file[p]) ++p;
            ++p;
        }
    if (flags & 2) p += 2;
    return p + 8 <= size ? p : 0;
}

}  // namespace

// CRC-32 of ranges of text that sits on the device: out[r] for d_text[start[r], start[r] + len[r]); returns when they are there
int kv_crc32_ranges(const uint8_t *d_text, const uint64_t *start, const uint32_t *len, size_t n, uint32_t *out, KvArena &scratch)
{
    if (n == 0) return KV_OK;
    hipStream_t st = kv_stream();
    const size_t b_start = kv_round_up(n * 8, 256), b_len = kv_round_up(n * 4, 256);
    KV_HIP(scratch.need(b_start + 2 * b_len));
    uint64_t *d_start = (uint64_t *)scratch.p;
    uint32_t *d_len = (uint32_t *)((unsigned char *)scratch.p + b_start), *d_out = (uint32_t *)((unsigned char *)scratch.p + b_start + b_len);
    KV_HIP(hipMemcpyAsync(d_start, start, n * 8, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemcpyAsync(d_len, len, n * 4, hipMemcpyHostToDevice, st));
    {
        KvProfScope prof("k_gz_crc");
        hipLaunchKernelGGL(k_gz_crc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_text, (const uint64_t *)d_start, (const uint32_t *)d_len, (uint32_t)n, d_out);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(out, d_out, n * 4, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

// CRC-32 of A followed by B from the CRC-32 of each and the length of B (zlib's crc32_combine).  Appending len_b zero bytes is
// a linear map of the CRC: for the lengths that keep coming back (the slice length; the last slice of a full BGZF member) it
// is kept as a GF(2) matrix, 32 XORs instead of zlib's chain of matrix squarings (~3 us a call).
uint32_t kv_crc32_join(uint32_t crc_a, uint32_t crc_b, uint32_t len_b)
{
    struct Op { uint32_t len = 0; uint32_t col[32]; };
    static thread_local Op ops[3];                     // [0]: the slice length; [1], [2]: other lengths that came up twice in a row
    static thread_local uint32_t last_miss = 0, victim = 1;
    if (len_b == 0) return crc_a;
    if (crc_a == 0) return crc_b;                      // A is empty
    Op *op = nullptr;
    for (Op &o : ops)
        if (o.len == len_b) { op = &o; break; }
    if (!op) {
        if (len_b != KV_CRC_SLICE && len_b != last_miss) {           // a length seen once is not worth the 32 calls that build its matrix
            last_miss = len_b;
            return (uint32_t)crc32_combine(crc_a, crc_b, len_b);
        }
        op = len_b == KV_CRC_SLICE ? &ops[0] : &ops[victim];
        if (len_b != KV_CRC_SLICE) victim = 3 - victim;
        op->len = len_b;
        for (int b = 0; b < 32; ++b) op->col[b] = (uint32_t)crc32_combine(1ul << b, 0, len_b);
    }
    uint32_t moved = 0;
    for (uint32_t v = crc_a, b = 0; v; v >>= 1, ++b)
        if (v & 1u) moved ^= op->col[b];
    return moved ^ crc_b;
}

// ---------------------------------------------------------------- host side
struct KvGunzip {
    const uint8_t *image = nullptr;
    uint64_t size = 0;
    uint64_t pos_bit = 0;             // of the next block, in the file
    bool done = false;
    double ratio = 4.0;               // text bytes per compressed byte so far
    uint64_t seen_comp = 0, seen_text = 0;
    uint32_t isize_total = 0;         // of the members that have ended so far (mod 2^32, as the field is)
    uint32_t crc_run = 0;             // CRC-32 of the text of the member that is open at pos_bit
    bool crc_on = true;               // off: KV_GUNZIP_CRC=0, or a stretch passed more than one member end (their starts in the text are not recorded)
    std::vector<std::pair<uint64_t, uint32_t>> pending_ends;      // of the pending segment: (text offset a member ends at, the CRC-32 its trailer holds)

    uint32_t chunk_bytes = 16384;
    KvGunzipArenas own, *a = &own;    // device buffers: the caller's (pooled across files) or this object's
    // the segment decoded by kv_gunzip_decode and not yet emitted
    std::vector<uint64_t> v_off, v_base;
    std::vector<uint32_t> v_n;
    uint64_t pending_text = 0, pending_pos = 0;
    uint32_t pending_isize = 0;
    bool pending_done = false, pending = false, window_ready = false;
    const uint64_t *d_off = nullptr, *d_base = nullptr;           // device copies of v_off / v_base / v_n
    const uint32_t *d_n = nullptr;
    const uint16_t *d_tails = nullptr;                            // the resolved tails of the pending segment
    uint64_t stat_jobs = 0, stat_dropped = 0, stat_repairs = 0, stat_segments = 0, stat_rounds = 0, stat_cuts = 0;
    // optional: bytes [off, off + n) of the file to d_dst on the stream, through the caller's pinned staging buffers (false: not done)
    std::function<bool(uint8_t *, uint64_t, uint64_t, hipStream_t)> upload;
    ~KvGunzip() { own.release(); }
};

void kv_gunzip_set_uploader(KvGunzip *g, std::function<bool(uint8_t *, uint64_t, uint64_t, hipStream_t)> upload)
{
    if (g) g->upload = std::move(upload);
}

KvGunzip *kv_gunzip_open(const uint8_t *image, uint64_t size, KvGunzipArenas *arenas)
{
    const uint64_t data = host_member_header(image, size, 0);
    if (!data) return nullptr;
    KvGunzip *g = new KvGunzip();
    g->image = image;
    g->size = size;
    if (arenas) g->a = arenas;
    g->pos_bit = data * 8;
    const char *cb = kv_knob("KV_GUNZIP_CHUNK_KB");
    if (cb && atoi(cb) >= 1 && atoi(cb) <= 64) g->chunk_bytes = (uint32_t)atoi(cb) * 1024u;
    const char *cc = kv_knob("KV_GUNZIP_CRC");
    g->crc_on = !(cc && !strcmp(cc, "0"));
    return g;
}

void kv_gunzip_close(KvGunzip *g) { delete g; }
bool kv_gunzip_done(const KvGunzip *g) { return g->done; }
double kv_gunzip_ratio(const KvGunzip *g) { return g->ratio; }
void kv_gunzip_stats(const KvGunzip *g, uint64_t out[4])
{
    out[0] = g->stat_segments; out[1] = g->stat_jobs; out[2] = g->stat_dropped; out[3] = g->stat_repairs;
}

static int gz_run_jobs(KvGunzip *g, const uint8_t *d_comp, uint64_t n_bytes, bool is_file_end, const GzJob *jobs, size_t n, GzJob *d_jobs,
                       const unsigned long long *d_starts, const unsigned long long *d_headers, uint64_t terminal_bit, bool exact, GzResult *d_results, unsigned long long *d_ctr, uint16_t *d_syms,
                       GzResult *results)
{
    hipStream_t st = kv_stream();
    KV_HIP(hipMemcpyAsync(d_jobs, jobs, n * sizeof(GzJob), hipMemcpyHostToDevice, st));
    KV_HIP(hipMemsetAsync(d_ctr, 0, 8, st));
    {
        KvProfScope prof("k_gz_decode");
        // the LDS window sets how many stretches a CU holds (1 K symbols: 32 = 8 waves per SIMD); a match that reaches further
        // back reads the symbols the wave itself stored to HBM, behind a workgroup-scope release
        const char *rb = kv_knob("KV_GUNZIP_RING_BITS");
        const int bits = rb ? atoi(rb) : 10;
        const int per_cu = bits >= 14 ? 4 : bits == 13 ? 8 : bits == 12 ? 12 : bits == 11 ? 24 : 32;
        const unsigned grid = (unsigned)std::min<uint64_t>(n, (uint64_t)per_cu * (uint64_t)kv_device_cus());
#define KV_LAUNCH_GZ(B_, E_) hipLaunchKernelGGL((k_gz_decode<B_, E_>), dim3(grid), dim3(64), 0, st, d_comp, n_bytes, is_file_end ? 1 : 0, (const GzJob *)d_jobs, (uint32_t)n, d_starts, d_headers, terminal_bit, d_syms, d_results, d_ctr)
        if (exact) KV_LAUNCH_GZ(10, true);
        else if (bits >= 14) KV_LAUNCH_GZ(14, false);
        else if (bits == 13) KV_LAUNCH_GZ(13, false);
        else if (bits == 12) KV_LAUNCH_GZ(12, false);
        else if (bits == 11) KV_LAUNCH_GZ(11, false);
        else KV_LAUNCH_GZ(10, false);
#undef KV_LAUNCH_GZ
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(results, d_results, n * sizeof(GzResult), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    (void)g;
    return KV_OK;
}

// Decode the next segment: about `want_text` bytes of text (at least one chunk).  *text_bytes = what kv_gunzip_emit will
// deliver.  KV_ERR_TYPE: the stream is not something this decoder handles (the caller falls back to zlib).
int kv_gunzip_decode(KvGunzip *g, uint64_t want_text, uint64_t *text_bytes, bool *last)
{
    *text_bytes = 0;
    *last = g->done;
    KV_REQUIRE(!g->pending, KV_ERR_ARG, "kv_gunzip_decode: the previous segment has not been emitted");
    if (g->done) return KV_OK;
    hipStream_t st = kv_stream();
    const bool verbose = kv_knob("KV_INGEST_VERBOSE") != nullptr;
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kv_gunzip]   %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
    };
    const uint32_t CH = g->chunk_bytes;
    const uint64_t margin = 4ull << 20;
    // ---- the compressed bytes of the segment, and behind them a margin in which the next segment's first block is looked for
    const uint64_t first_byte = (g->pos_bit >> 3) & ~255ull;
    const uint64_t seg = std::max<uint64_t>((uint64_t)((double)want_text / g->ratio), 4ull * CH);
    const uint64_t seg_end = std::min<uint64_t>(kv_round_up(first_byte + seg, CH), g->size);
    const uint64_t upto = std::min<uint64_t>(seg_end + margin, g->size);
    const bool to_file_end = seg_end == g->size, is_file_end = upto == g->size;
    const uint64_t n_bytes = upto - first_byte;
    const uint32_t n_chunks = (uint32_t)((n_bytes + CH - 1) / CH);
    KV_HIP(g->a->comp.need(kv_round_up((uint64_t)n_chunks * CH + 2 * GZ_SLACK, 4096)));
    uint8_t *d_comp = (uint8_t *)g->a->comp.p;
    // (big stretches through the caller's pinned staging buffers if it has any: kv_gunzip_set_uploader)
    if (!(g->upload && n_bytes >= (64u << 20) && g->upload(d_comp, first_byte, n_bytes, st)))
        KV_HIP(hipMemcpyAsync(d_comp, g->image + first_byte, n_bytes, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemsetAsync(d_comp + n_bytes, 0, (uint64_t)n_chunks * CH + 2 * GZ_SLACK - n_bytes, st));
    lap("upload");
    const uint64_t b_cand = kv_round_up((uint64_t)n_chunks * GZ_FIND_KEEP * 8, 256);
    const uint64_t max_guesses = (uint64_t)n_chunks * GZ_FIND_KEEP * 7 + 8;          // seven cuts per stretch at most
    KV_HIP(g->a->small.need(b_cand + 256 + kv_round_up(max_guesses * sizeof(GzGuess), 256) + kv_round_up(max_guesses * 8, 256)));
    unsigned long long *d_cand = (unsigned long long *)g->a->small.p;
    unsigned long long *d_ctr = (unsigned long long *)((unsigned char *)g->a->small.p + b_cand);
    GzGuess *d_guesses = (GzGuess *)((unsigned char *)g->a->small.p + b_cand + 256);
    unsigned long long *d_found = (unsigned long long *)((unsigned char *)d_guesses + kv_round_up(max_guesses * sizeof(GzGuess), 256));
    {
        KvProfScope prof("k_gz_find");
        hipLaunchKernelGGL(k_gz_find, dim3(n_chunks), dim3(GZ_FIND_THREADS), CH + GZ_SLACK + 8, st, (const uint8_t *)d_comp, n_bytes, CH, n_chunks, d_cand);
    }
    KV_HIP(hipGetLastError());
    std::vector<unsigned long long> cand((uint64_t)n_chunks * GZ_FIND_KEEP);
    KV_HIP(hipMemcpyAsync(cand.data(), d_cand, (uint64_t)n_chunks * GZ_FIND_KEEP * 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    lap("find");
    // ---- stretches: from the exact position the last segment ended at, then from every found start up to the first one
    // behind the segment
    const uint64_t start_rel = g->pos_bit - first_byte * 8, seg_end_rel = (seg_end - first_byte) * 8;
    std::vector<uint64_t> starts(1, start_rel);
    bool have_terminal = false;
    for (uint64_t c = 0; c < (uint64_t)n_chunks * GZ_FIND_KEEP && !have_terminal; ++c) {
        if (cand[c] == ~0ull || cand[c] <= start_rel) continue;
        starts.push_back(cand[c]);
        if (!to_file_end && cand[c] >= seg_end_rel) have_terminal = true;
    }
    if (!to_file_end && !have_terminal && !is_file_end) {
        kv_set_error("no DEFLATE block start within %llu MB behind byte %llu", (unsigned long long)(margin >> 20), (unsigned long long)seg_end);
        return KV_ERR_TYPE;
    }                                                  // (none before the end of the file: the last stretch runs to the end)
    const uint64_t stop_rel = have_terminal ? starts.back() : ~0ull;             // reaching it ends the segment
    // ---- long stretches are cut: symbol boundaries inside their first block, found by letting 64 decoders per cut fall
    // into step (k_gz_sync).  KV_GUNZIP_SPLIT_KB: compressed bytes a piece should have (0: no cutting)
    std::vector<uint64_t> headers(starts);              // headers[i]: the block header stretch i takes its codes from
    {
        const char *sk = kv_knob("KV_GUNZIP_SPLIT_KB");
        // default: only when the block starts alone leave the device short of work (fewer than 24 stretches per CU), and
        // then as many pieces as make 32 per CU: every cut costs a header to parse, a tail to resolve and a decoder that
        // looks for its target between symbols too (k_gz_decode<.., true>, ~15 % more instructions per symbol)
        const uint64_t wanted = 32ull * (uint64_t)kv_device_cus();
        const uint64_t piece = sk ? strtoull(sk, nullptr, 10) * 1024ull * 8ull
                                  : starts.size() * 4 >= wanted * 3 ? 0 : std::max<uint64_t>(4096ull * 8ull, (seg_end_rel - start_rel) / wanted);
        std::vector<GzGuess> guesses;
        const size_t n_blocks = have_terminal ? starts.size() - 1 : starts.size();
        for (size_t j = 0; j < n_blocks && piece; ++j) {
            const uint64_t next = j + 1 < starts.size() ? starts[j + 1] : n_bytes * 8;
            const uint64_t span = next - starts[j];
            const uint64_t cuts = std::min<uint64_t>(8, span / piece);
            for (uint64_t i = 1; i < cuts; ++i) {
                GzGuess gs;
                gs.header_bit = starts[j];
                gs.guess_bit = starts[j] + i * (span / cuts);
                if (guesses.size() < max_guesses) guesses.push_back(gs);
            }
        }
        if (!guesses.empty()) {
            KV_HIP(hipMemcpyAsync(d_guesses, guesses.data(), guesses.size() * sizeof(GzGuess), hipMemcpyHostToDevice, st));
            {
                KvProfScope prof("k_gz_sync");
                hipLaunchKernelGGL(k_gz_sync, dim3((unsigned)guesses.size()), dim3(64), 0, st, (const uint8_t *)d_comp, n_bytes, (const GzGuess *)d_guesses,
                                   (uint32_t)guesses.size(), d_found);
            }
            KV_HIP(hipGetLastError());
            std::vector<unsigned long long> found(guesses.size());
            KV_HIP(hipMemcpyAsync(found.data(), d_found, guesses.size() * 8, hipMemcpyDeviceToHost, st));
            KV_HIP(hipStreamSynchronize(st));
            std::vector<std::pair<uint64_t, uint64_t>> all;             // (start, header)
            for (size_t i = 0; i < starts.size(); ++i) all.emplace_back(starts[i], starts[i]);
            for (size_t i = 0; i < guesses.size(); ++i) {
                if (found[i] == ~0ull || found[i] <= guesses[i].header_bit || found[i] >= stop_rel) continue;
                // (a boundary behind the next block start is of no use: the stretch before it stops at that start)
                const auto nx = std::upper_bound(starts.begin(), starts.end(), guesses[i].header_bit);
                if (nx != starts.end() && found[i] >= *nx) continue;
                all.emplace_back(found[i], guesses[i].header_bit);
                g->stat_cuts += 1;
            }
            std::sort(all.begin(), all.end());
            starts.clear(); headers.clear();
            for (const auto &a : all)
                if (starts.empty() || a.first != starts.back()) { starts.push_back(a.first); headers.push_back(a.second); }
        }
        lap("cuts");
    }
    const size_t n_first = have_terminal ? starts.size() - 1 : starts.size();
    bool exact = false;                                 // does any stretch begin inside a block?
    for (size_t i = 0; i < starts.size() && !exact; ++i) exact = headers[i] != starts[i];
    const double factor = std::min(std::max(2.5 * g->ratio, 8.0), 64.0);
    std::vector<GzJob> jobs(n_first);
    uint64_t total_cap = 0;
    for (size_t j = 0; j < n_first; ++j) {
        // room: up to the next stretch that begins with a block (the starts inside this block may turn out not to be any, and
        // the stretch then runs on over them)
        size_t nb = j + 1;
        while (nb < starts.size() && headers[nb] != starts[nb]) ++nb;
        const uint64_t next = nb < starts.size() ? starts[nb] : n_bytes * 8;
        jobs[j].start_bit = starts[j];
        jobs[j].header_bit = headers[j];
        jobs[j].target_idx = j + 1;                     // (starts[] on the device ends with ~0)
        jobs[j].out_off = total_cap;
        jobs[j].out_cap = (uint32_t)std::min<uint64_t>((uint64_t)((double)((next - starts[j]) / 8 + 1) * factor) + 16384, 0x7ffffff0u);
        jobs[j].pad = 0;
        total_cap += kv_round_up(jobs[j].out_cap, 64);
    }
    const uint64_t repair_room = std::max<uint64_t>(256ull << 20, total_cap / 4);
    const size_t max_jobs = n_first + 256;
    KV_HIP(g->a->syms.need((total_cap + repair_room) * 2 + 256));
    KV_HIP(g->a->meta.need(kv_round_up(max_jobs * sizeof(GzJob), 256) + kv_round_up(max_jobs * sizeof(GzResult), 256) + kv_round_up(max_jobs * 8, 256) * 2 +
                        kv_round_up(max_jobs * 4, 256) + 2 * kv_round_up((starts.size() + 1) * 8, 256)));
    GzJob *d_jobs = (GzJob *)g->a->meta.p;
    GzResult *d_results = (GzResult *)((unsigned char *)d_jobs + kv_round_up(max_jobs * sizeof(GzJob), 256));
    uint64_t *d_off = (uint64_t *)((unsigned char *)d_results + kv_round_up(max_jobs * sizeof(GzResult), 256));
    uint64_t *d_base = (uint64_t *)((unsigned char *)d_off + kv_round_up(max_jobs * 8, 256));
    uint32_t *d_n = (uint32_t *)((unsigned char *)d_base + kv_round_up(max_jobs * 8, 256));
    unsigned long long *d_starts = (unsigned long long *)((unsigned char *)d_n + kv_round_up(max_jobs * 4, 256));
    unsigned long long *d_headers = (unsigned long long *)((unsigned char *)d_starts + kv_round_up((starts.size() + 1) * 8, 256));
    {
        std::vector<unsigned long long> with_end(starts.begin(), starts.end()), kinds(headers.begin(), headers.end());
        with_end.push_back(~0ull);
        kinds.push_back(~0ull);
        KV_HIP(hipMemcpyAsync(d_starts, with_end.data(), with_end.size() * 8, hipMemcpyHostToDevice, st));
        KV_HIP(hipMemcpyAsync(d_headers, kinds.data(), kinds.size() * 8, hipMemcpyHostToDevice, st));
        KV_HIP(hipStreamSynchronize(st));               // (the vectors go out of scope)
    }
    uint16_t *d_syms = (uint16_t *)g->a->syms.p;
    std::vector<GzResult> results(n_first);
    { const int rc = gz_run_jobs(g, d_comp, n_bytes, is_file_end, jobs.data(), n_first, d_jobs, d_starts, d_headers, stop_rel, exact, d_results, d_ctr, d_syms, results.data()); if (rc != KV_OK) return rc; }
    lap("jobs + decode");
    // ---- the chain: stretch 0 starts at a known block; a later one counts iff a good stretch ends exactly on its start
    g->v_off.clear(); g->v_n.clear(); g->v_base.clear(); g->pending_ends.clear();
    bool crc_lost = false;
    auto note_ends = [&](const GzResult &r, uint64_t base) {
        if (r.members >= 1) g->pending_ends.emplace_back(base + r.trailer_at, r.trailer_crc);
        if (r.members > 1) crc_lost = true;
    };
    uint64_t repair_used = 0, text = 0, end_rel = 0;
    uint32_t repairs = 0, isize_seg = 0;
    bool ended = false;
    GzJob cur_job = jobs[0];
    GzResult cur = results[0];
    for (;;) {
        if (cur.status == GZ_FULL || (cur.status == GZ_LANDED && cur.end_bit < stop_rel && !std::binary_search(starts.begin() + 1, starts.end(), cur.end_bit))) {
            // out of room, or the stretch ran over a false start and stopped at a block nobody began at: decode again, from
            // this stretch's start (for room) or from where it stopped, as a stretch of its own
            GzJob again;
            if (cur.status == GZ_FULL) {
                if (cur_job.out_off >= total_cap) repair_used = cur_job.out_off - total_cap;        // the failed attempt's room is free again
                again = cur_job;
                again.out_cap = (uint32_t)std::min<uint64_t>((uint64_t)cur_job.out_cap * 32, 0x7ffffff0u);
            } else {
                g->v_off.push_back(cur_job.out_off); g->v_n.push_back(cur.n_out); g->v_base.push_back(text);
                note_ends(cur, text);
                text += cur.n_out;
                isize_seg += cur.isize_sum;
                again.start_bit = cur.end_bit;
                again.header_bit = cur.end_bit;
                const auto nx = std::upper_bound(starts.begin() + 1, starts.end(), cur.end_bit);
                again.target_idx = (uint64_t)(nx - starts.begin());
                const uint64_t next = nx == starts.end() ? n_bytes * 8 : *nx;
                again.out_cap = (uint32_t)std::min<uint64_t>((uint64_t)((double)((next - again.start_bit) / 8 + 1) * factor) + 16384, 0x7ffffff0u);
            }
            again.pad = 0;
            again.out_off = total_cap + repair_used;
            repair_used += kv_round_up(again.out_cap, 64);
            if (++repairs > 200 || repair_used > repair_room || (cur.status == GZ_FULL && cur_job.out_cap >= 0x7ffffff0u)) {
                kv_set_error("gzip stream at byte %llu: too many stretches needed decoding again", (unsigned long long)(first_byte + cur_job.start_bit / 8));
                return KV_ERR_TYPE;
            }
            GzResult r;
            { const int rc = gz_run_jobs(g, d_comp, n_bytes, is_file_end, &again, 1, d_jobs, d_starts, d_headers, stop_rel, exact, d_results, d_ctr, d_syms, &r); if (rc != KV_OK) return rc; }
            cur_job = again;
            cur = r;
            continue;
        }
        if (cur.status == GZ_BAD || cur.status == GZ_SHORT) {
            kv_set_error("gzip stream: %s near byte %llu", cur.status == GZ_BAD ? "invalid DEFLATE data" : "data end inside a block",
                         (unsigned long long)(first_byte + cur.end_bit / 8));
            return KV_ERR_TYPE;
        }
        g->v_off.push_back(cur_job.out_off); g->v_n.push_back(cur.n_out); g->v_base.push_back(text);
        note_ends(cur, text);
        text += cur.n_out;
        isize_seg += cur.isize_sum;
        end_rel = cur.end_bit;
        if (cur.status == GZ_END) { ended = true; break; }
        if (cur.end_bit >= stop_rel) break;
        const size_t k = (size_t)(std::lower_bound(starts.begin() + 1, starts.end(), cur.end_bit) - starts.begin());     // exists: checked above
        cur_job = jobs[k];
        cur = results[k];
    }
    lap("chain");
    if (ended && (uint32_t)(g->seen_text + text) != (uint32_t)(g->isize_total + isize_seg)) {
        // (the CRC-32 is not computed; a stream damaged so that it still decodes to the right length goes through, as it would
        // through any reader that does not check it)
        kv_set_error("gzip stream: %llu bytes of text, but the member trailers announce another length (damaged file?)", (unsigned long long)(g->seen_text + text));
        return KV_ERR_TYPE;
    }
    g->pending_isize = isize_seg;
    if (crc_lost && g->crc_on) {
        // one stretch ran over several member trailers: the per-member CRC-32 chain is broken from here on (only the ISIZE
        // sum is still checked at the end); say so once instead of going quiet
        g->crc_on = false;
        fprintf(stderr, "[kv_gunzip] many small gzip members in one stretch: CRC-32 of the members is not checked beyond this point (sizes are)\n");
    }
    const size_t nv = g->v_off.size();
    g->stat_segments += 1;
    g->stat_jobs += n_first;
    g->stat_dropped += n_first + repairs - nv;
    g->stat_repairs += repairs;
    // ---- tails, resolved by pointer doubling
    if (!g->window_ready) {
        KV_HIP(g->a->window.need(GZ_WIN));
        KV_HIP(hipMemsetAsync(g->a->window.p, 0, GZ_WIN, st));         // nothing valid points in front of the first byte
        g->window_ready = true;
    }
    // two copies of the tails (source and destination of a round) and two lists of marker places, room for an eighth of all
    // places (with more markers than that -- DNA: zlib codes the bases as short matches all over the window, half of a tail is
    // markers -- going over the whole tails moves fewer bytes than a list would)
    const uint64_t tail_syms = (uint64_t)(nv + 1) * GZ_WIN;
    const uint64_t list_cap = nv + 1 <= (1u << 17) ? tail_syms / 8 : 0;
    KV_HIP(g->a->tails.need(tail_syms * 2 * 2 + list_cap * 4 * 2 + 256));
    uint16_t *t0 = (uint16_t *)g->a->tails.p, *t1 = t0 + tail_syms;
    uint32_t *l0 = (uint32_t *)(t1 + tail_syms), *l1 = l0 + list_cap;
    KV_HIP(hipMemcpyAsync(d_off, g->v_off.data(), nv * 8, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemcpyAsync(d_base, g->v_base.data(), nv * 8, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemcpyAsync(d_n, g->v_n.data(), nv * 4, hipMemcpyHostToDevice, st));
    unsigned long long *d_markers = d_ctr + 1;
    unsigned long long markers = 0;
    KV_HIP(hipMemsetAsync(d_markers, 0, 8, st));
    {
        KvProfScope prof("k_gz_tails");
        hipLaunchKernelGGL(k_gz_tails, dim3((unsigned)(nv + 1)), dim3(256), 0, st, (const uint16_t *)d_syms, (const uint64_t *)d_off, (const uint32_t *)d_n,
                           (const uint8_t *)g->a->window.p, t0, t1, l0, list_cap, d_markers);
    }
    KV_HIP(hipMemcpyAsync(&markers, d_markers, 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    // FASTQ needs nearly every round (each read's header is a copy of the one before it: a chain as long as the file)
    const bool listed = markers <= list_cap;
    if (kv_knob("KV_GUNZIP_VERBOSE"))
        fprintf(stderr, "[kv_gunzip] %zu stretches (%llu of them begin inside a block), %llu of %llu tail symbols are markers (%s)\n", nv,
                (unsigned long long)g->stat_cuts, markers, (unsigned long long)tail_syms, listed ? "listed" : "whole tails");
    for (uint64_t d = 1; d < nv + 1 && markers; d <<= 1) {
        KvProfScope prof("k_gz_scan");
        KV_HIP(hipMemsetAsync(d_markers, 0, 8, st));
        if (listed) {
            hipLaunchKernelGGL(k_gz_scan_list, dim3((unsigned)((markers + 4095) / 4096)), dim3(256), 0, st, t0, t1, (const uint32_t *)l0, (uint64_t)markers, l1, (uint32_t)d, d_markers);
            std::swap(l0, l1);
        } else {
            hipLaunchKernelGGL(k_gz_scan, dim3((unsigned)(nv + 1)), dim3(256), 0, st, (const uint16_t *)t0, t1, (uint32_t)d, d_markers);
        }
        KV_HIP(hipMemcpyAsync(&markers, d_markers, 8, hipMemcpyDeviceToHost, st));
        KV_HIP(hipStreamSynchronize(st));
        std::swap(t0, t1);
        g->stat_rounds += 1;
    }
    KV_HIP(hipGetLastError());
    lap("tails + scan");
    g->d_tails = t0;                          // plain text now
    g->d_off = d_off; g->d_base = d_base; g->d_n = d_n;
    g->pending = true;
    g->pending_text = text;
    g->pending_pos = first_byte * 8 + end_rel;
    g->pending_done = ended;
    *text_bytes = text;
    *last = ended;
    if (text == 0) {                          // nothing to emit (an empty member at the end of the file)
        g->pending = false;
        g->isize_total += isize_seg;
        for (const auto &end : g->pending_ends) {
            if (g->crc_on && g->crc_run != end.second) { kv_set_error("gzip stream: CRC-32 of a member does not match its text (damaged file)"); return KV_ERR_TYPE; }
            g->crc_run = 0;
        }
        g->pos_bit = g->pending_pos;
        g->done = ended;
    }
    return KV_OK;
}

// The text of the decoded segment -> d_text (device, *text_bytes of kv_gunzip_decode), on the calling thread's stream.
int kv_gunzip_emit(KvGunzip *g, uint8_t *d_text)
{
    KV_REQUIRE(g->pending, KV_ERR_ARG, "kv_gunzip_emit: nothing decoded");
    hipStream_t st = kv_stream();
    const size_t nv = g->v_off.size();
    const uint16_t *tails = g->d_tails;
    {
        KvProfScope prof("k_gz_resolve");
        hipLaunchKernelGGL(k_gz_resolve, dim3((unsigned)(nv * GZ_RESOLVE_SPLIT)), dim3(256), 0, st, (const uint16_t *)g->a->syms.p, g->d_off, g->d_n, g->d_base, tails, d_text);
        hipLaunchKernelGGL(k_gz_window_out, dim3(16), dim3(256), 0, st, tails + (uint64_t)nv * GZ_WIN, (uint8_t *)g->a->window.p);
    }
    KV_HIP(hipGetLastError());
    if (g->crc_on) {
        // the text member by member (ends from the decoders), every member in slices of 16 KB: one thread each, joined here
        std::vector<uint64_t> r_start;
        std::vector<uint32_t> r_len;
        std::vector<uint32_t> r_closes;                // index into pending_ends + 1 for the range that ends a member, else 0
        uint64_t at = 0;
        size_t e = 0;
        while (at < g->pending_text || e < g->pending_ends.size()) {
            const uint64_t stop = e < g->pending_ends.size() ? g->pending_ends[e].first : g->pending_text;
            if (at == stop) {                          // a member that ends here without (more) text: an empty range closes it
                if (e >= g->pending_ends.size()) break;
                r_start.push_back(at); r_len.push_back(0); r_closes.push_back((uint32_t)++e);
                continue;
            }
            const uint32_t n = (uint32_t)std::min<uint64_t>(stop - at, KV_CRC_SLICE);
            r_start.push_back(at); r_len.push_back(n);
            at += n;
            r_closes.push_back(at == stop && e < g->pending_ends.size() ? (uint32_t)++e : 0u);
        }
        const size_t nr = r_start.size();
        if (nr) {
            std::vector<uint32_t> crcs(nr);
            { const int rc = kv_crc32_ranges(d_text, r_start.data(), r_len.data(), nr, crcs.data(), g->a->crc); if (rc != KV_OK) return rc; }
            uint32_t run = g->crc_run;
            for (size_t r = 0; r < nr; ++r) {
                run = kv_crc32_join(run, crcs[r], r_len[r]);
                if (r_closes[r]) {
                    if (run != g->pending_ends[r_closes[r] - 1].second) {
                        kv_set_error("gzip stream: the CRC-32 of a member does not match its text (damaged file)");
                        return KV_ERR_TYPE;
                    }
                    run = 0;
                }
            }
            g->crc_run = run;
        }
    }
    g->seen_comp += (g->pending_pos - g->pos_bit) / 8;
    g->seen_text += g->pending_text;
    g->isize_total += g->pending_isize;
    if (g->seen_comp > 0) g->ratio = std::max(1.0, (double)g->seen_text / (double)g->seen_comp);
    g->pos_bit = g->pending_pos;
    g->done = g->pending_done;
    g->pending = false;
    return KV_OK;
}

// Whole-buffer form for tests and tools: a gzip image in host memory -> its text in host memory, `segment_text` bytes of
// text (roughly) per segment.  stats: segments, stretches decoded, dropped, decoded again.
extern "C" int kv_gunzip_host(const void *file, uint64_t size, void *out, uint64_t out_cap, uint64_t segment_text, uint64_t *text_bytes,
                              uint64_t *stats, double *device_ms)
{
    KV_REQUIRE(file && out && text_bytes, KV_ERR_ARG, "kv_gunzip_host: null argument");
    KvGunzip *g = kv_gunzip_open((const uint8_t *)file, size, nullptr);
    KV_REQUIRE(g, KV_ERR_TYPE, "not a gzip file image");
    hipStream_t st = kv_stream();
    KvArena text;
    uint64_t total = 0;
    int rc = KV_OK;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms_sum = 0;
    while (rc == KV_OK && !kv_gunzip_done(g)) {
        uint64_t n = 0;
        (void)hipEventRecord(a, st);
        bool last = false;
        rc = kv_gunzip_decode(g, segment_text ? segment_text : (1ull << 30), &n, &last);
        if (rc != KV_OK) break;
        if (n == 0) continue;
        if (text.need(n + 256) != hipSuccess) { kv_set_error("kv_gunzip_host: out of device memory"); rc = KV_ERR_HIP; break; }
        rc = kv_gunzip_emit(g, (uint8_t *)text.p);
        if (rc != KV_OK) break;
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        ms_sum += ms;
        if (total + n > out_cap) { kv_set_error("kv_gunzip_host: the text needs more than %llu bytes", (unsigned long long)out_cap); rc = KV_ERR_CAPACITY; break; }
        if (n && hipMemcpy((uint8_t *)out + total, text.p, n, hipMemcpyDeviceToHost) != hipSuccess) { kv_set_error("kv_gunzip_host: copy failed"); rc = KV_ERR_HIP; break; }
        total += n;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (stats) kv_gunzip_stats(g, stats);
    if (device_ms) *device_ms = ms_sum;
    if (text.p) (void)hipFree(text.p);
    kv_gunzip_close(g);
    *text_bytes = total;
    return rc;
}
