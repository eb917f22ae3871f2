// kv_novel.hip -- K3, the fused novel scan: novel() + kmer_is_interesting()
// (kevlar/novel.py:21-53,123-169) as three launches over the packed case reads:
//
//   k_novel_mark  one workgroup per tile: hash every k-mer, band filter, probe the control and
//                 case sketches, and set bit (read * stride + offset) for each interesting k-mer;
//                 per-tile hit counts on the side.  This is the whole cost of the scan.
//   k_tile_scan   exclusive prefix sum of the tile hit counts (one workgroup).
//   k_novel_emit_bits  a tile's hits are the set bits of its range of the mask: ranked with a prefix sum,
//                 one hit per lane, k-mer rebuilt from the packed words, S abundances read.
//                 (k_novel_emit, which stages the tile again, serves 2-bit-hash kinds and k > 64.)
//
// Hits therefore reach the host already sorted, in exactly the order the reference annotates
// them, and the bit mask doubles as the per-band mask that the multi-GPU merge all-reduces.
#include <algorithm>
#include <functional>
#include <map>

#include "kv_binned.h"          // kv_device_cus
#include "kv_novel_device.h"
#include "kv_kmer2bit_device.h"

namespace {

// Set index of the verdict cache for k_novel_mark (16 <= k <= 32): a hash of the k-mer's minimizer -- the
// smallest canonical (k-2)-mer among its three.  It is a pure function of the k-mer and strand-symmetric like
// the k-mer hash itself, and neighbouring k-mers of a read share it every other time, so the 64 lanes of a
// wave, which hold 64 consecutive k-mers, ask for ~32 distinct 64-byte sets instead of 64 distinct sectors,
// and an 8-way set loses far fewer entries to conflicts than a direct-mapped slot.  Which set a hash is
// stored in only affects the hit rate: an entry anywhere in the cache is a hash proven rejected, so a
// match is always right.
#define VC_WINDOW 3   // measured at config 2: 3 -> 15.1 ms, 4 -> 15.2, 5 -> 16.1, 7 -> 16.9, 2 -> 15.6, 1 -> 18.1 (direct-mapped: 21.1)
#define VC_WINDOW_MAX 9
__device__ __forceinline__ uint32_t kmer_minimizer_key(const uint32_t *__restrict__ words, uint32_t pos, int k, int window)
{
    // the k-mer's 2 bits/base, base j at bits 2j (codes A0 C1 G2 T3 as packed by k_pack_reads)
    const uint32_t w0 = pos >> 4, sh = 2u * (pos & 15u);
    const uint64_t lo = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
    uint64_t code = lo >> sh;
    if (sh) code |= (uint64_t)words[w0 + 2] << (64u - sh);
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    code &= kmask;
    // reverse complement in the same layout: reverse the 2-bit groups, complement (3 - code)
    uint64_t rev = __brevll(code);
    rev = ((rev >> 1) & 0x5555555555555555ull) | ((rev & 0x5555555555555555ull) << 1);
    rev = (~rev >> (64 - 2 * k)) & kmask;
    const int m = k - (window - 1);
    const uint64_t mmask = (1ull << (2 * m)) - 1ull;
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < VC_WINDOW_MAX; ++j) {
        if (j >= window) break;
        const uint64_t f = (code >> (2 * j)) & mmask;
        const uint64_t r = (rev >> (2 * (window - 1 - j))) & mmask;         // reverse complement of the same m-mer
        const uint64_t c = f < r ? f : r;
        uint32_t v = (uint32_t)c ^ (uint32_t)(c >> 27);
        v *= 0x9E3779B1u;
        v ^= v >> 15;
        best = v < best ? v : best;
    }
    return best;
}

// With --abund-screen the reference order is kept (cases in order with full minima, novel.py:36-44)
// because `discard` depends on which case fails first.
__device__ __forceinline__ bool novel_test_screen(const NovelParams &p, uint64_t h, bool &discard)
{
    bool interesting = true;
    discard = false;
    for (int c = 0; c < p.ncase && interesting; ++c) {
        const int a = (int)sketch_get(p.sk[c], h);
        if (a < p.case_min) { interesting = false; discard = a < p.screen; }
    }
    for (int c = 0; c < p.nctrl && interesting; ++c)
        if ((int)sketch_get(p.sk[p.ncase + c], h) > p.ctrl_max) interesting = false;
    return interesting;
}

// k_novel_mark for batches of equal-length reads, hashed from the 2-bit form (kv_kmer2bit_device.h): a thread takes NM2_CH consecutive
// k-mers of one read; the k-mers of the band are collected per wave and evaluated 64 at a time (kmer_is_interesting(), cheapest evidence
// first), the interesting ones set their bit of the hit mask; k_tile_hits counts the bits per tile afterwards.  No abundance screen,
// no verdict cache (a batch that repeats its k-mers takes the super-k-mer scan instead).
#define NM2_CH 10
#define NM2_THREADS 512
template <int KW, int FK = 0>
__global__ __launch_bounds__(NM2_THREADS, 6) void k_novel_mark_2bit(ReadsDev rd, NovelParams p)
{
    __shared__ __attribute__((aligned(8))) uint32_t lut[FK ? 1024 : 256];          // FK: the product tables of skm_key_hash_pl
    __shared__ NovelShared ns;
    __shared__ unsigned long long queue[2 * (NM2_THREADS / 64) * 128];
    if (threadIdx.x < 256) {
        if (FK) { ((uint64_t *)lut)[threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C1); ((uint64_t *)lut)[256 + threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C2); }
        else lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    }
    load_descs(ns, p);
    __syncthreads();
    const int k = FK ? FK : p.hp.k;
    const uint32_t L = rd.uni_len, wpr = (L + 15u) >> 4, nk = L - (uint32_t)k + 1u, cpr = (nk + NM2_CH - 1u) / NM2_CH;
    WaveQueue2 wq;
    wq.q = queue + (threadIdx.x >> 6) * 256u;
    wq.q2 = wq.q + 128u;
    wq.n = 0;
    // (the verdict cache, p.vcache_2bit: the kernel is bound by its random 64-byte requests -- table 0 of the case, then the four tables of
    // the control that rejects an inherited k-mer -- not by the hashing; a set of the cache is one such request)
    const bool cached_verdicts = p.vcache != nullptr && p.vcache_2bit != 0;
    auto judge = [&](bool have, uint64_t h, uint64_t bit) {
        unsigned long long *slot = nullptr;
        unsigned long long cached = 0;
        if (cached_verdicts && have) {
            // (the set from the LOW bits of the hash: a band is a range of hashes, its k-mers share the top ones)
            const unsigned long long *set = p.vcache + ((h & ((1ull << (61 - p.vcache_shift)) - 1ull)) << 3);
            const ulonglong2 e0 = ((const ulonglong2 *)set)[0], e1 = ((const ulonglong2 *)set)[1];
            const ulonglong2 e2 = ((const ulonglong2 *)set)[2], e3 = ((const ulonglong2 *)set)[3];
            const unsigned long long e[8] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, e3.x, e3.y};
            bool hit = false;
            uint32_t way = (uint32_t)(h >> 40) & 7u, empty = 8;
#pragma unroll
            for (int w = 7; w >= 0; --w) {
                hit |= e[w] == h;
                if (e[w] == 0) empty = (uint32_t)w;
            }
            if (e[way] != 0 && empty < 8) way = empty;
            slot = const_cast<unsigned long long *>(set) + way;
            cached = hit ? h : ~h;
        }
        if (have && novel_test_fast(ns, p, h, slot, cached)) atomicOr(&p.mask[bit >> 5], 1u << (bit & 31));
    };
    const uint64_t n_items = rd.n_reads * cpr;
    for (uint64_t i0 = (uint64_t)blockIdx.x * NM2_THREADS + (threadIdx.x & ~63u); i0 < n_items; i0 += (uint64_t)gridDim.x * NM2_THREADS) {
        const uint64_t i = i0 + (threadIdx.x & 63u);
        bool mine = i < n_items;
        const uint64_t r = mine ? i / cpr : 0ull;
        const uint32_t j0 = mine ? (uint32_t)(i - r * cpr) * NM2_CH : 0u;
        if (mine && ((rd.flags[r] & 1) || r < p.first_read)) mine = false;        // the scan skips these reads (kevlar/novel.py:134-139)
        const uint32_t cnt = mine ? min((uint32_t)NM2_CH, nk - j0) : 0u;
        uint32_t u = 0;
        kmer2bit_walk<KW, NM2_CH, FK>(rd.words + r * wpr, j0, cnt, k, lut, p.hp, [&](bool live, uint64_t h) {
            wave_queue_push2(wq, live && band_pass(p, h), h, r * p.mask_stride + j0 + u, judge);
            u += 1;
        });
    }
    wave_queue_flush2(wq, judge);
}

__global__ __launch_bounds__(KV_TILE_THREADS) void k_novel_mark(ReadsDev rd, NovelParams p)
{
    __shared__ TileShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)tile_smem;
    __shared__ uint32_t tile_hits;
    __shared__ NovelShared ns;
    uint32_t read0;
    if (threadIdx.x == 0) tile_hits = 0;
    load_descs(ns, p);
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 1, p.first_read, read0);
    const uint32_t total = sh.kpre[nr];
    uint32_t mine = 0;
    // two-deep software pipeline: the next k-mer is hashed and its verdict-cache word requested
    // before the current k-mer is evaluated, so that round trip hides behind useful work
    struct Cand {
        uint32_t r, i;
        uint64_t h;
        unsigned long long *slot;
        unsigned long long cached;
        bool live;
    };
    auto fetch = [&](uint32_t q) {
        Cand c;
        c.live = false; c.slot = nullptr; c.cached = 0; c.h = 0; c.r = 0; c.i = 0;
        if (q >= total) return c;
        locate_kmer(sh, nr, q, c.r, c.i);
        const uint32_t fwd = sh.foff[c.r] + c.i;
        const uint32_t rc = sh.roff[c.r] + (sh.len[c.r] - (uint32_t)p.hp.k - c.i);
        c.h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
        c.live = band_pass(p, c.h);
        if (c.live && p.vcache && p.screen == 0) {
            if (p.vcache_sets) {
                const uint32_t gread = read0 + c.r;
                const uint32_t key = kmer_minimizer_key(rd.words + rd.woff[gread], sh.seg_start + c.i, p.hp.k, p.vcache_window);
                const unsigned long long *set = p.vcache + ((uint64_t)(key & p.vcache_set_mask) << 3);
                // plain loads: a stale line can only hide an entry (a miss), never invent one
                const ulonglong2 e0 = ((const ulonglong2 *)set)[0], e1 = ((const ulonglong2 *)set)[1];
                const ulonglong2 e2 = ((const ulonglong2 *)set)[2], e3 = ((const ulonglong2 *)set)[3];
                const unsigned long long e[8] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, e3.x, e3.y};
                bool hit = false;
                uint32_t way = (uint32_t)(c.h >> 7) & 7u, empty = 8;
#pragma unroll
                for (int w = 7; w >= 0; --w) {
                    hit |= e[w] == c.h;
                    if (e[w] == 0) empty = (uint32_t)w;
                }
                if (e[way] != 0 && empty < 8) way = empty;      // own way taken: first free one, else overwrite own way
                c.slot = const_cast<unsigned long long *>(set) + way;
                c.cached = hit ? c.h : ~c.h;
            } else {
                c.slot = p.vcache + (c.h >> p.vcache_shift);
                c.cached = __hip_atomic_load(c.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return c;
    };
    Cand cur = fetch(threadIdx.x);
    for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
        const Cand nxt = fetch(q + blockDim.x);
        if (cur.live) {
            bool discard = false;
            const bool interesting = p.screen > 0 ? novel_test_screen(p, cur.h, discard)
                                                  : novel_test_fast(ns, p, cur.h, cur.slot, cur.cached);
            const uint32_t gread = read0 + cur.r;
            if (discard) atomicMin(&p.disc_first[gread], sh.seg_start + cur.i);   // the reference stops reading the read there
            if (interesting) {
                const uint64_t bit = (uint64_t)gread * p.mask_stride + sh.seg_start + cur.i;
                atomicOr(&p.mask[bit >> 5], 1u << (bit & 31));
                mine += 1;
            }
        }
        cur = nxt;
    }
    mine = (uint32_t)wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&tile_hits, mine);
    __syncthreads();
    if (threadIdx.x == 0) p.tile_count[blockIdx.x] = tile_hits;
}

// exclusive scan of n 32-bit counts into 64-bit bases (single 1024-thread workgroup)
__global__ __launch_bounds__(1024) void k_tile_scan(const uint32_t *counts, uint32_t n, uint64_t *base)
{
    // One workgroup; a wave takes 64 * PER consecutive counts per round, 64 at a time: consecutive lanes on consecutive counts, so a
    // load is one 256-byte piece of memory and a store one of 512, and the wave's running total rides along from one 64 to the next.
    // (A thread on PER consecutive counts -- eight 4-byte loads 32 bytes apart from its neighbour's, eight 8-byte stores 64 bytes apart --
    // made every round ~7 us of single-CU memory pipe: 0.108 ms for the 117 k tiles of config 2.)  The next round's counts are
    // requested before this round's barriers.
    constexpr uint32_t PER = 8;
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t nxt[PER];
    auto request = [&](uint32_t start) {
        const uint32_t i0 = start + wave * 64u * PER + lane;
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) nxt[u] = i0 + u * 64u < n ? counts[i0 + u * 64u] : 0u;
    };
    request(0);
    for (uint32_t start = 0; start < n; start += 1024 * PER) {
        const uint32_t i0 = start + wave * 64u * PER + lane;
        uint32_t c[PER];
        uint64_t excl[PER];                             // exclusive prefix inside the wave's stretch
        uint64_t run = 0;                               // total of the pieces in front (wave-uniform)
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u) {
            c[u] = nxt[u];
            uint64_t incl = c[u];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t up = __shfl_up(incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            excl[u] = run + incl - c[u];
            run += __shfl(incl, 63);
        }
        if (start + 1024 * PER < n) request(start + 1024 * PER);
        if (lane == 0) wsum[wave] = run;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
        for (uint32_t u = 0; u < PER; ++u)
            if (i0 + u * 64u < n) base[i0 + u * 64u] = before + excl[u];
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + run;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[n] = carry;
}

__global__ __launch_bounds__(KV_TILE_THREADS) void k_novel_emit(ReadsDev rd, NovelParams p)
{
    __shared__ TileShared sh;
    __shared__ uint32_t wcount[KV_TILE_THREADS / 64];
    if (p.tile_count[blockIdx.x] == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)tile_smem;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 1, p.first_read, read0);
    const uint32_t total = sh.kpre[nr];
    const int S = p.ncase + p.nctrl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t out = p.tile_base[blockIdx.x];
    for (uint32_t q0 = 0; q0 < total; q0 += blockDim.x) {
        const uint32_t q = q0 + threadIdx.x;
        uint32_t r = 0, i = 0;
        bool hit = false;
        if (q < total) {
            locate_kmer(sh, nr, q, r, i);
            const uint64_t bit = (uint64_t)(read0 + r) * p.mask_stride + sh.seg_start + i;
            hit = (p.mask[bit >> 5] >> (bit & 31)) & 1u;
        }
        const unsigned long long ballot = __ballot(hit);
        if (lane == 0) wcount[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w = 0; w < KV_TILE_THREADS / 64; ++w) {
            if (w < wave) before += wcount[w];
            all += wcount[w];
        }
        if (hit) {
            const uint64_t slot = out + before + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
            const uint32_t fwd = sh.foff[r] + i;
            const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.hp.k - i);
            const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
            p.hit_read[slot] = read0 + r;
            p.hit_off[slot] = sh.seg_start + i;
            if (p.set_keys) {
                const uint64_t at = set_find(p, h);
                for (int c = 0; c < S; ++c) p.hit_abund[slot * (uint64_t)S + c] = at != KV_SET_NONE ? p.set_abund[at * (uint64_t)S + c] : 0;
            } else {
                for (int c = 0; c < S; ++c) p.hit_abund[slot * (uint64_t)S + c] = (uint8_t)sketch_get(p.sk[c], h);
            }
        }
        out += all;
        __syncthreads();
    }
}

// k_novel_emit without re-staging the tile: the hits of a tile are the set bits of its (contiguous) range
// of the bit mask, so the workgroup scans those mask words (about 140 for 64 reads of 100 bp), ranks the
// set bits with a prefix sum and hands one hit to each lane; the lane rebuilds the k-mer and its reverse
// complement in registers from the packed words and reads the S abundances.  Murmur kinds with k <= 64 (the
// register windows of KmerRoll); other cases use k_novel_emit.
template <int NW, bool ABUND>
__global__ __launch_bounds__(256) void k_novel_emit_bits(ReadsDev rd, NovelParams p)
{
    __shared__ uint32_t wcnt[256];      // set bits per word of the current chunk, then their exclusive prefix
    __shared__ uint32_t wbits[256];
    __shared__ uint32_t wave_tot[4];
    __shared__ NovelShared ns;
    const uint32_t nhit_tile = p.tile_count[blockIdx.x];
    if (nhit_tile == 0) return;
    load_descs(ns, p);
    const TileDesc td = rd.tile[blockIdx.x];
    uint64_t b0, b1;
    if (td.seg) {
        b0 = (uint64_t)td.first * p.mask_stride + td.seg_start;
        b1 = min(b0 + (uint64_t)KV_SEG_BASES, ((uint64_t)td.first + 1) * p.mask_stride);
    } else {
        b0 = (uint64_t)td.first * p.mask_stride;
        b1 = ((uint64_t)td.first + td.count) * p.mask_stride;
    }
    const int S = p.ncase + p.nctrl;
    const int k = p.hp.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t out = p.tile_base[blockIdx.x];
    uint32_t emitted = 0;
    for (uint64_t w0 = b0 >> 5; w0 < ((b1 + 31) >> 5) && emitted < nhit_tile; w0 += 256) {
        const uint64_t w = w0 + threadIdx.x;
        uint32_t bits = 0;
        if (w < ((b1 + 31) >> 5)) {
            bits = p.mask[w];
            const uint64_t wlo = w << 5;
            if (wlo < b0) bits &= ~0u << (uint32_t)(b0 - wlo);                 // bits of the previous tile
            if (wlo + 32 > b1) bits &= b1 > wlo ? (~0u >> (uint32_t)(wlo + 32 - b1)) : 0u;   // ... of the next
        }
        const uint32_t c = (uint32_t)__popc(bits);
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t before = 0, chunk_hits = 0;
        for (int v = 0; v < 4; ++v) {
            if (v < wave) before += wave_tot[v];
            chunk_hits += wave_tot[v];
        }
        wcnt[threadIdx.x] = before + incl - c;
        wbits[threadIdx.x] = bits;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < chunk_hits; j += 256) {
            uint32_t lo = 0, hi = 256;                         // word holding the j-th set bit of the chunk
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (wcnt[mid] <= j) lo = mid; else hi = mid;
            }
            uint32_t word = wbits[lo];
            for (uint32_t skip = j - wcnt[lo]; skip > 0; --skip) word &= word - 1;   // drop the lower set bits
            const uint64_t bit = ((w0 + lo) << 5) + (uint32_t)(__ffs((int)word) - 1);
            const uint64_t read = bit / p.mask_stride;
            const uint32_t off = (uint32_t)(bit - read * p.mask_stride);
            if (!ABUND) {                               // positions only: k_hit_abund fills in the abundances, a thread per hit
                p.hit_read[out + j] = (uint32_t)read;
                p.hit_off[out + j] = off;
                continue;
            }
            // the k-mer and its reverse complement as ASCII register windows (byte 0 = first base)
            uint32_t wf[NW], wr[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) { wf[q] = 0; wr[q] = 0; }
            const uint32_t *words = rd.words + rd.woff[read];
            for (int i = 0; i < k; ++i) {
                const uint32_t pos = off + (uint32_t)i;
                const uint32_t code = (words[pos >> 4] >> (2 * (pos & 15))) & 3u;
                const uint32_t fwd = (0x54474341u >> (8 * code)) & 0xffu;      // "ACGT"
                const uint32_t rev = (0x41434754u >> (8 * code)) & 0xffu;      // "TGCA"
                const int ri = k - 1 - i;
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    if (q == (i >> 2)) wf[q] |= fwd << (8 * (i & 3));
                    if (q == (ri >> 2)) wr[q] |= rev << (8 * (ri & 3));
                }
            }
            const uint64_t h = murmur_regs<NW>(wf, p.hp) ^ murmur_regs<NW>(wr, p.hp);
            const uint64_t slot = out + j;
            p.hit_read[slot] = (uint32_t)read;
            p.hit_off[slot] = off;
            hit_abundances(ns, p, h, p.hit_abund + slot * (uint64_t)S);
        }
        out += chunk_hits;
        emitted += chunk_hits;
        __syncthreads();
    }
}

// The S abundances of every hit, a thread per hit.  (k_novel_emit_bits<.., true> does this inside the tile that found the
// hit: ~20 of a workgroup's 256 threads then walk a chain of dependent loads -- word offset, words, twelve probes -- while
// the others wait; with the hits listed first the same work runs at full occupancy.)
template <int KW>
__global__ __launch_bounds__(256) void k_hit_abund(ReadsDev rd, NovelParams p, uint64_t nhits)
{
    __shared__ uint32_t lut[256];
    __shared__ NovelShared ns;
    lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    load_descs(ns, p);
    __syncthreads();
    const int S = p.ncase + p.nctrl, k = p.hp.k;
    const uint32_t uni_wpr = (rd.uni_len + 15u) >> 4;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nhits; i += (uint64_t)gridDim.x * 256ull) {
        const uint32_t read = p.hit_read[i], off = p.hit_off[i];
        const uint32_t *words = rd.words + (rd.uni_len ? (uint64_t)read * uni_wpr : rd.woff[read]);
        uint64_t bw[2] = {skm_bases32(words, off), KW == 2 ? skm_bases32(words, off + 32u) : 0ull};
        const SkmKey<KW> f = skm_first_kmer<KW>(bw, k);
        const uint64_t h = skm_key_hash<KW>(f, lut, p.hp);        // murmur(k-mer) ^ murmur(reverse complement): either strand
        // (the scan that found the k-mer interesting has left its abundances: one lookup in a table that sits in L2 for twelve probes)
        if (p.ab_keys && ab_lookup(p, h, p.hit_abund + i * (uint64_t)S, S)) continue;
        hit_abundances(ns, p, h, p.hit_abund + i * (uint64_t)S);
    }
}

// verdict cache kept across the batches of one scan: valid as long as the same sketches (unmodified),
// thresholds and band settings are used; otherwise it is cleared
struct VerdictCache {
    unsigned long long *p = nullptr;
    int bits = 0;
    uint64_t signature = 0;
};
std::map<hipStream_t, VerdictCache> g_vcache;
std::mutex g_vcache_mu;

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return kv_hip_malloc(&p, n ? n : 4); }
    template <typename T> T *as() { return (T *)p; }
};

// grow-only scratch of the scan, one pair of arenas per stream: hipMalloc / hipFree of the 66 MB bit mask and
// the hit arrays on every call cost more than the emit kernel
struct Arena {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t need(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = kv_hip_malloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
};
struct ScanArenas {
    Arena work, hits, set, abund;
    std::mutex mu;                                  // one scan at a time per stream
    // kv_hits_lazy: the hit arrays leave `hits` on a copy stream of this stream's own, behind the scan's last kernel (kdone); the
    // next scan that writes `hits` queues behind the copy (copied)
    hipStream_t copy_stream = nullptr;
    hipEvent_t kdone = nullptr, copied = nullptr;
    bool copy_pending = false;
};
thread_local bool g_hits_lazy = false;
std::map<hipStream_t, ScanArenas> g_scan_arenas;
std::mutex g_scan_arenas_mu;
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

inline int vc_sets_env()
{
    const char *e = kv_knob("KV_NOVEL_VCSETS");   // 0: direct-mapped cache in k_novel_mark as well
    return e ? atoi(e) : -1;
}

// point p.vcache at this stream's verdict cache, (re)allocating or clearing it as the signature requires
int attach_vcache(NovelParams &p, kv_sketch *const *ctrls, int ncase, int nctrl, int ctrl_max, uint64_t n_kmers, hipStream_t st)
{
    const char *vc_env = kv_knob("KV_NOVEL_VCACHE");
    if (nctrl > 0 && p.screen == 0 && !(vc_env && atoi(vc_env) == 0)) {
        // signature of everything the cached verdicts depend on
        uint64_t sig = 0x9e3779b97f4a7c15ull ^ (uint64_t)(uint32_t)ctrl_max;
        for (int c = ncase; c < ncase + nctrl; ++c) {
            const kv_sketch *sk = ctrls[c - ncase];
            sig = (sig ^ sk->uid) * 0xff51afd7ed558ccdull;
            sig = (sig ^ sk->version) * 0xc4ceb9fe1a85ec53ull;
        }
        VerdictCache *vc;
        {
            std::lock_guard<std::mutex> lk(g_vcache_mu);
            vc = &g_vcache[kv_stream_key(st)];
        }
        int want = 20;
        // ~2 slots per distinct inherited k-mer at 30x; a batch of many (vcache_2bit) meets the sample's, not its own: room for 2^29
        while (want < (p.vcache_2bit ? 29 : 28) && (1ull << want) < n_kmers / 2) ++want;
        if (vc->p == nullptr || vc->bits < want) {
            if (vc->p) (void)hipFree(vc->p);
            vc->p = nullptr;
            if (kv_hip_malloc((void **)&vc->p, (size_t)8 << want) == hipSuccess) { vc->bits = want; vc->signature = 0; }
            else { (void)hipGetLastError(); vc->bits = 0; }
        }
        if (vc->p) {
            if (vc->signature != sig) {
                KV_HIP(hipMemsetAsync(vc->p, 0, (size_t)8 << vc->bits, st));
                vc->signature = sig;
            }
            p.vcache = vc->p;
            p.vcache_shift = 64 - vc->bits;
        }
    }

    return KV_OK;
}

int scan_reads(NovelParams &p, const kv_reads *reads, int fam, uint64_t n_kmers, bool use_skm, const std::function<int()> &prepare_tile_scan,
               uint32_t *d_mask, uint64_t mask_stride, kv_hits **out, bool *skm_overflowed = nullptr);

}  // namespace

extern "C" int kv_novel_scan(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                             const kv_reads *reads, uint64_t first_read, int case_min, int ctrl_max,
                             int screen_thresh, int band_mode, int nbands, int band, uint32_t *d_mask,
                             uint64_t mask_stride, kv_hits **out)
{
    KV_REQUIRE(cases && reads && out && ncase >= 1 && nctrl >= 0 && (ctrls || nctrl == 0), KV_ERR_ARG,
               "kv_novel_scan: bad argument");
    KV_REQUIRE(ncase + nctrl <= KV_MAX_SAMPLES, KV_ERR_ARG, "at most %d samples per scan", KV_MAX_SAMPLES);
    KV_REQUIRE(band_mode == KV_BAND_NONE || (nbands > 0 && band >= 0 && band < nbands), KV_ERR_ARG,
               "band %d out of range for %d bands", band, nbands);
    NovelParams p;
    memset(&p, 0, sizeof(p));
    const int k = cases[0]->h.ksize, fam = cases[0]->h.hashfam;
    for (int c = 0; c < ncase + nctrl; ++c) {
        const kv_sketch *s = c < ncase ? cases[c] : ctrls[c - ncase];
        KV_REQUIRE(s, KV_ERR_ARG, "kv_novel_scan: null sketch");
        { const int rc = kv_sketch_ready(s); if (rc != KV_OK) return rc; }
        KV_REQUIRE(s->h.ksize == k && s->h.hashfam == fam, KV_ERR_ARG,
                   "all sketches of one scan must share k and hash function");
        p.sk[c] = s->d_desc;
    }
    const uint64_t min_stride = reads->max_len >= (uint32_t)k ? reads->max_len - (uint32_t)k + 1 : 1;
    if (d_mask) KV_REQUIRE(mask_stride >= min_stride, KV_ERR_ARG, "mask_stride %llu is smaller than the longest read's %llu k-mers",
                           (unsigned long long)mask_stride, (unsigned long long)min_stride);
    p.hp = make_hash_params(k, fam);
    p.ncase = ncase; p.nctrl = nctrl;
    p.case_min = case_min; p.ctrl_max = ctrl_max; p.screen = screen_thresh > 0 ? screen_thresh : 0;
    p.band_mode = band_mode; p.nbands = nbands; p.band = band;
    if (band_mode == KV_BAND_RANGE) kv_band_bounds(nbands, band, &p.band_lo, &p.band_hi);
    p.first_read = first_read;
    p.host_ctrls = (const void *)ctrls; p.host_nctrl = nctrl;
    p.host_case0 = (const void *)cases[0];
    hipStream_t st = kv_stream();
    uint64_t n_kmers = 0;
    kv_reads_num_kmers(reads, k, &n_kmers);
    // large batches: evaluate every DISTINCT k-mer once over the batch's super-k-mer buckets (kv_skm.hip); otherwise
    // (and as the fallback) every k-mer of every read, with the verdict cache absorbing the repeats
    bool use_skm = p.screen == 0 && kv_skm_eligible(cases[0], reads, n_kmers, true);
    const bool skm_by_name = kv_knob("KV_NOVEL_PATH") != nullptr;          // asked for by name: no second-guessing
    if (use_skm && !skm_by_name) {
        // Is there anything to deduplicate?  (i) the scan remembers: the last batch it cut for this case sample overflowed the
        // tables and was scanned again tile by tile (config 4's batches of 0.6x coverage: 48 per sample, every one of them scanned
        // twice before this flag existed).  (ii) before the first cut: the case sketch knows how many distinct k-mers it holds
        // (n_unique; of its band only, if banded); a batch of a sample at sequencing coverage brings ~5 k-mers per distinct k-mer
        // of the sample, a batch with less than one for every two distinct k-mers is below ~3x coverage, where more than half of its
        // k-mers are distinct and the bucket tables do not hold them.
        // (n_unique is kept by the counts; a sketch that was loaded from a file or filled through weighted pairs has none: there the
        // occupancy of table 0 -- counted now if it is stale -- gives the same figure by linear counting)
        double distinct_held = (double)cases[0]->n_unique;
        if (distinct_held == 0.0) {
            kv_sketch *c0 = cases[0];
            std::lock_guard<std::mutex> lk(c0->mu);
            if (!c0->occ_dirty || kv_sketch_refresh_occupancy(c0) == KV_OK) distinct_held = kv_estimate_distinct(c0->n_occupied, c0->h.size[0]);
            else (void)hipGetLastError();
        }
        const double held = distinct_held * (band_mode == KV_BAND_RANGE ? (double)nbands : 1.0);
        const bool sparse = held > 0.0 && (double)n_kmers < 0.5 * held;
        p.vcache_2bit = sparse ? 1 : 0;         // one batch of many of its sample: what the controls reject in this batch they rejected in the ones before
        if (cases[0]->skm_scan_off || sparse) {
            use_skm = false;
            if (kv_knob("KV_SKM_VERBOSE"))
                fprintf(stderr, "[kv_novel] tile scan: %s\n", cases[0]->skm_scan_off ? "the previous batch of this case sample did not fit the super-k-mer tables"
                                                                                      : "the batch is a small share of the distinct k-mers the case sketch holds");
        }
    }
    auto prepare_tile_scan = [&]() -> int {
        const int rc = attach_vcache(p, ctrls, ncase, nctrl, ctrl_max, n_kmers, st);
        if (rc != KV_OK) return rc;
        if (p.vcache && k >= 16 && k <= 32 && !(vc_sets_env() == 0)) {
            p.vcache_sets = 1;
            p.vcache_window = VC_WINDOW;
            p.vcache_set_mask = (uint32_t)((1ull << (64 - p.vcache_shift - 3)) - 1ull);   // entries / 8 sets
        }
        return KV_OK;
    };
    bool skm_overflowed = false;
    const int rc = scan_reads(p, reads, fam, n_kmers, use_skm, prepare_tile_scan, d_mask, mask_stride, out, &skm_overflowed);
    if (skm_overflowed) cases[0]->skm_scan_off = true;
    return rc;
}

namespace {

// mark -> count per tile -> emit in (read, offset) order: shared by kv_novel_scan and kv_novel_scan_set
int scan_reads(NovelParams &p, const kv_reads *reads, int fam, uint64_t n_kmers, bool use_skm, const std::function<int()> &prepare_tile_scan,
               uint32_t *d_mask, uint64_t mask_stride, kv_hits **out, bool *skm_overflowed)
{
    const int k = p.hp.k, S = p.ncase + p.nctrl;
    const uint64_t min_stride = reads->max_len >= (uint32_t)k ? reads->max_len - (uint32_t)k + 1 : 1;
    hipStream_t st = kv_stream();
    if (!use_skm) { const int rc = prepare_tile_scan(); if (rc != KV_OK) return rc; }

    kv_hits *hits = new kv_hits();
    hits->nsamples = S;
    *out = hits;
    if (reads->n_tiles == 0) return KV_OK;

    ScanArenas *arenas;
    {
        std::lock_guard<std::mutex> lk(g_scan_arenas_mu);
        arenas = &g_scan_arenas[kv_stream_key(st)];
    }
    std::lock_guard<std::mutex> arena_lock(arenas->mu);
    hipError_t e = hipSuccess;
    if (arenas->copy_pending) {                     // the previous scan's hits may still be leaving the buffers this one writes
        e = hipStreamWaitEvent(st, arenas->copied, 0);
        arenas->copy_pending = false;
    }
    const uint64_t mask_words = d_mask ? 0 : (reads->n_reads * min_stride + 31) / 32;
    if (mask_words * 4 > ((uint64_t)64 << 30)) {
        // one bit per (read, offset) with the stride of the longest read: a chromosome among a million reads
        kv_set_error("kv_novel_scan: %llu reads with a longest read of %u bases need a %llu GB hit mask; "
                     "scan long sequences in a batch of their own", (unsigned long long)reads->n_reads, reads->max_len,
                     (unsigned long long)(mask_words * 4 >> 30));
        delete hits;
        *out = nullptr;
        return KV_ERR_CAPACITY;
    }
    const size_t b_mask = up256(mask_words * 4), b_flags = p.screen > 0 ? up256(reads->n_reads * 4) : 0;
    const size_t b_tcount = up256((uint64_t)reads->n_tiles * 4), b_tbase = up256(((uint64_t)reads->n_tiles + 1) * 8);
    e = arenas->work.need(b_mask + b_flags + b_tcount + b_tbase + 256);
    unsigned char *wp = (unsigned char *)arenas->work.p;
    if (e == hipSuccess) {
        if (d_mask) {
            p.mask = d_mask; p.mask_stride = mask_stride;
        } else {
            p.mask_stride = min_stride;
            p.mask = (uint32_t *)wp;
            e = hipMemsetAsync(p.mask, 0, mask_words * 4, st);
        }
        wp += b_mask;
    }
    if (e == hipSuccess && p.screen > 0) {
        p.disc_first = (uint32_t *)wp;
        e = hipMemsetAsync(p.disc_first, 0xFF, reads->n_reads * 4, st);
    }
    wp += b_flags;
    p.tile_count = (uint32_t *)wp; wp += b_tcount;
    uint64_t *d_tbase_p = (uint64_t *)wp;
    p.tile_base = d_tbase_p;
    uint64_t nhits = 0;
    bool marked_by_skm = use_skm;
    p.ab_keys = nullptr; p.ab_vals = nullptr; p.ab_mask = 0; p.ab_list = nullptr; p.ab_count = nullptr; p.ab_list_cap = 0;
    if (e == hipSuccess && use_skm && !p.set_keys && kv_skm_list_ready(reads, k) && !(kv_knob("KV_NOVEL_ABCACHE") && atoi(kv_knob("KV_NOVEL_ABCACHE")) == 0)) {
        // room for the abundances of the interesting k-mers (a k-mer in a few thousand is one): 1 / 64 of the k-mers in slots
        uint64_t slots = 1u << 16;
        while (slots < n_kmers / 64 && slots < (1ull << 24)) slots <<= 1;
        const uint64_t list_cap = slots / 8;
        if (arenas->abund.need(up256(slots * 8) + up256(slots * (uint64_t)S) + up256(list_cap * 4) + 256) == hipSuccess) {
            p.ab_keys = (unsigned long long *)arenas->abund.p;
            p.ab_vals = (uint8_t *)arenas->abund.p + up256(slots * 8);
            p.ab_mask = slots - 1;
            p.ab_list = (uint32_t *)(p.ab_vals + up256(slots * (uint64_t)S));
            p.ab_count = (unsigned int *)((unsigned char *)p.ab_list + up256(list_cap * 4));
            p.ab_list_cap = (uint32_t)list_cap;
            e = hipMemsetAsync(p.ab_keys, 0, slots * 8, st);
            if (e == hipSuccess) e = hipMemsetAsync(p.ab_count, 0, 4, st);
        } else {
            (void)hipGetLastError();
        }
    }
    if (e == hipSuccess && use_skm) {
        const int rc = kv_skm_novel_mark(reads, p, n_kmers);
        if (rc == KV_ERR_CAPACITY) {
            // every bit set so far is a true hit, so the tile scan can simply run on top of the same mask
            marked_by_skm = false;
            if (skm_overflowed) *skm_overflowed = true;
            const int rc2 = prepare_tile_scan();
            if (rc2 != KV_OK) { delete hits; *out = nullptr; return rc2; }
        } else if (rc != KV_OK) {
            delete hits;
            *out = nullptr;
            return rc;
        }
    }
    if (e == hipSuccess) {
        // equal-length reads of a murmur kind, 16 <= k <= 64, no abundance screen: every k-mer hashed from its 2-bit form
        // (k_novel_mark_2bit; KV_NOVEL_PATH=tiles, or KV_NOVEL_2BIT=0, keeps the tile kernel)
        const char *forced = kv_knob("KV_NOVEL_PATH"), *nm2 = kv_knob("KV_NOVEL_2BIT");
        const bool two_bit = !marked_by_skm && p.screen == 0 && fam == HF_MURMUR && k >= SKM_MIN_K && k <= SKM_MAX_K && reads->uni_len >= (uint32_t)k &&
                             reads->uni_per_tile != 0 && !(forced && strcmp(forced, "tiles") == 0) && !(nm2 && atoi(nm2) == 0);
        if (two_bit) {
            {
                KvProfScope prof("k_novel_mark_2bit");
                const uint64_t n_items = reads->n_reads * (((uint64_t)reads->uni_len - (uint64_t)k + 1 + NM2_CH - 1) / NM2_CH);
                const unsigned grid = (unsigned)std::min<uint64_t>((n_items + NM2_THREADS - 1) / NM2_THREADS, 3u * (unsigned)kv_device_cus() * 4u);
                if (k == 31 && !kv_knob("KV_SKM_ANY_K")) hipLaunchKernelGGL((k_novel_mark_2bit<1, 31>), dim3(grid), dim3(NM2_THREADS), 0, st, reads_dev(reads), p);
                else if (k <= 32) hipLaunchKernelGGL(k_novel_mark_2bit<1>, dim3(grid), dim3(NM2_THREADS), 0, st, reads_dev(reads), p);
                else hipLaunchKernelGGL(k_novel_mark_2bit<2>, dim3(grid), dim3(NM2_THREADS), 0, st, reads_dev(reads), p);
            }
            kv_tile_hits_launch(reads, p, st);
        } else if (!marked_by_skm) {
            KvProfScope prof("k_novel_mark");
            kv_ensure_dynamic_lds((const void *)k_novel_mark, reads->tile_lds_bytes);
            hipLaunchKernelGGL(k_novel_mark, dim3(reads->n_tiles), dim3(KV_TILE_THREADS), reads->tile_lds_bytes, st, reads_dev(reads), p);
        }
        {
            KvProfScope prof("k_tile_scan");
            hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, p.tile_count, reads->n_tiles, d_tbase_p);
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&nhits, d_tbase_p + reads->n_tiles, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess && nhits) {
        e = arenas->hits.need(2 * up256(nhits * 4) + up256(nhits * (uint64_t)S));
        p.hit_read = (uint32_t *)arenas->hits.p;
        p.hit_off = (uint32_t *)((unsigned char *)arenas->hits.p + up256(nhits * 4));
        p.hit_abund = (uint8_t *)arenas->hits.p + 2 * up256(nhits * 4);
        if (e == hipSuccess) {
            KvProfScope prof("k_novel_emit");
            const bool from_bits = fam == HF_MURMUR && k <= 64 && !kv_knob("KV_NOVEL_EMIT_TILES");
            const bool dense = from_bits && k >= SKM_MIN_K && !kv_knob("KV_NOVEL_EMIT_FUSED");
            const unsigned grid_hits = (unsigned)std::min<uint64_t>((nhits + 255) / 256, 1u << 16);
            if (dense && k <= 32) {
                hipLaunchKernelGGL((k_novel_emit_bits<8, false>), dim3(reads->n_tiles), dim3(256), 0, st, reads_dev(reads), p);
                hipLaunchKernelGGL(k_hit_abund<1>, dim3(grid_hits), dim3(256), 0, st, reads_dev(reads), p, nhits);
            } else if (dense) {
                hipLaunchKernelGGL((k_novel_emit_bits<16, false>), dim3(reads->n_tiles), dim3(256), 0, st, reads_dev(reads), p);
                hipLaunchKernelGGL(k_hit_abund<2>, dim3(grid_hits), dim3(256), 0, st, reads_dev(reads), p, nhits);
            } else if (from_bits && k <= 32) {
                hipLaunchKernelGGL((k_novel_emit_bits<8, true>), dim3(reads->n_tiles), dim3(256), 0, st, reads_dev(reads), p);
            } else if (from_bits) {
                hipLaunchKernelGGL((k_novel_emit_bits<16, true>), dim3(reads->n_tiles), dim3(256), 0, st, reads_dev(reads), p);
            } else {
                kv_ensure_dynamic_lds((const void *)k_novel_emit, reads->tile_lds_bytes);
                hipLaunchKernelGGL(k_novel_emit, dim3(reads->n_tiles), dim3(KV_TILE_THREADS), reads->tile_lds_bytes, st, reads_dev(reads), p);
            }
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hits->read.resize(nhits);
        if (e == hipSuccess) e = hits->offset.resize(nhits);
        if (e == hipSuccess) e = hits->abund.resize(nhits * (uint64_t)S);
        // kv_hits_lazy (and no abundance screen, whose bookkeeping reads the hits on the host right here): the call returns when its
        // KERNELS are done -- the sketches may change from then on -- while the arrays travel on a copy stream; whoever reads the handle
        // waits for them (kv_hits::wait).  25 MB of hits per scan of config 2 are half a millisecond of PCIe that the next count hides.
        hipStream_t cs = st;
        const bool lazy = g_hits_lazy && !p.disc_first;
        if (lazy && e == hipSuccess) {
            if (!arenas->copy_stream) {
                // (lowest priority: the copies are blit kernels, and the count kernels they run beside should win the CUs)
                int least = 0, greatest = 0;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                e = hipStreamCreateWithPriority(&arenas->copy_stream, hipStreamNonBlocking, least);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&arenas->kdone, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&arenas->copied, hipEventDisableTiming);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&hits->ready, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(arenas->kdone, st);
            if (e == hipSuccess) e = hipStreamWaitEvent(arenas->copy_stream, arenas->kdone, 0);
            cs = arenas->copy_stream;
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hits->read.data(), p.hit_read, nhits * 4, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess) e = hipMemcpyAsync(hits->offset.data(), p.hit_off, nhits * 4, hipMemcpyDeviceToHost, cs);
        if (e == hipSuccess) e = hipMemcpyAsync(hits->abund.data(), p.hit_abund, nhits * (uint64_t)S, hipMemcpyDeviceToHost, cs);
        if (lazy) {
            if (e == hipSuccess) e = hipEventRecord(hits->ready, cs);
            if (e == hipSuccess) e = hipEventRecord(arenas->copied, cs);
            if (e == hipSuccess) arenas->copy_pending = true;
            if (e == hipSuccess) e = hipEventSynchronize(arenas->kdone);
            else (void)hipStreamSynchronize(cs);            // (nothing may be in flight into a handle that is about to be deleted)
        } else if (e == hipSuccess) {
            e = hipStreamSynchronize(st);
        }
    }
    std::vector<uint32_t> first_trip;
    if (e == hipSuccess && p.disc_first) {
        first_trip.resize(reads->n_reads);
        e = hipMemcpy(first_trip.data(), p.disc_first, reads->n_reads * 4, hipMemcpyDeviceToHost);
        for (uint64_t i = 0; i < reads->n_reads; ++i)
            if (first_trip[i] != 0xffffffffu) hits->discarded.push_back((uint32_t)i);
    }
    if (e != hipSuccess) {
        delete hits;
        *out = nullptr;
        kv_set_error("kv_novel_scan failed: %s", hipGetErrorString(e));
        return KV_ERR_HIP;
    }
    // a read dropped by the abundance screen loses all of its hits (novel.py:152-154,164); the interesting k-mers in
    // front of the k-mer that tripped the screen had already been tallied by the reference's loop (novel.py:160-162),
    // so they are kept aside for the "unique novel kmers" figure
    if (!hits->discarded.empty()) {
        uint64_t w = 0;
        for (uint64_t i = 0; i < hits->read.size(); ++i) {
            const uint32_t trip = first_trip[hits->read[i]];
            if (trip != 0xffffffffu) {
                if (hits->offset[i] < trip) { hits->shadow_read.push_back(hits->read[i]); hits->shadow_offset.push_back(hits->offset[i]); }
                continue;
            }
            hits->read[w] = hits->read[i];
            hits->offset[w] = hits->offset[i];
            if (w != i) memmove(&hits->abund[w * (uint64_t)S], &hits->abund[i * (uint64_t)S], (size_t)S);
            ++w;
        }
        hits->read.n = w; hits->offset.n = w; hits->abund.n = w * (uint64_t)S;   // shrink in place
    }
    return KV_OK;
}

}  // namespace

extern "C" int kv_hits_lazy(int on)
{
    g_hits_lazy = on != 0;
    return KV_OK;
}

extern "C" int kv_hits_shadow(const kv_hits *h, const uint32_t **read, const uint32_t **offset, uint64_t *n)
{
    KV_REQUIRE(h && n, KV_ERR_ARG, "kv_hits_shadow: null argument");
    h->wait();
    *n = h->shadow_read.size();
    if (read) *read = h->shadow_read.data();
    if (offset) *offset = h->shadow_offset.data();
    return KV_OK;
}

extern "C" int kv_hits_count(const kv_hits *h, uint64_t *n_hits, uint64_t *n_discarded_reads)
{
    KV_REQUIRE(h, KV_ERR_ARG, "kv_hits_count: null handle");
    if (n_hits) *n_hits = h->read.size();
    if (n_discarded_reads) *n_discarded_reads = h->discarded.size();
    return KV_OK;
}

extern "C" int kv_hits_fetch(const kv_hits *h, uint32_t *read, uint32_t *offset, uint8_t *abund, uint64_t cap_hits,
                             uint32_t *discarded_reads, uint64_t cap_discarded)
{
    KV_REQUIRE(h, KV_ERR_ARG, "kv_hits_fetch: null handle");
    KV_REQUIRE(cap_hits >= h->read.size(), KV_ERR_CAPACITY, "hit buffer too small");
    h->wait();
    if (!h->read.empty()) {
        KV_REQUIRE(read && offset && abund, KV_ERR_ARG, "kv_hits_fetch: null output");
        memcpy(read, h->read.data(), h->read.size() * 4);
        memcpy(offset, h->offset.data(), h->offset.size() * 4);
        memcpy(abund, h->abund.data(), h->abund.size());
    }
    if (discarded_reads) {
        KV_REQUIRE(cap_discarded >= h->discarded.size(), KV_ERR_CAPACITY, "discard buffer too small");
        if (!h->discarded.empty()) memcpy(discarded_reads, h->discarded.data(), h->discarded.size() * 4);
    }
    return KV_OK;
}

extern "C" int kv_hits_view(const kv_hits *h, const uint32_t **read, const uint32_t **offset, const uint8_t **abund,
                            const uint32_t **discarded_reads)
{
    KV_REQUIRE(h, KV_ERR_ARG, "kv_hits_view: null handle");
    h->wait();
    if (read) *read = h->read.data();
    if (offset) *offset = h->offset.data();
    if (abund) *abund = h->abund.data();
    if (discarded_reads) *discarded_reads = h->discarded.data();
    return KV_OK;
}

extern "C" int kv_hits_destroy(kv_hits *h)
{
    delete h;
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// scan of a routed hash list (read-sharded multi-GPU path, kv_shard.hip): the items are
// (hash, tag) pairs of the k-mers whose band this rank owns; tag = read << 16 | offset, bit 63
// set for k-mers of reads the scan must skip (non-ACGT).  Same predicate, same verdict cache.
// Hits leave as (tag, abundances) in arbitrary order; kv_hits_from_tagged sorts them.
// ---------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_novel_list(NovelParams p, const uint64_t *__restrict__ items, uint64_t n,
                                                    uint64_t *hit_tag, uint8_t *hit_abund,
                                                    unsigned long long *hit_count, uint64_t cap, bool want_hash)
{
    __shared__ NovelShared ns;
    load_descs(ns, p);
    __syncthreads();
    const int S = p.ncase + p.nctrl;
    struct Cand {
        uint64_t h, tag;
        unsigned long long *slot;
        unsigned long long cached;
        bool live;
    };
    auto fetch = [&](uint64_t i) {
        Cand c;
        c.live = false; c.slot = nullptr; c.cached = 0; c.h = 0; c.tag = 0;
        if (i >= n) return c;
        const ulonglong2 v = *(const ulonglong2 *)(items + 2 * i);
        c.h = v.x; c.tag = v.y;
        c.live = (c.tag >> 63) == 0;
        if (c.live && p.vcache) {
            // no sequence here, so no minimizer: the set comes from the hash itself -- from its LOW bits: the hashes a band owner
            // receives are a range, they share the top ones --; 8 ways still lose fewer entries to conflicts than a direct-mapped slot
            const unsigned long long *set = p.vcache + ((c.h & ((1ull << (61 - p.vcache_shift)) - 1ull)) << 3);
            const ulonglong2 e0 = ((const ulonglong2 *)set)[0], e1 = ((const ulonglong2 *)set)[1];
            const ulonglong2 e2 = ((const ulonglong2 *)set)[2], e3 = ((const ulonglong2 *)set)[3];
            const unsigned long long e[8] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y, e3.x, e3.y};
            bool hit = false;
            uint32_t way = (uint32_t)(c.h >> 7) & 7u, empty = 8;
#pragma unroll
            for (int w = 7; w >= 0; --w) {
                hit |= e[w] == c.h;
                if (e[w] == 0) empty = (uint32_t)w;
            }
            if (e[way] != 0 && empty < 8) way = empty;
            c.slot = const_cast<unsigned long long *>(set) + way;
            c.cached = hit ? c.h : ~c.h;
        }
        return c;
    };
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t i0 = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    Cand cur = fetch(i0);
    const int lane = threadIdx.x & 63;
    for (uint64_t base = blockIdx.x * (uint64_t)blockDim.x; base < n; base += stride) {   // wave-uniform trip count
        const Cand nxt = fetch(base + threadIdx.x + stride);
        const bool interesting = cur.live && novel_test_fast(ns, p, cur.h, cur.slot, cur.cached);
        const unsigned long long ballot = __ballot(interesting);
        if (ballot) {
            unsigned long long first = 0;
            if (lane == 0) first = atomicAdd(hit_count, (unsigned long long)__popcll(ballot));
            first = __shfl(first, 0);
            if (interesting) {
                const uint64_t pos = first + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
                if (pos < cap) {
                    hit_tag[pos] = want_hash ? cur.h : cur.tag;
                    for (int c = 0; c < S; ++c) hit_abund[pos * (uint64_t)S + c] = (uint8_t)sketch_get(p.sk[c], cur.h);
                }
            }
        }
        cur = nxt;
    }
}

// The same for (hash, occurrences) pairs -- every distinct k-mer of the case sample once (or once per shard): nothing repeats, so
// there is no verdict cache to consult and fill (a random 64-byte line per item, for nothing), and a thread keeps E first probes in
// flight -- table 0 of the first case sample, where a sequencing-error k-mer ends -- before the few that pass take the rest of
// kmer_is_interesting() one by one.  (k_novel_list over the pairs of config 2: 0.91 ms per rank at N = 8, 3.66 at N = 2.)
template <int E>
__global__ __launch_bounds__(256) void k_novel_pairs(NovelParams p, const uint64_t *__restrict__ items, uint64_t n, uint64_t *hit_hash, uint8_t *hit_abund,
                                                     unsigned long long *hit_count, uint64_t cap)
{
    __shared__ NovelShared ns;
    load_descs(ns, p);
    __syncthreads();
    const int S = p.ncase + p.nctrl;
    const int lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)blockIdx.x * (E * 256u); base < n; base += (uint64_t)gridDim.x * (E * 256u)) {      // workgroup-uniform trip count
        uint64_t h[E];
        bool live[E];
        uint32_t v[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const uint64_t i = base + (uint64_t)u * 256u + threadIdx.x;
            live[u] = i < n;
            const ulonglong2 it = live[u] ? *(const ulonglong2 *)(items + 2 * i) : make_ulonglong2(0ull, 0ull);
            h[u] = it.x;
            live[u] = live[u] && (it.y >> 63) == 0;
        }
        // (p.case0_bits: the bit map of "table 0 of the first case sample >= case_min" instead of the table -- an eighth of its bytes)
#pragma unroll
        for (int u = 0; u < E; ++u) {
            if (p.case0_bits) {
                const uint64_t bin = fastmod(h[u], ns.d[0].size, ns.d[0].magic);
                const __attribute__((address_space(1))) uint32_t *bits = (const __attribute__((address_space(1))) uint32_t *)p.case0_bits;
                v[u] = live[u] && ((bits[bin >> 5] >> (uint32_t)(bin & 31u)) & 1u) ? (uint32_t)p.case_min : 0u;
            } else {
                v[u] = live[u] ? probe(ns, 0, 0, h[u]) : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const bool interesting = live[u] && (int)v[u] >= p.case_min && novel_test_fast(ns, p, h[u], nullptr, 0ull);
            const unsigned long long ballot = __ballot(interesting);
            if (!ballot) continue;
            unsigned long long first = 0;
            if (lane == 0) first = atomicAdd(hit_count, (unsigned long long)__popcll(ballot));
            first = __shfl(first, 0);
            if (interesting) {
                const uint64_t pos = first + (uint64_t)__popcll(ballot & ((1ull << lane) - 1ull));
                if (pos < cap) {
                    hit_hash[pos] = h[u];
                    for (int c = 0; c < S; ++c) hit_abund[pos * (uint64_t)S + c] = (uint8_t)sketch_get(p.sk[c], h[u]);
                }
            }
        }
    }
}

}  // namespace

namespace {
int scan_items(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl, const void *d_items, uint64_t n_items,
               int case_min, int ctrl_max, void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits, bool want_hash);
std::map<hipStream_t, KvArena> g_pairs_bits;     // per stream: the bit map of the pairs scan's first probe (grow-only; kv_scratch_trim)
std::mutex g_pairs_bits_mu;
}
void kv_novel_scratch_release()
{
    std::lock_guard<std::mutex> lk(g_pairs_bits_mu);
    for (auto &kv : g_pairs_bits) kv.second.release();
}

extern "C" int kv_novel_scan_hashes(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                                    const void *d_items, uint64_t n_items, int case_min, int ctrl_max,
                                    void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits)
{
    return scan_items(cases, ncase, ctrls, nctrl, d_items, n_items, case_min, ctrl_max, d_hit_tags, d_hit_abund, hit_cap, n_hits, false);
}

// The same test over the (hash, occurrences) pairs a band owner received of the case sample (kv_route_distinct): every
// interesting pair leaves as its HASH and the S abundances.  The same hash may arrive from several shards and then
// leaves several times; kv_novel_scan_set does not mind.
extern "C" int kv_novel_scan_distinct(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                                      const void *d_items, uint64_t n_items, int case_min, int ctrl_max,
                                      void *d_hit_hashes, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits)
{
    return scan_items(cases, ncase, ctrls, nctrl, d_items, n_items, case_min, ctrl_max, d_hit_hashes, d_hit_abund, hit_cap, n_hits, true);
}

namespace {

int scan_items(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl, const void *d_items, uint64_t n_items,
               int case_min, int ctrl_max, void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits, bool want_hash)
{
    KV_REQUIRE(cases && n_hits && ncase >= 1 && nctrl >= 0 && (ctrls || nctrl == 0), KV_ERR_ARG,
               "kv_novel_scan_hashes: bad argument");
    KV_REQUIRE(ncase + nctrl <= KV_MAX_SAMPLES, KV_ERR_ARG, "at most %d samples per scan", KV_MAX_SAMPLES);
    KV_REQUIRE(n_items == 0 || (d_items && d_hit_tags && d_hit_abund), KV_ERR_ARG, "kv_novel_scan_hashes: null buffer");
    *n_hits = 0;
    if (n_items == 0) return KV_OK;
    NovelParams p;
    memset(&p, 0, sizeof(p));
    const int k = cases[0]->h.ksize, fam = cases[0]->h.hashfam;
    for (int c = 0; c < ncase + nctrl; ++c) {
        const kv_sketch *s = c < ncase ? cases[c] : ctrls[c - ncase];
        KV_REQUIRE(s, KV_ERR_ARG, "kv_novel_scan_hashes: null sketch");
        { const int rc = kv_sketch_ready(s); if (rc != KV_OK) return rc; }
        KV_REQUIRE(s->h.ksize == k && s->h.hashfam == fam, KV_ERR_ARG,
                   "all sketches of one scan must share k and hash function");
        p.sk[c] = s->d_desc;
    }
    p.hp = make_hash_params(k, fam);
    p.ncase = ncase; p.nctrl = nctrl;
    p.case_min = case_min; p.ctrl_max = ctrl_max;
    hipStream_t st = kv_stream();
    // pairs (want_hash): every k-mer once, no verdict cache, several first probes in flight (KV_NOVEL_PAIRS=0: the list kernel)
    const bool pairs = want_hash && !(kv_knob("KV_NOVEL_PAIRS") && atoi(kv_knob("KV_NOVEL_PAIRS")) == 0);
    if (!pairs) { const int rc = attach_vcache(p, ctrls, ncase, nctrl, ctrl_max, n_items, st); if (rc != KV_OK) return rc; }
    DevBuf d_count;
    KV_HIP(d_count.alloc(8));
    KV_HIP(hipMemsetAsync(d_count.p, 0, 8, st));
    const uint64_t bits_from = kv_knob("KV_NOVEL_BITS_MIN") ? strtoull(kv_knob("KV_NOVEL_BITS_MIN"), nullptr, 10) : (1ull << 20);      // (tests: 1)
    if (pairs && cases[0]->h.storage == ST_BYTE && n_items >= bits_from && !(kv_knob("KV_NOVEL_BITS") && atoi(kv_knob("KV_NOVEL_BITS")) == 0)) {
        // the first probe -- where a sequencing-error k-mer ends -- from a bit map of table 0 (a streaming pass over the table first: worth it
        // from a million pairs up)
        KvArena *bits;
        { std::lock_guard<std::mutex> lk(g_pairs_bits_mu); bits = &g_pairs_bits[kv_stream_key(st)]; }
        if (bits->need(kv_round_up(((cases[0]->h.size[0] + 31) >> 5) * 4, 256)) == hipSuccess) {
            kv_case_bits_launch((const uint8_t *)cases[0]->h.tab[0], (uint64_t)cases[0]->h.size[0], case_min, (uint32_t *)bits->p, st);
            p.case0_bits = (const uint32_t *)bits->p;
        } else {
            (void)hipGetLastError();
        }
    }
    if (pairs) {
        KvProfScope prof("k_novel_pairs");
        const unsigned grid = (unsigned)std::min<uint64_t>((n_items + 1023) / 1024, 256 * 16);
        hipLaunchKernelGGL(k_novel_pairs<4>, dim3(grid), dim3(256), 0, st, p, (const uint64_t *)d_items, n_items,
                           (uint64_t *)d_hit_tags, (uint8_t *)d_hit_abund, d_count.as<unsigned long long>(), hit_cap);
    } else {
        KvProfScope prof("k_novel_list");
        const unsigned grid = (unsigned)std::min<uint64_t>((n_items + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(k_novel_list, dim3(grid), dim3(256), 0, st, p, (const uint64_t *)d_items, n_items,
                           (uint64_t *)d_hit_tags, (uint8_t *)d_hit_abund, d_count.as<unsigned long long>(), hit_cap, want_hash);
    }
    KV_HIP(hipGetLastError());
    KvReadback rb;
    hipError_t rb_err = hipSuccess;
    const unsigned long long *cnt_p = rb.add(d_count.as<unsigned long long>(), 1, st, &rb_err);
    KV_HIP(rb_err);
    KV_HIP(rb.wait(st));
    const unsigned long long cnt = *cnt_p;
    KV_REQUIRE(cnt <= hit_cap, KV_ERR_CAPACITY, "kv_novel_scan_hashes: %llu hits exceed the buffer of %llu", cnt,
               (unsigned long long)hit_cap);
    *n_hits = cnt;
    return KV_OK;
}

__global__ void k_set_insert(unsigned long long *keys, uint8_t *abund, uint64_t mask, int S, const uint64_t *hashes,
                             const uint8_t *hash_abund, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hashes[i];
        if (h == KV_SET_NONE) continue;            // padding of the all-gather
        for (uint64_t slot = (h ^ (h >> 32)) & mask;; slot = (slot + 1) & mask) {
            const unsigned long long prev = atomicCAS(&keys[slot], KV_SET_NONE, (unsigned long long)h);
            if (prev == KV_SET_NONE) {
                for (int c = 0; c < S; ++c) abund[slot * (uint64_t)S + c] = hash_abund[i * (uint64_t)S + c];
                break;
            }
            if (prev == (unsigned long long)h) break;     // the same k-mer reported by another shard: same abundances
        }
    }
}

}  // namespace

// The scan of a read shard once the interesting k-mers are known (read-sharded multi-GPU run): d_hashes[n] with
// d_abund[n * nsamples] beside them -- every band owner's kv_novel_scan_distinct output, all-gathered; entries
// ~0 are padding -- become a hash set, and every k-mer of `reads` (reads with non-ACGT skipped, as in kv_novel_scan)
// whose hash is a member is a hit, reported with the abundances the set carries.  Hits leave in (read, offset) order.
namespace {
// the gathered (hash, abundances) rows as a hash set in this stream's arena: p.set_keys / set_abund / set_mask
int build_hit_set(NovelParams &p, int kind, int ksize, int nsamples, const void *d_hashes, const void *d_abund, uint64_t n, hipStream_t st)
{
    memset(&p, 0, sizeof(p));
    const int fam = kv_hashfam_of(kind);
    p.hp = make_hash_params(ksize, fam);
    p.ncase = nsamples;
    ScanArenas *arenas;
    {
        std::lock_guard<std::mutex> lk(g_scan_arenas_mu);
        arenas = &g_scan_arenas[kv_stream_key(st)];
    }
    uint64_t slots = 1024;
    while (slots < 2 * n + 64) slots <<= 1;
    {
        std::lock_guard<std::mutex> arena_lock(arenas->mu);
        KV_HIP(arenas->set.need(up256(slots * 8) + up256(slots * (uint64_t)nsamples)));
    }
    unsigned long long *keys = (unsigned long long *)arenas->set.p;
    uint8_t *abund = (uint8_t *)arenas->set.p + up256(slots * 8);
    KV_HIP(hipMemsetAsync(keys, 0xFF, slots * 8, st));
    if (n) {
        KvProfScope prof("k_set_insert");
        hipLaunchKernelGGL(k_set_insert, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, keys, abund, slots - 1,
                           nsamples, (const uint64_t *)d_hashes, (const uint8_t *)d_abund, n);
        KV_HIP(hipGetLastError());
    }
    p.set_keys = keys; p.set_abund = abund; p.set_mask = slots - 1;
    return KV_OK;
}
}  // namespace

// The same answer from the owner of the minimizer buckets (minimizer-sharded exchange): after kv_mex_route(keep_scan) on this stream,
// every occurrence of this rank's buckets whose hash is in the set leaves as (read << 16 | offset) with the S abundances the set
// carries -- reads the scan must skip included: the caller drops them (kevlar/novel.py:134-139) -- in arbitrary order, for the
// all-gather and kv_hits_from_tagged.  KV_ERR_CAPACITY: not available (see kv_skm_mex_scan_set); scan the shard instead.
extern "C" int kv_mex_scan_set(int kind, int ksize, int nsamples, const void *d_hashes, const void *d_abund, uint64_t n,
                               void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits)
{
    KV_REQUIRE(n_hits && nsamples >= 1 && nsamples <= KV_MAX_SAMPLES && ksize >= 1, KV_ERR_ARG, "kv_mex_scan_set: bad argument");
    KV_REQUIRE((n == 0 || (d_hashes && d_abund)) && (hit_cap == 0 || (d_hit_tags && d_hit_abund)), KV_ERR_ARG, "kv_mex_scan_set: null buffer");
    *n_hits = 0;
    NovelParams p;
    hipStream_t st = kv_stream();
    { const int rc = build_hit_set(p, kind, ksize, nsamples, d_hashes, d_abund, n, st); if (rc != KV_OK) return rc; }
    return kv_skm_mex_scan_set(p, ksize, (uint64_t *)d_hit_tags, (uint8_t *)d_hit_abund, hit_cap, n_hits);
}

extern "C" int kv_novel_scan_set(const kv_reads *reads, int kind, int ksize, int nsamples, const void *d_hashes,
                                 const void *d_abund, uint64_t n, kv_hits **out)
{
    KV_REQUIRE(reads && out && nsamples >= 1 && nsamples <= KV_MAX_SAMPLES && ksize >= 1, KV_ERR_ARG, "kv_novel_scan_set: bad argument");
    KV_REQUIRE(n == 0 || (d_hashes && d_abund), KV_ERR_ARG, "kv_novel_scan_set: null buffer");
    NovelParams p;
    const int fam = kv_hashfam_of(kind);
    hipStream_t st = kv_stream();
    { const int rc = build_hit_set(p, kind, ksize, nsamples, d_hashes, d_abund, n, st); if (rc != KV_OK) return rc; }
    uint64_t n_kmers = 0;
    kv_reads_num_kmers(reads, ksize, &n_kmers);
    // Membership is one read of a set that stays in the L2, so the test is nearly free and what is left is finding the hash of every
    // k-mer: from the 2-bit form of equal-length reads that is ~305 lane-instructions per occurrence (k_novel_mark_2bit), which is what
    // cutting, splitting and combining the shard costs per occurrence at 30x and more than it costs at a shard's 4-15x (measured per
    // rank of config 2: 3.1 -> 1.8 ms at N = 8).  KV_SET_SCAN=skm keeps the bucketed scan; other batches take it as before.
    const char *how = kv_knob("KV_SET_SCAN"), *nm2 = kv_knob("KV_NOVEL_2BIT"), *forced = kv_knob("KV_NOVEL_PATH");
    const bool two_bit = fam == HF_MURMUR && ksize >= SKM_MIN_K && ksize <= SKM_MAX_K && reads->uni_len >= (uint32_t)ksize && reads->uni_per_tile != 0 &&
                         !(how && strcmp(how, "skm") == 0) && !(nm2 && atoi(nm2) == 0) && !(forced && strcmp(forced, "tiles") == 0);
    const bool use_skm = !two_bit && kv_skm_eligible_kind(fam, ksize, reads, n_kmers, true);
    return scan_reads(p, reads, fam, n_kmers, use_skm, []() { return KV_OK; }, nullptr, 0, out);
}
