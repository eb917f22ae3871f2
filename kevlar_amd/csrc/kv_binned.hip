// kv_binned.hip -- K2 count for large batches: no global atomics on the tables.
//
// The straightforward kernel (k_consume) performs T saturating read-modify-writes per k-mer at
// random table positions.  Measured on MI355X each one moves a 64-B sector in and out of HBM
// (269 GB per 525 M k-mers, profiles/r1_naive) and the device tops out at ~27 G atomics/s.
// Counting is a histogram, so this file computes it the way large GPU histograms are done:
//
//   A  k_bin_hash_direct  hash every k-mer once (same LDS-staged tiles as k_consume), band/mask filter,
//                   and for each of the T tables append the bin to one of C coarse buckets
//                   (bin range = F slices of 65536 bins).  Every workgroup owns a private segment
//                   of every bucket; its write cursor lives in LDS and the item is stored straight
//                   to HBM (L2 merges the partial lines): no global atomic, no barrier.
//                   (k_bin_hash is the earlier variant that stages items in LDS rings and flushes
//                   coalesced bursts; k_bin_list takes hashes from a list instead of reads.)
//   B  k_bin_split  each coarse bucket is split into its F slices; items shrink to the 16-bit
//                   offset inside the slice.  The fan-out is too wide for direct stores here, so
//                   items collect in per-workgroup LDS rings and leave as coalesced bursts.
//   C  k_bin_apply  one workgroup per (table, slice): the slice's 64 KB of counters is loaded
//                   into LDS, every item is applied with an LDS compare-and-swap (saturating at
//                   255 / 15 / 1), and the slice is written back once.  Table 0 also yields the
//                   exact change of n_occupied.
//
// HBM traffic per k-mer: ~T*(4+4+2+2) B of items + one read and one write of the tables,
// against T*128 B for the atomic kernel.  The result is bit-identical (saturating adds commute).
//
// Skewed inputs (one k-mer repeated millions of times) cannot break it: a bucket that fills
// up diverts further items to a spill list that is applied with global atomics after stage C;
// if even that overflows, the launcher reports it and kv_consume falls back to k_consume.
#include <algorithm>
#include <cmath>
#include <map>
#include <type_traits>

#include "kv_binned.h"
#include "kv_device.h"
#include "kv_kmer2bit_device.h"

namespace {

#define BIN_RING_MIN 64
#define BIN_RING_MAX 4096   // a round appends at most 4096 items to one stream
#define BIN_B_THREADS 512   // 8 waves: one lane per slice stream for F <= 512
#define BIN_B_ITEMS 8       // items per thread per round in stage B
#define BIN_B_BUDGET 16384  // LDS ring entries per stage-B workgroup
#define BIN_C_THREADS 1024
#define BIN_CW_THREADS 512   // weighted stage C: 32-KB slices, four workgroups per CU
#define BIN_MAX_SEG 512      // stage-B writers per slice (nwgB)

// ---- LDS write-combining rings ------------------------------------------------------------
// ns streams, each a ring of R items (R a power of two) plus an append counter and a flushed
// counter.  Appends are LDS atomics; after a workgroup barrier every wave flushes the streams it
// owns: the owning lane decides, one global atomic claims the space, then the whole wave copies
// the burst (coalesced).  A stream that receives more than R items between two flushes diverts
// the excess to the caller's overflow path, so skew costs speed, never correctness.
template <typename ItemT>
struct Rings {
    ItemT *ring;
    uint32_t *cnt, *base;
    uint32_t R;
    uint32_t skew;      // item p of stream s sits in slot (p + skew * s) & (R - 1): streams fill in step, so unskewed
                        // slots would share LDS banks; stage B uses 8 so that 8-item vectors stay aligned
};

template <typename ItemT>
__device__ __forceinline__ bool ring_append(const Rings<ItemT> &rs, uint32_t s, ItemT item)
{
    const uint32_t pos = atomicAdd(&rs.cnt[s], 1u);
    if (pos - rs.base[s] >= rs.R) return false;
    rs.ring[s * rs.R + ((pos + rs.skew * s) & (rs.R - 1))] = item;
    return true;
}

// `written` is the owning lane's count of items already flushed for ITS stream (one stream per
// lane: callers guarantee ns <= 64 * waves), i.e. the position inside the private segment whose
// first element index the same lane holds in `seg_base`.  Bursts are copied two at a time, one per
// half-wave (a burst is 32-48 items, so a whole wave per burst would idle half its lanes), and all
// per-burst parameters travel by v_readlane: the flush phase is instruction-bound, not bandwidth-bound.
template <typename ItemT, typename Store, typename Overflow>
__device__ __forceinline__ void rings_flush(const Rings<ItemT> &rs, uint32_t ns, bool final, uint32_t &written,
                                            uint64_t seg_base, uint32_t cap, Store store, Overflow overflow, uint32_t quantum = 32u)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t per_wave = (ns + nwaves - 1) / nwaves;             // <= 64
    const uint32_t s0 = wave * per_wave, s_end = min(ns, s0 + per_wave);
    const uint32_t s = s0 + lane;
    uint32_t n = 0, base = 0, pos = 0;
    if (s < s_end) {
        base = rs.base[s];
        const uint32_t avail = rs.cnt[s] - base;
        uint32_t newbase = base;
        if (avail > rs.R) { n = rs.R; newbase = base + avail; }   // overran: the excess was diverted at append time
        else if (final) { n = avail; newbase = base + n; }
        else if (avail >= rs.R / 2) { n = avail & ~(quantum - 1u); newbase = base + n; }   // whole half-wave bursts: one copy iteration, all lanes busy
        if (n) { pos = written; written += n; rs.base[s] = newbase; }
    }
    const uint32_t seg_lo = (uint32_t)seg_base, seg_hi = (uint32_t)(seg_base >> 32);
    const bool upper = lane >= 32;
    const uint32_t sub = lane & 31u;
    unsigned long long todo = __ballot(n > 0);
    while (todo) {
        const int l0 = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int l1 = todo ? __ffsll((long long)todo) - 1 : l0;
        const bool two = todo != 0;
        if (two) todo &= todo - 1;
        // wave-uniform lane indices: v_readlane, not an LDS permute
        const uint32_t n0 = __builtin_amdgcn_readlane(n, l0), n1 = two ? __builtin_amdgcn_readlane(n, l1) : 0u;
        const uint32_t b0 = __builtin_amdgcn_readlane(base, l0), b1 = __builtin_amdgcn_readlane(base, l1);
        const uint32_t p0 = __builtin_amdgcn_readlane(pos, l0), p1 = __builtin_amdgcn_readlane(pos, l1);
        const uint32_t g0l = __builtin_amdgcn_readlane(seg_lo, l0), g1l = __builtin_amdgcn_readlane(seg_lo, l1);
        const uint32_t g0h = __builtin_amdgcn_readlane(seg_hi, l0), g1h = __builtin_amdgcn_readlane(seg_hi, l1);
        const uint32_t sl = s0 + (uint32_t)(upper ? l1 : l0);
        const uint32_t nl = upper ? n1 : n0, bl = upper ? b1 : b0, pl = upper ? p1 : p0;
        const uint64_t gl = ((uint64_t)(upper ? g1h : g0h) << 32) | (upper ? g1l : g0l);
        for (uint32_t j = sub; j < nl; j += 32) {
            const ItemT item = rs.ring[sl * rs.R + ((bl + j + rs.skew * sl) & (rs.R - 1))];
            if (pl + j < cap) store(gl + pl + j, item);
            else overflow(sl, item);
        }
    }
}

// Stage B's flush between rounds: the flush phase of rings_flush is instruction-bound (PMC: ~600 VALU
// wave-instructions per wave and round, 4/5 of them broadcasting per-burst parameters and looping), so here
// every lane that owns a ready stream copies its own burst: 32 (or R) items as 16-byte vectors, LDS read ->
// global store, all streams of the wave in parallel.  Bursts are multiples of 32 items and the ring skew is 8,
// so vectors stay 16-byte aligned in LDS and in the segment (u16 and u32 items alike).  The rare misfits (segment
// nearly full, ring overrun that left the base unaligned) are left to rings_flush.
template <typename ItemT, typename Overflow>
__device__ __forceinline__ void rings_flush_lanes(const Rings<ItemT> &rs, uint32_t ns, uint32_t &written, uint64_t seg_base,
                                                  uint32_t cap, ItemT *gbuf, Overflow overflow)
{
    constexpr uint32_t VEC = 16u / sizeof(ItemT);        // items per 16-byte vector
    constexpr uint32_t QUANTUM = 64u / sizeof(ItemT);    // bursts are whole 64-byte sectors
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t per_wave = (ns + nwaves - 1) / nwaves;
    const uint32_t s0 = wave * per_wave, s_end = min(ns, s0 + per_wave);
    const uint32_t s = s0 + lane;
    bool misfit = false;
    if (s < s_end) {
        const uint32_t base = rs.base[s];
        const uint32_t avail = rs.cnt[s] - base;
        uint32_t n = 0, newbase = base;
        if (avail > rs.R) { n = rs.R; newbase = base + avail; }
        else if (avail >= rs.R / 2) { n = avail & ~(QUANTUM - 1u); newbase = base + n; }
        if (n) {
            if ((base & 7u) || written + n > cap || (newbase & 7u)) misfit = true;
            else {
                const ItemT *ring = rs.ring + s * rs.R;
                ItemT *dst = gbuf + seg_base + written;
                for (uint32_t v = 0; v < n; v += VEC)
                    *(uint4 *)(dst + v) = *(const uint4 *)(ring + ((base + v + rs.skew * s) & (rs.R - 1)));
                written += n;
                rs.base[s] = newbase;
            }
        }
    }
    if (__ballot(misfit)) {
        auto store = [&](uint64_t idx, ItemT off) { gbuf[idx] = off; };
        rings_flush(rs, ns, false, written, seg_base, cap, store, overflow, QUANTUM);
    }
}

__device__ __forceinline__ void rings_store_counts(uint32_t ns, uint32_t written, uint32_t cap, uint32_t *counts, uint32_t stride, uint32_t writer)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    const uint32_t per_wave = (ns + nwaves - 1) / nwaves;
    const uint32_t s = wave * per_wave + lane;
    if (lane < per_wave && s < ns) counts[(uint64_t)s * stride + writer] = written < cap ? written : cap;
}

// one k-mer -> T ring appends.  All T ring positions are requested back to back (independent LDS
// atomics in flight together) before any of them is consumed.
__device__ __forceinline__ uint32_t bin_push(const Rings<uint32_t> &rs, const BinGeom &g, const SketchDev *__restrict__ sk, uint64_t h)
{
    uint32_t sidx[BIN_MAX_T], item[BIN_MAX_T], pos[BIN_MAX_T];
    uint64_t bins[BIN_MAX_T];
#pragma unroll
    for (int t = 0; t < BIN_MAX_T; ++t) {
        if (t >= g.T) break;
        const uint64_t bin = fastmod(h, sk->size[t], sk->magic[t]);
        const uint32_t slice = (uint32_t)(bin >> 16);
        const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
        bins[t] = bin;
        item[t] = ((slice - c * (uint32_t)g.F) << 16) | (uint32_t)(bin & 0xffffu);
        sidx[t] = (uint32_t)t * (uint32_t)g.C + c;
    }
#pragma unroll
    for (int t = 0; t < BIN_MAX_T; ++t)
        if (t < g.T) pos[t] = atomicAdd(&rs.cnt[sidx[t]], 1u);
#pragma unroll
    for (int t = 0; t < BIN_MAX_T; ++t) {
        if (t >= g.T) break;
        const uint32_t s_ = sidx[t];
        if (pos[t] - rs.base[s_] < rs.R) rs.ring[s_ * rs.R + ((pos[t] + rs.skew * s_) & (rs.R - 1))] = item[t];
        else spill_item(g, t, bins[t]);
    }
    return 0u;
}

// ---- stage A -----------------------------------------------------------------------------
template <int THREADS, int NW>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 6 : 4)) void k_bin_hash(   // 3 x 8 waves (or 1 x 16) per CU must fit the register file
    ReadsDev rd, uint32_t n_tiles, const SketchDev *__restrict__ sk,
                                                      const SketchDev *__restrict__ mask, ConsumeFilter f, BinGeom g)
{
    __shared__ TileShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)smem;     // [0, tile_lds): staged ASCII; rings follow
    const uint32_t ns = (uint32_t)(g.T * g.C);
    Rings<uint32_t> rs;
    rs.R = g.ringA;
    rs.skew = 1;
    rs.ring = (uint32_t *)(smem + g.tile_lds);
    rs.cnt = rs.ring + (size_t)ns * rs.R;
    rs.base = rs.cnt + ns;
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS) { rs.cnt[s] = 0; rs.base[s] = 0; }
    __syncthreads();
    uint32_t written = 0;
    uint64_t seg_base = 0;        // element index of this lane's private segment (lane <-> stream as in rings_flush)
    {
        const uint32_t nwaves = THREADS >> 6, per_wave = (ns + nwaves - 1) / nwaves;
        const uint32_t mine = (threadIdx.x >> 6) * per_wave + (threadIdx.x & 63);
        if ((threadIdx.x & 63) < per_wave && mine < ns) seg_base = ((uint64_t)mine * g.nwgA + blockIdx.x) * g.cap1;
    }
    auto store = [&](uint64_t idx, uint32_t item) { g.gbuf1[idx] = item; };
    auto overflow = [&](uint32_t s, uint32_t item) {   // private segment full: keep the increment, apply it later with an atomic
        const uint32_t t = s / (uint32_t)g.C, c = s % (uint32_t)g.C;
        spill_item(g, (int)t, (((uint64_t)c * g.F + (item >> 16)) << 16) | (item & 0xffffu));
    };
    uint64_t n_added = 0;
    // tiles are handed out dynamically (one global atomic per tile): a workgroup that becomes resident
    // late, or shares its CU with fewer siblings, simply takes fewer -- a static deal is hostage to the
    // slowest workgroup, which showed as 46 vs 64 ms for the same launch on different boxes.  A quota of
    // 1.5x the average share keeps early workgroups from swallowing everything when the grid is not fully
    // resident (GPU shared with another stream or process): that overflowed their segments into the spill list
    __shared__ uint32_t next_tile;
    for (uint32_t taken = 0; taken < g.quotaA; ++taken) {
        __syncthreads();
        if (threadIdx.x == 0) next_tile = (uint32_t)atomicAdd(&g.ctr[4], 1ull);
        __syncthreads();
        const uint32_t tile = next_tile;
        if (tile >= n_tiles) break;
        uint32_t read0;
        const uint32_t nr = stage_tile(sh, rd, tile, f.hp.k, 0, 0, read0);
        const uint32_t total = sh.kpre[nr];
        // every thread owns a run of consecutive k-mers (rolling register windows when NW > 0); one
        // k-mer per thread per round, then the workgroup flushes its rings
        const uint32_t run = (total + THREADS - 1) / THREADS;
        const uint32_t q0 = threadIdx.x * run, q1 = min(total, q0 + run);
        KmerRoll<(NW > 0 ? NW : 8)> w;
        if (NW > 0 && q0 < q1) {
            locate_kmer(sh, nr, q0, w.r, w.i);
            roll_load(w, sh, f.hp.k);
        }
        for (uint32_t step = 0; step < run; ++step) {
            const uint32_t q = NW > 0 ? q0 + step : step * THREADS + threadIdx.x;
            const bool live = NW > 0 ? q < q1 : q < total;
            if (live) {
                uint64_t h;
                if (NW > 0) {
                    h = roll_hash(w, f.hp);
                    if (q + 1 < q1) roll_step(w, sh, nr, f.hp.k);
                } else {
                    uint32_t r, i;
                    locate_kmer(sh, nr, q, r, i);
                    const uint32_t fwd = sh.foff[r] + i;
                    const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)f.hp.k - i);
                    h = kmer_hash_lds(sh.ascii, fwd, rc, f.hp);
                }
                if (consume_filter_pass(f, mask, h)) {
                    n_added += 1;
                    n_added += bin_push(rs, g, sk, h);
                }
            }
            __syncthreads();
            rings_flush(rs, ns, false, written, seg_base, (uint32_t)g.cap1, store, overflow);
            __syncthreads();
        }
    }
    rings_flush(rs, ns, true, written, seg_base, (uint32_t)g.cap1, store, overflow);
    rings_store_counts(ns, written, (uint32_t)g.cap1, g.gcnt1, g.nwgA, blockIdx.x);
    n_added = wave_sum_u64(n_added);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
}

// Stage A without LDS rings: every workgroup still owns a private segment of every coarse bucket, but
// the segment's write cursor lives in LDS and each item is stored straight to HBM -- no barrier, no
// flush phase between hashing a k-mer and storing its T items.  The write frontier (workgroups x
// buckets x one 128-B line) sits in L2, which merges the partial lines before they reach HBM.
template <int THREADS, int NW>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 6 : 4)) void k_bin_hash_direct(
    ReadsDev rd, uint32_t n_tiles, const SketchDev *__restrict__ sk, const SketchDev *__restrict__ mask,
    ConsumeFilter f, BinGeom g)
{
    __shared__ TileShared sh;
    __shared__ uint32_t cur[BIN_MAX_T * BIN_C];
    __shared__ uint32_t next_tile;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)smem;
    const uint32_t ns = (uint32_t)(g.T * g.C);
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS) cur[s] = 0;
    uint64_t n_added = 0;
    for (uint32_t taken = 0; taken < g.quotaA; ++taken) {
        __syncthreads();
        if (threadIdx.x == 0) next_tile = (uint32_t)atomicAdd(&g.ctr[4], 1ull);
        __syncthreads();
        const uint32_t tile = next_tile;
        if (tile >= n_tiles) break;
        uint32_t read0;
        const uint32_t nr = stage_tile(sh, rd, tile, f.hp.k, 0, 0, read0);
        const uint32_t total = sh.kpre[nr];
        const uint32_t run = (total + THREADS - 1) / THREADS;
        const uint32_t q0 = threadIdx.x * run, q1 = min(total, q0 + run);
        KmerRoll<(NW > 0 ? NW : 8)> w;
        if (NW > 0 && q0 < q1) {
            locate_kmer(sh, nr, q0, w.r, w.i);
            roll_load(w, sh, f.hp.k);
        }
        for (uint32_t step = 0; step < run; ++step) {
            const uint32_t q = NW > 0 ? q0 + step : step * THREADS + threadIdx.x;
            const bool live = NW > 0 ? q < q1 : q < total;
            if (!live) continue;
            uint64_t h;
            if (NW > 0) {
                h = roll_hash(w, f.hp);
                if (q + 1 < q1) roll_step(w, sh, nr, f.hp.k);
            } else {
                uint32_t r, i;
                locate_kmer(sh, nr, q, r, i);
                const uint32_t fwd = sh.foff[r] + i;
                const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)f.hp.k - i);
                h = kmer_hash_lds(sh.ascii, fwd, rc, f.hp);
            }
            if (!consume_filter_pass(f, mask, h)) continue;
            n_added += 1;
            uint32_t sidx[BIN_MAX_T], item[BIN_MAX_T], pos[BIN_MAX_T];
            uint64_t bins[BIN_MAX_T];
#pragma unroll
            for (int t = 0; t < BIN_MAX_T; ++t) {
                if (t >= g.T) break;
                const uint64_t bin = fastmod(h, sk->size[t], sk->magic[t]);
                const uint32_t slice = (uint32_t)(bin >> 16);
                const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
                bins[t] = bin;
                item[t] = ((slice - c * (uint32_t)g.F) << 16) | (uint32_t)(bin & 0xffffu);
                sidx[t] = (uint32_t)t * (uint32_t)g.C + c;
            }
#pragma unroll
            for (int t = 0; t < BIN_MAX_T; ++t)
                if (t < g.T) pos[t] = atomicAdd(&cur[sidx[t]], 1u);
#pragma unroll
            for (int t = 0; t < BIN_MAX_T; ++t) {
                if (t >= g.T) break;
                if (pos[t] < g.cap1) g.gbuf1[((uint64_t)sidx[t] * g.nwgA + blockIdx.x) * g.cap1 + pos[t]] = item[t];
                else spill_item(g, t, bins[t]);
            }
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS)
        g.gcnt1[(uint64_t)s * g.nwgA + blockIdx.x] = (uint32_t)min((uint64_t)cur[s], g.cap1);
    n_added = wave_sum_u64(n_added);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
}

// Stage A for batches of equal-length reads, hashed from the 2-bit form (kv_kmer2bit_device.h): a thread takes BIN2_CH consecutive
// k-mers of one read; what passes the band / mask filter is collected per wave and routed 64 at a time -- T reductions, T items through
// the LDS cursors, direct stores, exactly as k_bin_hash_direct stores them.  A ticket is BIN2_TILES tiles' worth of reads.
#define BIN2_CH 10
#define BIN2_TILES 8u
// FK = 31: the instance for kevlar's default k, with the murmurs' first multiplications from the product tables (skm_key_hash_pl)
template <int THREADS, int KW, int FK = 0>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 6 : 4)) void k_bin_hash_2bit(
    ReadsDev rd, uint32_t n_units, const SketchDev *__restrict__ sk, const SketchDev *__restrict__ mask, ConsumeFilter f, BinGeom g)
{
    __shared__ __attribute__((aligned(8))) uint32_t lut[FK ? 1024 : 256];
    __shared__ uint32_t cur[BIN_MAX_T * BIN_C];
    __shared__ uint32_t next_unit;
    __shared__ unsigned long long queue[(THREADS / 64) * 128];
    const uint32_t ns = (uint32_t)(g.T * g.C);
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS) cur[s] = 0;
    if (threadIdx.x < 256) {
        if (FK) { ((uint64_t *)lut)[threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C1); ((uint64_t *)lut)[256 + threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C2); }
        else lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    }
    const int k = FK ? FK : f.hp.k;
    const uint32_t L = rd.uni_len, wpr = (L + 15u) >> 4, nk = L - (uint32_t)k + 1u, cpr = (nk + BIN2_CH - 1u) / BIN2_CH;
    const float inv_cpr = 1.0f / (float)cpr;
    const uint64_t reads_per_unit = (uint64_t)BIN2_TILES * rd.uni_per_tile;
    WaveQueue wq;
    wq.q = queue + (threadIdx.x >> 6) * 128u;
    wq.n = 0;
    uint64_t n_added = 0;
    auto route = [&](bool have, uint64_t h) {
        if (!have) return;
        n_added += 1;
        uint32_t sidx[BIN_MAX_T], item[BIN_MAX_T], pos[BIN_MAX_T];
        uint64_t bins[BIN_MAX_T];
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t) {
            if (t >= g.T) break;
            const uint64_t bin = fastmod(h, sk->size[t], sk->magic[t]);
            const uint32_t slice = (uint32_t)(bin >> 16);
            const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
            bins[t] = bin;
            item[t] = ((slice - c * (uint32_t)g.F) << 16) | (uint32_t)(bin & 0xffffu);
            sidx[t] = (uint32_t)t * (uint32_t)g.C + c;
        }
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t)
            if (t < g.T) pos[t] = atomicAdd(&cur[sidx[t]], 1u);
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t) {
            if (t >= g.T) break;
            if (pos[t] < g.cap1) g.gbuf1[((uint64_t)sidx[t] * g.nwgA + blockIdx.x) * g.cap1 + pos[t]] = item[t];
            else spill_item(g, t, bins[t]);
        }
    };
    for (uint32_t taken = 0; taken < g.quotaA; ++taken) {
        __syncthreads();
        if (threadIdx.x == 0) next_unit = (uint32_t)atomicAdd(&g.ctr[4], 1ull);
        __syncthreads();
        const uint32_t unit = next_unit;
        if (unit >= n_units) break;
        const uint64_t r0 = (uint64_t)unit * reads_per_unit;
        const uint32_t nr = (uint32_t)min(reads_per_unit, rd.n_reads - r0), n_items = nr * cpr;
        // (whole waves go round together: the queue's ballots want every lane there)
        for (uint32_t i0 = (threadIdx.x & ~63u); i0 < n_items; i0 += THREADS) {
            const uint32_t i = i0 + (threadIdx.x & 63u);
            const bool mine = i < n_items;
            const uint32_t r = mine ? k2_div(i, cpr, inv_cpr) : 0u, j0 = mine ? (i - r * cpr) * BIN2_CH : 0u;
            const uint32_t cnt = mine ? min((uint32_t)BIN2_CH, nk - j0) : 0u;
            kmer2bit_walk<KW, BIN2_CH, FK>(rd.words + (r0 + r) * wpr, j0, cnt, k, lut, f.hp, [&](bool live, uint64_t h) {
                wave_queue_push(wq, live && consume_filter_pass(f, mask, h), h, route);
            });
        }
    }
    wave_queue_flush(wq, route);
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS)
        g.gcnt1[(uint64_t)s * g.nwgA + blockIdx.x] = (uint32_t)min((uint64_t)cur[s], g.cap1);
    n_added = wave_sum_u64(n_added);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
}

// stage A over a list of hashes already in HBM (read-sharded multi-GPU count: the hashes of this
// rank's band arrive from the other ranks, kv_shard.hip).  Element i sits at list[i * stride].
#define BIN_LIST_ROUNDS 16
// W: the list holds (hash, count) pairs -- element i at list[i * stride], its count at list[i * stride + 1] -- and the
// items carry min(count, 255) as their weight (what a band owner receives from ranks that deduplicated their shards)
template <int THREADS, bool W>
__global__ __launch_bounds__(THREADS, (THREADS == 512 ? 6 : 4)) void k_bin_list(
    const uint64_t *__restrict__ list, uint64_t n, uint32_t stride, const SketchDev *__restrict__ sk, BinGeom g)
{
    // same direct stores through LDS cursors as k_bin_hash_direct; the hashes just come from HBM
    __shared__ uint32_t cur[BIN_MAX_T * BIN_C];
    __shared__ uint64_t next_chunk;
    const uint32_t ns = (uint32_t)(g.T * g.C);
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS) cur[s] = 0;
    const uint64_t chunk = (uint64_t)THREADS * BIN_LIST_ROUNDS;
    uint64_t n_added = 0;
    for (uint32_t taken = 0; taken < g.quotaA; ++taken) {
        __syncthreads();
        if (threadIdx.x == 0) next_chunk = (uint64_t)atomicAdd(&g.ctr[4], 1ull);
        __syncthreads();
        const uint64_t i0 = next_chunk * chunk;
        if (i0 >= n) break;
        uint64_t idx = i0 + threadIdx.x;
        uint64_t h_next = idx < n ? list[idx * stride] : 0, c_next = (W && idx < n) ? list[idx * stride + 1] : 1;
        for (int round = 0; round < BIN_LIST_ROUNDS; ++round) {
            const uint64_t h = h_next, count = c_next;
            const bool live = idx < n;
            idx += THREADS;
            if (round + 1 < BIN_LIST_ROUNDS && idx < n) {   // flies while this one is routed
                h_next = list[idx * stride];
                if (W) c_next = list[idx * stride + 1];
            }
            if (!live) continue;
            n_added += count;
            uint32_t left = W ? (uint32_t)min(count, (uint64_t)255) : 1u;
            while (left) {
                const uint32_t wgt = W ? min(left, BIN_W_MAX) : 1u;
                left -= wgt;
                uint32_t sidx[BIN_MAX_T], item[BIN_MAX_T], pos[BIN_MAX_T];
                uint64_t bins[BIN_MAX_T];
#pragma unroll
                for (int t = 0; t < BIN_MAX_T; ++t) {
                    if (t >= g.T) break;
                    const uint64_t bin = fastmod(h, sk->size[t], sk->magic[t]);
                    const uint32_t slice = (uint32_t)(bin >> g.sbits);
                    const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
                    bins[t] = bin;
                    item[t] = ((slice - c * (uint32_t)g.F) << g.sbits) | ((uint32_t)bin & ((1u << g.sbits) - 1u)) | (W ? (wgt - 1u) << BIN_W_SHIFT : 0u);
                    sidx[t] = (uint32_t)t * (uint32_t)g.C + c;
                }
#pragma unroll
                for (int t = 0; t < BIN_MAX_T; ++t)
                    if (t < g.T) pos[t] = atomicAdd(&cur[sidx[t]], 1u);
#pragma unroll
                for (int t = 0; t < BIN_MAX_T; ++t) {
                    if (t >= g.T) break;
                    if (pos[t] < g.cap1) g.gbuf1[((uint64_t)sidx[t] * g.nwgA + blockIdx.x) * g.cap1 + pos[t]] = item[t];
                    else spill_item(g, t, bins[t], wgt);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < ns; s += THREADS)
        g.gcnt1[(uint64_t)s * g.nwgA + blockIdx.x] = (uint32_t)min((uint64_t)cur[s], g.cap1);
    if (W) {
        n_added = wave_sum_u64(n_added);
        if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
    }
}

// ---- stage B -----------------------------------------------------------------------------
// W = weighted items (kv_binned.h): a coarse item carries its increment in bits 25..31 and leaves as offset | increment << 16
template <bool W, int SBITS>
__global__ __launch_bounds__(BIN_B_THREADS) void k_bin_split(BinGeom g)
{
    typedef typename std::conditional<W, uint32_t, uint16_t>::type Out;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t s = blockIdx.y;                  // coarse stream = table * C + bucket
    const uint32_t t = s / (uint32_t)g.C, c = s % (uint32_t)g.C;
    const uint32_t F = (uint32_t)g.F;
    Rings<Out> rs;
    rs.R = g.ringB;
    rs.skew = 8;
    rs.ring = (Out *)smem;
    rs.cnt = (uint32_t *)(smem + (((size_t)F * rs.R * sizeof(Out) + 15) & ~(size_t)15));
    rs.base = rs.cnt + F;
    for (uint32_t i = threadIdx.x; i < F; i += BIN_B_THREADS) { rs.cnt[i] = 0; rs.base[i] = 0; }
    __syncthreads();
    uint32_t written = 0;
    uint64_t seg_base = 0;
    {
        const uint32_t nwaves = BIN_B_THREADS >> 6, per_wave = (F + nwaves - 1) / nwaves;
        const uint32_t mine = (threadIdx.x >> 6) * per_wave + (threadIdx.x & 63);
        if ((threadIdx.x & 63) < per_wave && mine < F) seg_base = (((uint64_t)s * F + mine) * g.nwgB + blockIdx.x) * g.cap2;
    }
    Out *gbuf2 = (Out *)g.gbuf2;
    constexpr uint32_t SB = SBITS, OFFMASK = (1u << SB) - 1u;
    auto slice_of = [](uint32_t item) { return W ? (item >> SB) & ((1u << (BIN_W_SHIFT - SB)) - 1u) : item >> SB; };
    auto fine_of = [](uint32_t item) { return W ? (Out)((item & OFFMASK) | (((item >> BIN_W_SHIFT) + 1u) << 16)) : (Out)(item & OFFMASK); };
    auto spill_coarse = [&](uint32_t item) {
        spill_item(g, (int)t, (((uint64_t)c * F + slice_of(item)) << SB) | (item & OFFMASK), W ? (item >> BIN_W_SHIFT) + 1u : 1u);
    };
    auto store = [&](uint64_t idx, Out off) { gbuf2[idx] = off; };
    auto overflow = [&](uint32_t fi, Out off) {
        spill_item(g, (int)t, (((uint64_t)c * F + fi) << SB) | ((uint32_t)off & OFFMASK), W ? (uint32_t)off >> 16 : 1u);
    };
    // Weighted items are u32: rings of 32 entries (64 would leave room for one workgroup per CU only) and half as many
    // items per round, so that a round cannot overrun a ring that was below its flush threshold
    constexpr int ITEMS = W ? 4 : BIN_B_ITEMS;
    const uint64_t step = (uint64_t)BIN_B_THREADS * ITEMS;
    // this workgroup drains the private segments seg = blockIdx.x, blockIdx.x + nwgB, ... of bucket s
    for (uint32_t seg = blockIdx.x; seg < g.nwgA; seg += g.nwgB) {
        uint64_t end = g.gcnt1[(uint64_t)s * g.nwgA + seg];
        if (end > g.cap1) end = g.cap1;
        const uint32_t *src = g.gbuf1 + ((uint64_t)s * g.nwgA + seg) * g.cap1;
        uint4 va = make_uint4(0, 0, 0, 0), vb = va;
        auto fetch = [&](uint64_t r0, uint4 &a, uint4 &b) {
            const uint64_t i0 = r0 + (uint64_t)threadIdx.x * ITEMS;
            if (i0 + ITEMS <= end) { a = *(const uint4 *)(src + i0); if (ITEMS == 8) b = *(const uint4 *)(src + i0 + 4); }
        };
        fetch(0, va, vb);
        for (uint64_t r0 = 0; r0 < end; r0 += step) {
            const uint64_t i0 = r0 + (uint64_t)threadIdx.x * ITEMS;
            uint32_t items[ITEMS];
            int have = 0;
            if (i0 + ITEMS <= end) {
                items[0] = va.x; items[1] = va.y; items[2] = va.z; items[3] = va.w;
                if (ITEMS == 8) { items[ITEMS - 4] = vb.x; items[ITEMS - 3] = vb.y; items[ITEMS - 2] = vb.z; items[ITEMS - 1] = vb.w; }
                have = ITEMS;
            } else {
                for (uint64_t i = i0; i < end; ++i) items[have++] = src[i];
            }
            if (r0 + step < end) fetch(r0 + step, va, vb);          // next round's items fly during this round
            if (have == ITEMS) {
                // full vector: all ring positions are requested back to back (independent LDS atomics in flight
                // together), then consumed; an item whose ring is full is rare and handled after the fast path
                uint32_t pos[ITEMS], rbase[ITEMS];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) pos[j] = atomicAdd(&rs.cnt[slice_of(items[j])], 1u);
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) rbase[j] = rs.base[slice_of(items[j])];
                uint32_t full = 0;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const uint32_t fi = slice_of(items[j]);
                    if (pos[j] - rbase[j] < rs.R) rs.ring[fi * rs.R + ((pos[j] + rs.skew * fi) & (rs.R - 1))] = fine_of(items[j]);
                    else full |= 1u << j;
                }
                if (full) {
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if (full & (1u << j)) spill_coarse(items[j]);
                }
            } else {
                for (int j = 0; j < have; ++j)
                    if (!ring_append(rs, slice_of(items[j]), fine_of(items[j]))) spill_coarse(items[j]);
            }
            __syncthreads();
            rings_flush_lanes(rs, F, written, seg_base, (uint32_t)g.cap2, gbuf2, overflow);
            __syncthreads();
        }
    }
    rings_flush(rs, F, true, written, seg_base, (uint32_t)g.cap2, store, overflow, 64u / (uint32_t)sizeof(Out));
    rings_store_counts(F, written, (uint32_t)g.cap2, g.gcnt2 + (uint64_t)s * F * g.nwgB, g.nwgB, blockIdx.x);
}

// Stage B for weighted items, sorted in LDS (round 5).  The ring kernel above appends 2048 items per round to ~380 LDS rings and
// then lets a lane per ring look whether its ring has a burst to flush, between two barriers: ~70 lane-instructions per item, and -- like
// every kernel of the count stage -- bound by instruction issue, not by the 3.4 GB it moves.  Here a workgroup takes 8192 coarse items at
// a time, ranks them by slice with LDS atomics, lays them out slice by slice in LDS (as coarse items: they carry their slice) and copies
// that image out with consecutive lanes on consecutive items, converting to fine items on the way: a slice's items of one chunk leave
// as one contiguous run of ~20.  The scheme of k_skm_split_sorted, for 4-byte items.  MEASURED SLOWER than the rings (1.33 against 1.08 ms
// per sample: five barriers per 8192 items with two workgroups' worth of registers per thread) and therefore not the default.
#define BIN_BS_CHUNK 8192u
#define BIN_BS_MAXSEG 64u          // coarse segments one stage-B workgroup drains (nwgB >= nwgA / 32: at most 32)
template <int SBITS>
__global__ __launch_bounds__(BIN_B_THREADS) void k_bin_split_sorted(BinGeom g)
{
    __shared__ uint32_t cur[BIN_MAX_F], hist[BIN_MAX_F], off[BIN_MAX_F];
    __shared__ uint32_t spre[BIN_BS_MAXSEG + 1], scnt[BIN_BS_MAXSEG];
    __shared__ uint32_t wsum[BIN_B_THREADS / 64];
    __shared__ uint32_t chunk_items;
    extern __shared__ __attribute__((aligned(16))) uint32_t bimg[];      // [BIN_BS_CHUNK] coarse items in sorted order
    const uint32_t s = blockIdx.y;
    const uint32_t t = s / (uint32_t)g.C, c = s % (uint32_t)g.C, F = (uint32_t)g.F;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr uint32_t SB = SBITS, OFFMASK = (1u << SB) - 1u;
    auto slice_of = [](uint32_t item) { return (item >> SB) & ((1u << (BIN_W_SHIFT - SB)) - 1u); };
    auto fine_of = [](uint32_t item) { return (item & OFFMASK) | (((item >> BIN_W_SHIFT) + 1u) << 16); };
    for (uint32_t f = threadIdx.x; f < F; f += BIN_B_THREADS) { cur[f] = 0; hist[f] = 0; }
    // the segments seg = blockIdx.x, blockIdx.x + nwgB, ... of coarse stream s, enumerated flat; a segment's count is rounded up to whole
    // 16-byte vectors in that enumeration (its base is 128-byte aligned, so every vector load is aligned; the padding items are masked)
    const uint32_t nmine = min((g.nwgA - blockIdx.x + g.nwgB - 1u) / g.nwgB, BIN_BS_MAXSEG);
    if (threadIdx.x < nmine) {
        const uint64_t n = g.gcnt1[(uint64_t)s * g.nwgA + blockIdx.x + threadIdx.x * g.nwgB];
        scnt[threadIdx.x] = (uint32_t)(n < g.cap1 ? n : g.cap1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < nmine; ++i) { spre[i] = acc; acc += (scnt[i] + 3u) & ~3u; }
        spre[nmine] = acc;
    }
    __syncthreads();
    const uint32_t total = spre[nmine];
    constexpr uint32_t PER = BIN_BS_CHUNK / BIN_B_THREADS / 4u;          // 16-byte vectors per thread and chunk
    uint32_t *const out = (uint32_t *)g.gbuf2 + ((uint64_t)s * F * g.nwgB + blockIdx.x) * g.cap2;      // + slice * fstride
    const uint64_t fstride = (uint64_t)g.nwgB * g.cap2;
    uint4 nv[PER];
    uint32_t nn[PER];                                  // real items of the vector (0..4)
    auto request = [&](uint32_t c0) {
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t gi = c0 + (r * BIN_B_THREADS + threadIdx.x) * 4u;
            nv[r] = make_uint4(0, 0, 0, 0); nn[r] = 0;
            if (gi < total) {
                uint32_t lo = 0, hi = nmine;
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (spre[mid] <= gi) lo = mid; else hi = mid; }
                const uint32_t at = gi - spre[lo];
                nn[r] = min(4u, scnt[lo] > at ? scnt[lo] - at : 0u);
                const uint32_t *src = g.gbuf1 + ((uint64_t)s * g.nwgA + blockIdx.x + (uint64_t)lo * g.nwgB) * g.cap1 + at;
                if (nn[r]) nv[r] = *(const uint4 *)src;
            }
        }
    };
    request(0);
    for (uint32_t c0 = 0; c0 < total; c0 += BIN_BS_CHUNK) {
        uint32_t it[PER][4], rank[PER][4], have[PER];
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) { it[r][0] = nv[r].x; it[r][1] = nv[r].y; it[r][2] = nv[r].z; it[r][3] = nv[r].w; have[r] = nn[r]; }
        request(c0 + BIN_BS_CHUNK);                     // the next chunk's items fly while this one is ranked, laid out and stored
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r)
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) rank[r][e] = e < have[r] ? atomicAdd(&hist[slice_of(it[r][e])], 1u) : 0u;
        __syncthreads();
        {   // exclusive scan of the chunk's histogram (F <= 512: one slice per thread); the slice's run starts at slot cur[f] of its segment
            const uint32_t f = threadIdx.x;
            const uint32_t h = f < F ? hist[f] : 0u;
            uint32_t incl = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = incl - h;
            for (uint32_t wv = 0; wv < wave; ++wv) before += wsum[wv];
            if (f < F) {
                off[f] = before;
                const uint32_t at = cur[f];
                cur[f] = at + h;
                hist[f] = at - before;                  // slot = sorted position + this (mod 2^32)
            }
            if (threadIdx.x == BIN_B_THREADS - 1) chunk_items = before + h;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r)
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e)
                if (e < have[r]) bimg[off[slice_of(it[r][e])] + rank[r][e]] = it[r][e];
        __syncthreads();
        const uint32_t n = chunk_items;
        for (uint32_t q = threadIdx.x; q < n; q += BIN_B_THREADS) {
            const uint32_t item = bimg[q], f = slice_of(item), slot = q + hist[f];
            if (slot < (uint32_t)g.cap2) out[(uint64_t)f * fstride + slot] = fine_of(item);
            else spill_item(g, (int)t, (((uint64_t)c * F + f) << SB) | (item & OFFMASK), (item >> BIN_W_SHIFT) + 1u);
        }
        __syncthreads();
        for (uint32_t f = threadIdx.x; f < F; f += BIN_B_THREADS) hist[f] = 0;
        __syncthreads();
    }
    for (uint32_t f = threadIdx.x; f < F; f += BIN_B_THREADS)
        g.gcnt2[((uint64_t)s * F + f) * g.nwgB + blockIdx.x] = min(cur[f], (uint32_t)g.cap2);
}

// ---- stage C -----------------------------------------------------------------------------
__device__ __forceinline__ bool lds_inc(uint32_t *lds, uint32_t off, int storage)
{
    if (storage == ST_BIT) {
        const uint32_t bit = 1u << (off & 31);
        return (atomicOr(&lds[off >> 5], bit) & bit) == 0;
    }
    uint32_t *w;
    uint32_t shift, maxv;
    if (storage == ST_BYTE) {
        w = &lds[off >> 2]; shift = (off & 3) * 8u; maxv = 255u;
    } else {
        const uint32_t byte = off >> 1;
        w = &lds[byte >> 2]; shift = (byte & 3) * 8u + ((off & 1) ? 0u : 4u); maxv = 15u;
    }
    uint32_t old = *w;
    for (;;) {
        const uint32_t cur = (old >> shift) & maxv;
        if (cur == maxv) return false;
        const uint32_t prev = atomicCAS(w, old, old + (1u << shift));
        if (prev == old) return cur == 0;
        old = prev;
    }
}

// Eight saturating increments with the LDS round trips overlapped: all eight words are read first, then
// all eight compare-and-swaps are issued back to back, and only an item whose word changed in between
// (another lane, or an earlier item of this same vector, hit the same word) takes the retry loop.  Returns
// how many bins went 0 -> 1.  STORAGE is a template parameter and the fast path is branch-free: with a
// run-time switch and a branch per item this loop was instruction-bound (~175 instructions per item).
template <int STORAGE>
__device__ __forceinline__ uint32_t lds_inc8(uint32_t *lds, const uint32_t (&w)[4], uint32_t n)
{
    uint32_t fresh = 0;
    if (STORAGE == ST_BIT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t off = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
            if ((uint32_t)e < n) {
                const uint32_t bit = 1u << (off & 31);
                fresh += (atomicOr(&lds[off >> 5], bit) & bit) == 0 ? 1u : 0u;
            }
        }
        return fresh;
    }
    constexpr uint32_t maxv = STORAGE == ST_BYTE ? 255u : 15u;
    uint32_t widx[8], inc[8], old[8], prev[8];
    uint32_t retry = 0;                                  // bit e: item e lost its CAS
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t off = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
        uint32_t shift;
        if (STORAGE == ST_BYTE) { widx[e] = off >> 2; shift = (off & 3u) << 3; }
        else { widx[e] = off >> 3; shift = (((off >> 1) & 3u) << 3) + ((off & 1u) ? 0u : 4u); }
        inc[e] = 1u << shift;
    }
    // two groups of four: the window between reading a word and swapping it stays short (fewer lost races)
#pragma unroll
    for (int g4 = 0; g4 < 8; g4 += 4) {
#pragma unroll
        for (int e = g4; e < g4 + 4; ++e) old[e] = lds[widx[e]];          // absent items (e >= n) read a harmless word
        bool live[4];
#pragma unroll
        for (int e = g4; e < g4 + 4; ++e) {
            // a saturated counter (field == inc * maxv) can only stay saturated; it and absent items issue a
            // compare-and-swap that writes back what it compared against: branch-free, and a no-op either way
            live[e - g4] = (uint32_t)e < n && (old[e] & (inc[e] * maxv)) != inc[e] * maxv;
            prev[e] = atomicCAS(&lds[widx[e]], old[e], old[e] + (live[e - g4] ? inc[e] : 0u));
        }
#pragma unroll
        for (int e = g4; e < g4 + 4; ++e) {
            const bool won = prev[e] == old[e];
            fresh += (live[e - g4] && won && (old[e] & (inc[e] * maxv)) == 0) ? 1u : 0u;
            retry |= (live[e - g4] && !won) ? (1u << e) : 0u;
        }
    }
    if (retry) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!(retry & (1u << e))) continue;
            uint32_t cur_word = prev[e];
            for (;;) {
                const uint32_t field = cur_word & (inc[e] * maxv);
                if (field == inc[e] * maxv) break;
                const uint32_t seen = atomicCAS(&lds[widx[e]], cur_word, cur_word + inc[e]);
                if (seen == cur_word) { fresh += field == 0 ? 1u : 0u; break; }
                cur_word = seen;
            }
        }
    }
    return fresh;
}

// Four weighted saturating adds (fine items offset | increment << 16), same structure as lds_inc8: all words
// read, all compare-and-swaps issued back to back, only lost races loop.  min(max, x + c) applied once equals c
// single increments, so the tables come out bit-identical to the unweighted path.
template <int STORAGE>
__device__ __forceinline__ uint32_t lds_addw4(uint32_t *lds, const uint32_t (&w)[4], uint32_t n)
{
    uint32_t fresh = 0;
    if (STORAGE == ST_BIT) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t off = w[e] & 0xffffu;
            if ((uint32_t)e < n) {
                const uint32_t bit = 1u << (off & 31);
                fresh += (atomicOr(&lds[off >> 5], bit) & bit) == 0 ? 1u : 0u;
            }
        }
        return fresh;
    }
    constexpr uint32_t maxv = STORAGE == ST_BYTE ? 255u : 15u;
    uint32_t widx[4], shift[4], add[4], old[4], prev[4];
    bool live[4];
    uint32_t retry = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t off = w[e] & 0xffffu;
        add[e] = w[e] >> 16;
        if (STORAGE == ST_BYTE) { widx[e] = off >> 2; shift[e] = (off & 3u) << 3; }
        else { widx[e] = off >> 3; shift[e] = (((off >> 1) & 3u) << 3) + ((off & 1u) ? 0u : 4u); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) old[e] = lds[widx[e]];               // absent items (e >= n) read a harmless word
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t field = (old[e] >> shift[e]) & maxv;
        const uint32_t nf = min(maxv, field + add[e]);
        live[e] = (uint32_t)e < n && nf != field;
        prev[e] = atomicCAS(&lds[widx[e]], old[e], old[e] + (live[e] ? (nf - field) << shift[e] : 0u));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool won = prev[e] == old[e];
        fresh += (live[e] && won && ((old[e] >> shift[e]) & maxv) == 0) ? 1u : 0u;
        retry |= (live[e] && !won) ? (1u << e) : 0u;
    }
    if (retry) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!(retry & (1u << e))) continue;
            uint32_t cur_word = prev[e];
            for (;;) {
                const uint32_t field = (cur_word >> shift[e]) & maxv;
                if (field == maxv) break;
                const uint32_t nf = min(maxv, field + add[e]);
                const uint32_t seen = atomicCAS(&lds[widx[e]], cur_word, cur_word + ((nf - field) << shift[e]));
                if (seen == cur_word) { fresh += field == 0 ? 1u : 0u; break; }
                cur_word = seen;
            }
        }
    }
    return fresh;
}

// A workgroup lives for a few dependent round trips to HBM and little else (PMC: 73 % of its wave cycles parked), and a
// CU holds two of them, so the kernel's time is (round trips per workgroup) x (latency under load): the chain is kept
// short.  Table sizes and pointers come by value; the segment counts, the overflow flag and -- unless the tables are
// zero by decree -- the slice itself are requested together; every thread then requests ALL its item vectors at once
// (up to MAXV; slices with more fall back to the pipelined loop for the rest) before it applies the first.
template <int STORAGE, bool W, int SBITS>
__global__ __launch_bounds__(SBITS == BIN_SLICE_BITS_W ? BIN_CW_THREADS : BIN_C_THREADS) void k_bin_apply(const SketchDev *__restrict__ sk, BinGeom g)
{
    typedef typename std::conditional<W, uint32_t, uint16_t>::type Item;
    constexpr uint32_t VEC = 16u / sizeof(Item);                  // items per 16-byte vector
    constexpr uint32_t SB = SBITS, SLICE = 1u << SB;
    constexpr uint32_t THREADS = SBITS == BIN_SLICE_BITS_W ? BIN_CW_THREADS : BIN_C_THREADS;
    constexpr int MAXV = W ? 4 : 3;
    __shared__ __attribute__((aligned(16))) uint32_t lds[SLICE / 4];   // one slice: 65536 (32768 weighted) counters of <= 8 bits
    const int t = blockIdx.y;
    const uint32_t slice = blockIdx.x;
    if (slice >= g.nslices[t]) return;
    const uint32_t c = slice / (uint32_t)g.F, fidx = slice % (uint32_t)g.F;
    const uint64_t stream = ((uint64_t)t * g.C + c) * g.F + fidx;
    constexpr int storage = STORAGE;
    const uint64_t bin0 = (uint64_t)slice << SB;
    const uint64_t left = g.tsize[t] - bin0, nb = left < SLICE ? left : SLICE;
    // byte range of the slice inside the table (the allocation is padded to 16 B)
    const uint64_t byte0 = storage == ST_BYTE ? bin0 : (storage == ST_NIBBLE ? bin0 >> 1 : bin0 >> 3);
    const uint64_t nbytes = storage == ST_BYTE ? nb : (storage == ST_NIBBLE ? (nb + 1) / 2 : (nb + 7) / 8);
    const uint32_t nvec = (uint32_t)((nbytes + 15) / 16);
    uint4 *tab = (uint4 *)(g.ttab[t] + byte0);
    uint4 *l4 = (uint4 *)lds;
    // the slice's items sit in nwgB private segments of cap2 slots (cap2 % 64 == 0: 128-B aligned, so a
    // 16-byte vector never leaves its segment).  Their vectors are enumerated compactly through a
    // prefix sum over the segments (LDS), so every thread has work whatever the segment fill levels.
    __shared__ uint32_t seg_cnt[BIN_MAX_SEG], vpre[BIN_MAX_SEG];
    __shared__ uint16_t segof[MAXV * THREADS];       // the segment every one of the first MAXV x THREADS vectors lies in (below)
    __shared__ uint32_t wsum[THREADS / 64];
    __shared__ uint32_t total_vec_sh, flag_sh, fresh_sh;
    const Item *items = (const Item *)g.gbuf2 + stream * g.nwgB * g.cap2;
    const uint32_t *counts = g.gcnt2 + stream * g.nwgB;
    {
        uint32_t myc = 0;
        if (threadIdx.x < g.nwgB) myc = (g.dbg & 16u) ? 4u : counts[threadIdx.x];
        unsigned long long flag = 0;
        if (threadIdx.x == THREADS - 1 && !(g.dbg & 8u)) flag = g.ctr[1];          // overflow flag: leave the tables untouched for the fallback
        if (!g.zero_tables)
            for (uint32_t j = threadIdx.x; j < nvec; j += THREADS) l4[j] = tab[j];
        if (threadIdx.x < g.nwgB) seg_cnt[threadIdx.x] = myc;
        const uint32_t myv = (myc + VEC - 1) / VEC;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t incl = myv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        if (threadIdx.x == THREADS - 1) { flag_sh = flag != 0 ? 1u : 0u; fresh_sh = 0; }
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (threadIdx.x < g.nwgB) vpre[threadIdx.x] = before + incl - myv;
        if (threadIdx.x == THREADS - 1) total_vec_sh = before + incl;
        __syncthreads();
        // which segment a vector lies in, looked up instead of searched for: every wave writes the number of a few segments over
        // their vectors' entries (a thread's MAXV fetches used to be MAXV binary searches over the ~30 segments, five dependent LDS
        // reads and as many branches each: a fifth of the kernel's instructions)
        for (uint32_t sgi = (uint32_t)wave; sgi < g.nwgB; sgi += THREADS / 64) {
            const uint32_t first = vpre[sgi], nv = (seg_cnt[sgi] + VEC - 1) / VEC;
            for (uint32_t i = (uint32_t)lane; i < nv && first + i < (uint32_t)MAXV * THREADS; i += 64) segof[first + i] = (uint16_t)sgi;
        }
        __syncthreads();
    }
    if (flag_sh) return;
    const uint32_t total_vec = total_vec_sh;
    if (total_vec == 0) {                                          // untouched slice: no table traffic at all ...
        if (g.zero_tables)                                         // ... unless this pass is also the table's zeroing
            for (uint32_t j = threadIdx.x; j < nvec; j += THREADS) tab[j] = make_uint4(0, 0, 0, 0);
        return;
    }
    // a vector: its items (always loaded whole) and how many of them are real
    struct Vec { uint4 q; uint32_t n; };
    auto fetch = [&](uint32_t v) {
        Vec r;
        r.q = make_uint4(0, 0, 0, 0); r.n = 0;
        if (v >= total_vec) return r;
        if (g.dbg & 32u) { r.n = VEC; r.q = *(const uint4 *)(items + (uint64_t)v * VEC); return r; }     // same bytes, one contiguous run
        uint32_t lo = 0, hi = g.nwgB;      // largest segment with vpre[seg] <= v (empty segments share their successor's prefix)
        if (v < (uint32_t)MAXV * THREADS) {
            lo = segof[v];
        } else {
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (vpre[mid] <= v) lo = mid; else hi = mid;
            }
        }
        const uint32_t j0 = (v - vpre[lo]) * VEC;
        r.n = min(VEC, seg_cnt[lo] - j0);
        r.q = *(const uint4 *)(items + (uint64_t)lo * g.cap2 + j0);
        return r;
    };
    Vec first[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) first[i] = (g.dbg & 2u) ? Vec{make_uint4(0, 0, 0, 0), 0u} : fetch(threadIdx.x + (uint32_t)i * THREADS);
    if (g.zero_tables)
        for (uint32_t j = threadIdx.x; j < nvec; j += THREADS) l4[j] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint32_t fresh = 0;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const uint32_t w[4] = {first[i].q.x, first[i].q.y, first[i].q.z, first[i].q.w};
        if (g.dbg & 1u) { fresh += w[0] ^ w[1] ^ w[2] ^ w[3]; continue; }
        // (an absent vector would still issue its compare-and-swaps, all on word 0: same-address LDS atomics serialise)
        if (first[i].n) fresh += W ? lds_addw4<STORAGE>(lds, w, first[i].n) : lds_inc8<STORAGE>(lds, w, first[i].n);
    }
    if (total_vec > (uint32_t)MAXV * THREADS && !(g.dbg & 3u)) {
        // a fuller slice: the rest through a software pipeline, two vectors ahead
        const uint32_t v_start = (uint32_t)MAXV * THREADS + threadIdx.x;
        Vec v0 = fetch(v_start), v1 = fetch(v_start + THREADS);
        for (uint32_t v = v_start; v < total_vec; v += THREADS) {
            const Vec v2 = fetch(v + 2 * THREADS);
            const uint32_t w[4] = {v0.q.x, v0.q.y, v0.q.z, v0.q.w};
            fresh += W ? lds_addw4<STORAGE>(lds, w, v0.n) : lds_inc8<STORAGE>(lds, w, v0.n);
            v0 = v1; v1 = v2;
        }
    }
    // table 0's newly occupied bins: ONE update of the device-wide counter per workgroup.  (It is one address for the whole
    // chip and takes ~90 updates per microsecond however they are issued: an update per wave -- 16 x 7630 of them for a
    // 2 GB sketch -- held this kernel at 1.4 ms whatever else it did.)
    if (t == 0) {
        const uint32_t tot = (uint32_t)wave_sum_u64(fresh);
        if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&fresh_sh, tot);
    }
    __syncthreads();
    if (!(g.dbg & 4u))
        for (uint32_t j = threadIdx.x; j < nvec; j += THREADS) tab[j] = l4[j];
    if (t == 0 && threadIdx.x == 0 && fresh_sh) atomicAdd(&g.ctr[3], (unsigned long long)fresh_sh);
}

// saturating add of `weight` to one bin with global atomics; true if the bin was zero before
__device__ __forceinline__ bool table_add_bin(const SketchDev *s, int t, uint64_t bin, uint32_t weight)
{
    uint8_t *tab = s->tab[t];
    if (s->storage == ST_BIT) {
        const uint32_t bit = 1u << (bin & 31);
        return (atomicOr((uint32_t *)tab + (bin >> 5), bit) & bit) == 0;
    }
    uint32_t *w;
    uint32_t shift, maxv;
    if (s->storage == ST_BYTE) {
        w = (uint32_t *)(tab + (bin & ~3ull)); shift = (uint32_t)(bin & 3) * 8u; maxv = 255u;
    } else {
        const uint64_t byte = bin >> 1;
        w = (uint32_t *)(tab + (byte & ~3ull)); shift = (uint32_t)(byte & 3) * 8u + ((bin & 1) ? 0u : 4u); maxv = 15u;
    }
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const uint32_t cur = (old >> shift) & maxv;
        if (cur == maxv) return false;
        const uint32_t nf = min(maxv, cur + weight);
        const uint32_t prev = atomicCAS(w, old, old + ((nf - cur) << shift));
        if (prev == old) return cur == 0;
        old = prev;
    }
}

__global__ void k_bin_spill(const SketchDev *__restrict__ sk, BinGeom g)
{
    if (g.ctr[1] != 0) return;
    unsigned long long n = g.ctr[0];
    if (n > g.spill_cap) n = g.spill_cap;
    uint64_t fresh = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long e = g.spill[i];
        const int t = (int)((e >> 32) & 0xffu);
        const bool was_zero = table_add_bin(sk, t, e & 0xffffffffull, (uint32_t)((e >> 40) & 0xffu) + 1u);
        fresh += (was_zero && t == 0) ? 1 : 0;
    }
    fresh = wave_sum_u64(fresh);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&g.ctr[3], (unsigned long long)fresh);
}

// one grow-only arena per stream, so host threads counting different samples do not share buffers
std::map<hipStream_t, KvArena> g_scratch;
std::mutex g_scratch_mu;

KvArena &scratch_for(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    return g_scratch[kv_stream_key(st)];
}
}
void kv_bin_scratch_release()
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    for (auto &kv : g_scratch) kv.second.release();
}
namespace {

template <typename K>
void ensure_dynamic_lds(K kernel, size_t bytes) { kv_ensure_dynamic_lds((const void *)kernel, bytes); }

}  // namespace

int kv_device_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 256;
        kv_thread_device();
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus = n;
    }
    return cus;
}

// linear-counting estimate of the distinct k-mers behind an occupancy of table 0
double kv_estimate_distinct(uint64_t occupied, uint64_t size)
{
    if (occupied >= size) return (double)size * std::log((double)size);
    return -(double)size * std::log1p(-(double)occupied / (double)size);
}

// can stage A hash this batch from its 2-bit form (k_bin_hash_2bit)?  Reads of one length, murmur kinds, 16 <= k <= 64.  KV_BIN_2BIT=0: never
bool kv_bin_two_bit(const kv_sketch *s, const kv_reads *reads)
{
    const char *e = kv_knob("KV_BIN_2BIT");
    if (e && atoi(e) == 0) return false;
    return reads && reads->uni_len != 0 && reads->uni_per_tile != 0 && s->h.hashfam == HF_MURMUR && s->h.ksize >= SKM_MIN_K && s->h.ksize <= SKM_MAX_K &&
           reads->uni_len >= (uint32_t)s->h.ksize && reads->n_tiles > 0;
}

bool kv_binned_eligible(const kv_sketch *s, const kv_reads *reads, uint64_t n_kmers, int nbands)
{
    // reads == nullptr: the items come from a hash list (kv_consume_hashes)
    const char *force = kv_knob("KV_COUNT_PATH");
    if (force && strcmp(force, "atomic") == 0) return false;
    if (s->h.ntables > BIN_MAX_T) return false;
    uint64_t pmin = UINT64_MAX, pmax = 0;
    for (int t = 0; t < s->h.ntables; ++t) { pmin = std::min(pmin, s->h.size[t]); pmax = std::max(pmax, s->h.size[t]); }
    if (pmax > (uint64_t)BIN_C * BIN_MAX_F * 65536ull) return false;      // <= 2^31 bins per table
    if (force && (strcmp(force, "binned") == 0 || strcmp(force, "skm") == 0)) return reads ? reads->n_tiles > 0 : n_kmers > 0;
    const uint64_t expected = nbands > 0 ? n_kmers / (uint64_t)nbands : n_kmers;
    // worth it once the batch touches the tables about as densely as streaming them costs.  With the 2-bit stage A the hashing costs
    // half as much, and what is left to compare is T global compare-and-swaps per k-mer that reaches the tables (~27 G/s device-wide)
    // against streaming the tables once (read + write at ~4.7 TB/s) plus the split: even at one k-mer per 32 bins -- config 4's
    // 0.6x batches under 8-fold banding: 164 M k-mers into 2 G bins -- the atomics cost 24 ms and the streamed tables 3.4 + 7
    // (a hash list has no hashing left to pay at all: 164 M hashes into a band's 2 G-bin tables -- config 4 with every band resident,
    // scratch/cfg4_whole.py -- took 30.9 ms through the atomics)
    const uint64_t density = !reads || kv_bin_two_bit(s, reads) ? 32 : 8;
    return pmin >= (1ull << 20) && expected >= (1ull << 22) && expected * density >= pmax;
}

int kv_bin_plan(kv_sketch *s, uint64_t n_items_max, int nbands, bool use_mask, uint64_t work_units, uint32_t lds_front,
                uint32_t nwgA_fixed, bool weighted, BinPlan *plan)
{
    hipStream_t st = kv_stream();
    KvArena &scratch = scratch_for(st);
    BinGeom &g = plan->g;
    memset(&g, 0, sizeof(g));
    plan->weighted = weighted;
    g.zero_tables = s->lazy_zero ? 1 : 0;
    g.T = s->h.ntables;
    for (int t = 0; t < g.T && t < BIN_MAX_T; ++t) { g.tsize[t] = s->h.size[t]; g.ttab[t] = s->h.tab[t]; }
    g.tile_lds = lds_front;
    g.dbg = kv_knob("KV_BIN_DEBUG") ? (uint32_t)atoi(kv_knob("KV_BIN_DEBUG")) : 0u;
    uint64_t pmin = UINT64_MAX, pmax = 0;
    for (int t = 0; t < g.T; ++t) { pmin = std::min(pmin, s->h.size[t]); pmax = std::max(pmax, s->h.size[t]); }
    // 32768-bin slices for weighted items (KV_BIN_SLICE15=1) were measured at config 2: stage C 2.20 instead of 2.25 ms,
    // stage B 1.26-1.36 instead of 1.0 ms -- stage C is not held back by the overlap of its phases; the default stays 65536
    g.sbits = weighted && kv_knob("KV_BIN_SLICE15") && atoi(kv_knob("KV_BIN_SLICE15")) && ((pmax + 32767) >> 15) <= 64ull * 384ull
                  ? BIN_SLICE_BITS_W : BIN_SLICE_BITS;
    uint32_t maxsl = 1;
    for (int t = 0; t < g.T; ++t) {
        g.nslices[t] = (uint32_t)((s->h.size[t] + (1ull << g.sbits) - 1) >> g.sbits);
        maxsl = std::max(maxsl, g.nslices[t]);
    }
    plan->maxsl = maxsl;
    // Coarse buckets per table: as few as keep stage B's fan-out F at <= 384 slices per bucket (its u16 rings then
    // fit three workgroups per CU), between 4 and 32 with 512-thread stage-A workgroups; 64 buckets / 1024 threads
    // beyond 2^30 bins.  Fewer buckets = fewer distinct lines per stage-A store instruction (that stage is bound by
    // L2 write requests): measured per 525 M k-mers into 5e8-bin tables, A/B/C = 10.5/5.0/4.0 ms with 32 buckets,
    // 8.7/4.9/4.0 with 20, 8.3/6.1/4.0 with 16 (F = 478: rings too big for three workgroups).
    // (weighted items: up to 64 buckets with the 512-thread front end -- its cursors are sized by T * C -- and F <= 1024)
    int cmax = weighted && g.sbits == BIN_SLICE_BITS_W ? (int)std::min<uint32_t>(BIN_C, std::max<uint32_t>(4u, (maxsl + 383u) / 384u))
                              : (maxsl <= 32u * BIN_MAX_F ? (int)std::min<uint32_t>(32u, std::max<uint32_t>(4u, (maxsl + 383u) / 384u)) : BIN_C);
    if (const char *e = kv_knob("KV_BIN_C")) cmax = std::max(1, std::min<int>(BIN_C, atoi(e)));      // experiments: coarse buckets per table
    plan->cmax = cmax;
    g.F = (int)((maxsl + (uint32_t)cmax - 1) / (uint32_t)cmax);
    g.C = (int)((maxsl + (uint32_t)g.F - 1) / (uint32_t)g.F);
    g.recipF = g.F == 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)g.F + 1);   // F == 1: kernels take slice as is
    auto ring_for = [&](uint32_t streams, uint32_t budget) {
        uint32_t r = weighted && budget == BIN_B_BUDGET ? 32u : BIN_RING_MIN;
        while (r * 2 <= BIN_RING_MAX && (uint64_t)r * 2 * streams <= budget) r *= 2;
        return r;
    };
    const uint32_t budgetA = cmax <= 32 ? 8192u : 16384u;
    g.ringA = ring_for((uint32_t)(g.T * g.C), budgetA);
    g.ringB = ring_for((uint32_t)g.F, BIN_B_BUDGET);       // budget in entries: u32 rings get half the entries per byte
    if (weighted) while (g.ringB > 32u && (uint64_t)g.ringB * g.F * 4 > 2u * BIN_B_BUDGET) g.ringB /= 2;
    const int cus = kv_device_cus();
    const double expected = (double)(nbands > 0 && !use_mask ? n_items_max / (uint64_t)nbands + 1 : n_items_max);
    const uint64_t ns = (uint64_t)g.T * g.C;
    // writers: stage A = persistent workgroups (tiles dealt round-robin, so their loads are equal
    // to within one tile); stage B = nwgB workgroups per coarse bucket, ~256 k items each
    const uint32_t threadsA = cmax <= 32 ? 512u : 1024u;
    plan->threadsA = threadsA;
    if (nwgA_fixed) g.nwgA = nwgA_fixed;
    else g.nwgA = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(work_units, 1), (cmax <= 32 ? 3u : 1u) * (uint32_t)cus);
    {
        const uint64_t avg = (std::max<uint64_t>(work_units, 1) + g.nwgA - 1) / g.nwgA;
        g.quotaA = (uint32_t)std::min<uint64_t>(avg + avg / 2 + 1, 0xffffffffull);   // matches the 1.5x slack of cap1
    }
    const double slice_bins = (double)(1u << g.sbits);
    const double per_bucket = expected * std::min(1.0, (double)g.F * slice_bins / (double)pmin);
    const double per_slice = expected * std::min(1.0, slice_bins / (double)pmin);
    g.nwgB = (uint32_t)std::max(1.0, std::min((double)std::min<uint32_t>(g.nwgA, 512), std::ceil(per_bucket / 524288.0)));
    // a stage-B workgroup drains its segments one after the other, a dependent round trip or two each: with few items per
    // segment (a shard of a multi-GPU run, a small batch) that chain is the stage's whole time -- at most 32 segments each
    g.nwgB = std::max<uint32_t>(g.nwgB, std::min<uint32_t>(std::min<uint32_t>(g.nwgA, 512), (g.nwgA + 31u) / 32u));
    const double m1 = per_bucket / g.nwgA, m2 = per_slice / g.nwgB;
    g.cap1 = kv_round_up((uint64_t)(m1 * 1.5 + 8.0 * std::sqrt(m1)) + 2048, 64);   // tiles are dealt dynamically: shares are uneven
    g.cap2 = kv_round_up((uint64_t)(m2 * 1.05 + 8.0 * std::sqrt(m2)) + 64, 64);
    g.spill_cap = std::max<uint64_t>(1u << 22, (uint64_t)(expected * g.T / 8));
    const size_t fine_bytes = weighted ? 4 : 2;
    const size_t b_buf1 = kv_round_up(ns * g.nwgA * g.cap1 * 4, 256), b_buf2 = kv_round_up(ns * g.F * g.nwgB * g.cap2 * fine_bytes, 256);
    const size_t b_cnt1 = kv_round_up(ns * g.nwgA * 4, 256), b_cnt2 = kv_round_up(ns * g.F * g.nwgB * 4, 256);
    const size_t b_spill = kv_round_up(g.spill_cap * 8, 256), b_ctr = 256;
    // (no memory for the staging buffers is a CAPACITY answer: the caller has a path that needs less -- in the end the atomic kernel, which needs none)
    if (scratch.need(b_buf1 + b_buf2 + b_cnt1 + b_cnt2 + b_spill + b_ctr) != hipSuccess) {
        (void)hipGetLastError();
        kv_set_error("no device memory for the partitioned count's staging buffers (%.1f GB)", (double)(b_buf1 + b_buf2 + b_cnt1 + b_cnt2 + b_spill + b_ctr) / 1e9);
        return KV_ERR_CAPACITY;
    }
    unsigned char *base = (unsigned char *)scratch.p;
    g.gbuf1 = (uint32_t *)base; base += b_buf1;
    g.gbuf2 = (uint16_t *)base; base += b_buf2;
    g.gcnt1 = (uint32_t *)base; base += b_cnt1;
    g.gcnt2 = (uint32_t *)base; base += b_cnt2;
    g.spill = (unsigned long long *)base; base += b_spill;
    g.ctr = (unsigned long long *)base;
    {
        bool ok = g.T == 4 && ns * g.nwgA * g.cap1 < (1ull << 32) && !(kv_knob("KV_BIN_FAST4") && atoi(kv_knob("KV_BIN_FAST4")) == 0);
        for (int t = 0; t < g.T && t < BIN_MAX_T; ++t) {
            g.tmagic[t] = kv_fastmod_magic(s->h.size[t]);
            ok = ok && kv_fastmod_fp(s->h.size[t]) && s->h.size[t] < (1ull << 31);
        }
        g.fast4 = ok ? 1 : 0;
    }
    KV_HIP(hipMemsetAsync(g.ctr, 0, b_ctr, st));   // every segment count is written by its owner: no other memset
    return KV_OK;
}

int kv_bin_finish(kv_sketch *s, BinPlan &plan, bool added_from_ctr, uint64_t n_added_fixed, uint64_t *n_added)
{
    hipStream_t st = kv_stream();
    BinGeom &g = plan.g;
    const uint64_t ns = (uint64_t)g.T * g.C;
    {
        KvProfScope prof(plan.weighted ? "k_bin_split_w" : "k_bin_split");
        const size_t isz = plan.weighted ? 4 : 2;
        const size_t lds = (((size_t)g.F * g.ringB * isz + 15) & ~(size_t)15) + (size_t)g.F * 2 * 4;
        if (plan.weighted && g.sbits == BIN_SLICE_BITS_W) {
            ensure_dynamic_lds((k_bin_split<true, BIN_SLICE_BITS_W>), lds);
            hipLaunchKernelGGL((k_bin_split<true, BIN_SLICE_BITS_W>), dim3(g.nwgB, (unsigned)ns), dim3(BIN_B_THREADS), lds, st, g);
        } else if (plan.weighted && (g.nwgA + g.nwgB - 1) / g.nwgB <= BIN_BS_MAXSEG && (uint32_t)g.F <= BIN_B_THREADS &&
                   kv_knob("KV_BIN_SPLIT") && !strcmp(kv_knob("KV_BIN_SPLIT"), "sorted")) {
            // weighted items, 65536-bin slices, ranked and laid out in LDS, copied out in runs: only by name (KV_BIN_SPLIT=sorted) -- it gives
            // the same segments and measured SLOWER than the ring kernel, 3.98 against 3.24 ms per step of config 2 (profiles/README.md)
            const size_t lds2 = (size_t)BIN_BS_CHUNK * 4;
            ensure_dynamic_lds((k_bin_split_sorted<BIN_SLICE_BITS>), lds2);
            hipLaunchKernelGGL((k_bin_split_sorted<BIN_SLICE_BITS>), dim3(g.nwgB, (unsigned)ns), dim3(BIN_B_THREADS), lds2, st, g);
        } else if (plan.weighted) {
            ensure_dynamic_lds((k_bin_split<true, BIN_SLICE_BITS>), lds);
            hipLaunchKernelGGL((k_bin_split<true, BIN_SLICE_BITS>), dim3(g.nwgB, (unsigned)ns), dim3(BIN_B_THREADS), lds, st, g);
        } else {
            ensure_dynamic_lds((k_bin_split<false, BIN_SLICE_BITS>), lds);
            hipLaunchKernelGGL((k_bin_split<false, BIN_SLICE_BITS>), dim3(g.nwgB, (unsigned)ns), dim3(BIN_B_THREADS), lds, st, g);
        }
    }
    {
        KvProfScope prof(plan.weighted ? "k_bin_apply_w" : "k_bin_apply");
        const dim3 gridC(plan.maxsl, (unsigned)g.T);
        const SketchDev *d = (const SketchDev *)s->d_desc;
#define KV_LAUNCH_APPLY(ST_, W_, SB_) \
        hipLaunchKernelGGL((k_bin_apply<ST_, W_, SB_>), gridC, dim3(SB_ == BIN_SLICE_BITS_W ? BIN_CW_THREADS : BIN_C_THREADS), 0, st, d, g)
#define KV_LAUNCH_APPLY_ST(W_, SB_)                                               \
        do {                                                                       \
            if (s->h.storage == ST_BYTE) KV_LAUNCH_APPLY(ST_BYTE, W_, SB_);        \
            else if (s->h.storage == ST_NIBBLE) KV_LAUNCH_APPLY(ST_NIBBLE, W_, SB_); \
            else KV_LAUNCH_APPLY(ST_BIT, W_, SB_);                                 \
        } while (0)
        if (plan.weighted && g.sbits == BIN_SLICE_BITS_W) KV_LAUNCH_APPLY_ST(true, BIN_SLICE_BITS_W);
        else if (plan.weighted) KV_LAUNCH_APPLY_ST(true, BIN_SLICE_BITS);
        else KV_LAUNCH_APPLY_ST(false, BIN_SLICE_BITS);
#undef KV_LAUNCH_APPLY_ST
#undef KV_LAUNCH_APPLY
    }
    {
        KvProfScope prof("k_bin_spill");   // usually a handful of items; the kernels check the overflow flag themselves
        hipLaunchKernelGGL(k_bin_spill, dim3(256), dim3(256), 0, st, (const SketchDev *)s->d_desc, g);
    }
    KV_HIP(hipGetLastError());
    unsigned long long ctr[4] = {0, 0, 0, 0};
    KV_HIP(hipMemcpyAsync(ctr, g.ctr, sizeof(ctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    if (ctr[1] != 0) {
        kv_set_error("partitioned count: spill list overflow (%llu items)", ctr[0]);
        return KV_ERR_CAPACITY;
    }
    *n_added = added_from_ctr ? ctr[2] : n_added_fixed;
    s->lazy_zero = false;          // stage C has written every slice
    // exact occupancy bookkeeping; n_unique_kmers as a linear-counting estimate
    if (s->occ_dirty) {
        int rc = kv_sketch_refresh_occupancy(s);   // recount includes this batch
        if (rc != KV_OK) return rc;
        s->n_unique = (uint64_t)(kv_estimate_distinct(s->n_occupied, s->h.size[0]) + 0.5);
    } else {
        const double before = kv_estimate_distinct(s->n_occupied, s->h.size[0]);
        s->n_occupied += ctr[3];
        const double after = kv_estimate_distinct(s->n_occupied, s->h.size[0]);
        s->n_unique += (uint64_t)(after - before + 0.5);
    }
    return KV_OK;
}

// returns KV_OK, or KV_ERR_CAPACITY when the spill list overflowed (tables untouched: caller falls back)
int kv_consume_binned(kv_sketch *s, const kv_reads *reads, const uint64_t *d_list, uint32_t list_stride,
                      const ConsumeFilter &filter, const kv_sketch *mask, uint64_t n_kmers, int nbands, uint64_t *n_added,
                      bool weighted_list)
{
    // source: the packed reads (hash in stage A) or, when reads == nullptr, n_kmers hashes at d_list[i * list_stride]
    hipStream_t st = kv_stream();
    BinPlan plan;
    const bool two_bit = reads && kv_bin_two_bit(s, reads);
    const uint32_t n_units2 = two_bit ? (reads->n_tiles + BIN2_TILES - 1u) / BIN2_TILES : 0u;
    {
        // the stage-A workgroup size depends on the geometry, which depends only on the table sizes: probe it first
        uint32_t maxsl = 1;
        for (int t = 0; t < s->h.ntables; ++t) maxsl = std::max(maxsl, (uint32_t)((s->h.size[t] + 65535) >> 16));
        const uint32_t threadsA = maxsl <= 32u * BIN_MAX_F ? 512u : 1024u;
        const uint64_t work_units = two_bit ? n_units2 : reads ? reads->n_tiles
                                          : (n_kmers + (uint64_t)threadsA * BIN_LIST_ROUNDS - 1) / ((uint64_t)threadsA * BIN_LIST_ROUNDS);
        const int rc = kv_bin_plan(s, n_kmers, nbands, filter.use_mask != 0, work_units, reads && !two_bit ? reads->tile_lds_bytes : 0u, 0u, weighted_list, &plan);
        if (rc != KV_OK) return rc;
    }
    BinGeom &g = plan.g;
    const int cmax = plan.cmax;
    const uint64_t ns = (uint64_t)g.T * g.C;
    const SketchDev *d_mask = mask ? mask->d_desc : nullptr;
    if (two_bit) {
        KvProfScope prof("k_bin_hash_2bit");
        const SketchDev *d = (const SketchDev *)s->d_desc;
        const bool kw2 = s->h.ksize > 32;
        const bool k31 = !kw2 && s->h.ksize == 31 && !kv_knob("KV_SKM_ANY_K");
        if (cmax <= 32 && k31) hipLaunchKernelGGL((k_bin_hash_2bit<512, 1, 31>), dim3(g.nwgA), dim3(512), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
        else if (k31) hipLaunchKernelGGL((k_bin_hash_2bit<1024, 1, 31>), dim3(g.nwgA), dim3(1024), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
        else if (cmax <= 32 && !kw2) hipLaunchKernelGGL((k_bin_hash_2bit<512, 1>), dim3(g.nwgA), dim3(512), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
        else if (cmax <= 32) hipLaunchKernelGGL((k_bin_hash_2bit<512, 2>), dim3(g.nwgA), dim3(512), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
        else if (!kw2) hipLaunchKernelGGL((k_bin_hash_2bit<1024, 1>), dim3(g.nwgA), dim3(1024), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
        else hipLaunchKernelGGL((k_bin_hash_2bit<1024, 2>), dim3(g.nwgA), dim3(1024), 0, st, reads_dev(reads), n_units2, d, d_mask, filter, g);
    } else if (reads) {
        // default: direct stores through LDS cursors (k_bin_hash_direct); KV_BIN_DIRECT=0 selects the LDS-ring
        // variant, which measured ~10% slower on this stage (profiles/README.md)
        const bool direct = !(kv_knob("KV_BIN_DIRECT") && atoi(kv_knob("KV_BIN_DIRECT")) == 0);
        KvProfScope prof(direct ? "k_bin_hash_direct" : "k_bin_hash");
        const size_t lds = (size_t)g.tile_lds + ns * g.ringA * 4 + ns * 8;
        const unsigned grid = g.nwgA;
        const int k = s->h.ksize;
        const int nw = (s->h.hashfam == HF_MURMUR && !kv_knob("KV_NO_ROLL")) ? (k <= 32 ? 8 : (k <= 64 ? 16 : 0)) : 0;
#define KV_LAUNCH_BIN_HASH(THREADS_, NW_)                                                                         \
        do {                                                                                                      \
            ensure_dynamic_lds(k_bin_hash<THREADS_, NW_>, lds);                                                   \
            hipLaunchKernelGGL((k_bin_hash<THREADS_, NW_>), dim3(grid), dim3(THREADS_), lds, st, reads_dev(reads), \
                               reads->n_tiles, (const SketchDev *)s->d_desc, d_mask, filter, g);                  \
        } while (0)
#define KV_LAUNCH_BIN_DIRECT(THREADS_, NW_)                                                                       \
        do {                                                                                                      \
            ensure_dynamic_lds(k_bin_hash_direct<THREADS_, NW_>, (size_t)g.tile_lds);                             \
            hipLaunchKernelGGL((k_bin_hash_direct<THREADS_, NW_>), dim3(grid), dim3(THREADS_), g.tile_lds, st,    \
                               reads_dev(reads), reads->n_tiles, (const SketchDev *)s->d_desc, d_mask, filter, g); \
        } while (0)
        if (direct && cmax <= 32) {
            if (nw == 8) KV_LAUNCH_BIN_DIRECT(512, 8);
            else if (nw == 16) KV_LAUNCH_BIN_DIRECT(512, 16);
            else KV_LAUNCH_BIN_DIRECT(512, 0);
        } else if (direct) {
            if (nw == 8) KV_LAUNCH_BIN_DIRECT(1024, 8);
            else if (nw == 16) KV_LAUNCH_BIN_DIRECT(1024, 16);
            else KV_LAUNCH_BIN_DIRECT(1024, 0);
        } else if (cmax <= 32) {
            if (nw == 8) KV_LAUNCH_BIN_HASH(512, 8);
            else if (nw == 16) KV_LAUNCH_BIN_HASH(512, 16);
            else KV_LAUNCH_BIN_HASH(512, 0);
        } else {
            if (nw == 8) KV_LAUNCH_BIN_HASH(1024, 8);
            else if (nw == 16) KV_LAUNCH_BIN_HASH(1024, 16);
            else KV_LAUNCH_BIN_HASH(1024, 0);
        }
#undef KV_LAUNCH_BIN_DIRECT
#undef KV_LAUNCH_BIN_HASH
    } else {
        KvProfScope prof(weighted_list ? "k_bin_list_w" : "k_bin_list");
        const SketchDev *d = (const SketchDev *)s->d_desc;
        if (cmax <= 32 && weighted_list) hipLaunchKernelGGL((k_bin_list<512, true>), dim3(g.nwgA), dim3(512), 0, st, d_list, n_kmers, list_stride, d, g);
        else if (cmax <= 32) hipLaunchKernelGGL((k_bin_list<512, false>), dim3(g.nwgA), dim3(512), 0, st, d_list, n_kmers, list_stride, d, g);
        else if (weighted_list) hipLaunchKernelGGL((k_bin_list<1024, true>), dim3(g.nwgA), dim3(1024), 0, st, d_list, n_kmers, list_stride, d, g);
        else hipLaunchKernelGGL((k_bin_list<1024, false>), dim3(g.nwgA), dim3(1024), 0, st, d_list, n_kmers, list_stride, d, g);
    }
    return kv_bin_finish(s, plan, reads != nullptr || weighted_list, n_kmers, n_added);
}
