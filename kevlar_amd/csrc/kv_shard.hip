// kv_shard.hip -- read-sharded multi-GPU count and scan: hash once, route by band.
//
// kevlar shards a trio by k-mer hash band (docs/banding.rst; kevlar/count.py:62-66): band b owns
// the hash range [b*bs, (b+1)*bs).  Run the reference way, every band's process still parses and
// hashes EVERY read, and so does the one-band-per-GPU layout of bench.py's "banded" mode: measured,
// hashing is ~2/3 of a banded rank's time and eight GPUs give 2x.  With the GPUs of one node on
// xGMI the reads can be sharded instead: each rank hashes 1/N of every sample exactly once and
// sends each hash to the rank that owns its band (one all-to-all, 8 B per k-mer, 16 B for the case
// sample whose hits need their (read, offset) back).  The owner counts what it receives with the
// same partitioned kernels (kv_consume_hashes) and scans the case hashes it received
// (kv_novel_scan_hashes); sketches, bands and results are exactly those of the banded run.
//
//   k_route_hashes  LDS-staged tiles and rolling murmur windows as in k_bin_hash_direct; the destination
//                   of a hash is its band; every workgroup appends to a private segment per destination
//                   through an LDS cursor (direct stores, no barrier), and k_route_scan / k_route_compact /
//                   k_route_tail pack the segments into that destination's contiguous send buffer.
//   kv_hits_from_tagged  radix sort (rocPRIM) of the gathered (tag, abundances) hits back into the
//                   (read, offset) order of the reference's output.
#include <cmath>
#include <cstring>
#include <map>

#include <rocprim/device/device_radix_sort.hpp>

#include "kv_binned.h"
#include "kv_device.h"

namespace {

#define ROUTE_THREADS 512
#define ROUTE_MAX_DEST 16
#define ROUTE_MAX_WG 1024

// Every workgroup owns one private segment per destination and appends to it through a cursor in
// LDS: no global atomic, no barrier and no staging between hashing a k-mer and storing it (the
// write frontier -- workgroups x destinations x one 128-B line -- lives in L2, which merges the
// partial lines).  Tiles are handed out dynamically up to a quota per workgroup, which bounds what a
// segment can receive; anything beyond a segment's capacity (skewed input: one k-mer repeated
// thousands of times lands in one band) goes to a shared overflow list.  k_route_compact then packs
// the segments into the caller's send buffer, destination after destination with no gaps -- the layout an
// all-to-all with split sizes takes as it is.
struct RouteParams {
    HashParams hp;
    int ndest;
    uint32_t nwg, quota;         // workgroups, tiles per workgroup at most
    uint64_t bs;                 // band width UINT64_MAX / ndest (kv_band_bounds)
    uint64_t read_base;          // global index of this shard's first read
    int unit_tags;               // the "tag" of every item is the count 1 (kv_route_distinct's one-item-per-k-mer form)
    uint64_t seg_cap;            // items per private segment
    uint64_t *seg;               // [ndest][nwg][seg_cap] items (1 or 2 words each)
    uint32_t *seg_count;         // [ndest][nwg]
    uint64_t *seg_off;           // [ndest][nwg] position of the segment in the packed output
    uint64_t *ovf;               // overflow items
    uint8_t *ovf_dest;
    uint64_t ovf_cap;
    unsigned long long *ctr;     // [0] tile hand-out, [1] overflow items; per destination d: [2 + d] items in segments,
                                 // [18 + d] overflow items, [34 + d] first item in `out`, [50 + d] cursor of its overflow tail
    uint64_t cap;                // items in the caller's buffer (>= all k-mers of the shard)
    uint64_t *out;               // destination 0's items, then destination 1's, ... back to back
};

template <int NW, bool TAGS>
__global__ __launch_bounds__(ROUTE_THREADS, 6) void k_route_hashes(ReadsDev rd, uint32_t n_tiles, RouteParams p)
{
    __shared__ TileShared sh;
    __shared__ uint32_t cur[ROUTE_MAX_DEST];
    __shared__ uint64_t lo[ROUTE_MAX_DEST];
    __shared__ uint8_t flagged[KV_TILE_MAX_READS];
    __shared__ uint32_t next_tile;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr uint32_t W = TAGS ? 2 : 1;
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)smem;
    if (threadIdx.x < (uint32_t)p.ndest) { cur[threadIdx.x] = 0; lo[threadIdx.x] = p.bs * (uint64_t)threadIdx.x; }
    for (uint32_t done = 0; done < p.quota; ++done) {
        __syncthreads();
        if (threadIdx.x == 0) next_tile = (uint32_t)atomicAdd(&p.ctr[0], 1ull);
        __syncthreads();
        const uint32_t tile = next_tile;
        if (tile >= n_tiles) break;
        uint32_t read0;
        const uint32_t nr = stage_tile(sh, rd, tile, p.hp.k, 0, 0, read0);
        if (TAGS) {
            if (threadIdx.x < nr) flagged[threadIdx.x] = rd.flags[read0 + threadIdx.x] & 1;
            __syncthreads();
        }
        const uint32_t total = sh.kpre[nr];
        const uint32_t run = (total + ROUTE_THREADS - 1) / ROUTE_THREADS;
        const uint32_t q0 = threadIdx.x * run, q1 = min(total, q0 + run);
        KmerRoll<(NW > 0 ? NW : 8)> w;
        if (NW > 0 && q0 < q1) {
            locate_kmer(sh, nr, q0, w.r, w.i);
            roll_load(w, sh, p.hp.k);
        }
        for (uint32_t step = 0; step < run; ++step) {
            const uint32_t q = NW > 0 ? q0 + step : step * ROUTE_THREADS + threadIdx.x;
            const bool live = NW > 0 ? q < q1 : q < total;
            if (!live) continue;
            uint64_t h;
            uint32_t r, i;
            if (NW > 0) {
                h = roll_hash(w, p.hp);
                r = w.r; i = w.i;
                if (q + 1 < q1) roll_step(w, sh, nr, p.hp.k);
            } else {
                locate_kmer(sh, nr, q, r, i);
                const uint32_t fwd = sh.foff[r] + i;
                const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.hp.k - i);
                h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
            }
            if (h == UINT64_MAX) continue;   // the top hash value belongs to no band (count.py:64-66: bands are half-open)
            uint32_t d = 0;
            for (int b = 1; b < p.ndest; ++b) d += h >= lo[b] ? 1u : 0u;
            uint64_t tag = 0;
            if (TAGS) {
                tag = ((p.read_base + read0 + r) << 16) | (uint64_t)(sh.seg_start + i);
                if (flagged[r]) tag |= 1ull << 63;
                if (p.unit_tags) tag = 1;
            }
            const uint32_t pos = atomicAdd(&cur[d], 1u);
            if (pos < p.seg_cap) {
                uint64_t *dst = p.seg + (((uint64_t)d * p.nwg + blockIdx.x) * p.seg_cap + pos) * W;
                if (TAGS) *(ulonglong2 *)dst = make_ulonglong2(h, tag);
                else *dst = h;
            } else {
                const unsigned long long o = atomicAdd(&p.ctr[1], 1ull);
                if (o < p.ovf_cap) {
                    atomicAdd(&p.ctr[18 + d], 1ull);
                    if (TAGS) *(ulonglong2 *)(p.ovf + 2 * o) = make_ulonglong2(h, tag);
                    else p.ovf[o] = h;
                    p.ovf_dest[o] = (uint8_t)d;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)p.ndest)
        p.seg_count[(uint64_t)threadIdx.x * p.nwg + blockIdx.x] = (uint32_t)min((uint64_t)cur[threadIdx.x], p.seg_cap);
}

// one workgroup per destination: exclusive scan of its nwg (<= 1024) segment counts
__global__ __launch_bounds__(ROUTE_MAX_WG) void k_route_scan(RouteParams p)
{
    __shared__ uint64_t wsum[ROUTE_MAX_WG / 64];
    const uint32_t d = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t v = threadIdx.x < p.nwg ? p.seg_count[(uint64_t)d * p.nwg + threadIdx.x] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint64_t up = __shfl_up(incl, s);
        if (lane >= (uint32_t)s) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
    if (threadIdx.x < p.nwg) p.seg_off[(uint64_t)d * p.nwg + threadIdx.x] = before + incl - v;
    if (threadIdx.x == ROUTE_MAX_WG - 1) p.ctr[2 + d] = before + incl;
}

// where each destination's block starts in the flat output: segments first, overflow items after them
__global__ void k_route_bases(RouteParams p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long base = 0;
    for (int d = 0; d < p.ndest; ++d) {
        p.ctr[34 + d] = base;
        p.ctr[50 + d] = base + p.ctr[2 + d];
        base += p.ctr[2 + d] + p.ctr[18 + d];
    }
}

template <int W>
__global__ __launch_bounds__(256) void k_route_compact(RouteParams p)
{
    const uint32_t wg = blockIdx.x, d = blockIdx.y;
    uint64_t n = p.seg_count[(uint64_t)d * p.nwg + wg];
    const uint64_t *src = p.seg + ((uint64_t)d * p.nwg + wg) * p.seg_cap * W;
    const uint64_t first = p.ctr[34 + d] + p.seg_off[(uint64_t)d * p.nwg + wg];
    if (first >= p.cap) return;                          // more items than the output holds: the host sees the totals and says so
    n = min(n, p.cap - first);
    uint64_t *dst = p.out + first * W;
    for (uint64_t j = threadIdx.x; j < n * W; j += 256) dst[j] = src[j];
}

// overflow items join the tail of their destination (after the packed segments).  A workgroup counts its items per destination, takes
// its stretch of every tail with ONE device atomic each and deals the places out through LDS cursors (a device atomic per item, then per
// wave and destination, was most of the 8-26 ms this took for 4.5 M items); entries marked 0xff hold no item (kv_skm.hip, k_skm_loose_route)
template <int W>
__global__ __launch_bounds__(256) void k_route_tail(RouteParams p)
{
    __shared__ uint32_t dcnt[ROUTE_MAX_DEST], dcur[ROUTE_MAX_DEST];
    __shared__ unsigned long long dbase[ROUTE_MAX_DEST];
    unsigned long long n = p.ctr[1];
    if (n > p.ovf_cap) n = p.ovf_cap;
    if (threadIdx.x < ROUTE_MAX_DEST) { dcnt[threadIdx.x] = 0; dcur[threadIdx.x] = 0; }
    __syncthreads();
    // (a workgroup takes a contiguous stretch of the list: its items are read twice, the second time from cache)
    const uint64_t per_wg = (n + gridDim.x - 1) / gridDim.x, j_lo = blockIdx.x * per_wg, j_hi = j_lo + per_wg < n ? j_lo + per_wg : n;
    for (uint64_t j = j_lo + threadIdx.x; j < j_hi; j += blockDim.x) {
        const uint32_t d = p.ovf_dest[j];
        if (d < (uint32_t)p.ndest) atomicAdd(&dcnt[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)p.ndest) dbase[threadIdx.x] = dcnt[threadIdx.x] ? atomicAdd(&p.ctr[50 + threadIdx.x], (unsigned long long)dcnt[threadIdx.x]) : 0ull;
    __syncthreads();
    for (uint64_t j = j_lo + threadIdx.x; j < j_hi; j += blockDim.x) {
        const uint32_t d = p.ovf_dest[j];
        if (d >= (uint32_t)p.ndest) continue;
        const unsigned long long pos = dbase[d] + atomicAdd(&dcur[d], 1u);
        if (pos < p.cap)
            for (int w = 0; w < W; ++w) p.out[pos * W + w] = p.ovf[j * W + w];
    }
}

struct Scratch {
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
std::map<hipStream_t, Scratch> g_route_scratch;   // grow-only, one arena per stream
std::mutex g_route_mu;
}
void kv_route_scratch_release()
{
    std::lock_guard<std::mutex> lk(g_route_mu);
    for (auto &kv : g_route_scratch) { if (kv.second.p) (void)hipFree(kv.second.p); kv.second.p = nullptr; kv.second.bytes = 0; }
}
namespace {

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// segments, overflow list and counters for p.nwg writers of at most n_kmers items of W words
int route_scratch(RouteParams &p, uint64_t n_kmers, uint32_t W, hipStream_t st)
{
    const double m = 1.5 * (double)n_kmers / ((double)p.nwg * p.ndest);   // a full quota, spread evenly over the bands
    p.seg_cap = round_up((uint64_t)(m * 1.1 + 8.0 * std::sqrt(m)) + 1024, 64);
    // worst case: no capacity error possible -- up to 2^30 items; a bigger sink takes a quarter of its items through the overflow list
    // (what missed a segment of 1.5 x the even share: skew beyond that ends in the capacity error the callers fall back from)
    // (route_pack and kv_mex_route compare what was pushed with the list's size: KV_ERR_CAPACITY, never a silent loss)
    p.ovf_cap = n_kmers <= (1ull << 30) ? n_kmers : std::max<uint64_t>(1ull << 30, n_kmers / 4);
    if (const char *e = kv_knob("KV_ROUTE_OVF_CAP")) p.ovf_cap = std::max<uint64_t>(1, std::min<uint64_t>(p.ovf_cap, strtoull(e, nullptr, 10)));      // tests: force the error
    const size_t b_seg = round_up((uint64_t)p.ndest * p.nwg * p.seg_cap * 8 * W, 256);
    const size_t b_cnt = round_up((uint64_t)p.ndest * p.nwg * 4, 256), b_off = round_up((uint64_t)p.ndest * p.nwg * 8, 256);
    const size_t b_ovf = round_up(p.ovf_cap * 8 * W, 256), b_od = round_up(p.ovf_cap, 256), b_ctr = 1024;
    Scratch *scratch;
    {
        std::lock_guard<std::mutex> lk(g_route_mu);
        scratch = &g_route_scratch[kv_stream_key(st)];
    }
    KV_HIP(scratch->need(b_seg + b_cnt + b_off + b_ovf + b_od + b_ctr));
    unsigned char *base = (unsigned char *)scratch->p;
    p.seg = (uint64_t *)base; base += b_seg;
    p.seg_count = (uint32_t *)base; base += b_cnt;
    p.seg_off = (uint64_t *)base; base += b_off;
    p.ovf = (uint64_t *)base; base += b_ovf;
    p.ovf_dest = (uint8_t *)base; base += b_od;
    p.ctr = (unsigned long long *)base;
    KV_HIP(hipMemsetAsync(p.ctr, 0, b_ctr, st));
    return KV_OK;
}

// pack the segments and the overflow tail into p.out, destination after destination; counts_out[d] = items of d
// (route_pack_enqueue + a read of p.ctr[0 .. 33] by the caller: the same without a synchronisation of its own)
void route_pack_enqueue(RouteParams &p, uint32_t W, hipStream_t st)
{
    KvProfScope prof("k_route_compact");
    hipLaunchKernelGGL(k_route_scan, dim3((unsigned)p.ndest), dim3(ROUTE_MAX_WG), 0, st, p);
    hipLaunchKernelGGL(k_route_bases, dim3(1), dim3(64), 0, st, p);
    if (W == 2) {
        hipLaunchKernelGGL(k_route_compact<2>, dim3(p.nwg, (unsigned)p.ndest), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_route_tail<2>, dim3(256), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(k_route_compact<1>, dim3(p.nwg, (unsigned)p.ndest), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_route_tail<1>, dim3(256), dim3(256), 0, st, p);
    }
}

int route_pack(RouteParams &p, uint32_t W, hipStream_t st, uint64_t *counts_out, unsigned long long *tiles_done)
{
    {
        KvProfScope prof("k_route_compact");
        hipLaunchKernelGGL(k_route_scan, dim3((unsigned)p.ndest), dim3(ROUTE_MAX_WG), 0, st, p);
        hipLaunchKernelGGL(k_route_bases, dim3(1), dim3(64), 0, st, p);
        if (W == 2) {
            hipLaunchKernelGGL(k_route_compact<2>, dim3(p.nwg, (unsigned)p.ndest), dim3(256), 0, st, p);
            hipLaunchKernelGGL(k_route_tail<2>, dim3(256), dim3(256), 0, st, p);
        } else {
            hipLaunchKernelGGL(k_route_compact<1>, dim3(p.nwg, (unsigned)p.ndest), dim3(256), 0, st, p);
            hipLaunchKernelGGL(k_route_tail<1>, dim3(256), dim3(256), 0, st, p);
        }
    }
    KV_HIP(hipGetLastError());
    unsigned long long host[34];
    KV_HIP(hipMemcpyAsync(host, p.ctr, sizeof(host), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    *tiles_done = host[0];
    // ctr[1] counts every item that missed its segment, stored or not: past the list's end the kernels drop what they cannot store
    KV_REQUIRE(host[1] <= p.ovf_cap, KV_ERR_CAPACITY, "route: %llu items beside their segments, the overflow list holds %llu",
               host[1], (unsigned long long)p.ovf_cap);
    for (int d = 0; d < p.ndest; ++d) counts_out[d] = host[2 + d] + host[18 + d];
    return KV_OK;
}

__global__ void k_iota_u32(uint32_t *v, uint64_t n)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) v[i] = (uint32_t)i;
}

__global__ void k_gather_hits(const uint64_t *sorted_tags, const uint32_t *sorted_idx, const uint8_t *abund, uint64_t n,
                              int S, uint32_t *out_read, uint32_t *out_off, uint8_t *out_abund)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t tag = sorted_tags[i];
        out_read[i] = (uint32_t)(tag >> 16);
        out_off[i] = (uint32_t)(tag & 0xffffu);
        const uint64_t src = sorted_idx[i];
        for (int c = 0; c < S; ++c) out_abund[i * (uint64_t)S + c] = abund[src * (uint64_t)S + c];
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return kv_hip_malloc(&p, n ? n : 4); }
    template <typename T> T *as() { return (T *)p; }
};

}  // namespace

extern "C" int kv_route_hashes(const kv_reads *reads, int kind, int ksize, int ndest, uint64_t read_index_base,
                               int with_tags, void *d_out, uint64_t cap_items, uint64_t *counts_out)
{
    KV_REQUIRE(reads && d_out && counts_out, KV_ERR_ARG, "kv_route_hashes: null argument");
    KV_REQUIRE(ndest >= 1 && ndest <= ROUTE_MAX_DEST, KV_ERR_ARG, "kv_route_hashes: 1..%d destinations, got %d",
               ROUTE_MAX_DEST, ndest);
    KV_REQUIRE(ksize >= 1 && reads->max_len < 65536, KV_ERR_ARG, "kv_route_hashes: bad k or read longer than a tag can address");
    KV_REQUIRE(read_index_base + reads->n_reads <= (1ull << 32), KV_ERR_ARG, "kv_route_hashes: read index does not fit the u32 read field of kv_hits");
    uint64_t n_kmers = 0;
    kv_reads_num_kmers(reads, ksize, &n_kmers);
    KV_REQUIRE(cap_items >= n_kmers, KV_ERR_CAPACITY, "kv_route_hashes: the output needs room for the %llu k-mers of the shard",
               (unsigned long long)n_kmers);
    for (int d = 0; d < ndest; ++d) counts_out[d] = 0;
    if (reads->n_tiles == 0 || n_kmers == 0) return KV_OK;
    const uint32_t W = with_tags ? 2u : 1u;
    RouteParams p;
    memset(&p, 0, sizeof(p));
    p.hp = make_hash_params(ksize, kv_hashfam_of(kind));
    p.ndest = ndest;
    p.bs = UINT64_MAX / (uint64_t)ndest;
    p.read_base = read_index_base;
    p.cap = cap_items;
    p.out = (uint64_t *)d_out;
    int dev = 0, cus = 256;
    kv_thread_device();
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = reads->tile_lds_bytes;
    const int per_cu = std::max(1, std::min(3, (int)(160 * 1024 / (lds + 2048))));
    p.nwg = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(reads->n_tiles, (uint64_t)per_cu * (uint64_t)cus), ROUTE_MAX_WG);
    // dynamic hand-out, at most 1.5x the average share: bounds what one workgroup can put into a segment
    p.quota = (uint32_t)((reads->n_tiles + p.nwg - 1) / p.nwg);
    p.quota += p.quota / 2 + 1;
    hipStream_t st = kv_stream();
    { const int rc = route_scratch(p, n_kmers, W, st); if (rc != KV_OK) return rc; }
    {
        KvProfScope prof("k_route_hashes");
        const int nw = p.hp.hashfam == HF_MURMUR ? (ksize <= 32 ? 8 : (ksize <= 64 ? 16 : 0)) : 0;
#define KV_LAUNCH_ROUTE(NW_, TAGS_)                                                                              \
        do {                                                                                                     \
            kv_ensure_dynamic_lds((const void *)k_route_hashes<NW_, TAGS_>, lds);                                \
            hipLaunchKernelGGL((k_route_hashes<NW_, TAGS_>), dim3(p.nwg), dim3(ROUTE_THREADS), lds, st,          \
                               reads_dev(reads), reads->n_tiles, p);                                             \
        } while (0)
        if (with_tags) {
            if (nw == 8) KV_LAUNCH_ROUTE(8, true); else if (nw == 16) KV_LAUNCH_ROUTE(16, true); else KV_LAUNCH_ROUTE(0, true);
        } else {
            if (nw == 8) KV_LAUNCH_ROUTE(8, false); else if (nw == 16) KV_LAUNCH_ROUTE(16, false); else KV_LAUNCH_ROUTE(0, false);
        }
#undef KV_LAUNCH_ROUTE
    }
    KV_HIP(hipGetLastError());
    unsigned long long done = 0;
    { const int rc = route_pack(p, W, st, counts_out, &done); if (rc != KV_OK) return rc; }
    KV_REQUIRE(done >= reads->n_tiles, KV_ERR_HIP, "kv_route_hashes: %llu of %u tiles processed", done, reads->n_tiles);
    return KV_OK;
}

// Like kv_route_hashes without tags, but the shard is deduplicated first (super-k-mer buckets, kv_skm.hip): an item
// is the pair (hash, occurrences in this shard), one per DISTINCT k-mer of a bucket.  *n_items_out = items written.
// Shards the bucketed path does not take (k outside 16..64, non-murmur hash, tiny or oversized inputs, or its loose
// list overflowing) leave one (hash, 1) item per k-mer instead: the owner sees the same counts either way.
extern "C" int kv_route_distinct(const kv_reads *reads, int kind, int ksize, int ndest, void *d_out, uint64_t cap_items,
                                 uint64_t *counts_out)
{
    KV_REQUIRE(reads && d_out && counts_out, KV_ERR_ARG, "kv_route_distinct: null argument");
    KV_REQUIRE(ndest >= 1 && ndest <= ROUTE_MAX_DEST, KV_ERR_ARG, "kv_route_distinct: 1..%d destinations, got %d", ROUTE_MAX_DEST, ndest);
    KV_REQUIRE(ksize >= 1, KV_ERR_ARG, "kv_route_distinct: bad k");
    uint64_t n_kmers = 0;
    kv_reads_num_kmers(reads, ksize, &n_kmers);
    KV_REQUIRE(cap_items >= n_kmers, KV_ERR_CAPACITY, "kv_route_distinct: the output needs room for the %llu k-mers of the shard",
               (unsigned long long)n_kmers);
    for (int d = 0; d < ndest; ++d) counts_out[d] = 0;
    if (reads->n_tiles == 0 || n_kmers == 0) return KV_OK;
    hipStream_t st = kv_stream();
    const int hashfam = kv_hashfam_of(kind);
    const char *force = kv_knob("KV_ROUTE_PATH");                 // "plain": tests pin the one-item-per-k-mer form
    const uint64_t min_stride = reads->max_len >= (uint32_t)ksize ? reads->max_len - (uint32_t)ksize + 1 : 1;
    bool bucketed = hashfam == HF_MURMUR && ksize >= 16 && ksize <= 64 && reads->tile_max_bases > 0 && reads->tile_max_bases <= 8192u &&
                    (double)reads->n_reads * (double)min_stride < (double)(1ull << 40) && !(force && strcmp(force, "plain") == 0) &&
                    (n_kmers >= (1ull << 20) || (force && strcmp(force, "skm") == 0));
    RouteParams p;
    if (bucketed) {
        memset(&p, 0, sizeof(p));
        p.ndest = ndest;
        p.bs = UINT64_MAX / (uint64_t)ndest;
        p.cap = cap_items;
        p.out = (uint64_t *)d_out;
        struct Ctx { RouteParams *p; uint64_t n_kmers; hipStream_t st; } ctx = {&p, n_kmers, st};
        auto alloc = [](void *c, uint32_t nwg, KvRouteSink *sink) -> int {
            Ctx *x = (Ctx *)c;
            RouteParams &q = *x->p;
            q.nwg = nwg;
            const int rc = route_scratch(q, x->n_kmers, 2, x->st);
            if (rc != KV_OK) return rc;
            sink->ndest = q.ndest; sink->nwg = q.nwg; sink->bs = q.bs; sink->seg_cap = q.seg_cap; sink->seg = q.seg;
            sink->seg_count = q.seg_count; sink->ovf = q.ovf; sink->ovf_dest = q.ovf_dest; sink->ovf_cap = q.ovf_cap; sink->ctr = q.ctr;
            return KV_OK;
        };
        const int rc = kv_skm_route_distinct(reads, ksize, n_kmers, ndest, alloc, &ctx);
        if (rc == KV_ERR_CAPACITY) bucketed = false;
        else if (rc != KV_OK) return rc;
    }
    if (!bucketed) {
        memset(&p, 0, sizeof(p));
        p.hp = make_hash_params(ksize, hashfam);
        p.ndest = ndest;
        p.bs = UINT64_MAX / (uint64_t)ndest;
        p.unit_tags = 1;
        p.cap = cap_items;
        p.out = (uint64_t *)d_out;
        const size_t lds = reads->tile_lds_bytes;
        const int per_cu = std::max(1, std::min(3, (int)(160 * 1024 / (lds + 2048))));
        p.nwg = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(reads->n_tiles, (uint64_t)per_cu * (uint64_t)kv_device_cus()), ROUTE_MAX_WG);
        p.quota = (uint32_t)((reads->n_tiles + p.nwg - 1) / p.nwg);
        p.quota += p.quota / 2 + 1;
        { const int rc = route_scratch(p, n_kmers, 2, st); if (rc != KV_OK) return rc; }
        KvProfScope prof("k_route_hashes");
        const int nw = p.hp.hashfam == HF_MURMUR ? (ksize <= 32 ? 8 : (ksize <= 64 ? 16 : 0)) : 0;
        if (nw == 8) { kv_ensure_dynamic_lds((const void *)k_route_hashes<8, true>, lds); hipLaunchKernelGGL((k_route_hashes<8, true>), dim3(p.nwg), dim3(ROUTE_THREADS), lds, st, reads_dev(reads), reads->n_tiles, p); }
        else if (nw == 16) { kv_ensure_dynamic_lds((const void *)k_route_hashes<16, true>, lds); hipLaunchKernelGGL((k_route_hashes<16, true>), dim3(p.nwg), dim3(ROUTE_THREADS), lds, st, reads_dev(reads), reads->n_tiles, p); }
        else { kv_ensure_dynamic_lds((const void *)k_route_hashes<0, true>, lds); hipLaunchKernelGGL((k_route_hashes<0, true>), dim3(p.nwg), dim3(ROUTE_THREADS), lds, st, reads_dev(reads), reads->n_tiles, p); }
    }
    KV_HIP(hipGetLastError());
    unsigned long long done = 0;
    return route_pack(p, 2, st, counts_out, &done);
}

extern "C" int kv_mex_plan_make(int kind, int ksize, uint64_t n_reads_global, uint32_t read_len, int ndest, kv_mex_plan *plan)
{
    KV_REQUIRE(kv_hashfam_of(kind) == HF_MURMUR, KV_ERR_ARG, "kv_mex_plan_make: the super-k-mer front end takes the murmur sketch kinds");
    return kv_skm_mex_plan(ksize, n_reads_global, read_len, ndest, plan);
}

extern "C" int kv_mex_plan_short(kv_mex_plan *plan)
{
    KV_REQUIRE(plan, KV_ERR_ARG, "kv_mex_plan_short: null plan");
    return kv_skm_mex_plan_short(plan);
}

extern "C" int kv_mex_emit(const kv_reads *shard, const kv_mex_plan *plan, uint64_t read_base, void *d_seg, void *d_cnt)
{
    KV_REQUIRE(shard && plan && d_seg && d_cnt, KV_ERR_ARG, "kv_mex_emit: null argument");
    return kv_skm_mex_emit(shard, plan, read_base, (uint64_t *)d_seg, (uint32_t *)d_cnt, nullptr, 0, nullptr, nullptr);
}

extern "C" int kv_mex_pack(const kv_mex_plan *plan, const void *d_seg, const void *d_cnt, void *d_out, uint64_t *records_per_dest)
{
    KV_REQUIRE(plan && d_seg && d_cnt && d_out && records_per_dest, KV_ERR_ARG, "kv_mex_pack: null argument");
    return kv_skm_mex_pack(plan, (const uint64_t *)d_seg, (const uint32_t *)d_cnt, (uint64_t *)d_out, records_per_dest);
}

extern "C" int kv_mex_route(const kv_mex_plan *plan, int my_dest, const void *d_recv_seg, const void *d_recv_cnt, int n_src, int compact,
                            int keep_scan, void *d_out, uint64_t cap_items, uint64_t *counts_out, uint64_t *n_kmers_in)
{
    KV_REQUIRE(plan && d_recv_seg && d_recv_cnt && d_out && counts_out && n_kmers_in, KV_ERR_ARG, "kv_mex_route: null argument");
    const int ndest = plan->ndest;
    for (int d = 0; d < ndest; ++d) counts_out[d] = 0;
    hipStream_t st = kv_stream();
    RouteParams p;
    memset(&p, 0, sizeof(p));
    p.ndest = ndest;
    p.bs = UINT64_MAX / (uint64_t)ndest;
    p.cap = cap_items;
    p.out = (uint64_t *)d_out;
    // every k-mer occurrence that arrived could be a pair of its own: the sink is sized for this rank's expected share
    // with slack, and the caller's buffer must hold what actually arrived (checked below, before anything is packed)
    // (no more than the caller's buffer takes, though: beyond it the call ends in a capacity error whatever the sink holds -- a 63 G-k-mer
    // sample's owner expects 7.9 G occurrences and 1.5 G pairs; a sink for the occurrences would be 400 GB)
    const uint64_t expect = plan->n_kmers_global / (uint64_t)ndest;
    const uint64_t sink_items = std::min<uint64_t>(expect + expect / 4, cap_items + cap_items / 4) + (1u << 20);
    KvReadback rb;
    struct Ctx { RouteParams *p; uint64_t n_kmers; hipStream_t st; KvReadback *rb; const unsigned long long *host; } ctx = {&p, sink_items, st, &rb, nullptr};
    auto alloc = [](void *c, uint32_t nwg, KvRouteSink *sink) -> int {
        Ctx *x = (Ctx *)c;
        RouteParams &q = *x->p;
        q.nwg = nwg;
        const int rc = route_scratch(q, x->n_kmers, 2, x->st);
        if (rc != KV_OK) return rc;
        sink->ndest = q.ndest; sink->nwg = q.nwg; sink->bs = q.bs; sink->seg_cap = q.seg_cap; sink->seg = q.seg;
        sink->seg_count = q.seg_count; sink->ovf = q.ovf; sink->ovf_dest = q.ovf_dest; sink->ovf_cap = q.ovf_cap; sink->ctr = q.ctr;
        return KV_OK;
    };
    // the pairs are packed behind the route kernels without waiting for them: kv_skm_mex_route waits once, for both
    auto after = [](void *c) -> int {
        Ctx *x = (Ctx *)c;
        RouteParams &q = *x->p;
        if (q.nwg == 0 || q.seg == nullptr) return KV_OK;
        route_pack_enqueue(q, 2, x->st);
        KV_HIP(hipGetLastError());
        hipError_t e = hipSuccess;
        x->host = x->rb->add(q.ctr, 34, x->st, &e);
        KV_HIP(e);
        return KV_OK;
    };
    { const int rc = kv_skm_mex_route(plan, my_dest, (const uint64_t *)d_recv_seg, (const uint32_t *)d_recv_cnt, n_src, compact, keep_scan, alloc, after, &ctx, n_kmers_in); if (rc != KV_OK) return rc; }
    // what has to fit is the pairs -- one per distinct k-mer, a fifth of the occurrences at sequencing coverage -- not the occurrences:
    // the sink's overflow list and the packing both bound what they write and count what they were asked to
    if (ctx.host) {
        uint64_t total = 0;
        for (int d = 0; d < ndest; ++d) { counts_out[d] = ctx.host[2 + d] + ctx.host[18 + d]; total += counts_out[d]; }
        KV_REQUIRE(ctx.host[1] <= p.ovf_cap, KV_ERR_CAPACITY, "kv_mex_route: %llu pairs beside their segments, the list holds %llu (%llu k-mers arrived)",
                   (unsigned long long)ctx.host[1], (unsigned long long)p.ovf_cap, (unsigned long long)*n_kmers_in);
        KV_REQUIRE(total <= cap_items, KV_ERR_CAPACITY, "kv_mex_route: %llu pairs for %llu k-mers that arrived, the output holds %llu",
                   (unsigned long long)total, (unsigned long long)*n_kmers_in, (unsigned long long)cap_items);
    }
    return KV_OK;
}

// ---- (hash, occurrences) pairs in 9 bytes for the wire (include/kvsketch.h) ------------------------------------------------------------
namespace {
struct PairBlocks {                      // per block: first pair, pairs, first word of its packed form
    uint64_t first[ROUTE_MAX_DEST], n[ROUTE_MAX_DEST], woff[ROUTE_MAX_DEST];
    int nblocks;
};
__device__ __forceinline__ uint64_t pairs_wave_sum(uint64_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)v, d), hi = (uint32_t)__shfl_down((int)(uint32_t)(v >> 32), d);
        v += (uint64_t)lo | ((uint64_t)hi << 32);
    }
    return v;
}
// the head words (occurrences, summed below) start as zero
__global__ void k_pairs_pack_init(PairBlocks b, uint64_t *out)
{
    const int d = threadIdx.x;
    if (d < b.nblocks) out[b.woff[d]] = 0;
}
// a workgroup takes 2048 pairs at a time, consecutive lanes on consecutive pairs (16-byte loads, 8-byte hash stores); the count bytes meet
// in LDS and leave as 256 whole words (no byte stores to HBM; the last word's missing bytes are zero)
#define PAIRS_CHUNK 2048u
__global__ __launch_bounds__(256) void k_pairs_pack(PairBlocks b, const ulonglong2 *__restrict__ pairs, uint64_t *out)
{
    __shared__ __attribute__((aligned(8))) uint8_t cb[PAIRS_CHUNK];
    const int d = blockIdx.y;
    const uint64_t n = b.n[d], chunks = (n + PAIRS_CHUNK - 1) / PAIRS_CHUNK;
    const ulonglong2 *src = pairs + b.first[d];
    uint64_t *hashes = out + b.woff[d] + 1, *cwords = hashes + n;
    uint64_t occ = 0;
    for (uint64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint64_t base = c * PAIRS_CHUNK;
#pragma unroll
        for (uint32_t j = 0; j < PAIRS_CHUNK / 256u; ++j) {
            const uint32_t at = j * 256u + threadIdx.x;
            const uint64_t i = base + at;
            uint8_t byte = 0;
            if (i < n) {
                const ulonglong2 it = src[i];
                hashes[i] = it.x;
                byte = (uint8_t)(it.y < 255ull ? it.y : 255ull);
                occ += it.y;
            }
            cb[at] = byte;
        }
        __syncthreads();
        const uint64_t w = base / 8 + threadIdx.x;
        if (w < (n + 7) / 8) cwords[w] = ((const uint64_t *)cb)[threadIdx.x];
        __syncthreads();
    }
    occ = pairs_wave_sum(occ);
    if ((threadIdx.x & 63) == 0 && occ) atomicAdd((unsigned long long *)&out[b.woff[d]], (unsigned long long)occ);
}
__global__ __launch_bounds__(256) void k_pairs_unpack(PairBlocks b, const uint64_t *__restrict__ in, ulonglong2 *pairs, unsigned long long *occ_sum)
{
    __shared__ __attribute__((aligned(8))) uint8_t cb[PAIRS_CHUNK];
    const int s = blockIdx.y;
    const uint64_t n = b.n[s], chunks = (n + PAIRS_CHUNK - 1) / PAIRS_CHUNK;
    const uint64_t *hashes = in + b.woff[s] + 1, *cwords = hashes + n;
    ulonglong2 *dst = pairs + b.first[s];
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(occ_sum, (unsigned long long)in[b.woff[s]]);
    for (uint64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint64_t base = c * PAIRS_CHUNK;
        const uint64_t w = base / 8 + threadIdx.x;
        ((uint64_t *)cb)[threadIdx.x] = w < (n + 7) / 8 ? cwords[w] : 0ull;
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < PAIRS_CHUNK / 256u; ++j) {
            const uint32_t at = j * 256u + threadIdx.x;
            const uint64_t i = base + at;
            if (i < n) dst[i] = make_ulonglong2(hashes[i], (unsigned long long)cb[at]);
        }
        __syncthreads();
    }
}
unsigned long long *g_pairs_occ = nullptr;       // one device word for kv_pairs_unpack's total (calls are serialised by g_pairs_mu)
std::mutex g_pairs_mu;
}  // namespace

extern "C" int kv_pairs_pack(const void *d_pairs, const uint64_t *counts, int ndest, void *d_out, uint64_t out_cap_words, uint64_t *words_per_dest)
{
    KV_REQUIRE(d_out && counts && words_per_dest && ndest >= 1 && ndest <= ROUTE_MAX_DEST, KV_ERR_ARG, "kv_pairs_pack: bad argument");
    PairBlocks b;
    memset(&b, 0, sizeof(b));
    b.nblocks = ndest;
    uint64_t first = 0, woff = 0, most = 0;
    for (int d = 0; d < ndest; ++d) {
        b.first[d] = first; b.n[d] = counts[d]; b.woff[d] = woff;
        words_per_dest[d] = 1 + counts[d] + (counts[d] + 7) / 8;
        first += counts[d]; woff += words_per_dest[d];
        most = std::max(most, counts[d]);
    }
    KV_REQUIRE(first == 0 || d_pairs, KV_ERR_ARG, "kv_pairs_pack: null pairs");
    KV_REQUIRE(woff <= out_cap_words, KV_ERR_CAPACITY, "kv_pairs_pack: %llu words for a buffer of %llu", (unsigned long long)woff, (unsigned long long)out_cap_words);
    hipStream_t st = kv_stream();
    KvProfScope prof("k_pairs_pack");
    hipLaunchKernelGGL(k_pairs_pack_init, dim3(1), dim3(64), 0, st, b, (uint64_t *)d_out);
    if (most) {
        const unsigned gx = (unsigned)std::min<uint64_t>((most + PAIRS_CHUNK - 1) / PAIRS_CHUNK, 2048);
        hipLaunchKernelGGL(k_pairs_pack, dim3(gx, (unsigned)ndest), dim3(256), 0, st, b, (const ulonglong2 *)d_pairs, (uint64_t *)d_out);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

extern "C" int kv_pairs_unpack(const void *d_in, const uint64_t *words_per_src, int nsrc, void *d_pairs, uint64_t cap_pairs, uint64_t *pairs_per_src,
                               uint64_t *occurrences)
{
    KV_REQUIRE(d_in && words_per_src && pairs_per_src && occurrences && nsrc >= 1 && nsrc <= ROUTE_MAX_DEST, KV_ERR_ARG, "kv_pairs_unpack: bad argument");
    PairBlocks b;
    memset(&b, 0, sizeof(b));
    b.nblocks = nsrc;
    uint64_t first = 0, woff = 0, most = 0;
    for (int s = 0; s < nsrc; ++s) {
        const uint64_t w = words_per_src[s];
        KV_REQUIRE(w >= 1, KV_ERR_ARG, "kv_pairs_unpack: block %d has no head word", s);
        // w = 1 + n + ceil(n / 8): n = floor(8 (w - 1) / 9), the one n that gives w back (the sender only produces such w)
        const uint64_t n = (w - 1) * 8 / 9;
        KV_REQUIRE(1 + n + (n + 7) / 8 == w, KV_ERR_ARG, "kv_pairs_unpack: block %d of %llu words is not a packed block", s, (unsigned long long)w);
        b.first[s] = first; b.n[s] = n; b.woff[s] = woff;
        pairs_per_src[s] = n;
        first += n; woff += w;
        most = std::max(most, n);
    }
    KV_REQUIRE(first <= cap_pairs && (first == 0 || d_pairs), KV_ERR_CAPACITY, "kv_pairs_unpack: %llu pairs for a buffer of %llu", (unsigned long long)first,
               (unsigned long long)cap_pairs);
    hipStream_t st = kv_stream();
    KvProfScope prof("k_pairs_unpack");
    std::lock_guard<std::mutex> lk(g_pairs_mu);
    if (!g_pairs_occ) KV_HIP(kv_hip_malloc((void **)&g_pairs_occ, 256));
    KV_HIP(hipMemsetAsync(g_pairs_occ, 0, 8, st));
    {
        // (every block starts its own workgroups, an empty one too: its head still counts)
        const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((most + PAIRS_CHUNK - 1) / PAIRS_CHUNK, 2048));
        hipLaunchKernelGGL(k_pairs_unpack, dim3(gx, (unsigned)nsrc), dim3(256), 0, st, b, (const uint64_t *)d_in, (ulonglong2 *)d_pairs, g_pairs_occ);
    }
    KV_HIP(hipGetLastError());
    unsigned long long total = 0;
    KV_HIP(hipMemcpyAsync(&total, g_pairs_occ, 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    *occurrences = total;
    return KV_OK;
}

extern "C" int kv_reads_flags(const kv_reads *reads, uint8_t *flags)
{
    KV_REQUIRE(reads && (flags || reads->n_reads == 0), KV_ERR_ARG, "kv_reads_flags: null argument");
    if (reads->n_reads == 0) return KV_OK;
    hipStream_t st = kv_stream();
    KV_HIP(hipMemcpyAsync(flags, reads->d_flags, reads->n_reads, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

extern "C" int kv_mex_emit_pack(const kv_reads *shard, const kv_mex_plan *plan, uint64_t read_base, void *d_seg, void *d_cnt, void *d_out,
                                uint64_t out_cap_words, uint64_t *records_per_dest, int *packed)
{
    KV_REQUIRE(shard && plan && d_seg && d_cnt && d_out && records_per_dest && packed, KV_ERR_ARG, "kv_mex_emit_pack: null argument");
    return kv_skm_mex_emit(shard, plan, read_base, (uint64_t *)d_seg, (uint32_t *)d_cnt, (uint64_t *)d_out, out_cap_words, records_per_dest, packed);
}

// ---- argsort for the host half of partition (kevlar_amd/partition.py assemble_partitions) ---------------------------------
// The reads of a partition run are ordered by name, by component label and by a hash of their canonical sequence: three
// numpy argsorts over 6.7 M reads were 0.75 s of config 4's band (a fifth of `kevlar partition` there).  The keys go to the
// device, rocPRIM's radix sort -- stable -- orders (key, index) pairs, the indices come back: a few milliseconds.
namespace {
__global__ void k_row_keys(const uint8_t *rows, const uint32_t *idx, uint64_t n, uint32_t width, uint32_t at, unsigned long long *keys)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *row = rows + (uint64_t)idx[i] * width + at;
        unsigned long long key = 0;
        for (uint32_t b = 0; b < 8; ++b) key = (key << 8) | (at + b < width ? (unsigned long long)row[b] : 0ull);     // big-endian: compares like the bytes
        keys[i] = key;
    }
}

int argsort_pairs(DevBuf &keys_in, DevBuf &idx_in, DevBuf &keys_out, DevBuf &idx_out, DevBuf &tmp, size_t &tmp_bytes, uint64_t n, hipStream_t st)
{
    size_t need = 0;
    KV_HIP(rocprim::radix_sort_pairs(nullptr, need, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                     idx_out.as<uint32_t>(), (size_t)n, 0, 64, st));
    if (need > tmp_bytes) {
        if (tmp.p) { (void)hipFree(tmp.p); tmp.p = nullptr; }
        KV_HIP(tmp.alloc(need));
        tmp_bytes = need;
    }
    KV_HIP(rocprim::radix_sort_pairs(tmp.p, need, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(), idx_in.as<uint32_t>(),
                                     idx_out.as<uint32_t>(), (size_t)n, 0, 64, st));
    return KV_OK;
}
}  // namespace

// order[i] = index of the i-th smallest key, equal keys in index order (numpy.argsort(keys, kind='stable')); host pointers
extern "C" int kv_argsort_u64(const uint64_t *keys, uint64_t n, uint32_t *order)
{
    KV_REQUIRE((keys && order) || n == 0, KV_ERR_ARG, "kv_argsort_u64: null argument");
    KV_REQUIRE(n < (1ull << 32), KV_ERR_ARG, "kv_argsort_u64: too many keys");
    if (n == 0) return KV_OK;
    hipStream_t st = kv_stream();
    DevBuf k_in, k_out, v_in, v_out, tmp;
    size_t tmp_bytes = 0;
    KV_HIP(k_in.alloc(n * 8)); KV_HIP(k_out.alloc(n * 8)); KV_HIP(v_in.alloc(n * 4)); KV_HIP(v_out.alloc(n * 4));
    KV_HIP(hipMemcpyAsync(k_in.p, keys, n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota_u32, dim3(256), dim3(256), 0, st, v_in.as<uint32_t>(), n);
    { const int rc = argsort_pairs(k_in, v_in, k_out, v_out, tmp, tmp_bytes, n, st); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemcpyAsync(order, v_out.p, n * 4, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

// the same for n rows of `width` bytes compared like byte strings (numpy.argsort of an 'S<width>' array, kind='stable'):
// least significant eight bytes first, every pass stable
extern "C" int kv_argsort_rows(const void *rows, uint64_t n, uint32_t width, uint32_t *order)
{
    KV_REQUIRE((rows && order) || n == 0, KV_ERR_ARG, "kv_argsort_rows: null argument");
    KV_REQUIRE(n < (1ull << 32) && width >= 1, KV_ERR_ARG, "kv_argsort_rows: bad size");
    if (n == 0) return KV_OK;
    hipStream_t st = kv_stream();
    DevBuf d_rows, k_in, k_out, v_a, v_b, tmp;
    size_t tmp_bytes = 0;
    KV_HIP(d_rows.alloc(n * (uint64_t)width)); KV_HIP(k_in.alloc(n * 8)); KV_HIP(k_out.alloc(n * 8)); KV_HIP(v_a.alloc(n * 4)); KV_HIP(v_b.alloc(n * 4));
    KV_HIP(hipMemcpyAsync(d_rows.p, rows, n * (uint64_t)width, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_iota_u32, dim3(256), dim3(256), 0, st, v_a.as<uint32_t>(), n);
    DevBuf *cur = &v_a, *nxt = &v_b;
    const uint32_t chunks = (width + 7u) / 8u;
    for (uint32_t c = chunks; c-- > 0;) {
        hipLaunchKernelGGL(k_row_keys, dim3(1024), dim3(256), 0, st, d_rows.as<uint8_t>(), cur->as<uint32_t>(), n, width, c * 8u, k_in.as<unsigned long long>());
        KV_HIP(hipGetLastError());
        { const int rc = argsort_pairs(k_in, *cur, k_out, *nxt, tmp, tmp_bytes, n, st); if (rc != KV_OK) return rc; }
        std::swap(cur, nxt);
    }
    KV_HIP(hipMemcpyAsync(order, cur->p, n * 4, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

// n_total gathered hits (tag, S abundance bytes each), of which the n_valid smallest tags are real
// (padding carries tag ~0): sort by tag = (read, offset) and hand back an ordinary kv_hits.
extern "C" int kv_hits_from_tagged(const void *d_tags, const void *d_abund, uint64_t n_total, uint64_t n_valid,
                                   int nsamples, kv_hits **out)
{
    KV_REQUIRE(out && n_valid <= n_total && nsamples >= 1 && nsamples <= KV_MAX_SAMPLES, KV_ERR_ARG,
               "kv_hits_from_tagged: bad argument");
    KV_REQUIRE(n_total == 0 || (d_tags && d_abund), KV_ERR_ARG, "kv_hits_from_tagged: null buffer");
    KV_REQUIRE(n_total < (1ull << 32), KV_ERR_ARG, "kv_hits_from_tagged: too many hits");
    kv_hits *hits = new kv_hits();
    hits->nsamples = nsamples;
    *out = hits;
    if (n_valid == 0) return KV_OK;
    hipStream_t st = kv_stream();
    DevBuf k_out, v_in, v_out, tmp, o_read, o_off, o_abund;
    hipError_t e = k_out.alloc(n_total * 8);
    if (e == hipSuccess) e = v_in.alloc(n_total * 4);
    if (e == hipSuccess) e = v_out.alloc(n_total * 4);
    if (e == hipSuccess) e = o_read.alloc(n_valid * 4);
    if (e == hipSuccess) e = o_off.alloc(n_valid * 4);
    if (e == hipSuccess) e = o_abund.alloc(n_valid * (uint64_t)nsamples);
    size_t tmp_bytes = 0;
    if (e == hipSuccess)
        e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const unsigned long long *)d_tags, k_out.as<unsigned long long>(),
                                      v_in.as<uint32_t>(), v_out.as<uint32_t>(), (size_t)n_total, 0, 64, st);
    if (e == hipSuccess) e = tmp.alloc(tmp_bytes);
    if (e == hipSuccess) {
        KvProfScope prof("sort_hits");
        hipLaunchKernelGGL(k_iota_u32, dim3(256), dim3(256), 0, st, v_in.as<uint32_t>(), n_total);
        e = rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const unsigned long long *)d_tags, k_out.as<unsigned long long>(),
                                      v_in.as<uint32_t>(), v_out.as<uint32_t>(), (size_t)n_total, 0, 64, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather_hits, dim3(256), dim3(256), 0, st, k_out.as<uint64_t>(), v_out.as<uint32_t>(),
                               (const uint8_t *)d_abund, n_valid, nsamples, o_read.as<uint32_t>(), o_off.as<uint32_t>(),
                               o_abund.as<uint8_t>());
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hits->read.resize(n_valid);
    if (e == hipSuccess) e = hits->offset.resize(n_valid);
    if (e == hipSuccess) e = hits->abund.resize(n_valid * (uint64_t)nsamples);
    if (e == hipSuccess) e = hipMemcpyAsync(hits->read.data(), o_read.p, n_valid * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hits->offset.data(), o_off.p, n_valid * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hits->abund.data(), o_abund.p, n_valid * (uint64_t)nsamples, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        delete hits;
        *out = nullptr;
        kv_set_error("kv_hits_from_tagged failed: %s", hipGetErrorString(e));
        return KV_ERR_HIP;
    }
    return KV_OK;
}
