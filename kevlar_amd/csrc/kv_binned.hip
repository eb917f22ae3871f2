// kv_binned.hip -- K2 count for large batches: no global atomics on the tables.
//
// The straightforward kernel (k_consume) performs T saturating read-modify-writes per k-mer at
// random table positions.  Measured on MI355X each one moves a 64-B sector in and out of HBM
// (269 GB per 525 M k-mers, profiles/r1_naive) and the device tops out at ~27 G atomics/s.
// Counting is a histogram, so this file computes it the way large GPU histograms are done:
//
//   A  k_bin_hash   hash every k-mer once (same LDS-staged tiles as k_consume), band/mask filter,
//                   and for each of the T tables append the bin to one of C coarse buckets
//                   (bin range = F slices of 65536 bins).  Appends go through per-workgroup LDS
//                   ring buffers and reach HBM as 64..256-byte coalesced bursts; one global
//                   atomic claims space per burst.
//   B  k_bin_split  each coarse bucket is split into its F slices; items shrink to the 16-bit
//                   offset inside the slice.
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

#include "kv_device.h"

namespace {

#define BIN_C 64            // coarse buckets per table
#define BIN_MAX_T 4         // tables handled by the partitioned path
#define BIN_NS (BIN_C * BIN_MAX_T)
#define BIN_RING_BUDGET 16384   // LDS ring entries shared by all streams of a workgroup
#define BIN_RING_MIN 64
#define BIN_RING_MAX 4096       // a round appends at most 1024 items to one stream
#define BIN_A_THREADS 1024
#define BIN_B_THREADS 256
#define BIN_B_CHUNK 131072  // items of one coarse bucket handled by one stage-B workgroup
#define BIN_C_THREADS 512
#define BIN_MAX_F 512

struct BinGeom {
    int T, F, C;                     // tables, slices per coarse bucket, coarse buckets in use (<= BIN_C)
    uint32_t ringA, ringB;           // LDS ring entries per stream in stages A / B (powers of two)
    uint32_t recipF;                 // ceil(2^32 / F): slice / F by multiply-high
    uint32_t nslices[BIN_MAX_T];
    uint64_t cap1, cap2, spill_cap;
    uint32_t *gbuf1;                 // [T*C][cap1] coarse items: (slice-in-bucket << 16) | offset
    uint16_t *gbuf2;                 // [T*C*F][cap2] offsets inside a slice
    uint32_t *gcnt1, *gcnt2;         // items appended per coarse bucket / per slice
    unsigned long long *spill;       // (table << 32) | bin
    unsigned long long *ctr;         // [0] spill count, [1] overflow flag, [2] k-mers added, [3] occupancy delta
};

__device__ __forceinline__ void spill_item(const BinGeom &g, int t, uint64_t bin)
{
    const unsigned long long pos = atomicAdd(&g.ctr[0], 1ull);
    if (pos < g.spill_cap) g.spill[pos] = ((unsigned long long)t << 32) | bin;
    else g.ctr[1] = 1;
}

// ---- stage A -----------------------------------------------------------------------------
struct StreamsA {
    uint32_t ring[BIN_RING_BUDGET];   // stream s owns [s * ringA, (s + 1) * ringA)
    uint32_t cnt[BIN_NS], base[BIN_NS];
    uint32_t fl_n[BIN_NS], fl_pos[BIN_NS], fl_base[BIN_NS];
};

__device__ __forceinline__ void flush_streams_a(StreamsA &st, const BinGeom &g, int ns, bool final)
{
    // decide (one thread per stream), then copy (one wave per stream, lanes = items)
    const uint32_t R = g.ringA;
    for (int s = threadIdx.x; s < ns; s += blockDim.x) {
        const uint32_t base = st.base[s], avail = st.cnt[s] - base;
        uint32_t n = 0, newbase = base;
        if (avail > R) { n = R; newbase = st.cnt[s]; }       // ring overran: the excess went to the spill list
        else if (final) { n = avail; newbase = base + n; }
        else if (avail >= R / 2) { n = avail & ~15u; newbase = base + n; }
        st.fl_n[s] = n;
        st.fl_base[s] = base;
        if (n) st.fl_pos[s] = atomicAdd(&g.gcnt1[s], n);
        st.base[s] = newbase;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (int s = wave; s < ns; s += nwaves) {
        const uint32_t n = st.fl_n[s];
        for (uint32_t j = lane; j < n; j += 64) {
            const uint32_t item = st.ring[(uint32_t)s * R + ((st.fl_base[s] + j) & (R - 1))];
            const uint64_t pos = (uint64_t)st.fl_pos[s] + j;
            if (pos < g.cap1) {
                g.gbuf1[(uint64_t)s * g.cap1 + pos] = item;
            } else {   // coarse bucket full: keep the increment, apply it later with an atomic
                const int t = s / g.C, c = s % g.C;
                spill_item(g, t, (((uint64_t)c * g.F + (item >> 16)) << 16) | (item & 0xffffu));
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(BIN_A_THREADS) void k_bin_hash(ReadsDev rd, uint32_t n_tiles, const SketchDev *__restrict__ sk,
                                                            const SketchDev *__restrict__ mask, ConsumeFilter f, BinGeom g)
{
    __shared__ TileShared sh;
    __shared__ StreamsA st;
    const int ns = g.T * g.C;
    for (int s = threadIdx.x; s < BIN_NS; s += blockDim.x) { st.cnt[s] = 0; st.base[s] = 0; }
    __syncthreads();
    uint64_t n_added = 0;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        uint32_t read0;
        const uint32_t nr = stage_tile(sh, rd, tile, f.hp.k, 0, 0, read0);
        const uint32_t total = sh.kpre[nr];
        for (uint32_t q0 = 0; q0 < total; q0 += blockDim.x) {
            const uint32_t q = q0 + threadIdx.x;
            if (q < total) {
                uint32_t r, i;
                locate_kmer(sh, nr, q, r, i);
                const uint32_t fwd = sh.foff[r] + i;
                const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)f.hp.k - i);
                const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, f.hp);
                if (consume_filter_pass(f, mask, h)) {
                    n_added += 1;
                    for (int t = 0; t < g.T; ++t) {
                        const uint64_t bin = fastmod(h, sk->size[t], sk->magic[t]);
                        const uint32_t slice = (uint32_t)(bin >> 16);
                        const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
                        const uint32_t item = ((slice - c * (uint32_t)g.F) << 16) | (uint32_t)(bin & 0xffffu);
                        const int s = t * g.C + (int)c;
                        const uint32_t pos = atomicAdd(&st.cnt[s], 1u);
                        if (pos - st.base[s] < g.ringA) st.ring[(uint32_t)s * g.ringA + (pos & (g.ringA - 1))] = item;
                        else spill_item(g, t, bin);
                    }
                }
            }
            __syncthreads();
            flush_streams_a(st, g, ns, false);
        }
    }
    flush_streams_a(st, g, ns, true);
    n_added = wave_sum_u64(n_added);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
}

// ---- stage B -----------------------------------------------------------------------------
__global__ __launch_bounds__(BIN_B_THREADS) void k_bin_split(BinGeom g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // layout: ring[F][ringB] u16 | cnt[F] | base[F] | fl_n[F] | fl_pos[F] | fl_base[F]
    const uint32_t R = g.ringB;
    uint16_t *ring = (uint16_t *)smem;
    uint32_t *cnt = (uint32_t *)(smem + (size_t)g.F * R * 2);
    uint32_t *base = cnt + g.F, *fl_n = base + g.F, *fl_pos = fl_n + g.F, *fl_base = fl_pos + g.F;
    const int s = blockIdx.y;                       // coarse stream = table * C + bucket
    const int t = s / g.C, c = s % g.C;
    uint64_t total = g.gcnt1[s];
    if (total > g.cap1) total = g.cap1;
    const uint64_t start = (uint64_t)blockIdx.x * BIN_B_CHUNK;
    if (start >= total) return;
    const uint64_t end = total < start + BIN_B_CHUNK ? total : start + BIN_B_CHUNK;
    for (int fidx = threadIdx.x; fidx < g.F; fidx += blockDim.x) { cnt[fidx] = 0; base[fidx] = 0; }
    __syncthreads();
    const uint32_t *src = g.gbuf1 + (uint64_t)s * g.cap1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (uint64_t r0 = start; r0 < end; r0 += (uint64_t)blockDim.x * 4) {
        const uint64_t i0 = r0 + (uint64_t)threadIdx.x * 4;
        uint32_t items[4];
        int have = 0;
        if (i0 + 4 <= end) {
            const uint4 v = *(const uint4 *)(src + i0);
            items[0] = v.x; items[1] = v.y; items[2] = v.z; items[3] = v.w;
            have = 4;
        } else {
            for (uint64_t i = i0; i < end; ++i) items[have++] = src[i];
        }
        for (int j = 0; j < have; ++j) {
            const uint32_t fidx = items[j] >> 16, off = items[j] & 0xffffu;
            const uint32_t pos = atomicAdd(&cnt[fidx], 1u);
            if (pos - base[fidx] < R) ring[fidx * R + (pos & (R - 1))] = (uint16_t)off;
            else spill_item(g, t, (((uint64_t)c * g.F + fidx) << 16) | off);
        }
        __syncthreads();
        const bool final = r0 + (uint64_t)blockDim.x * 4 >= end;
        for (int fidx = threadIdx.x; fidx < g.F; fidx += blockDim.x) {
            const uint32_t b = base[fidx], avail = cnt[fidx] - b;
            uint32_t n = 0, nb = b;
            if (avail > R) { n = R; nb = cnt[fidx]; }
            else if (final) { n = avail; nb = b + n; }
            else if (avail >= R / 2) { n = avail & ~15u; nb = b + n; }
            fl_n[fidx] = n;
            fl_base[fidx] = b;
            if (n) fl_pos[fidx] = atomicAdd(&g.gcnt2[(uint64_t)s * g.F + fidx], n);
            base[fidx] = nb;
        }
        __syncthreads();
        for (int fidx = wave; fidx < g.F; fidx += nwaves) {
            const uint32_t n = fl_n[fidx];
            for (uint32_t j = lane; j < n; j += 64) {
                const uint16_t off = ring[fidx * R + ((fl_base[fidx] + j) & (R - 1))];
                const uint64_t pos = (uint64_t)fl_pos[fidx] + j;
                if (pos < g.cap2) g.gbuf2[((uint64_t)s * g.F + fidx) * g.cap2 + pos] = off;
                else spill_item(g, t, (((uint64_t)c * g.F + fidx) << 16) | off);
            }
        }
        __syncthreads();
    }
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

__global__ __launch_bounds__(BIN_C_THREADS) void k_bin_apply(const SketchDev *__restrict__ sk, BinGeom g)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[16384];   // one slice: 65536 counters of <= 8 bits
    const int t = blockIdx.y;
    const uint32_t slice = blockIdx.x;
    if (slice >= g.nslices[t]) return;
    const uint32_t c = slice / (uint32_t)g.F, fidx = slice % (uint32_t)g.F;
    const uint64_t stream = ((uint64_t)t * g.C + c) * g.F + fidx;
    uint64_t n = g.gcnt2[stream];
    if (n > g.cap2) n = g.cap2;
    if (n == 0) return;                                            // untouched slice: no traffic at all
    const int storage = sk->storage;
    const uint64_t bin0 = (uint64_t)slice << 16;
    const uint64_t left = sk->size[t] - bin0, nb = left < 65536 ? left : 65536;
    // byte range of the slice inside the table (the allocation is padded to 16 B)
    const uint64_t byte0 = storage == ST_BYTE ? bin0 : (storage == ST_NIBBLE ? bin0 >> 1 : bin0 >> 3);
    const uint64_t nbytes = storage == ST_BYTE ? nb : (storage == ST_NIBBLE ? (nb + 1) / 2 : (nb + 7) / 8);
    const uint32_t nvec = (uint32_t)((nbytes + 15) / 16);
    uint4 *tab = (uint4 *)(sk->tab[t] + byte0);
    uint4 *l4 = (uint4 *)lds;
    for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) l4[i] = tab[i];
    __syncthreads();
    const uint16_t *items = g.gbuf2 + stream * g.cap2;
    uint32_t fresh = 0;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) fresh += lds_inc(lds, items[i], storage) ? 1u : 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) tab[i] = l4[i];
    if (t == 0) {
        const uint64_t tot = wave_sum_u64(fresh);
        if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&g.ctr[3], (unsigned long long)tot);
    }
}

__device__ __forceinline__ bool table_inc_bin(const SketchDev *s, int t, uint64_t bin)
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
        const uint32_t prev = atomicCAS(w, old, old + (1u << shift));
        if (prev == old) return cur == 0;
        old = prev;
    }
}

__global__ void k_bin_spill(const SketchDev *__restrict__ sk, BinGeom g)
{
    unsigned long long n = g.ctr[0];
    if (n > g.spill_cap) n = g.spill_cap;
    uint64_t fresh = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long e = g.spill[i];
        const int t = (int)(e >> 32);
        const bool was_zero = table_inc_bin(sk, t, e & 0xffffffffull);
        fresh += (was_zero && t == 0) ? 1 : 0;
    }
    fresh = wave_sum_u64(fresh);
    if ((threadIdx.x & 63) == 0 && fresh) atomicAdd(&g.ctr[3], (unsigned long long)fresh);
}

// grow-only scratch shared by all calls of this process (one process per GPU)
struct Scratch {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t need(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
};
Scratch g_scratch;
std::mutex g_scratch_mu;

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

}  // namespace

// linear-counting estimate of the distinct k-mers behind an occupancy of table 0
double kv_estimate_distinct(uint64_t occupied, uint64_t size)
{
    if (occupied >= size) return (double)size * std::log((double)size);
    return -(double)size * std::log1p(-(double)occupied / (double)size);
}

bool kv_binned_eligible(const kv_sketch *s, const kv_reads *reads, uint64_t n_kmers, int nbands)
{
    const char *force = getenv("KV_COUNT_PATH");
    if (force && strcmp(force, "atomic") == 0) return false;
    if (s->h.ntables > BIN_MAX_T) return false;
    uint64_t pmin = UINT64_MAX, pmax = 0;
    for (int t = 0; t < s->h.ntables; ++t) { pmin = std::min(pmin, s->h.size[t]); pmax = std::max(pmax, s->h.size[t]); }
    if (pmax > (uint64_t)BIN_C * BIN_MAX_F * 65536ull) return false;      // <= 2^31 bins per table
    if (force && strcmp(force, "binned") == 0) return reads->n_tiles > 0;
    const uint64_t expected = nbands > 0 ? n_kmers / (uint64_t)nbands : n_kmers;
    // worth it once the batch touches the tables about as densely as streaming them costs
    return pmin >= (1ull << 20) && expected >= (1ull << 22) && expected * 8 >= pmax;
}

// returns KV_OK, or KV_ERR_CAPACITY when the spill list overflowed (tables untouched: caller falls back)
int kv_consume_binned(kv_sketch *s, const kv_reads *reads, const ConsumeFilter &filter, const kv_sketch *mask,
                      uint64_t n_kmers, int nbands, uint64_t *n_added)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    hipStream_t st = kv_stream();
    BinGeom g;
    memset(&g, 0, sizeof(g));
    g.T = s->h.ntables;
    uint64_t pmin = UINT64_MAX;
    uint32_t maxsl = 1;
    for (int t = 0; t < g.T; ++t) {
        g.nslices[t] = (uint32_t)((s->h.size[t] + 65535) >> 16);
        maxsl = std::max(maxsl, g.nslices[t]);
        pmin = std::min(pmin, s->h.size[t]);
    }
    g.F = (int)((maxsl + BIN_C - 1) / BIN_C);
    g.C = (int)((maxsl + (uint32_t)g.F - 1) / (uint32_t)g.F);
    auto ring_for = [](uint32_t streams) {
        uint32_t r = BIN_RING_MIN;
        while (r * 2 <= BIN_RING_MAX && (uint64_t)r * 2 * streams <= BIN_RING_BUDGET) r *= 2;
        return r;
    };
    g.ringA = ring_for((uint32_t)(g.T * g.C));
    g.ringB = ring_for((uint32_t)g.F);
    g.recipF = g.F == 1 ? 0u : (uint32_t)((1ull << 32) / (uint64_t)g.F + 1);   // F == 1: kernels take slice as is
    const double expected = (double)(nbands > 0 && !filter.use_mask ? n_kmers / (uint64_t)nbands + 1 : n_kmers);
    g.cap1 = round_up((uint64_t)(expected * std::min(1.0, (double)g.F * 65536.0 / (double)pmin) * 1.03) + 65536, 64);
    g.cap2 = round_up((uint64_t)(expected * std::min(1.0, 65536.0 / (double)pmin) * 1.10) + 2048, 64);
    g.spill_cap = std::max<uint64_t>(1u << 20, (uint64_t)(expected * g.T / 8));
    const uint64_t ns = (uint64_t)g.T * g.C;
    const size_t b_buf1 = round_up(ns * g.cap1 * 4, 256), b_buf2 = round_up(ns * g.F * g.cap2 * 2, 256);
    const size_t b_cnt1 = round_up(ns * 4, 256), b_cnt2 = round_up(ns * g.F * 4, 256);
    const size_t b_spill = round_up(g.spill_cap * 8, 256), b_ctr = 256;
    KV_HIP(g_scratch.need(b_buf1 + b_buf2 + b_cnt1 + b_cnt2 + b_spill + b_ctr));
    unsigned char *base = (unsigned char *)g_scratch.p;
    g.gbuf1 = (uint32_t *)base; base += b_buf1;
    g.gbuf2 = (uint16_t *)base; base += b_buf2;
    g.gcnt1 = (uint32_t *)base; base += b_cnt1;
    g.gcnt2 = (uint32_t *)base; base += b_cnt2;
    g.spill = (unsigned long long *)base; base += b_spill;
    g.ctr = (unsigned long long *)base;
    KV_HIP(hipMemsetAsync(g.gcnt1, 0, b_cnt1 + b_cnt2, st));
    KV_HIP(hipMemsetAsync(g.ctr, 0, b_ctr, st));

    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const SketchDev *d_mask = mask ? mask->d_desc : nullptr;
    {
        KvProfScope prof("k_bin_hash");
        const unsigned grid = std::min<unsigned>(reads->n_tiles, (unsigned)cus);
        hipLaunchKernelGGL(k_bin_hash, dim3(grid), dim3(BIN_A_THREADS), 0, st, reads_dev(reads), reads->n_tiles,
                           (const SketchDev *)s->d_desc, d_mask, filter, g);
    }
    {
        KvProfScope prof("k_bin_split");
        const unsigned chunks = (unsigned)((g.cap1 + BIN_B_CHUNK - 1) / BIN_B_CHUNK);
        const size_t lds = (size_t)g.F * g.ringB * 2 + (size_t)g.F * 5 * 4;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_bin_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_bin_split, dim3(chunks, (unsigned)ns), dim3(BIN_B_THREADS), lds, st, g);
    }
    KV_HIP(hipGetLastError());
    unsigned long long ctr[4] = {0, 0, 0, 0};
    KV_HIP(hipMemcpyAsync(ctr, g.ctr, sizeof(ctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    if (ctr[1] != 0) {
        kv_set_error("partitioned count: spill list overflow (%llu items)", ctr[0]);
        return KV_ERR_CAPACITY;
    }
    {
        KvProfScope prof("k_bin_apply");
        hipLaunchKernelGGL(k_bin_apply, dim3(maxsl, (unsigned)g.T), dim3(BIN_C_THREADS), 0, st, (const SketchDev *)s->d_desc, g);
    }
    if (ctr[0] > 0) {
        KvProfScope prof("k_bin_spill");
        const unsigned grid = (unsigned)std::min<uint64_t>((ctr[0] + 255) / 256, 2048);
        hipLaunchKernelGGL(k_bin_spill, dim3(grid), dim3(256), 0, st, (const SketchDev *)s->d_desc, g);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(ctr, g.ctr, sizeof(ctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    *n_added = ctr[2];
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
