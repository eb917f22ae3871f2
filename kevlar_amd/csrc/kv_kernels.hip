// kv_kernels.hip -- gfx950 kernels of the count / novel hot path and their launchers.
//
// Work decomposition (all hashing kernels):
//   one 256-thread workgroup per TILE of up to 128 consecutive reads.  The tile's 2-bit
//   packed words are loaded coalesced from HBM and expanded ONCE into ASCII in LDS, forward
//   strand and reverse complement (murmur hashes the ASCII k-mer, H1 in SURVEY.md 8(a));
//   then the tile's k-mers are spread flat over the 256 threads.  A k-mer's two 64-bit
//   hashes read their 16-byte murmur blocks straight out of LDS with dword loads +
//   v_alignbyte_b32, so the kernel is generic in k (1..255) and in read length.
//
// No MFMA anywhere: this is 64-bit integer hashing plus random byte-granular table access.
#include <algorithm>
#include <map>

#include "kv_binned.h"
#include "kv_device.h"

namespace {

// ---------------------------------------------------------------------------------------
// K2 consume: sketch.consume_seqfile[_banding][_with_mask]  (kevlar/count.py:43-71)
// ---------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(KV_TILE_THREADS) void k_consume(ReadsDev rd, const SketchDev *__restrict__ sk,
                                                            const SketchDev *__restrict__ mask, ConsumeFilter p,
                                                            uint64_t *counters)
{
    __shared__ TileShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)tile_smem;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 0, 0, read0);
    const uint32_t total = sh.kpre[nr];
    uint64_t n_added = 0, n_new = 0;
    if (NW > 0) {
        // every thread owns a run of consecutive k-mers and rolls its register windows along it
        const uint32_t run = (total + blockDim.x - 1) / blockDim.x;
        const uint32_t q0 = threadIdx.x * run, q1 = min(total, q0 + run);
        if (q0 < q1) {
            KmerRoll<(NW > 0 ? NW : 8)> w;
            locate_kmer(sh, nr, q0, w.r, w.i);
            roll_load(w, sh, p.hp.k);
            for (uint32_t q = q0; q < q1; ++q) {
                const uint64_t h = roll_hash(w, p.hp);
                if (consume_filter_pass(p, mask, h)) {
                    n_new += sketch_add(sk, h) ? 1 : 0;
                    n_added += 1;
                }
                if (q + 1 < q1) roll_step(w, sh, nr, p.hp.k);
            }
        }
    } else {
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            uint32_t r, i;
            locate_kmer(sh, nr, q, r, i);
            const uint32_t fwd = sh.foff[r] + i;
            const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.hp.k - i);
            const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
            if (!consume_filter_pass(p, mask, h)) continue;
            n_new += sketch_add(sk, h) ? 1 : 0;
            n_added += 1;
        }
    }
    n_added = wave_sum_u64(n_added);
    n_new = wave_sum_u64(n_new);
    if ((threadIdx.x & 63) == 0) {
        if (n_added) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)n_added);
        if (n_new) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)n_new);
    }
}

// ---------------------------------------------------------------------------------------
// point queries on hash arrays, k-mer string hashing, occupancy
// ---------------------------------------------------------------------------------------
__global__ void k_get_hashes(const SketchDev *__restrict__ sk, const uint64_t *hashes, uint64_t n, uint8_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint8_t)sketch_get(sk, hashes[i]);
}

__global__ void k_add_hashes(const SketchDev *__restrict__ sk, const uint64_t *hashes, uint64_t n, uint32_t stride,
                             uint8_t *is_new, uint64_t *counters)
{
    uint64_t n_new = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool nw = sketch_add(sk, hashes[i * stride]);
        if (is_new) is_new[i] = nw ? 1 : 0;
        n_new += nw ? 1 : 0;
    }
    n_new = wave_sum_u64(n_new);
    if ((threadIdx.x & 63) == 0 && n_new) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)n_new);
}

// items = (hash, count) pairs
__global__ void k_add_hashes_weighted(const SketchDev *__restrict__ sk, const uint64_t *items, uint64_t n, uint64_t *counters)
{
    uint64_t n_new = 0, total = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t h = items[2 * i], count = items[2 * i + 1];
        const uint32_t weight = (uint32_t)min(count, (uint64_t)255);
        bool nw = false;
        if (weight)
            for (int t = 0; t < sk->ntables; ++t) nw |= table_add(sk, t, h, weight);
        n_new += nw ? 1 : 0;
        total += count;
    }
    n_new = wave_sum_u64(n_new);
    total = wave_sum_u64(total);
    if ((threadIdx.x & 63) == 0) {
        if (n_new) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)n_new);
        if (total) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)total);
    }
}

// MurmurHash3_x64_128 (low word) of k characters produced one at a time by `at(j)`
template <typename At>
__device__ uint64_t murmur_chars(At at, int k)
{
    uint64_t h1 = 0, h2 = 0;
    const int nblocks = k / 16, rem = k & 15;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1 = 0, k2 = 0;
        for (int j = 7; j >= 0; --j) {
            k1 = (k1 << 8) | at(16 * b + j);
            k2 = (k2 << 8) | at(16 * b + 8 + j);
        }
        mm_block(h1, h2, k1, k2);
    }
    uint64_t k1 = 0, k2 = 0;
    for (int j = rem - 1; j >= 8; --j) k2 = (k2 << 8) | at(16 * nblocks + j);
    if (rem > 8) { k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2; }
    for (int j = (rem > 8 ? 8 : rem) - 1; j >= 0; --j) k1 = (k1 << 8) | at(16 * nblocks + j);
    if (rem > 0) { k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1; }
    return mm_final(h1, h2, k);
}

// khmer's hash of a k-mer given as characters at(j) (forward) and rc(j) (reverse complement)
template <typename Fwd, typename Rev>
__device__ __forceinline__ uint64_t kmer_hash_chars(Fwd fwd, Rev rev, int k, int hashfam)
{
    if (hashfam == HF_TWOBIT) {
        uint64_t f = 0, r = 0;
        for (int j = 0; j < k; ++j) {
            uint32_t x = (fwd(j) >> 1) & 3u;
            x ^= ((x ^ (x >> 1)) & 1u) * 3u;
            f = (f << 2) | x;
            uint32_t y = (rev(j) >> 1) & 3u;
            y ^= ((y ^ (y >> 1)) & 1u) * 3u;
            r = (r << 2) | y;
        }
        return f < r ? f : r;
    }
    return murmur_chars(fwd, k) ^ murmur_chars(rev, k);
}

__device__ __forceinline__ uint32_t complement_char(uint32_t c)
{
    return c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : (c == 'T' ? 'A' : 'N')));
}

__global__ void k_hash_kmers(const uint8_t *kmers, int k, uint64_t n, int hashfam, uint64_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *km = kmers + i * (uint64_t)k;
        out[i] = kmer_hash_chars([&](int j) { return (uint32_t)km[j]; },
                                 [&](int j) { return complement_char(km[k - 1 - j]); }, k, hashfam);
    }
}

// hashes of the k-mers at (read, offset) positions of a packed batch: the annotated k-mers of filter and partition
// never leave the device as text
__global__ void k_hash_positions(const uint32_t *words, const uint64_t *woff, const uint32_t *ann_read, const uint32_t *ann_off,
                                 uint64_t n, int k, int hashfam, uint64_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t *w = words + woff[ann_read[i]];
        const uint32_t off = ann_off[i];
        auto base = [&](uint32_t p) { return (w[p >> 4] >> (2u * (p & 15u))) & 3u; };
        out[i] = kmer_hash_chars([&](int j) { return (0x54474341u >> (8u * base(off + (uint32_t)j))) & 0xffu; },                    // "ACGT"
                                 [&](int j) { return (0x41434754u >> (8u * base(off + (uint32_t)(k - 1 - j)))) & 0xffu; },          // "TGCA"
                                 k, hashfam);
    }
}

// K4: n_occupied = non-zero bins of table 0 (kevlar/sketch.py:62-74 estimate_fpr input)
__global__ void k_occupancy(const uint32_t *tab, uint64_t nwords, int storage, uint64_t *out)
{
    uint64_t n = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t w = tab[i];
        if (storage == ST_BIT) {
            n += __popc(w);
        } else if (storage == ST_BYTE) {
            n += ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
        } else {
            uint32_t x = w | (w >> 1);
            x |= x >> 2;
            n += __popc(x & 0x11111111u);
        }
    }
    n = wave_sum_u64(n);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd((unsigned long long *)out, (unsigned long long)n);
}

}  // namespace

// ---------------------------------------------------------------------------------------
// launchers (C ABI)
// ---------------------------------------------------------------------------------------
int kv_sketch_refresh_occupancy(kv_sketch *s)
{
    KV_HIP(hipMemsetAsync(&s->d_counters[3], 0, sizeof(uint64_t), kv_stream()));
    const uint64_t nwords = s->alloc_bytes[0] / 4;
    {
        KvProfScope prof("k_occupancy");
        const unsigned grid = (unsigned)std::min<uint64_t>((nwords + 255) / 256, 2048);
        hipLaunchKernelGGL(k_occupancy, dim3(grid ? grid : 1), dim3(256), 0, kv_stream(), (const uint32_t *)s->h.tab[0],
                           nwords, s->h.storage, &s->d_counters[3]);
    }
    KV_HIP(hipGetLastError());
    uint64_t v = 0;
    KV_HIP(hipMemcpyAsync(&v, &s->d_counters[3], sizeof(v), hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    s->n_occupied = v;
    s->occ_dirty = false;
    return KV_OK;
}

extern "C" int kv_consume(kv_sketch *s, const kv_reads *reads, int nbands, int band, const kv_sketch *mask,
                          int threshold, int consume_masked, uint64_t *n_kmers_out)
{
    KV_REQUIRE(s && reads, KV_ERR_ARG, "kv_consume: null handle");
    KV_REQUIRE(nbands >= 0 && (nbands == 0 || (band >= 0 && band < nbands)), KV_ERR_ARG,
               "band %d out of range for %d bands", band, nbands);
    { const int rc = kv_sketch_ready(mask); if (rc != KV_OK) return rc; }
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    const ConsumeFilter p = make_consume_filter(s->h.ksize, s->h.hashfam, nbands, band, mask != nullptr, threshold,
                                                consume_masked);
    uint64_t n_kmers = 0;
    kv_reads_num_kmers(reads, s->h.ksize, &n_kmers);
    if (kv_binned_eligible(s, reads, n_kmers, nbands)) {
        uint64_t added = 0;
        // large batches of short k: count every distinct k-mer once with its multiplicity (kv_skm.hip); the
        // one-item-per-k-mer partition is its fallback, the atomic kernel the fallback of both
        if (kv_skm_eligible(s, reads, n_kmers, false)) {
            const int rc = kv_consume_skm(s, reads, p, mask, n_kmers, nbands, &added);
            if (rc == KV_OK) {
                if (n_kmers_out) *n_kmers_out = added;
                return KV_OK;
            }
            if (rc != KV_ERR_CAPACITY) return rc;
        }
        const int rc = kv_consume_binned(s, reads, nullptr, 1, p, mask, n_kmers, nbands, &added);
        if (rc == KV_OK) {
            if (n_kmers_out) *n_kmers_out = added;
            return KV_OK;
        }
        if (rc != KV_ERR_CAPACITY) return rc;   // capacity: tables untouched, fall through to the atomic kernel
    }
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), kv_stream()));
    if (reads->n_tiles > 0) {
        KvProfScope prof("k_consume");
        const SketchDev *dm = mask ? mask->d_desc : nullptr;
        const unsigned lds = reads->tile_lds_bytes;
        const bool roll = s->h.hashfam == HF_MURMUR && !kv_knob("KV_NO_ROLL");
#define KV_LAUNCH_CONSUME(NW_)                                                                                   \
        do {                                                                                                     \
            kv_ensure_dynamic_lds((const void *)k_consume<NW_>, lds);                                            \
            hipLaunchKernelGGL((k_consume<NW_>), dim3(reads->n_tiles), dim3(KV_TILE_THREADS), lds, kv_stream(), \
                               reads_dev(reads), (const SketchDev *)s->d_desc, dm, p, s->d_counters);            \
        } while (0)
        if (roll && s->h.ksize <= 32) KV_LAUNCH_CONSUME(8);
        else if (roll && s->h.ksize <= 64) KV_LAUNCH_CONSUME(16);
        else KV_LAUNCH_CONSUME(0);
#undef KV_LAUNCH_CONSUME
    }
    KV_HIP(hipGetLastError());
    uint64_t c[2] = {0, 0};
    KV_HIP(hipMemcpyAsync(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    s->n_unique += c[1];
    s->occ_dirty = true;
    if (n_kmers_out) *n_kmers_out = c[0];
    return KV_OK;
}

namespace {
// grow-only device scratch of the point-query entry points, one per stream (hipMalloc + hipFree per call cost more
// than the kernels they wrap when filter / simlike issue thousands of small queries)
struct PointScratch { KvArena arena; std::mutex mu; };
std::map<hipStream_t, PointScratch> g_point_scratch;
std::mutex g_point_scratch_mu;
PointScratch &point_scratch()
{
    std::lock_guard<std::mutex> lk(g_point_scratch_mu);
    return g_point_scratch[kv_stream_key(kv_stream())];
}
inline size_t pad256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace

extern "C" int kv_hash_positions(const kv_reads *reads, int kind, int ksize, const uint32_t *ann_read, const uint32_t *ann_offset,
                                 uint64_t n, uint64_t *hashes_out)
{
    KV_REQUIRE(reads && ((ann_read && ann_offset && hashes_out) || n == 0), KV_ERR_ARG, "kv_hash_positions: null argument");
    KV_REQUIRE(ksize >= 1 && ksize <= KV_MAX_K, KV_ERR_ARG, "k=%d out of range", ksize);
    const int fam = kv_hashfam_of(kind);
    KV_REQUIRE(fam != HF_TWOBIT || ksize <= 32, KV_ERR_ARG, "graph sketches need k <= 32 (got %d)", ksize);
    if (n == 0) return KV_OK;
    for (uint64_t i = 0; i < n; ++i) {
        KV_REQUIRE(ann_read[i] < reads->n_reads && (uint64_t)ann_offset[i] + (uint64_t)ksize <= reads->h_len[ann_read[i]], KV_ERR_ARG,
                   "kv_hash_positions: annotation %llu (read %u, offset %u) does not lie inside its read", (unsigned long long)i,
                   ann_read[i], ann_offset[i]);
    }
    PointScratch &ps = point_scratch();
    std::lock_guard<std::mutex> lk(ps.mu);
    hipStream_t st = kv_stream();
    KV_HIP(ps.arena.need(2 * pad256(n * 4) + pad256(n * 8)));
    uint32_t *d_r = (uint32_t *)ps.arena.p, *d_o = (uint32_t *)((char *)ps.arena.p + pad256(n * 4));
    uint64_t *d_h = (uint64_t *)((char *)ps.arena.p + 2 * pad256(n * 4));
    KV_HIP(hipMemcpyAsync(d_r, ann_read, n * 4, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemcpyAsync(d_o, ann_offset, n * 4, hipMemcpyHostToDevice, st));
    {
        KvProfScope prof("k_hash_positions");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_hash_positions, dim3(grid), dim3(256), 0, st, (const uint32_t *)reads->d_words, (const uint64_t *)reads->d_woff,
                           (const uint32_t *)d_r, (const uint32_t *)d_o, n, ksize, fam, d_h);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(hashes_out, d_h, n * 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

extern "C" int kv_hash_kmers(int kind, const char *kmers, int k, uint64_t n, uint64_t *hashes_out)
{
    KV_REQUIRE((kmers && hashes_out) || n == 0, KV_ERR_ARG, "kv_hash_kmers: null argument");
    KV_REQUIRE(k >= 1 && k <= KV_MAX_K, KV_ERR_ARG, "k=%d out of range", k);
    const int fam = kv_hashfam_of(kind);
    KV_REQUIRE(fam != HF_TWOBIT || k <= 32, KV_ERR_ARG, "graph sketches need k <= 32 (got %d)", k);
    if (n == 0) return KV_OK;
    PointScratch &ps = point_scratch();
    std::lock_guard<std::mutex> lk(ps.mu);
    hipStream_t st = kv_stream();
    KV_HIP(ps.arena.need(pad256(n * (uint64_t)k) + pad256(n * 8)));
    uint8_t *d_km = (uint8_t *)ps.arena.p;
    uint64_t *d_h = (uint64_t *)((char *)ps.arena.p + pad256(n * (uint64_t)k));
    KV_HIP(hipMemcpyAsync(d_km, kmers, n * (uint64_t)k, hipMemcpyHostToDevice, st));
    {
        KvProfScope prof("k_hash_kmers");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_hash_kmers, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_km, k, n, fam, d_h);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(hashes_out, d_h, n * 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

extern "C" int kv_get_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *counts_out)
{
    KV_REQUIRE(s && ((hashes && counts_out) || n == 0), KV_ERR_ARG, "kv_get_hashes: null argument");
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    PointScratch &ps = point_scratch();
    std::lock_guard<std::mutex> slk(ps.mu);
    hipStream_t st = kv_stream();
    KV_HIP(ps.arena.need(pad256(n * 8) + pad256(n)));
    uint64_t *d_h = (uint64_t *)ps.arena.p;
    uint8_t *d_o = (uint8_t *)ps.arena.p + pad256(n * 8);
    KV_HIP(hipMemcpyAsync(d_h, hashes, n * 8, hipMemcpyHostToDevice, st));
    {
        KvProfScope prof("k_get_hashes");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_get_hashes, dim3(grid), dim3(256), 0, st, (const SketchDev *)s->d_desc, (const uint64_t *)d_h, n, d_o);
    }
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(counts_out, d_o, n, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

extern "C" int kv_add_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *is_new_out)
{
    KV_REQUIRE(s && (hashes || n == 0), KV_ERR_ARG, "kv_add_hashes: null argument");
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    PointScratch &ps = point_scratch();
    std::lock_guard<std::mutex> slk(ps.mu);
    hipStream_t st = kv_stream();
    KV_HIP(ps.arena.need(pad256(n * 8) + pad256(n)));
    uint64_t *d_h = (uint64_t *)ps.arena.p;
    uint8_t *d_o = is_new_out ? (uint8_t *)ps.arena.p + pad256(n * 8) : nullptr;
    KV_HIP(hipMemcpyAsync(d_h, hashes, n * 8, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), st));
    {
        KvProfScope prof("k_add_hashes");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_add_hashes, dim3(grid), dim3(256), 0, st, (const SketchDev *)s->d_desc, (const uint64_t *)d_h, n, 1u, d_o,
                           s->d_counters);
    }
    KV_HIP(hipGetLastError());
    uint64_t c[2] = {0, 0};
    KV_HIP(hipMemcpyAsync(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost, st));
    if (is_new_out) KV_HIP(hipMemcpyAsync(is_new_out, d_o, n, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    s->n_unique += c[1];
    s->occ_dirty = true;
    return KV_OK;
}

// Count n hashes that are already in HBM (element i at d_hashes[i * stride_words]): the receive side of
// the read-sharded multi-GPU count, where the hashes of this rank's band were computed by the ranks
// that hold the reads (kv_route_hashes).  Same partitioned path as kv_consume for large lists.
extern "C" int kv_consume_hashes(kv_sketch *s, const void *d_hashes, uint64_t n, uint32_t stride_words, uint64_t *n_added_out)
{
    KV_REQUIRE(s && (d_hashes || n == 0) && stride_words >= 1, KV_ERR_ARG, "kv_consume_hashes: bad argument");
    if (n_added_out) *n_added_out = n;
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    if (kv_binned_eligible(s, nullptr, n, 0)) {
        const ConsumeFilter p = make_consume_filter(s->h.ksize, s->h.hashfam, 0, 0, false, 0, 0);
        uint64_t added = 0;
        const int rc = kv_consume_binned(s, nullptr, (const uint64_t *)d_hashes, stride_words, p, nullptr, n, 0, &added);
        if (rc == KV_OK) return KV_OK;
        if (rc != KV_ERR_CAPACITY) return rc;
    }
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), kv_stream()));
    {
        KvProfScope prof("k_add_hashes");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_add_hashes, dim3(grid), dim3(256), 0, kv_stream(), (const SketchDev *)s->d_desc,
                           (const uint64_t *)d_hashes, n, stride_words, (uint8_t *)nullptr, s->d_counters);
    }
    KV_HIP(hipGetLastError());
    uint64_t c[2] = {0, 0};
    KV_HIP(hipMemcpyAsync(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    s->n_unique += c[1];
    s->occ_dirty = true;
    return KV_OK;
}

// The receive side once the sending ranks deduplicated their shards (kv_route_distinct): n items (hash, count); each adds
// min(count, 255) to its bins in one saturating add, which leaves the tables exactly as `count` single increments do.
// *n_added_out = sum of the counts = the k-mer occurrences the items stand for.
extern "C" int kv_consume_hashes_weighted(kv_sketch *s, const void *d_items, uint64_t n, uint64_t *n_added_out)
{
    KV_REQUIRE(s && (d_items || n == 0), KV_ERR_ARG, "kv_consume_hashes_weighted: bad argument");
    if (n_added_out) *n_added_out = 0;
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    if (kv_binned_eligible(s, nullptr, n, 0)) {
        const ConsumeFilter p = make_consume_filter(s->h.ksize, s->h.hashfam, 0, 0, false, 0, 0);
        uint64_t added = 0;
        const int rc = kv_consume_binned(s, nullptr, (const uint64_t *)d_items, 2, p, nullptr, n, 0, &added, true);
        if (rc == KV_OK) {
            if (n_added_out) *n_added_out = added;
            return KV_OK;
        }
        if (rc != KV_ERR_CAPACITY) return rc;
    }
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), kv_stream()));
    {
        KvProfScope prof("k_add_hashes_weighted");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 8192);
        hipLaunchKernelGGL(k_add_hashes_weighted, dim3(grid), dim3(256), 0, kv_stream(), (const SketchDev *)s->d_desc,
                           (const uint64_t *)d_items, n, s->d_counters);
    }
    KV_HIP(hipGetLastError());
    uint64_t c[2] = {0, 0};
    KV_HIP(hipMemcpyAsync(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    s->n_unique += c[1];
    s->occ_dirty = true;
    if (n_added_out) *n_added_out = c[0];
    return KV_OK;
}
