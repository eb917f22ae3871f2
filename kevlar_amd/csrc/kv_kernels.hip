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

#include "kv_device.h"

namespace {

// ---------------------------------------------------------------------------------------
// K2 consume: sketch.consume_seqfile[_banding][_with_mask]  (kevlar/count.py:43-71)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(KV_TILE_THREADS) void k_consume(ReadsDev rd, const SketchDev *__restrict__ sk,
                                                            const SketchDev *__restrict__ mask, ConsumeFilter p,
                                                            uint64_t *counters)
{
    __shared__ TileShared sh;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 0, 0, read0);
    const uint32_t total = sh.kpre[nr];
    uint64_t n_added = 0, n_new = 0;
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
    n_added = wave_sum_u64(n_added);
    n_new = wave_sum_u64(n_new);
    if ((threadIdx.x & 63) == 0) {
        if (n_added) atomicAdd((unsigned long long *)&counters[0], (unsigned long long)n_added);
        if (n_new) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)n_new);
    }
}

// ---------------------------------------------------------------------------------------
// K3 novel scan: novel() + kmer_is_interesting()  (kevlar/novel.py:21-53,123-169)
// ---------------------------------------------------------------------------------------
struct NovelParams {
    HashParams hp;
    int ncase, nctrl;
    const SketchDev *sk[KV_MAX_SAMPLES];  // cases first, then controls
    int case_min, ctrl_max, screen;
    int band_mode, nbands, band;
    uint64_t band_lo, band_hi;
    uint64_t first_read;
    uint32_t cap_hits;
    uint32_t *hit_read, *hit_off;
    uint8_t *hit_abund;
    uint8_t *disc_flag;   // per read: dropped by the abundance screen
    uint32_t *mask;
    uint64_t mask_stride;
};

__global__ __launch_bounds__(KV_TILE_THREADS) void k_novel(ReadsDev rd, NovelParams p, uint64_t *counters)
{
    __shared__ TileShared sh;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 1, p.first_read, read0);
    const uint32_t total = sh.kpre[nr];
    const int S = p.ncase + p.nctrl;
    for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
        uint32_t r, i;
        locate_kmer(sh, nr, q, r, i);
        const uint32_t fwd = sh.foff[r] + i;
        const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.hp.k - i);
        const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
        if (p.band_mode == KV_BAND_RANGE && !(h >= p.band_lo && h < p.band_hi)) continue;
        if (p.band_mode == KV_BAND_REFQUIRK &&
            (h & (uint64_t)(p.nbands - 1)) != (uint64_t)(int64_t)(p.band - 1)) continue;

        bool interesting = true, discard = false;
        if (p.screen > 0) {
            // reference order: cases (in order, full Count-Min minimum), stop at the first
            // failing case and test it against the screen threshold (novel.py:36-44)
            for (int c = 0; c < p.ncase && interesting; ++c) {
                const int a = (int)sketch_get(p.sk[c], h);
                if (a < p.case_min) { interesting = false; discard = a < p.screen; }
            }
            for (int c = 0; c < p.nctrl && interesting; ++c)
                if ((int)sketch_get(p.sk[p.ncase + c], h) > p.ctrl_max) interesting = false;
        } else {
            // same predicate, cheapest evidence first: a control passes as soon as ONE table
            // is <= ctrl_max (the minimum is then <= ctrl_max); a case fails as soon as ONE
            // table is < case_min.  Typical inherited k-mer: rejected after 4 loads.
            for (int c = 0; c < p.nctrl && interesting; ++c) {
                const SketchDev *s = p.sk[p.ncase + c];
                bool pass = false;
                for (int t = 0; t < s->ntables && !pass; ++t) pass = (int)table_get(s, t, h) <= p.ctrl_max;
                interesting = pass;
            }
            for (int c = 0; c < p.ncase && interesting; ++c) {
                const SketchDev *s = p.sk[c];
                for (int t = 0; t < s->ntables && interesting; ++t) interesting = (int)table_get(s, t, h) >= p.case_min;
            }
        }
        const uint32_t gread = read0 + r;
        if (discard) {
            p.disc_flag[gread] = 1;   // any number of k-mers may flag the same read: plain store
            continue;
        }
        if (!interesting) continue;
        const unsigned long long slot = atomicAdd((unsigned long long *)&counters[2], 1ull);
        if (slot < p.cap_hits) {
            p.hit_read[slot] = gread;
            p.hit_off[slot] = i;
            for (int c = 0; c < S; ++c) p.hit_abund[slot * (uint64_t)S + c] = (uint8_t)sketch_get(p.sk[c], h);
        }
        if (p.mask) {
            const uint64_t bit = (uint64_t)gread * p.mask_stride + i;
            atomicOr(&p.mask[bit >> 5], 1u << (bit & 31));
        }
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

__global__ void k_add_hashes(const SketchDev *__restrict__ sk, const uint64_t *hashes, uint64_t n, uint8_t *is_new,
                             uint64_t *counters)
{
    uint64_t n_new = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const bool nw = sketch_add(sk, hashes[i]);
        if (is_new) is_new[i] = nw ? 1 : 0;
        n_new += nw ? 1 : 0;
    }
    n_new = wave_sum_u64(n_new);
    if ((threadIdx.x & 63) == 0 && n_new) atomicAdd((unsigned long long *)&counters[1], (unsigned long long)n_new);
}

template <bool RC>
__device__ __forceinline__ uint32_t kmer_byte(const uint8_t *km, int k, int j)
{
    if (!RC) return km[j];
    const uint32_t c = km[k - 1 - j];
    return c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : (c == 'T' ? 'A' : 'N')));
}

template <bool RC>
__device__ uint64_t murmur_global(const uint8_t *km, int k)
{
    uint64_t h1 = 0, h2 = 0;
    const int nblocks = k / 16, rem = k & 15;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1 = 0, k2 = 0;
        for (int j = 7; j >= 0; --j) {
            k1 = (k1 << 8) | kmer_byte<RC>(km, k, 16 * b + j);
            k2 = (k2 << 8) | kmer_byte<RC>(km, k, 16 * b + 8 + j);
        }
        mm_block(h1, h2, k1, k2);
    }
    uint64_t k1 = 0, k2 = 0;
    for (int j = rem - 1; j >= 8; --j) k2 = (k2 << 8) | kmer_byte<RC>(km, k, 16 * nblocks + j);
    if (rem > 8) { k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2; }
    for (int j = (rem > 8 ? 8 : rem) - 1; j >= 0; --j) k1 = (k1 << 8) | kmer_byte<RC>(km, k, 16 * nblocks + j);
    if (rem > 0) { k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1; }
    return mm_final(h1, h2, k);
}

__global__ void k_hash_kmers(const uint8_t *kmers, int k, uint64_t n, int hashfam, uint64_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t *km = kmers + i * (uint64_t)k;
        uint64_t h;
        if (hashfam == HF_TWOBIT) {
            uint64_t f = 0, r = 0;
            for (int j = 0; j < k; ++j) {
                uint32_t x = (kmer_byte<false>(km, k, j) >> 1) & 3u;
                x ^= ((x ^ (x >> 1)) & 1u) * 3u;
                f = (f << 2) | x;
                uint32_t y = (kmer_byte<true>(km, k, j) >> 1) & 3u;
                y ^= ((y ^ (y >> 1)) & 1u) * 3u;
                r = (r << 2) | y;
            }
            h = f < r ? f : r;
        } else {
            h = murmur_global<false>(km, k) ^ murmur_global<true>(km, k);
        }
        out[i] = h;
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
    std::lock_guard<std::mutex> lk(s->mu);
    const ConsumeFilter p = make_consume_filter(s->h.ksize, s->h.hashfam, nbands, band, mask != nullptr, threshold,
                                                consume_masked);
    KV_HIP(hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), kv_stream()));
    if (reads->n_tiles > 0) {
        KvProfScope prof("k_consume");
        hipLaunchKernelGGL(k_consume, dim3(reads->n_tiles), dim3(KV_TILE_THREADS), 0, kv_stream(), reads_dev(reads),
                           (const SketchDev *)s->d_desc, (const SketchDev *)(mask ? mask->d_desc : nullptr), p,
                           s->d_counters);
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

extern "C" int kv_hash_kmers(int kind, const char *kmers, int k, uint64_t n, uint64_t *hashes_out)
{
    KV_REQUIRE((kmers && hashes_out) || n == 0, KV_ERR_ARG, "kv_hash_kmers: null argument");
    KV_REQUIRE(k >= 1 && k <= KV_MAX_K, KV_ERR_ARG, "k=%d out of range", k);
    const int fam = kv_hashfam_of(kind);
    KV_REQUIRE(fam != HF_TWOBIT || k <= 32, KV_ERR_ARG, "graph sketches need k <= 32 (got %d)", k);
    if (n == 0) return KV_OK;
    uint8_t *d_km = nullptr;
    uint64_t *d_h = nullptr;
    KV_HIP(hipMalloc((void **)&d_km, n * (uint64_t)k));
    hipError_t e = hipMalloc((void **)&d_h, n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_km, kmers, n * (uint64_t)k, hipMemcpyHostToDevice, kv_stream());
    if (e == hipSuccess) {
        KvProfScope prof("k_hash_kmers");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_hash_kmers, dim3(grid), dim3(256), 0, kv_stream(), d_km, k, n, fam, d_h);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hashes_out, d_h, n * 8, hipMemcpyDeviceToHost, kv_stream());
    if (e == hipSuccess) e = hipStreamSynchronize(kv_stream());
    (void)hipFree(d_km);
    if (d_h) (void)hipFree(d_h);
    KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "kv_hash_kmers failed: %s", hipGetErrorString(e));
    return KV_OK;
}

extern "C" int kv_get_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *counts_out)
{
    KV_REQUIRE(s && ((hashes && counts_out) || n == 0), KV_ERR_ARG, "kv_get_hashes: null argument");
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    uint64_t *d_h = nullptr;
    uint8_t *d_o = nullptr;
    KV_HIP(hipMalloc((void **)&d_h, n * 8));
    hipError_t e = hipMalloc((void **)&d_o, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_h, hashes, n * 8, hipMemcpyHostToDevice, kv_stream());
    if (e == hipSuccess) {
        KvProfScope prof("k_get_hashes");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_get_hashes, dim3(grid), dim3(256), 0, kv_stream(), (const SketchDev *)s->d_desc, d_h, n, d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(counts_out, d_o, n, hipMemcpyDeviceToHost, kv_stream());
    if (e == hipSuccess) e = hipStreamSynchronize(kv_stream());
    (void)hipFree(d_h);
    if (d_o) (void)hipFree(d_o);
    KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "kv_get_hashes failed: %s", hipGetErrorString(e));
    return KV_OK;
}

extern "C" int kv_add_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *is_new_out)
{
    KV_REQUIRE(s && (hashes || n == 0), KV_ERR_ARG, "kv_add_hashes: null argument");
    if (n == 0) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    uint64_t *d_h = nullptr;
    uint8_t *d_o = nullptr;
    KV_HIP(hipMalloc((void **)&d_h, n * 8));
    hipError_t e = hipSuccess;
    if (is_new_out) e = hipMalloc((void **)&d_o, n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_h, hashes, n * 8, hipMemcpyHostToDevice, kv_stream());
    if (e == hipSuccess) e = hipMemsetAsync(s->d_counters, 0, 2 * sizeof(uint64_t), kv_stream());
    if (e == hipSuccess) {
        KvProfScope prof("k_add_hashes");
        const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
        hipLaunchKernelGGL(k_add_hashes, dim3(grid), dim3(256), 0, kv_stream(), (const SketchDev *)s->d_desc, d_h, n, d_o,
                           s->d_counters);
        e = hipGetLastError();
    }
    uint64_t c[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost, kv_stream());
    if (e == hipSuccess && is_new_out) e = hipMemcpyAsync(is_new_out, d_o, n, hipMemcpyDeviceToHost, kv_stream());
    if (e == hipSuccess) e = hipStreamSynchronize(kv_stream());
    (void)hipFree(d_h);
    if (d_o) (void)hipFree(d_o);
    KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "kv_add_hashes failed: %s", hipGetErrorString(e));
    s->n_unique += c[1];
    s->occ_dirty = true;
    return KV_OK;
}

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
        KV_REQUIRE(s->h.ksize == k && s->h.hashfam == fam, KV_ERR_ARG,
                   "all sketches of one scan must share k and hash function");
        p.sk[c] = s->d_desc;
    }
    p.hp = make_hash_params(k, fam);
    p.ncase = ncase; p.nctrl = nctrl;
    p.case_min = case_min; p.ctrl_max = ctrl_max; p.screen = screen_thresh > 0 ? screen_thresh : 0;
    p.band_mode = band_mode; p.nbands = nbands; p.band = band;
    if (band_mode == KV_BAND_RANGE) kv_band_bounds(nbands, band, &p.band_lo, &p.band_hi);
    p.first_read = first_read;
    p.mask = d_mask; p.mask_stride = mask_stride;
    const int S = ncase + nctrl;

    kv_hits *hits = new kv_hits();
    hits->nsamples = S;
    uint64_t *d_cnt = nullptr;
    uint64_t cap = 1u << 20;
    const uint64_t nflags = reads->n_reads ? reads->n_reads : 1;
    hipError_t e = hipMalloc((void **)&d_cnt, 4 * sizeof(uint64_t));
    if (e == hipSuccess && p.screen > 0) e = hipMalloc((void **)&p.disc_flag, nflags);
    uint64_t c[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 2 && e == hipSuccess; ++attempt) {
        p.cap_hits = (uint32_t)cap;
        e = hipMalloc((void **)&p.hit_read, cap * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&p.hit_off, cap * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&p.hit_abund, cap * (uint64_t)S);
        if (e == hipSuccess && p.disc_flag) e = hipMemsetAsync(p.disc_flag, 0, nflags, kv_stream());
        if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 4 * sizeof(uint64_t), kv_stream());
        if (e == hipSuccess && reads->n_tiles > 0) {
            KvProfScope prof("k_novel");
            hipLaunchKernelGGL(k_novel, dim3(reads->n_tiles), dim3(KV_TILE_THREADS), 0, kv_stream(), reads_dev(reads), p, d_cnt);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(c, d_cnt, sizeof(c), hipMemcpyDeviceToHost, kv_stream());
        if (e == hipSuccess) e = hipStreamSynchronize(kv_stream());
        if (e == hipSuccess && c[2] <= cap) {
            hits->read.resize(c[2]); hits->offset.resize(c[2]); hits->abund.resize(c[2] * (uint64_t)S);
            if (c[2]) {
                e = hipMemcpy(hits->read.data(), p.hit_read, c[2] * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(hits->offset.data(), p.hit_off, c[2] * 4, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpy(hits->abund.data(), p.hit_abund, c[2] * (uint64_t)S, hipMemcpyDeviceToHost);
            }
            if (e == hipSuccess && p.disc_flag && reads->n_reads) {
                std::vector<uint8_t> flags(reads->n_reads);
                e = hipMemcpy(flags.data(), p.disc_flag, reads->n_reads, hipMemcpyDeviceToHost);
                for (uint64_t i = 0; i < reads->n_reads; ++i)
                    if (flags[i]) hits->discarded.push_back((uint32_t)i);
            }
            attempt = 2;
        } else if (e == hipSuccess) {
            cap = c[2];  // second pass with exactly enough room (mask bits and flags are idempotent)
        }
        (void)hipFree(p.hit_read); (void)hipFree(p.hit_off); (void)hipFree(p.hit_abund);
        p.hit_read = p.hit_off = nullptr; p.hit_abund = nullptr;
    }
    if (p.disc_flag) (void)hipFree(p.disc_flag);
    if (d_cnt) (void)hipFree(d_cnt);
    if (e != hipSuccess) {
        delete hits;
        kv_set_error("kv_novel_scan failed: %s", hipGetErrorString(e));
        return KV_ERR_HIP;
    }
    // canonical order: by read, then offset; a discarded read drops all of its hits
    {
        std::vector<uint32_t> &disc = hits->discarded;   // ascending by construction
        const uint64_t n = hits->read.size();
        std::vector<uint64_t> order(n);
        for (uint64_t i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
            if (hits->read[a] != hits->read[b]) return hits->read[a] < hits->read[b];
            return hits->offset[a] < hits->offset[b];
        });
        std::vector<uint32_t> rr, oo;
        std::vector<uint8_t> aa;
        rr.reserve(n); oo.reserve(n); aa.reserve(n * (uint64_t)S);
        for (uint64_t j = 0; j < n; ++j) {
            const uint64_t i = order[j];
            if (!disc.empty() && std::binary_search(disc.begin(), disc.end(), hits->read[i])) continue;
            rr.push_back(hits->read[i]);
            oo.push_back(hits->offset[i]);
            aa.insert(aa.end(), hits->abund.begin() + i * (uint64_t)S, hits->abund.begin() + (i + 1) * (uint64_t)S);
        }
        hits->read.swap(rr); hits->offset.swap(oo); hits->abund.swap(aa);
    }
    *out = hits;
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

extern "C" int kv_hits_destroy(kv_hits *h)
{
    delete h;
    return KV_OK;
}
