// kv_device.h -- device-side building blocks shared by the gfx950 kernels: MurmurHash3 over
// LDS-staged ASCII, Barrett modulo, table access, tile staging.  Include inside a .hip file.
#pragma once
#include "kv_internal.h"

namespace {


// ---------------------------------------------------------------------------------------
// arithmetic helpers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

#define MM_C1 0x87c37b91114253d5ULL
#define MM_C2 0x4cf5ad432745937fULL

__device__ __forceinline__ void mm_block(uint64_t &h1, uint64_t &h2, uint64_t k1, uint64_t k2)
{
    k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
    k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
    h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
}

__device__ __forceinline__ uint64_t mm_final(uint64_t h1, uint64_t h2, int len)
{
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    return h1 + h2;
}

// h % size: fastmod(h, size, magic) of kv_fastmod.h (included at the top)

// MurmurHash3_x64_128 (low word) of the k bytes at LDS byte address `a`
__device__ __forceinline__ uint64_t murmur_lds(const uint32_t *lds, uint32_t a, const HashParams &hp)
{
    const uint32_t sh = a & 3u;
    const uint32_t *p = lds + (a >> 2);
    uint64_t h1 = 0, h2 = 0;
    uint32_t d0 = p[0];
    for (int b = 0; b < hp.nblocks; ++b) {
        const uint32_t d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
        const uint64_t k1 = (uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sh) |
                            ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
        const uint64_t k2 = (uint64_t)__builtin_amdgcn_alignbyte(d3, d2, sh) |
                            ((uint64_t)__builtin_amdgcn_alignbyte(d4, d3, sh) << 32);
        mm_block(h1, h2, k1, k2);
        d0 = d4;
        p += 4;
    }
    if (hp.rem > 0) {
        const uint32_t d1 = p[1], d2 = p[2];
        if (hp.rem > 8) {
            const uint32_t d3 = p[3], d4 = p[4];
            uint64_t k2 = ((uint64_t)__builtin_amdgcn_alignbyte(d3, d2, sh) |
                           ((uint64_t)__builtin_amdgcn_alignbyte(d4, d3, sh) << 32)) & hp.m2;
            k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
        }
        uint64_t k1 = ((uint64_t)__builtin_amdgcn_alignbyte(d1, d0, sh) |
                       ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32)) & hp.m1;
        k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
    }
    return mm_final(h1, h2, hp.k);
}

__device__ __forceinline__ uint32_t lds_byte(const uint32_t *lds, uint32_t a)
{
    return (lds[a >> 2] >> (8u * (a & 3u))) & 0xffu;
}

// khmer *graph hash: 2 bits/base A=0 T=1 C=2 G=3, first base most significant, k <= 32
__device__ __forceinline__ uint64_t twobit_lds(const uint32_t *lds, uint32_t a, int k)
{
    uint64_t v = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t x = (lds_byte(lds, a + (uint32_t)j) >> 1) & 3u;  // A0 C1 T2 G3
        x ^= ((x ^ (x >> 1)) & 1u) * 3u;                            // swap C<->T: A0 T1 C2 G3
        v = (v << 2) | x;
    }
    return v;
}

__device__ __forceinline__ uint64_t kmer_hash_lds(const uint32_t *lds, uint32_t fwd, uint32_t rc, const HashParams &hp)
{
    if (hp.hashfam == HF_TWOBIT) {
        const uint64_t f = twobit_lds(lds, fwd, hp.k), r = twobit_lds(lds, rc, hp.k);
        return f < r ? f : r;
    }
    return murmur_lds(lds, fwd, hp) ^ murmur_lds(lds, rc, hp);
}

// ---------------------------------------------------------------------------------------
// table access
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t table_get(const SketchDev *s, int t, uint64_t h)
{
    const uint64_t bin = fastmod(h, s->size[t], s->magic[t]);
    const uint8_t *tab = s->tab[t];
    if (s->storage == ST_BYTE) return tab[bin];
    if (s->storage == ST_NIBBLE) return (tab[bin >> 1] >> ((bin & 1) ? 0 : 4)) & 15u;
    return (tab[bin >> 3] >> (bin & 7)) & 1u;
}

__device__ __forceinline__ uint32_t sketch_get(const SketchDev *s, uint64_t h)
{
    uint32_t best = 255;
    for (int t = 0; t < s->ntables; ++t) {
        const uint32_t v = table_get(s, t, h);
        best = v < best ? v : best;
    }
    return best;
}

// saturating increment of one bin; returns true if the bin was zero before OUR increment.
// There is no byte atomic on gfx950: CAS on the containing dword, issued at agent scope so
// the read side bypasses the (non-coherent) per-CU L1.
__device__ __forceinline__ bool table_inc(const SketchDev *s, int t, uint64_t h)
{
    const uint64_t bin = fastmod(h, s->size[t], s->magic[t]);
    uint8_t *tab = s->tab[t];
    if (s->storage == ST_BIT) {
        const uint32_t bit = 1u << (bin & 31);
        uint32_t *w = (uint32_t *)tab + (bin >> 5);
        const uint32_t old = atomicOr(w, bit);
        return (old & bit) == 0;
    }
    uint32_t *w;
    uint32_t shift, maxv;
    if (s->storage == ST_BYTE) {
        w = (uint32_t *)(tab + (bin & ~3ull));
        shift = (uint32_t)(bin & 3) * 8u;
        maxv = 255u;
    } else {
        const uint64_t byte = bin >> 1;
        w = (uint32_t *)(tab + (byte & ~3ull));
        shift = (uint32_t)(byte & 3) * 8u + ((bin & 1) ? 0u : 4u);
        maxv = 15u;
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

// `weight` saturating increments of one bin in one CAS (min(max, v + weight) == weight single increments)
__device__ __forceinline__ bool table_add(const SketchDev *s, int t, uint64_t h, uint32_t weight)
{
    const uint64_t bin = fastmod(h, s->size[t], s->magic[t]);
    uint8_t *tab = s->tab[t];
    if (s->storage == ST_BIT) {
        const uint32_t bit = 1u << (bin & 31);
        const uint32_t old = atomicOr((uint32_t *)tab + (bin >> 5), bit);
        return (old & bit) == 0;
    }
    uint32_t *w;
    uint32_t shift, maxv;
    if (s->storage == ST_BYTE) {
        w = (uint32_t *)(tab + (bin & ~3ull));
        shift = (uint32_t)(bin & 3) * 8u;
        maxv = 255u;
    } else {
        const uint64_t byte = bin >> 1;
        w = (uint32_t *)(tab + (byte & ~3ull));
        shift = (uint32_t)(byte & 3) * 8u + ((bin & 1) ? 0u : 4u);
        maxv = 15u;
    }
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const uint32_t cur = (old >> shift) & maxv;
        if (cur == maxv) return false;
        const uint32_t next = min(maxv, cur + weight);
        const uint32_t prev = atomicCAS(w, old, (old & ~(maxv << shift)) | (next << shift));
        if (prev == old) return cur == 0;
        old = prev;
    }
}

__device__ __forceinline__ bool sketch_add(const SketchDev *s, uint64_t h)
{
    bool is_new = false;
    for (int t = 0; t < s->ntables; ++t) is_new |= table_inc(s, t, h);
    return is_new;
}

// ---------------------------------------------------------------------------------------
// tile staging: packed words (HBM) -> ASCII forward + reverse complement (LDS)
// ---------------------------------------------------------------------------------------
// (the ASCII itself lives in dynamic LDS: kv_reads::tile_lds_bytes, so one long read can own a big tile)
struct TileShared {
    uint32_t *ascii;
    uint32_t foff[KV_TILE_MAX_READS];      // LDS byte offset of the forward strand
    uint32_t roff[KV_TILE_MAX_READS];      // ... of the reverse complement
    uint32_t len[KV_TILE_MAX_READS];
    uint32_t wpre[KV_TILE_MAX_READS + 1];  // packed-word prefix within the tile
    uint32_t kpre[KV_TILE_MAX_READS + 1];  // k-mer prefix within the tile
    uint32_t unk;                          // k-mers per read if every read of the tile has the same count, else 0
    float unk_inv;
    uint32_t seg_start;                    // segment tile: offset of the staged bases inside their read (else 0)
};

struct ReadsDev {
    const uint32_t *words;
    const uint64_t *woff;
    const uint32_t *len;
    const uint8_t *flags;
    const TileDesc *tile;
    uint32_t uni_len, uni_per_tile;     // kv_reads::uni_len / uni_per_tile (0: look the layout up)
    uint64_t n_reads;
};

// exclusive prefix sums over <= 128 reads by wave 0 (two entries per lane)
__device__ __forceinline__ uint32_t wave_excl_scan2(uint32_t a, uint32_t b, uint32_t &excl_b, uint32_t &total)
{
    const int lane = threadIdx.x & 63;
    uint32_t s = a + b, incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    total = __shfl(incl, 63);
    const uint32_t excl_a = incl - s;
    excl_b = excl_a + a;
    return excl_a;
}

// Stage tile `tile_id`; returns the number of reads in the tile.  skip_mode: 0 = count
// (every read >= k contributes), 1 = novel (flagged reads and reads < first_read skipped).
__device__ __forceinline__ uint32_t stage_tile(TileShared &sh, const ReadsDev &rd, uint32_t tile_id, int k,
                                               int skip_mode, uint64_t first_read, uint32_t &read0)
{
    const TileDesc td = rd.tile[tile_id];
    const uint32_t r0 = td.first, nr = td.count;
    const uint32_t seg_start = td.seg ? td.seg_start : 0u;       // multiple of 16: word aligned
    read0 = r0;
    const uint64_t w0 = rd.woff[r0] + (seg_start >> 4);
    if (threadIdx.x < 64) {
        const uint32_t i0 = 2 * threadIdx.x, i1 = i0 + 1;
        uint32_t l0 = 0, l1 = 0, k0 = 0, k1 = 0;
        if (i0 < nr) {
            l0 = rd.len[r0 + i0];
            if (td.seg) {   // (nr == 1) stage the segment's k-mer starts plus the k - 1 bases that complete its last k-mers
                const uint32_t rest = l0 - seg_start, want = (uint32_t)KV_SEG_BASES + (uint32_t)k - 1u;
                l0 = rest < want ? rest : want;
            }
            const bool skip = skip_mode && ((rd.flags[r0 + i0] & 1) || (uint64_t)(r0 + i0) < first_read);
            k0 = (l0 >= (uint32_t)k && !skip) ? l0 - (uint32_t)k + 1 : 0;
        }
        if (i1 < nr) {
            l1 = rd.len[r0 + i1];
            const bool skip = skip_mode && ((rd.flags[r0 + i1] & 1) || (uint64_t)(r0 + i1) < first_read);
            k1 = (l1 >= (uint32_t)k && !skip) ? l1 - (uint32_t)k + 1 : 0;
        }
        const uint32_t p0 = i0 < nr ? ((l0 + KV_READ_PAD + 3) & ~3u) : 0;
        const uint32_t p1 = i1 < nr ? ((l1 + KV_READ_PAD + 3) & ~3u) : 0;
        uint32_t eb, tot;
        const uint32_t ea = wave_excl_scan2(2 * p0, 2 * p1, eb, tot);
        if (i0 < nr) { sh.foff[i0] = ea; sh.roff[i0] = ea + p0; sh.len[i0] = l0; }
        if (i1 < nr) { sh.foff[i1] = eb; sh.roff[i1] = eb + p1; sh.len[i1] = l1; }
        uint32_t kb, ktot;
        const uint32_t ka = wave_excl_scan2(k0, k1, kb, ktot);
        if (i0 < nr) sh.kpre[i0] = ka;
        if (i1 < nr) sh.kpre[i1] = kb;
        if (threadIdx.x == 0) { sh.kpre[nr] = ktot; sh.seg_start = seg_start; }
        // uniform tile (the common case: fixed-length reads): k-mer -> read is a division, not a search
        const uint32_t ref_k = __shfl(k0, 0);
        const bool same = (i0 >= nr || k0 == ref_k) && (i1 >= nr || k1 == ref_k);
        const bool uniform = __all(same) && ref_k > 0;
        if (threadIdx.x == 0) { sh.unk = uniform ? ref_k : 0u; sh.unk_inv = uniform ? 1.0f / (float)ref_k : 0.0f; }
        if (td.seg) {
            if (threadIdx.x == 0) { sh.wpre[0] = 0; sh.wpre[1] = (l0 + 15) >> 4; }
        } else {
            if (i0 < nr) sh.wpre[i0] = (uint32_t)(rd.woff[r0 + i0] - w0);
            if (i1 < nr) sh.wpre[i1] = (uint32_t)(rd.woff[r0 + i1] - w0);
            if (threadIdx.x == 0) sh.wpre[nr] = (uint32_t)(rd.woff[r0 + nr] - w0);
        }
    }
    __syncthreads();
    const uint32_t nwords = sh.wpre[nr];
    uint8_t *lds8 = (uint8_t *)sh.ascii;
    for (uint32_t w = threadIdx.x; w < nwords; w += blockDim.x) {
        uint32_t lo = 0, hi = nr;  // largest r with wpre[r] <= w (wpre strictly increases over non-empty reads)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sh.wpre[mid] <= w) lo = mid; else hi = mid;
        }
        const uint32_t r = lo;
        const uint32_t j = w - sh.wpre[r];
        const uint32_t L = sh.len[r];
        const uint32_t bits = rd.words[w0 + w];
        uint32_t *fdst = sh.ascii + ((sh.foff[r] + 16 * j) >> 2);
        const uint32_t rbase = sh.roff[r];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t out = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint32_t code = (bits >> (2 * (4 * g + m))) & 3u;
                out |= ((0x54474341u >> (8 * code)) & 0xffu) << (8 * m);             // "ACGT"
                const uint32_t pos = 16 * j + 4 * g + m;
                if (pos < L) lds8[rbase + (L - 1 - pos)] = (uint8_t)((0x41434754u >> (8 * code)) & 0xffu);  // "TGCA"
            }
            fdst[g] = out;
        }
    }
    __syncthreads();
    return nr;
}

__device__ __forceinline__ void locate_kmer(const TileShared &sh, uint32_t nr, uint32_t q, uint32_t &r, uint32_t &i)
{
    if (sh.unk) {   // q < 2^16 here, so the float quotient is off by at most one
        uint32_t rr = (uint32_t)((float)q * sh.unk_inv);
        if (rr * sh.unk > q) rr -= 1;
        else if ((rr + 1) * sh.unk <= q) rr += 1;
        r = rr;
        i = q - rr * sh.unk;
        return;
    }
    uint32_t lo = 0, hi = nr;  // largest r with kpre[r] <= q
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sh.kpre[mid] <= q) lo = mid; else hi = mid;
    }
    r = lo;
    i = q - sh.kpre[lo];
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    return v;
}

__host__ __device__ inline HashParams make_hash_params(int k, int hashfam)
{
    HashParams hp;
    hp.k = k;
    hp.nblocks = k / 16;
    hp.rem = k & 15;
    const int r1 = hp.rem > 8 ? 8 : hp.rem;
    const int r2 = hp.rem > 8 ? hp.rem - 8 : 0;
    hp.m1 = r1 >= 8 ? ~0ull : ((1ull << (8 * r1)) - 1);
    hp.m2 = r2 >= 8 ? ~0ull : ((1ull << (8 * r2)) - 1);
    hp.hashfam = hashfam;
    return hp;
}

inline ReadsDev reads_dev(const kv_reads *r)
{
    ReadsDev d;
    d.words = r->d_words; d.woff = r->d_woff; d.len = r->d_len; d.flags = r->d_flags; d.tile = r->d_tile;
    d.uni_len = r->uni_len; d.uni_per_tile = r->uni_per_tile; d.n_reads = r->n_reads;
    return d;
}



// ---------------------------------------------------------------------------------------
// Rolling k-mer windows (k <= 4*NW - 1 + 1): each thread walks a RUN of consecutive k-mers.
// The k-mer's ASCII bytes live in registers, byte 0 of w[0] = first base, so murmur reads its
// words directly; moving to the next k-mer is one byte-shift per dword plus ONE LDS byte per
// strand, instead of 2 x (k/4 + 2) LDS dwords and a k-mer -> read search per k-mer.
// ---------------------------------------------------------------------------------------
template <int NW>
struct KmerRoll {
    uint32_t wf[NW], wr[NW];
    uint32_t r, i, nk_r;        // read slot, offset of the current k-mer, k-mers of this read
    uint32_t fa, ra;            // LDS byte address of the forward / reverse-complement k-mer
};

template <int NW>
__device__ __forceinline__ void roll_load(KmerRoll<NW> &w, const TileShared &sh, int k)
{
    w.nk_r = sh.kpre[w.r + 1] - sh.kpre[w.r];
    w.fa = sh.foff[w.r] + w.i;
    w.ra = sh.roff[w.r] + (sh.len[w.r] - (uint32_t)k - w.i);
    {
        const uint32_t s = w.fa & 3u;
        const uint32_t *p = sh.ascii + (w.fa >> 2);
        uint32_t prev = p[0];
#pragma unroll
        for (int j = 0; j < NW; ++j) { const uint32_t nxt = p[j + 1]; w.wf[j] = __builtin_amdgcn_alignbyte(nxt, prev, s); prev = nxt; }
    }
    {
        const uint32_t s = w.ra & 3u;
        const uint32_t *p = sh.ascii + (w.ra >> 2);
        uint32_t prev = p[0];
#pragma unroll
        for (int j = 0; j < NW; ++j) { const uint32_t nxt = p[j + 1]; w.wr[j] = __builtin_amdgcn_alignbyte(nxt, prev, s); prev = nxt; }
    }
}

// advance to the next k-mer of the tile (flat order); reloads at read boundaries
template <int NW>
__device__ __forceinline__ void roll_step(KmerRoll<NW> &w, const TileShared &sh, uint32_t nr, int k)
{
    w.i += 1;
    if (w.i >= w.nk_r) {
        w.i = 0;
        do { w.r += 1; } while (w.r < nr && sh.kpre[w.r + 1] == sh.kpre[w.r]);
        if (w.r < nr) roll_load(w, sh, k);
        return;
    }
    const uint32_t nf = lds_byte(sh.ascii, w.fa + 4u * NW);      // byte entering the forward window at the top
    const uint32_t nb = lds_byte(sh.ascii, w.ra - 1u);           // byte entering the reverse window at the bottom
    w.fa += 1; w.ra -= 1;
#pragma unroll
    for (int j = 0; j < NW - 1; ++j) w.wf[j] = __builtin_amdgcn_alignbyte(w.wf[j + 1], w.wf[j], 1);
    w.wf[NW - 1] = (w.wf[NW - 1] >> 8) | (nf << 24);
#pragma unroll
    for (int j = NW - 1; j > 0; --j) w.wr[j] = __builtin_amdgcn_alignbyte(w.wr[j], w.wr[j - 1], 3);
    w.wr[0] = (w.wr[0] << 8) | nb;
}

template <int NW>
__device__ __forceinline__ uint64_t murmur_regs(const uint32_t (&w)[NW], const HashParams &hp)
{
    uint64_t h1 = 0, h2 = 0;
#pragma unroll
    for (int b = 0; b < NW / 4; ++b) {
        if (b < hp.nblocks) {
            mm_block(h1, h2, (uint64_t)w[4 * b] | ((uint64_t)w[4 * b + 1] << 32),
                     (uint64_t)w[4 * b + 2] | ((uint64_t)w[4 * b + 3] << 32));
        } else if (b == hp.nblocks && hp.rem > 0) {
            if (hp.rem > 8) {
                uint64_t k2 = ((uint64_t)w[4 * b + 2] | ((uint64_t)w[4 * b + 3] << 32)) & hp.m2;
                k2 *= MM_C2; k2 = rotl64(k2, 33); k2 *= MM_C1; h2 ^= k2;
            }
            uint64_t k1 = ((uint64_t)w[4 * b] | ((uint64_t)w[4 * b + 1] << 32)) & hp.m1;
            k1 *= MM_C1; k1 = rotl64(k1, 31); k1 *= MM_C2; h1 ^= k1;
        }
    }
    return mm_final(h1, h2, hp.k);
}

template <int NW>
__device__ __forceinline__ uint64_t roll_hash(const KmerRoll<NW> &w, const HashParams &hp)
{
    return murmur_regs<NW>(w.wf, hp) ^ murmur_regs<NW>(w.wr, hp);
}

__device__ __forceinline__ bool consume_filter_pass(const ConsumeFilter &f, const SketchDev *mask, uint64_t h)
{
    if (f.use_band && !(h >= f.band_lo && h < f.band_hi)) return false;
    if (f.use_mask) {
        const int m = (int)sketch_get(mask, h);
        if (f.consume_masked ? (m < f.threshold) : (m > f.threshold)) return false;
    }
    return true;
}

inline ConsumeFilter make_consume_filter(int k, int hashfam, int nbands, int band, bool use_mask, int threshold,
                                         int consume_masked)
{
    ConsumeFilter f;
    f.hp = make_hash_params(k, hashfam);
    f.use_band = nbands > 0;
    f.band_lo = f.band_hi = 0;
    if (nbands > 0) kv_band_bounds(nbands, band, &f.band_lo, &f.band_hi);
    f.use_mask = use_mask ? 1 : 0;
    f.threshold = threshold;
    f.consume_masked = consume_masked;
    return f;
}

}  // namespace
