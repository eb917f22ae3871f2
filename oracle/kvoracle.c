/*
 * kvoracle.c -- CPU ORACLE (test infrastructure, see kvoracle.h header comment).
 *
 * Restates, in scalar C, the behaviour of khmer (dib-lab/khmer @ 6c893074, un-vendored
 * dependency of the reference) that kevlar's count/novel/filter drivers rely on, plus
 * the reference's own novel scan loop.  Each function cites the reference call site it
 * follows.  Pinned against the reference's golden files by tests/test_oracle_fixtures.py.
 *
 * Deliberately simple: one thread, no SIMD, byte-at-a-time where that is clearest.
 */
#include "kvoracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* MurmurHash3_x64_128 (public domain, smhasher; khmer vendors it under                */
/* third-party/smhasher -- include path visible at reference notebook/mutsim/Makefile:6) */
/* ------------------------------------------------------------------------------------ */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p)
{
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

uint64_t kvo_murmur3_x64_128_lo(const void *data, int len, uint32_t seed)
{
    const uint8_t *bytes = (const uint8_t *)data;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;

    for (int i = 0; i < nblocks; ++i) {
        uint64_t k1 = load_le64(bytes + 16 * i);
        uint64_t k2 = load_le64(bytes + 16 * i + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }

    const uint8_t *tail = bytes + 16 * nblocks;
    uint64_t k1 = 0, k2 = 0;
    const int rem = len & 15;
    for (int i = rem - 1; i >= 8; --i) k2 = (k2 << 8) | tail[i];
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; --i) k1 = (k1 << 8) | tail[i];
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }

    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; /* out[0]; out[1] = h2 + h1 is unused by khmer */
    return h1;
}

/* ------------------------------------------------------------------------------------ */
/* k-mer hashing: H1; call sites kevlar/novel.py:38,48,145,              */
/* kevlar/filter.py:32-34,67; strand symmetry asserted by kevlar/tests/test_novel.py:68-77 */
/* ------------------------------------------------------------------------------------ */
static inline char complement(char c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    default:  return 'N';
    }
}

/* khmer cleans reads before hashing: upper-case, anything outside ACGT becomes 'A'
 * (SURVEY.md 8(c) "unpinned edges": no reference test constrains this). */
static inline char clean_base(char c)
{
    switch (c) {
    case 'A': case 'a': return 'A';
    case 'C': case 'c': return 'C';
    case 'G': case 'g': return 'G';
    case 'T': case 't': return 'T';
    default:  return 'A';
    }
}

#define KVO_MAXK 512

uint64_t kvo_hash_murmur(const char *kmer, int k)
{
    char rc[KVO_MAXK];
    if (k > KVO_MAXK) k = KVO_MAXK;
    for (int i = 0; i < k; ++i) rc[i] = complement(kmer[k - 1 - i]);
    return kvo_murmur3_x64_128_lo(kmer, k, 0) ^ kvo_murmur3_x64_128_lo(rc, k, 0);
}

/* *graph types: 2 bits per base, A=0 T=1 C=2 G=3, first base most significant,
 * hash = min(forward, reverse complement); k <= 32. */
static inline int twobit(char c)
{
    switch (c) {
    case 'A': return 0;
    case 'T': return 1;
    case 'C': return 2;
    case 'G': return 3;
    default:  return -1;
    }
}

int kvo_hash_2bit(const char *kmer, int k, uint64_t *out)
{
    uint64_t f = 0, r = 0;
    if (k > 32) return -1;
    for (int i = 0; i < k; ++i) {
        int c = twobit(kmer[i]);
        if (c < 0) return -1;
        f = (f << 2) | (uint64_t)c;
        int cc = twobit(complement(kmer[k - 1 - i]));
        r = (r << 2) | (uint64_t)cc;
    }
    *out = f < r ? f : r;
    return 0;
}

void kvo_reverse_hash_2bit(uint64_t h, int k, char *out)
{
    static const char alphabet[4] = {'A', 'T', 'C', 'G'};
    for (int i = k - 1; i >= 0; --i) { out[i] = alphabet[h & 3]; h >>= 2; }
    out[k] = '\0';
}

static inline int kind_is_graph(int kind) { return kind >= KVO_COUNTGRAPH; }

uint64_t kvo_hash(int kind, const char *kmer, int k)
{
    if (kind_is_graph(kind)) {
        uint64_t h = 0;
        kvo_hash_2bit(kmer, k, &h);
        return h;
    }
    return kvo_hash_murmur(kmer, k);
}

/* ------------------------------------------------------------------------------------ */
/* H2: table sizes.  kevlar/count.py:29-35 computes tablesize = memory/4 * buckets/byte */
/* and khmer picks the n largest primes below it (odd numbers, downwards).             */
/* ------------------------------------------------------------------------------------ */
static int is_prime_u64(uint64_t n)
{
    if (n < 2) return 0;
    if (n == 2) return 1;
    if ((n & 1) == 0) return 0;
    for (uint64_t d = 3; d * d <= n; d += 2)
        if (n % d == 0) return 0;
    return 1;
}

int kvo_primes_below(double target, int n, uint64_t *out)
{
    if (target < 1.0) return 0;
    uint64_t x = (uint64_t)target; /* float argument truncated, as Cython's uint64_t coercion does */
    if (x < 2) return 0;
    uint64_t i = x - 1;
    if ((i & 1) == 0) { if (i == 0) return 0; i -= 1; }
    int found = 0;
    while (found < n && i > 0) {
        if (is_prime_u64(i)) out[found++] = i;
        if (i < 2) break;
        i -= 2;
    }
    return found;
}

/* ------------------------------------------------------------------------------------ */
/* sketch storage                                                                        */
/* ------------------------------------------------------------------------------------ */
enum { ST_BYTE = 0, ST_NIBBLE = 1, ST_BIT = 2 };

struct kvo_sketch {
    int kind, storage, ksize, ntables;
    uint64_t *sizes;
    uint8_t **tables;
    uint64_t n_occupied, n_unique;
};

static int storage_of(int kind)
{
    switch (kind) {
    case KVO_COUNTTABLE: case KVO_COUNTGRAPH: return ST_BYTE;
    case KVO_SMALLCOUNTTABLE: case KVO_SMALLCOUNTGRAPH: return ST_NIBBLE;
    default: return ST_BIT;
    }
}

static uint64_t table_nbytes(int storage, uint64_t size)
{
    switch (storage) {
    case ST_BYTE: return size;
    case ST_NIBBLE: return size / 2 + 1;
    default: return size / 8 + 1;
    }
}

kvo_sketch *kvo_sketch_create(int kind, int ksize, int ntables, const uint64_t *sizes)
{
    kvo_sketch *s = (kvo_sketch *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->kind = kind; s->storage = storage_of(kind); s->ksize = ksize; s->ntables = ntables;
    s->sizes = (uint64_t *)calloc((size_t)ntables, sizeof(uint64_t));
    s->tables = (uint8_t **)calloc((size_t)ntables, sizeof(uint8_t *));
    for (int i = 0; i < ntables; ++i) {
        s->sizes[i] = sizes[i];
        s->tables[i] = (uint8_t *)calloc(table_nbytes(s->storage, sizes[i]), 1);
        if (!s->tables[i]) { kvo_sketch_free(s); return NULL; }
    }
    return s;
}

void kvo_sketch_free(kvo_sketch *s)
{
    if (!s) return;
    if (s->tables) for (int i = 0; i < s->ntables; ++i) free(s->tables[i]);
    free(s->tables); free(s->sizes); free(s);
}

int kvo_kind(const kvo_sketch *s) { return s->kind; }
int kvo_ksize(const kvo_sketch *s) { return s->ksize; }
int kvo_ntables(const kvo_sketch *s) { return s->ntables; }
uint64_t kvo_tablesize(const kvo_sketch *s, int i) { return s->sizes[i]; }
uint64_t kvo_n_occupied(const kvo_sketch *s) { return s->n_occupied; }
uint64_t kvo_n_unique(const kvo_sketch *s) { return s->n_unique; }
const uint8_t *kvo_table_bytes(const kvo_sketch *s, int i, uint64_t *nbytes)
{
    if (nbytes) *nbytes = table_nbytes(s->storage, s->sizes[i]);
    return s->tables[i];
}

/* H3: add.  Counter maxima 255 / 15 / 1 (bigcount is never enabled by kevlar: the header
 * flag is 0 in every fixture).  Nibble layout: even bin = low nibble?  khmer stores the
 * even bin in the HIGH nibble -- verified against kevlar/tests/data/test.smallcounttable
 * and test.smallcountgraph by tests/test_oracle_fixtures.py. */
int kvo_add_hash(kvo_sketch *s, uint64_t h)
{
    int is_new = 0;
    for (int i = 0; i < s->ntables; ++i) {
        uint64_t bin = h % s->sizes[i];
        uint8_t *t = s->tables[i];
        switch (s->storage) {
        case ST_BYTE: {
            if (t[bin] == 0) { is_new = 1; if (i == 0) s->n_occupied++; }
            if (t[bin] < 255) t[bin]++;
            break;
        }
        case ST_NIBBLE: {
            int shift = (bin & 1) ? 0 : 4;
            uint8_t cur = (uint8_t)((t[bin >> 1] >> shift) & 15);
            if (cur == 0) { is_new = 1; if (i == 0) s->n_occupied++; }
            if (cur < 15) t[bin >> 1] = (uint8_t)((t[bin >> 1] & ~(15 << shift)) | ((cur + 1) << shift));
            break;
        }
        default: {
            uint8_t bit = (uint8_t)(1u << (bin & 7));
            if (!(t[bin >> 3] & bit)) { is_new = 1; if (i == 0) s->n_occupied++; }
            t[bin >> 3] |= bit;
            break;
        }
        }
    }
    if (is_new) s->n_unique++;
    return is_new;
}

/* H6: get = Count-Min minimum over tables (Nodetable: AND of bits). */
int kvo_get_hash(const kvo_sketch *s, uint64_t h)
{
    int best = 255;
    for (int i = 0; i < s->ntables; ++i) {
        uint64_t bin = h % s->sizes[i];
        const uint8_t *t = s->tables[i];
        int v;
        switch (s->storage) {
        case ST_BYTE: v = t[bin]; break;
        case ST_NIBBLE: v = (t[bin >> 1] >> ((bin & 1) ? 0 : 4)) & 15; break;
        default: v = (t[bin >> 3] >> (bin & 7)) & 1; break;
        }
        if (v < best) best = v;
    }
    return best;
}

/* H4: hash-range banding of consume_seqfile_banding (kevlar/count.py:62-66). */
void kvo_band_bounds(int nbands, int band, uint64_t *lo, uint64_t *hi)
{
    uint64_t bs = UINT64_MAX / (uint64_t)nbands;
    *lo = bs * (uint64_t)band;
    *hi = (band == nbands - 1) ? UINT64_MAX : bs * (uint64_t)(band + 1);
}

/* H3-H5: consume one sequence.  mask rules (kevlar/count.py:43-60):
 *   consume_masked == 0: skip the k-mer if mask.get(h) >  threshold
 *   consume_masked != 0: skip the k-mer unless mask.get(h) >= threshold            */
uint64_t kvo_consume(kvo_sketch *s, const char *seq, size_t len, int nbands, int band,
                     const kvo_sketch *mask, int threshold, int consume_masked)
{
    const int k = s->ksize;
    if (len < (size_t)k || k > KVO_MAXK) return 0;
    uint64_t lo = 0, hi = 0, n = 0;
    if (nbands > 0) kvo_band_bounds(nbands, band, &lo, &hi);
    char *clean = (char *)malloc(len);
    for (size_t i = 0; i < len; ++i) clean[i] = clean_base(seq[i]);
    for (size_t i = 0; i + (size_t)k <= len; ++i) {
        uint64_t h = kvo_hash(s->kind, clean + i, k);
        if (nbands > 0 && !(h >= lo && h < hi)) continue;
        if (mask) {
            int m = kvo_get_hash(mask, h);
            if (consume_masked) { if (m < threshold) continue; }
            else                { if (m > threshold) continue; }
        }
        kvo_add_hash(s, h);
        ++n;
    }
    free(clean);
    return n;
}

uint64_t kvo_consume_reads(kvo_sketch *s, const char *bases, const uint64_t *offs,
                           uint64_t n_reads, int nbands, int band, const kvo_sketch *mask,
                           int threshold, int consume_masked)
{
    uint64_t n = 0;
    for (uint64_t r = 0; r < n_reads; ++r)
        n += kvo_consume(s, bases + offs[r], (size_t)(offs[r + 1] - offs[r]), nbands, band, mask,
                         threshold, consume_masked);
    return n;
}

/* khmer Hashtable::abundance_distribution, the second pass of `kevlar dist` (kevlar/dist.py:47-77;
 * khmer 2.1.1, un-vendored: lib/hashtable.cc).  For every k-mer of the read, in order: if `tracking`
 * has not seen it (get == 0), record it in tracking and bump hist[counts.get(kmer)].  hist has 65536
 * entries (MAX_BIGCOUNT + 1).  Returns the number of k-mers newly recorded.                        */
uint64_t kvo_abundance_distribution(const kvo_sketch *counts, kvo_sketch *tracking, const char *seq,
                                    size_t len, uint64_t *hist)
{
    const int k = counts->ksize;
    if (len < (size_t)k || k > KVO_MAXK) return 0;
    uint64_t n = 0;
    char *clean = (char *)malloc(len);
    for (size_t i = 0; i < len; ++i) clean[i] = clean_base(seq[i]);
    for (size_t i = 0; i + (size_t)k <= len; ++i) {
        uint64_t h = kvo_hash(counts->kind, clean + i, k);
        if (kvo_get_hash(tracking, h) != 0) continue;
        kvo_add_hash(tracking, h);
        int c = kvo_get_hash(counts, h);
        hist[c & 0xffff] += 1;
        ++n;
    }
    free(clean);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* H11: OXLI v4 files (layouts decoded from the fixtures)               */
/* ------------------------------------------------------------------------------------ */
static void put_le(FILE *f, uint64_t v, int nbytes)
{
    for (int i = 0; i < nbytes; ++i) fputc((int)((v >> (8 * i)) & 0xff), f);
}

static int get_le(FILE *f, int nbytes, uint64_t *out)
{
    uint64_t v = 0;
    for (int i = 0; i < nbytes; ++i) {
        int c = fgetc(f);
        if (c == EOF) return -1;
        v |= (uint64_t)c << (8 * i);
    }
    *out = v;
    return 0;
}

int kvo_sketch_save(const kvo_sketch *s, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    fwrite("OXLI", 1, 4, f);
    fputc(4, f);
    fputc(s->storage == ST_BYTE ? 1 : (s->storage == ST_BIT ? 2 : 7), f);
    if (s->storage == ST_BYTE) fputc(0, f); /* use_bigcount */
    put_le(f, (uint64_t)s->ksize, 4);
    fputc(s->ntables, f);
    put_le(f, s->n_occupied, 8);
    for (int i = 0; i < s->ntables; ++i) {
        put_le(f, s->sizes[i], 8);
        fwrite(s->tables[i], 1, table_nbytes(s->storage, s->sizes[i]), f);
    }
    if (s->storage == ST_BYTE) put_le(f, 0, 8); /* n_bigcounts */
    fclose(f);
    return 0;
}

kvo_sketch *kvo_sketch_load(const char *path, int kind_hint)
{
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    char sig[4];
    uint64_t v, k, nt, occ;
    kvo_sketch *s = NULL;
    if (fread(sig, 1, 4, f) != 4 || memcmp(sig, "OXLI", 4) != 0) goto fail;
    if (get_le(f, 1, &v) || v != 4) goto fail;
    if (get_le(f, 1, &v)) goto fail;
    int storage = v == 1 ? ST_BYTE : (v == 2 ? ST_BIT : (v == 7 ? ST_NIBBLE : -1));
    if (storage < 0 || storage != storage_of(kind_hint)) goto fail;
    if (storage == ST_BYTE && get_le(f, 1, &v)) goto fail; /* use_bigcount */
    if (get_le(f, 4, &k) || get_le(f, 1, &nt) || get_le(f, 8, &occ)) goto fail;
    s = (kvo_sketch *)calloc(1, sizeof(*s));
    s->kind = kind_hint; s->storage = storage; s->ksize = (int)k; s->ntables = (int)nt;
    s->n_occupied = occ;
    s->sizes = (uint64_t *)calloc((size_t)nt, sizeof(uint64_t));
    s->tables = (uint8_t **)calloc((size_t)nt, sizeof(uint8_t *));
    for (int i = 0; i < (int)nt; ++i) {
        if (get_le(f, 8, &s->sizes[i])) goto fail;
        uint64_t nb = table_nbytes(storage, s->sizes[i]);
        s->tables[i] = (uint8_t *)malloc(nb);
        if (fread(s->tables[i], 1, nb, f) != nb) goto fail;
    }
    fclose(f);
    return s;
fail:
    fclose(f);
    kvo_sketch_free(s);
    return NULL;
}

/* ------------------------------------------------------------------------------------ */
/* H7-H9: the novel scan, kevlar/novel.py:123-169 with kmer_is_interesting :21-53        */
/* ------------------------------------------------------------------------------------ */
/* band_mode 3 (test infrastructure for BASELINE.json config 3, no reference analogue): ALL nbands bands of a banded run in one pass.
 * `cases` / `ctrls` then hold nbands x ncase / nbands x nctrl sketches, band-major -- the sketches kevlar would have counted in
 * its nbands separate `--band b` jobs (docs/banding.rst) --, every k-mer is hashed once, evaluated against the sketches of the band
 * its hash falls into (the test kvo_consume applies: bs*b <= h < bs*(b+1)) and the band is reported beside the hit, so the hits
 * with hit_band == b are exactly what kvo_novel_scan(band_mode 1, band b) over band b's sketches returns. */
static int64_t novel_scan_impl(kvo_sketch *const *cases_all, int ncase, kvo_sketch *const *ctrls_all, int nctrl,
                       const char *bases, const uint64_t *offs, uint64_t n_reads, int ksize,
                       int case_min, int ctrl_max, int screen_thresh, int band_mode, int nbands,
                       int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund, uint8_t *hit_band,
                       int64_t cap, uint8_t *read_status)
{
    kvo_sketch *const *cases = cases_all, *const *ctrls = ctrls_all;
    const uint64_t bs = nbands > 0 ? UINT64_MAX / (uint64_t)nbands : 0;
    const int S = ncase + nctrl;
    int64_t nhits = 0;
    uint64_t lo = 0, hi = 0;
    if (band_mode == 1) kvo_band_bounds(nbands, band, &lo, &hi);
    uint8_t abund[64];

    for (uint64_t r = 0; r < n_reads; ++r) {
        const char *seq = bases + offs[r];
        const size_t len = (size_t)(offs[r + 1] - offs[r]);
        if (read_status) read_status[r] = 3;
        if (len < (size_t)ksize) continue;                 /* novel.py:134-135 */
        int clean = 1;                                     /* novel.py:136-139 [^ACGT] */
        for (size_t i = 0; i < len; ++i)
            if (seq[i] != 'A' && seq[i] != 'C' && seq[i] != 'G' && seq[i] != 'T') { clean = 0; break; }
        if (!clean) continue;

        const int64_t first_hit = nhits;
        int discard_read = 0;
        for (size_t i = 0; i + (size_t)ksize <= len; ++i) { /* novel.py:143 get_kmers */
            const uint64_t h = kvo_hash(cases_all[0]->kind, seq + i, ksize);
            int b_of_h = 0;
            if (band_mode == 3) {
                if (h == UINT64_MAX) continue;                      /* in no band: the last band ends below it (kvo_band_bounds) */
                b_of_h = (int)(h / bs);
                if (b_of_h >= nbands) b_of_h = nbands - 1;          /* the last band also takes the remainder of 2^64 / nbands */
                cases = cases_all + (size_t)b_of_h * (size_t)ncase;
                ctrls = ctrls_all + (size_t)b_of_h * (size_t)nctrl;
            }
            if (band_mode == 1 && !(h >= lo && h < hi)) continue;
            if (band_mode == 2 && (h & (uint64_t)(nbands - 1)) != (uint64_t)(int64_t)(band - 1))
                continue;                                   /* novel.py:144-147, band is 0-based */
            int interesting = 1;
            for (int c = 0; c < ncase; ++c) {               /* novel.py:36-44 */
                int a = kvo_get_hash(cases[c], kvo_hash(cases[c]->kind, seq + i, ksize));
                if (a < case_min) {
                    if (screen_thresh > 0 && a < screen_thresh) discard_read = 1;
                    interesting = 0;
                    break;
                }
                abund[c] = (uint8_t)a;
            }
            if (discard_read) break;                        /* novel.py:152-154 */
            if (!interesting) continue;
            for (int c = 0; c < nctrl; ++c) {               /* novel.py:46-51 */
                int a = kvo_get_hash(ctrls[c], kvo_hash(ctrls[c]->kind, seq + i, ksize));
                if (a > ctrl_max) { interesting = 0; break; }
                abund[ncase + c] = (uint8_t)a;
            }
            if (!interesting) continue;
            if (nhits < cap) {
                hit_read[nhits] = (uint32_t)r;
                hit_off[nhits] = (uint16_t)i;
                memcpy(hit_abund + (size_t)nhits * (size_t)S, abund, (size_t)S);
                if (hit_band) hit_band[nhits] = (uint8_t)b_of_h;
            }
            ++nhits;
        }
        if (discard_read) {                                 /* novel.py:164 */
            nhits = first_hit;
            if (read_status) read_status[r] = 2;
        } else if (read_status) {
            read_status[r] = nhits > first_hit ? 1 : 0;
        }
    }
    return nhits;
}

int64_t kvo_novel_scan(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl,
                       const char *bases, const uint64_t *offs, uint64_t n_reads, int ksize,
                       int case_min, int ctrl_max, int screen_thresh, int band_mode, int nbands,
                       int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund,
                       int64_t cap, uint8_t *read_status)
{
    if (band_mode < 0 || band_mode > 2) return -1;
    return novel_scan_impl(cases, ncase, ctrls, nctrl, bases, offs, n_reads, ksize, case_min, ctrl_max, screen_thresh, band_mode, nbands,
                           band, hit_read, hit_off, hit_abund, NULL, cap, read_status);
}

/* ------------------------------------------------------------------------------------ */
/* Multi-threaded legs for the "all host cores" CPU baseline (bench.py cpu_baseline), the way */
/* kevlar runs khmer (kevlar/count.py:41-76: numthreads threads pull reads from one parser   */
/* and add to ONE sketch with atomic saturating increments).  Saturating adds commute, so the */
/* tables and n_occupied equal the single-thread result; n_unique is order dependent, as in   */
/* khmer.                                                                                     */
/* ------------------------------------------------------------------------------------ */
#include <pthread.h>

/* n_occupied / n_unique are accumulated per thread (occ, uniq) and added once at the end: a shared counter bumped
 * on every new k-mer would turn into the hottest cache line of the machine with many threads */
static int add_hash_atomic(kvo_sketch *s, uint64_t h, uint64_t *occ, uint64_t *uniq)
{
    int is_new = 0;
    for (int i = 0; i < s->ntables; ++i) {
        const uint64_t bin = h % s->sizes[i];
        uint8_t *t = s->tables[i];
        if (s->storage == ST_BYTE) {
            uint8_t cur = __atomic_load_n(&t[bin], __ATOMIC_RELAXED);
            while (cur < 255 && !__atomic_compare_exchange_n(&t[bin], &cur, (uint8_t)(cur + 1), 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            if (cur == 0) { is_new = 1; if (i == 0) ++*occ; }
        } else if (s->storage == ST_NIBBLE) {
            const int shift = (bin & 1) ? 0 : 4;
            uint8_t old = __atomic_load_n(&t[bin >> 1], __ATOMIC_RELAXED);
            for (;;) {
                const uint8_t cur = (uint8_t)((old >> shift) & 15);
                if (cur == 15) break;
                const uint8_t neu = (uint8_t)((old & ~(15 << shift)) | ((cur + 1) << shift));
                if (__atomic_compare_exchange_n(&t[bin >> 1], &old, neu, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
                    if (cur == 0) { is_new = 1; if (i == 0) ++*occ; }
                    break;
                }
            }
        } else {
            const uint8_t bit = (uint8_t)(1u << (bin & 7));
            const uint8_t old = __atomic_fetch_or(&t[bin >> 3], bit, __ATOMIC_RELAXED);
            if (!(old & bit)) { is_new = 1; if (i == 0) ++*occ; }
        }
    }
    if (is_new) ++*uniq;
    return is_new;
}

typedef struct {
    kvo_sketch *s;
    const char *bases;
    const uint64_t *offs;
    uint64_t n_reads, chunk;
    uint64_t *next;             /* shared cursor: threads pull chunks of reads, like khmer's parser */
    uint64_t n_added;
    int nbands, band;           /* hash-range banding of consume_seqfile_banding (kevlar/count.py:62-66); nbands 0: none */
    kvo_sketch *const *all;     /* non-NULL: the nbands sketches of ALL bands; a k-mer is hashed once and added to its band's sketch */
} mt_count_job;

static void *mt_count_worker(void *arg)
{
    mt_count_job *j = (mt_count_job *)arg;
    const int k = j->s->ksize;
    char *clean = NULL;
    size_t cap = 0;
    uint64_t occ = 0, uniq = 0;
    uint64_t lo = 0, hi = 0;
    uint64_t occ_b[256], uniq_b[256];
    const uint64_t bs = j->nbands > 0 ? UINT64_MAX / (uint64_t)j->nbands : 0;
    if (j->all) { memset(occ_b, 0, sizeof(occ_b)); memset(uniq_b, 0, sizeof(uniq_b)); }
    if (j->nbands > 0) kvo_band_bounds(j->nbands, j->band, &lo, &hi);
    for (;;) {
        const uint64_t r0 = __atomic_fetch_add(j->next, j->chunk, __ATOMIC_RELAXED);
        if (r0 >= j->n_reads) break;
        const uint64_t r1 = r0 + j->chunk < j->n_reads ? r0 + j->chunk : j->n_reads;
        for (uint64_t r = r0; r < r1; ++r) {
            const char *seq = j->bases + j->offs[r];
            const size_t len = (size_t)(j->offs[r + 1] - j->offs[r]);
            if (len < (size_t)k || k > KVO_MAXK) continue;
            if (len > cap) { free(clean); clean = (char *)malloc(len); cap = len; }
            for (size_t i = 0; i < len; ++i) clean[i] = clean_base(seq[i]);
            /* The adds are the same adds in the same order; the k-mers of a read are only hashed a few dozen at a time first, and the
             * cache lines their bins live in are requested before the first add (a sketch of hundreds of megabytes is a cache miss
             * per table and k-mer: with the misses of 32 k-mers in flight the threaded legs -- test infrastructure and the CPU
             * baseline of bench.py -- run about twice as fast; nothing of the arithmetic changes). */
            enum { AHEAD = 32 };
            uint64_t hs[AHEAD];
            for (size_t i0 = 0; i0 + (size_t)k <= len; i0 += AHEAD) {
                size_t m = 0;
                for (; m < AHEAD && i0 + m + (size_t)k <= len; ++m) {
                    const uint64_t h = kvo_hash(j->s->kind, clean + i0 + m, k);
                    hs[m] = h;
                    const kvo_sketch *dst = j->s;
                    if (j->all) {
                        int b = h == UINT64_MAX ? 0 : (int)(h / bs);
                        if (b >= j->nbands) b = j->nbands - 1;
                        dst = j->all[b];
                    }
                    for (int t = 0; t < dst->ntables; ++t) {
                        const uint64_t bin = h % dst->sizes[t];
                        __builtin_prefetch(dst->tables[t] + (dst->storage == ST_BYTE ? bin : (dst->storage == ST_NIBBLE ? bin >> 1 : bin >> 3)), 1, 1);
                    }
                }
                for (size_t q = 0; q < m; ++q) {
                    const uint64_t h = hs[q];
                    if (j->all) {
                        /* the band whose test (bs*b <= h < bs*(b+1), the last one up to 2^64 - 1 exclusive: kvo_band_bounds) h passes */
                        if (h == UINT64_MAX) continue;
                        int b = (int)(h / bs);
                        if (b >= j->nbands) b = j->nbands - 1;
                        add_hash_atomic(j->all[b], h, &occ_b[b], &uniq_b[b]);
                        j->n_added++;
                        continue;
                    }
                    if (j->nbands > 0 && !(h >= lo && h < hi)) continue;      /* same test as kvo_consume */
                    add_hash_atomic(j->s, h, &occ, &uniq);
                    j->n_added++;
                }
            }
        }
    }
    free(clean);
    if (j->all) {
        for (int b = 0; b < j->nbands; ++b) {
            __atomic_fetch_add(&j->all[b]->n_occupied, occ_b[b], __ATOMIC_RELAXED);
            __atomic_fetch_add(&j->all[b]->n_unique, uniq_b[b], __ATOMIC_RELAXED);
        }
        return NULL;
    }
    __atomic_fetch_add(&j->s->n_occupied, occ, __ATOMIC_RELAXED);
    __atomic_fetch_add(&j->s->n_unique, uniq, __ATOMIC_RELAXED);
    return NULL;
}

uint64_t kvo_consume_reads_mt(kvo_sketch *s, const char *bases, const uint64_t *offs, uint64_t n_reads, int nthreads)
{
    return kvo_consume_reads_mt_banded(s, bases, offs, n_reads, nthreads, 0, 0);
}

uint64_t kvo_consume_reads_mt_banded(kvo_sketch *s, const char *bases, const uint64_t *offs, uint64_t n_reads, int nthreads,
                                     int nbands, int band)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    mt_count_job jobs[256];
    uint64_t next = 0, total = 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].s = s; jobs[t].bases = bases; jobs[t].offs = offs; jobs[t].n_reads = n_reads;
        jobs[t].chunk = 1024; jobs[t].next = &next; jobs[t].n_added = 0; jobs[t].nbands = nbands; jobs[t].band = band; jobs[t].all = NULL;
        pthread_create(&th[t], NULL, mt_count_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); total += jobs[t].n_added; }
    return total;
}

/* All nbands bands of a banded count in ONE pass over the reads (test infrastructure for BASELINE.json config 3; kevlar itself runs
 * `kevlar count --num-bands N --band b` once per band, docs/banding.rst): sketches[b] receives exactly the k-mers
 * kvo_consume_reads_mt_banded(sketches[b], ..., nbands, b) would add -- the band test is a partition of the hash values below
 * 2^64 - 1 --, so tables and n_occupied per band equal the band-by-band result; the hashing is paid once instead of nbands times. */
uint64_t kvo_consume_reads_mt_allbands(kvo_sketch *const *sketches, int nbands, const char *bases, const uint64_t *offs, uint64_t n_reads,
                                       int nthreads)
{
    if (nbands < 1 || nbands > 256) return 0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    mt_count_job jobs[256];
    uint64_t next = 0, total = 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].s = sketches[0]; jobs[t].bases = bases; jobs[t].offs = offs; jobs[t].n_reads = n_reads;
        jobs[t].chunk = 1024; jobs[t].next = &next; jobs[t].n_added = 0; jobs[t].nbands = nbands; jobs[t].band = 0; jobs[t].all = sketches;
        pthread_create(&th[t], NULL, mt_count_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); total += jobs[t].n_added; }
    return total;
}

typedef struct {
    kvo_sketch *const *cases; int ncase;
    kvo_sketch *const *ctrls; int nctrl;
    const char *bases; const uint64_t *offs;
    uint64_t r0, r1;
    int ksize, case_min, ctrl_max;
    int64_t nhits;
} mt_scan_job;

static void *mt_scan_worker(void *arg)
{
    mt_scan_job *j = (mt_scan_job *)arg;
    /* the scan of a contiguous range of reads; hits are only counted here (cap 0) -- the baseline times the
     * evaluation, the ordered output is what tests check through kvo_novel_scan                          */
    j->nhits = kvo_novel_scan(j->cases, j->ncase, j->ctrls, j->nctrl, j->bases, j->offs + j->r0, j->r1 - j->r0,
                              j->ksize, j->case_min, j->ctrl_max, 0, 0, 0, 0, NULL, NULL, NULL, 0, NULL);
    return NULL;
}

typedef struct {
    kvo_sketch *const *cases; int ncase;
    kvo_sketch *const *ctrls; int nctrl;
    const char *bases; const uint64_t *offs;
    uint64_t r0, r1;
    int ksize, case_min, ctrl_max, band_mode, nbands, band;
    uint32_t *hr; uint16_t *ho; uint8_t *ha; uint8_t *hb; int64_t cap;
    int64_t nhits;
} mt_scan_hits_job;

static void *mt_scan_hits_worker(void *arg)
{
    mt_scan_hits_job *j = (mt_scan_hits_job *)arg;
    j->nhits = novel_scan_impl(j->cases, j->ncase, j->ctrls, j->nctrl, j->bases, j->offs + j->r0, j->r1 - j->r0, j->ksize, j->case_min,
                               j->ctrl_max, 0, j->band_mode, j->nbands, j->band, j->hr, j->ho, j->ha, j->hb, j->cap, NULL);
    return NULL;
}

/* The scan loop of kvo_novel_scan (kevlar/novel.py:123-169, no abundance screen: a read's verdict then depends on that read
 * alone) over contiguous ranges of reads on nthreads threads; the ranges' hits are concatenated in read order, so the result is
 * kvo_novel_scan's.  Returns the number of hits (which may exceed cap: the caller retries with room), -1 without memory. */
static int64_t novel_scan_mt_impl(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl, const char *bases,
                          const uint64_t *offs, uint64_t n_reads, int ksize, int case_min, int ctrl_max, int band_mode, int nbands,
                          int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund, uint8_t *hit_band, int64_t cap, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    const int S = ncase + nctrl;
    pthread_t th[256];
    mt_scan_hits_job jobs[256];
    const int64_t each = cap / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        mt_scan_hits_job *j = &jobs[t];
        j->cases = cases; j->ncase = ncase; j->ctrls = ctrls; j->nctrl = nctrl; j->bases = bases; j->offs = offs;
        j->r0 = n_reads * (uint64_t)t / (uint64_t)nthreads; j->r1 = n_reads * (uint64_t)(t + 1) / (uint64_t)nthreads;
        j->ksize = ksize; j->case_min = case_min; j->ctrl_max = ctrl_max; j->band_mode = band_mode; j->nbands = nbands; j->band = band;
        j->cap = each; j->nhits = 0;
        j->hr = (uint32_t *)malloc((size_t)(each > 0 ? each : 1) * 4);
        j->ho = (uint16_t *)malloc((size_t)(each > 0 ? each : 1) * 2);
        j->ha = (uint8_t *)malloc((size_t)(each > 0 ? each : 1) * (size_t)S);
        j->hb = hit_band ? (uint8_t *)malloc((size_t)(each > 0 ? each : 1)) : NULL;
        if (!j->hr || !j->ho || !j->ha || (hit_band && !j->hb)) return -1;
        pthread_create(&th[t], NULL, mt_scan_hits_worker, j);
    }
    int64_t total = 0, worst = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); if (jobs[t].nhits > worst) worst = jobs[t].nhits; }
    for (int t = 0; t < nthreads; ++t) {
        mt_scan_hits_job *j = &jobs[t];
        if (worst <= each) {
            for (int64_t i = 0; i < j->nhits; ++i) {
                hit_read[total + i] = j->hr[i] + (uint32_t)j->r0;        /* kvo_novel_scan numbers the reads of its range from 0 */
                hit_off[total + i] = j->ho[i];
                memcpy(hit_abund + (size_t)(total + i) * (size_t)S, j->ha + (size_t)i * (size_t)S, (size_t)S);
                if (hit_band) hit_band[total + i] = j->hb[i];
            }
            total += j->nhits;
        }
        free(j->hr); free(j->ho); free(j->ha); free(j->hb);
    }
    return worst <= each ? total : worst * nthreads;
}

int64_t kvo_novel_scan_mt(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl, const char *bases,
                          const uint64_t *offs, uint64_t n_reads, int ksize, int case_min, int ctrl_max, int band_mode, int nbands,
                          int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund, int64_t cap, int nthreads)
{
    if (band_mode < 0 || band_mode > 2) return -1;
    return novel_scan_mt_impl(cases, ncase, ctrls, nctrl, bases, offs, n_reads, ksize, case_min, ctrl_max, band_mode, nbands, band,
                              hit_read, hit_off, hit_abund, NULL, cap, nthreads);
}

/* The scans of ALL nbands bands of a banded run in one pass (novel_scan_impl, band_mode 3): cases / ctrls hold nbands x ncase /
 * nbands x nctrl sketches band-major; hit_band[i] says which band's sketches judged hit i.  The merged result of the nbands jobs
 * kevlar runs (docs/banding.rst + kevlar/unband.py:41-77) is all hits in (read, offset) order, which is the order returned. */
int64_t kvo_novel_scan_mt_allbands(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl, int nbands, const char *bases,
                                   const uint64_t *offs, uint64_t n_reads, int ksize, int case_min, int ctrl_max, uint32_t *hit_read,
                                   uint16_t *hit_off, uint8_t *hit_abund, uint8_t *hit_band, int64_t cap, int nthreads)
{
    if (nbands < 1 || nbands > 256 || !hit_band) return -1;
    return novel_scan_mt_impl(cases, ncase, ctrls, nctrl, bases, offs, n_reads, ksize, case_min, ctrl_max, 3, nbands, 0,
                              hit_read, hit_off, hit_abund, hit_band, cap, nthreads);
}

int64_t kvo_novel_scan_count_mt(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl,
                                const char *bases, const uint64_t *offs, uint64_t n_reads, int ksize,
                                int case_min, int ctrl_max, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    mt_scan_job jobs[256];
    int64_t total = 0;
    for (int t = 0; t < nthreads; ++t) {
        mt_scan_job *j = &jobs[t];
        j->cases = cases; j->ncase = ncase; j->ctrls = ctrls; j->nctrl = nctrl; j->bases = bases; j->offs = offs;
        j->r0 = n_reads * (uint64_t)t / (uint64_t)nthreads; j->r1 = n_reads * (uint64_t)(t + 1) / (uint64_t)nthreads;
        j->ksize = ksize; j->case_min = case_min; j->ctrl_max = ctrl_max; j->nhits = 0;
        pthread_create(&th[t], NULL, mt_scan_worker, j);
    }
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); total += jobs[t].nhits; }
    return total;
}
