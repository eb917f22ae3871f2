/*
 * kvoracle.h -- CPU ORACLE for the kevlar novel-k-mer path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic that the reference (kevlar-dev/kevlar,
 * mounted at /root/reference) runs on its count -> novel -> filter path.  Nothing under
 * kevlar_amd/ (the product) may include, link, import or execute anything in oracle/;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / reported baseline.
 *
 * The arithmetic itself lives in a third-party dependency that is NOT in the reference
 * tree: khmer (dib-lab/khmer @ 6c893074ea005589c230fb7cb3712f0b258f42fc, pinned in
 * /root/reference/Dockerfile:36, unpinned in requirements.txt:7).  khmer in turn vendors
 * the public-domain smhasher MurmurHash3.  The restatement below follows khmer's
 * published behaviour and is PINNED (not "parity unpinned") against the reference's own
 * golden files and known-answer tests -- see tests/test_oracle_fixtures.py:
 *   kevlar/tests/test_count.py:45-68   five .ct files, byte for byte
 *   kevlar/tests/test_sketch.py:17-29  six saved sketches (byte/nibble/bit x table/graph)
 *   kevlar/tests/test_count.py:130-166 mask semantics ("36898 distinct k-mers stored")
 *   kevlar/tests/test_novel.py:179-207 "29 unique novel kmers in 14 reads"
 */
#ifndef KVORACLE_H
#define KVORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sketch kinds: khmer class names used at kevlar/sketch.py:14-27,99-119 */
enum {
    KVO_COUNTTABLE = 0,      /* byte counters, murmur hash  (.ct/.counttable)       */
    KVO_SMALLCOUNTTABLE = 1, /* nibble counters, murmur hash (.sct/.smallcounttable) */
    KVO_NODETABLE = 2,       /* bits, murmur hash            (.nt/.nodetable)        */
    KVO_COUNTGRAPH = 3,      /* byte counters, 2-bit hash    (.cg/.countgraph)       */
    KVO_SMALLCOUNTGRAPH = 4, /* nibble counters, 2-bit hash  (.scg/.smallcountgraph) */
    KVO_NODEGRAPH = 5        /* bits, 2-bit hash             (.ng/.nodegraph)        */
};

typedef struct kvo_sketch kvo_sketch;

/* --- hashing --- */
uint64_t kvo_murmur3_x64_128_lo(const void *data, int len, uint32_t seed);
uint64_t kvo_hash_murmur(const char *kmer, int k);  /* Counttable/SmallCounttable/Nodetable */
int kvo_hash_2bit(const char *kmer, int k, uint64_t *out); /* *graph types, k<=32; -1 on bad base */
uint64_t kvo_hash(int kind, const char *kmer, int k);
void kvo_reverse_hash_2bit(uint64_t h, int k, char *out); /* *graph only */

/* --- table sizing (H2): the n largest primes below target, scanning odd numbers down --- */
int kvo_primes_below(double target, int n, uint64_t *out);

/* --- sketch lifecycle --- */
kvo_sketch *kvo_sketch_create(int kind, int ksize, int ntables, const uint64_t *sizes);
void kvo_sketch_free(kvo_sketch *s);
kvo_sketch *kvo_sketch_load(const char *path, int kind_hint); /* kind_hint: table vs graph hash */
int kvo_sketch_save(const kvo_sketch *s, const char *path);

int kvo_kind(const kvo_sketch *s);
int kvo_ksize(const kvo_sketch *s);
int kvo_ntables(const kvo_sketch *s);
uint64_t kvo_tablesize(const kvo_sketch *s, int i);
uint64_t kvo_n_occupied(const kvo_sketch *s);
uint64_t kvo_n_unique(const kvo_sketch *s);
/* raw storage of table i as saved on disk (bytes / packed nibbles / packed bits) */
const uint8_t *kvo_table_bytes(const kvo_sketch *s, int i, uint64_t *nbytes);

/* --- add / get (H3, H6) --- */
int kvo_add_hash(kvo_sketch *s, uint64_t h); /* returns 1 if counted as new */
int kvo_get_hash(const kvo_sketch *s, uint64_t h);

/* --- consume with banding and masks (H3-H5); returns number of k-mers added --- */
uint64_t kvo_consume(kvo_sketch *s, const char *seq, size_t len, int nbands, int band,
                     const kvo_sketch *mask, int threshold, int consume_masked);
/* batch form over concatenated reads; offs has n+1 entries */
uint64_t kvo_consume_reads(kvo_sketch *s, const char *bases, const uint64_t *offs,
                           uint64_t n_reads, int nbands, int band, const kvo_sketch *mask,
                           int threshold, int consume_masked);

/* band bounds (H4): keep iff lo <= h < hi ; last band hi = 2^64-1 */
void kvo_band_bounds(int nbands, int band, uint64_t *lo, uint64_t *hi);
/* kevlar dist second pass (kevlar/dist.py:47-77): hist[65536] */
uint64_t kvo_abundance_distribution(const kvo_sketch *counts, kvo_sketch *tracking, const char *seq,
                                    size_t len, uint64_t *hist);

/* --- novel scan (H7-H9: kevlar/novel.py:21-53,123-169) ---
 * band_mode: 0 = none; 1 = hash-range band (count-side semantics, kevlar/count.py:62-66);
 *            2 = reference quirk (kevlar/novel.py:144-147: (h & (N-1)) != band0-1 -> skip).
 * Outputs (caller allocated, capacity cap): hit_read, hit_off, hit_abund[cap*(ncase+nctrl)].
 * read_status[n_reads]: 0 not emitted, 1 emitted, 2 discarded by abund screen,
 *                       3 skipped (len<k or non-ACGT).
 * Returns number of hits (may exceed cap: then only the first cap are stored). */
int64_t kvo_novel_scan(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl,
                       const char *bases, const uint64_t *offs, uint64_t n_reads, int ksize,
                       int case_min, int ctrl_max, int screen_thresh, int band_mode, int nbands,
                       int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund,
                       int64_t cap, uint8_t *read_status);

/* --- multi-threaded legs of bench.py's cpu_baseline (kevlar/count.py:41-76: threads share one sketch, atomic
 * saturating adds).  Tables and n_occupied equal the single-thread result; the scan leg only counts its hits. */
uint64_t kvo_consume_reads_mt(kvo_sketch *s, const char *bases, const uint64_t *offs, uint64_t n_reads, int nthreads);
/* the same with the hash-range banding of consume_seqfile_banding (kevlar/count.py:62-66): only band `band` of `nbands` is added */
uint64_t kvo_consume_reads_mt_banded(kvo_sketch *s, const char *bases, const uint64_t *offs, uint64_t n_reads, int nthreads,
                                     int nbands, int band);
/* kvo_novel_scan (no abundance screen) over contiguous ranges of reads on nthreads threads, hits concatenated in read order */
int64_t kvo_novel_scan_mt(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl, const char *bases,
                          const uint64_t *offs, uint64_t n_reads, int ksize, int case_min, int ctrl_max, int band_mode, int nbands,
                          int band, uint32_t *hit_read, uint16_t *hit_off, uint8_t *hit_abund, int64_t cap, int nthreads);
/* test infrastructure for BASELINE.json config 3: all nbands bands of a banded count / scan in one pass over the reads */
uint64_t kvo_consume_reads_mt_allbands(kvo_sketch *const *sketches, int nbands, const char *bases, const uint64_t *offs, uint64_t n_reads,
                                       int nthreads);
int64_t kvo_novel_scan_mt_allbands(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl, int nbands, const char *bases,
                                   const uint64_t *offs, uint64_t n_reads, int ksize, int case_min, int ctrl_max, uint32_t *hit_read,
                                   uint16_t *hit_off, uint8_t *hit_abund, uint8_t *hit_band, int64_t cap, int nthreads);
int64_t kvo_novel_scan_count_mt(kvo_sketch *const *cases, int ncase, kvo_sketch *const *ctrls, int nctrl,
                                const char *bases, const uint64_t *offs, uint64_t n_reads, int ksize,
                                int case_min, int ctrl_max, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
