/*
 * kvsketch.h -- C ABI of libkvsketch_hip.so, the MI355X (gfx950) sketch engine behind
 * kevlar's novel-k-mer path (count -> novel -> filter -> partition).
 *
 * The reference has no C ABI on this path: its boundary is the Python object API of khmer
 * (SURVEY.md section 8(b)).  Every entry point below names the khmer / kevlar call it
 * replaces (paths are relative to the reference tree).  Plain pointers and sizes only; no
 * torch types.  All tables live in HBM; "host" pointers are ordinary malloc'ed memory,
 * "device" pointers are HBM addresses (e.g. a torch tensor's data_ptr()).
 *
 * Conventions
 *   - every function returns KV_OK (0) or a negative KV_ERR_* code; kv_last_error() gives
 *     the message for the calling thread.
 *   - handles are owned by the caller and released with the matching *_destroy.
 *   - buffers passed in are borrowed for the duration of the call.
 *   - all work is enqueued on one HIP stream (kv_set_stream; default: the null stream) and
 *     functions that return host data synchronise that stream before returning.
 *   - calls on one sketch from several host threads are serialised by the library
 *     (kevlar/count.py:41-76 starts `numthreads` threads on one sketch + one parser).
 */
#ifndef KVSKETCH_H
#define KVSKETCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KV_OK 0
#define KV_ERR_ARG (-1)      /* bad argument (maps to ValueError)                       */
#define KV_ERR_IO (-2)       /* file cannot be opened / truncated / wrong signature      */
#define KV_ERR_TYPE (-3)     /* sketch file type does not match the requested kind        */
#define KV_ERR_HIP (-4)      /* HIP runtime failure (no device, OOM, launch failure)     */
#define KV_ERR_NOTIMPL (-5)  /* e.g. reverse_hash on a *table sketch                      */
#define KV_ERR_CAPACITY (-6) /* caller-provided buffer too small                          */

/* sketch kinds = khmer classes used at kevlar/sketch.py:14-27,99-119 */
#define KV_COUNTTABLE 0
#define KV_SMALLCOUNTTABLE 1
#define KV_NODETABLE 2
#define KV_COUNTGRAPH 3
#define KV_SMALLCOUNTGRAPH 4
#define KV_NODEGRAPH 5

#define KV_MAX_TABLES 16
#define KV_MAX_K 255       /* murmur-hashed (*table) kinds; 2-bit (*graph) kinds need k<=32 */
#define KV_MAX_SAMPLES 16  /* cases + controls in one novel scan                          */

typedef struct kv_sketch kv_sketch; /* Count-Min / Bloom tables resident in HBM            */
typedef struct kv_reads kv_reads;   /* a batch of reads, 2-bit packed, resident in HBM     */
typedef struct kv_hits kv_hits;     /* sparse result of a novel scan                       */

typedef struct kv_sketch_info {
    int32_t kind, ksize, ntables, reserved;
    uint64_t sizes[KV_MAX_TABLES]; /* khmer .hashsizes()                                   */
    uint64_t n_occupied;           /* khmer .n_occupied(): non-zero bins of table 0        */
    uint64_t n_unique;             /* khmer .n_unique_kmers() (see kv_consume)             */
    uint64_t bytes_device;         /* HBM held by the tables                               */
} kv_sketch_info;

/* ---- library / device ---------------------------------------------------------------- */
const char *kv_last_error(void);
/* Table buffers of destroyed sketches are kept for the next sketch of the same size (KV_TABLE_CACHE_GB, default 32); the library gives
 * them back by itself when one of its own allocations fails.  A host program about to allocate a large buffer of its own (torch,
 * another library) calls this first.                                                                                                */
int kv_table_cache_trim(void);
/* The library's per-stream working buffers only grow (the super-k-mer buckets of the last batch, the pair sink of an exchange owner, the
 * staging of the partitioned add): after one very large batch -- a bucket owner of a 900 M-read sample holds ~200 GB of them -- a
 * long-lived process gives them back with this call.  No other kv_* call may be running; whatever a later call kept in them (the
 * buckets a scan would reuse, an owner's combined buckets) is forgotten and rebuilt.  Also trims the table cache and the first-toucher
 * arrays of kv_unique_new.                                                                                                           */
int kv_scratch_trim(void);
const char *kv_version(void);
/* The environment variables the library looks at are one table (kevlar_amd/csrc/kv_knobs.h, kv_host.hip): SETTINGS a user may set
 * (cache sizes, host threads, progress on stderr), TUNING switches that pin a path or shrink a geometry for tests and A/B runs -- same
 * results on every path; honoured only while KV_TUNING=1 is set -- and EXPERIMENTS that skip parts of kernels (wrong results; only a
 * library built with -DKV_EXPERIMENTS honours them).  kv_knobs_describe: whole_table = 0 writes what is set right now as
 * "NAME=value ..." ("ignored:NAME=value" for a knob that is set but not honoured), 1 the table as "name<TAB>class<TAB>what it does"
 * lines.  kv_knob_get: 1 and the value if `name` is set and honoured, 0 if not, KV_ERR_ARG for a name outside the table (the Python
 * wrapper reads its own switches through this, so there is one registry).                                                           */
int kv_knobs_describe(int whole_table, char *out, uint64_t cap);
int kv_knob_get(const char *name, char *value_out, uint64_t cap);
int kv_device_count(int *n);
int kv_set_device(int device);       /* one process per GPU: call once with LOCAL_RANK; every host thread that enters the library
                                      * afterwards is switched to that GPU (HIP's current device is per thread and starts at 0)  */
/* what kv_set_device said (-1: never called), what the library last switched the CALLING thread to, what hipGetDevice says now */
int kv_thread_device_get(int *configured, int *this_thread, int *hip_current);
int kv_set_stream(void *hip_stream); /* hipStream_t for the CALLING host thread; NULL = null stream */
int kv_synchronize(void);
/* streams for host threads that count different samples at the same time (kevlar/novel.py:64-72
 * counts its samples one after the other; their kernels are independent)                     */
int kv_stream_create(void **out);
int kv_stream_destroy(void *hip_stream);

/* live per-kernel timing with HIP events on the library's stream (bench.py roofline leg).
 * kv_prof_get: accumulated milliseconds and launch count for a kernel name.             */
int kv_prof_enable(int on);
int kv_prof_reset(void);
int kv_prof_get(const char *kernel, double *ms, uint64_t *launches);
int kv_prof_names(char *buf, size_t cap); /* comma separated list of kernel names seen     */

/* ---- host-side helpers (no table access) --------------------------------------------- */
/* khmer table sizing: the n largest primes below `target`, odd numbers downwards
 * (kevlar/count.py:29-35 -> khmer Counttable(k, tablesize, n)).                           */
int kv_primes_below(double target, int n, uint64_t *out, int *found);
/* khmer .hash(kmer) (kevlar/novel.py:145, kevlar/tests/test_novel.py:68-77)               */
int kv_hash_kmer(int kind, const char *kmer, int k, uint64_t *out);
/* khmer .reverse_hash(h): *graph kinds only; KV_ERR_NOTIMPL for *table kinds
 * (kevlar/tests/test_sketch.py:39-69)                                                     */
int kv_reverse_hash(int kind, uint64_t h, int k, char *out /* k+1 bytes */);
/* hash-range band of consume_seqfile_banding (kevlar/count.py:62-66): lo <= h < hi        */
int kv_band_bounds(int nbands, int band, uint64_t *lo, uint64_t *hi);

/* ---- sketches ------------------------------------------------------------------------ */
/* khmer Counttable/SmallCounttable/Nodetable/...(k, tablesize, ntables) with the primes
 * already chosen (kevlar/sketch.py:99-119, kevlar/filter.py:29)                           */
int kv_sketch_create(int kind, int ksize, int ntables, const uint64_t *sizes, kv_sketch **out);
int kv_sketch_destroy(kv_sketch *s);
/* khmer Class.load(path) / .save(path): OXLI v4 files (kevlar/sketch.py:14-27,77-92,
 * kevlar/count.py:95, kevlar/novel.py:92)                                                 */
int kv_sketch_load(const char *path, int kind, kv_sketch **out);
int kv_sketch_save(kv_sketch *s, const char *path);
/* .ksize() .n_tables() .hashsizes() .n_occupied() .n_unique_kmers()
 * (kevlar/sketch.py:62-74, kevlar/count.py:82-84)                                         */
int kv_sketch_info_get(kv_sketch *s, kv_sketch_info *out);
/* raw storage of one table, as laid out on disk (bytes / packed nibbles / packed bits)    */
int kv_sketch_table_read(kv_sketch *s, int table, uint8_t *host_out, uint64_t nbytes);
int kv_sketch_table_write(kv_sketch *s, int table, const uint8_t *host_in, uint64_t nbytes);
int kv_sketch_table_devptr(kv_sketch *s, int table, void **devptr, uint64_t *nbytes);
/* zero every table and counter (a fresh sketch of the same geometry; asynchronous)         */
int kv_sketch_clear(kv_sketch *s);
/* on != 0: the batches counted into this sketch are a CASE sample's and each is scanned right after it is counted
 * (kevlar/novel.py:92-121 loads the case samples last, then scans them).  kv_consume then also keeps, for the batch it just
 * counted, every distinct k-mer with its hash, and a kv_novel_scan of that same batch evaluates from that list instead of
 * combining and hashing the batch's k-mers a second time.  on = 2: the process counts and scans sample after sample (a server, the
 * bench's steady state), so the list is kept from the stream's very first batch on -- a one-shot run skips it there because the
 * allocation costs more than the list saves once.  Purely a performance hint: results do not depend on it. */
int kv_sketch_scan_hint(kv_sketch *s, int on);

/* ---- reads --------------------------------------------------------------------------- */
/* khmer.ReadParser stand-in (kevlar/count.py:40): the host hands over parsed sequences
 * (ASCII, concatenated; offs has n_reads+1 entries); the library 2-bit packs them into HBM.
 * Bases outside ACGT are packed as 'A' (khmer's read cleaning) and the read is flagged so
 * that the novel scan skips it (kevlar/novel.py:136-139).                                 */
int kv_reads_create(const char *bases, const uint64_t *offs, uint64_t n_reads, kv_reads **out);
/* Same, for reads that are already 2-bit packed on the host (A=0 C=1 G=2 T=3, base j of read
 * r in word r*words_per_read + j/16 at bits 2*(j%16)), all of length read_len, ACGT only.
 * Used by synthetic-read generators that never materialise ASCII.                        */
int kv_reads_create_packed(const uint32_t *words, uint64_t n_reads, uint32_t read_len, kv_reads **out);
/* Equal-length reads of a synthetic family generated ON the device (kv_synth.hip): reads [first_read, first_read +
 * n_reads) of sample 0 = proband, 1 = mother, 2 = father over an iid-uniform genome of genome_len bases; pure
 * function of (seed, sample, read index).  No reference counterpart (kevlar's test reads came from wgsim,
 * kevlar/tests/data/microtrios/README): it stands where BASELINE.json's config 4 needs 900 M reads per sample.       */
int kv_reads_generate(uint64_t genome_len, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                      uint32_t read_len, double error_rate, kv_reads **out);
/* packed words [first_word, first_word + n_words) of a batch back to the host (tests; kv_reads_create_packed's layout)   */
int kv_reads_words_read(const kv_reads *r, uint64_t first_word, uint64_t n_words, uint32_t *host_out);
int kv_reads_destroy(kv_reads *r);

/* Native FASTA/FASTQ reader (gzip transparent) = khmer.ReadParser (kevlar/count.py:40,
 * kevlar/__init__.py:125-128).  kv_fastx_next parses up to max_reads records; with upload != 0 the
 * sequences are 2-bit packed into a new kv_reads batch in HBM (NULL at end of file).  The text of the
 * batch just parsed (names = header lines after '@'/'>', sequences, qualities; blobs + n+1 offsets;
 * is_fastq[i] = record had a quality line) stays valid until the next kv_fastx_next on the handle.
 * One handle may be shared by several host threads (kevlar/count.py:41-76): kv_fastx_next is
 * serialised, but then only the returned kv_reads -- not the shared batch text -- may be used.      */
typedef struct kv_fastx kv_fastx;
int kv_fastx_open(const char *path, kv_fastx **out);
int kv_fastx_next(kv_fastx *f, uint64_t max_reads, int upload, kv_reads **reads_out, uint64_t *n_reads_out);
int kv_fastx_batch_text(kv_fastx *f, const char **names, const uint64_t **name_offs, const char **seqs,
                        const uint64_t **seq_offs, const char **quals, const uint64_t **qual_offs,
                        const uint8_t **is_fastq);
int kv_fastx_num_reads(kv_fastx *f, uint64_t *n); /* khmer parser.num_reads */
/* Packed-read cache (count and novel share one parse).  With KEVLAR_PACK_CACHE=1 in the environment a
 * complete pass over FILE leaves FILE.kvpack beside it, and kv_fastx_open(FILE) streams from it when it matches the
 * source's size and mtime: kv_fastx_next then uploads the stored packed words (no inflate, no parsing);
 * kv_fastx_batch_text still gives names and all offsets, while sequences and qualities are produced per record by
 * kv_fastx_record_text (byte for byte what the source holds).  kv_fastx_from_cache tells which mode a handle is in. */
int kv_fastx_from_cache(kv_fastx *f, int *yes);
int kv_fastx_record_text(kv_fastx *f, uint64_t i, char *seq_out, char *qual_out);
/* Device ingest: a file that holds four-line FASTQ, uncompressed or as blocked gzip (BGZF: bgzip, htslib,
 * kevlar_amd.open(..., 'w')), and is read as packed batches (upload != 0) from the first call is never parsed on the host:
 * its bytes go to HBM, one wavefront inflates one BGZF member (kv_inflate.hip), lines and records are found and the
 * sequences packed by kernels (kv_fastq.hip).  kv_fastx_on_device tells whether a handle works that way; the text of a batch then
 * stays in HBM, and kv_fastx_fetch(idx, n) brings the named records to the host, after which kv_fastx_batch_text
 * describes exactly those n records.  Anything else (plain gzip, FASTA, blank lines, KV_INGEST=host) is parsed on the
 * host as before, with identical results.                                                                        */
int kv_fastx_on_device(kv_fastx *f, int *yes);
int kv_fastx_fetch(kv_fastx *f, const uint64_t *idx, uint64_t n);
int kv_fastx_close(kv_fastx *f);
int kv_reads_count(const kv_reads *r, uint64_t *n_reads, uint64_t *n_bases);
/* number of k-mers a consume of this batch visits at size k (sum over reads of len-k+1)   */
int kv_reads_num_kmers(const kv_reads *r, int ksize, uint64_t *n_kmers);

/* ---- count: sketch.consume_seqfile[_banding][_with_mask] (kevlar/count.py:43-71) ------ */
/* nbands = 0: no banding; else 0-based `band` of `nbands` (kevlar/count.py:122).
 * mask = NULL or a sketch queried with this sketch's hash:
 *   consume_masked == 0: skip the k-mer if mask.get(h) >  threshold
 *   consume_masked != 0: skip the k-mer unless mask.get(h) >= threshold
 * n_kmers_out: k-mers added.  n_unique_kmers is accumulated with the semantics of khmer's
 * multi-threaded consume (a k-mer counts as new if any of its bins was zero when ITS
 * increment landed); kv_unique_exact() gives the single-thread file-order value.   */
int kv_consume(kv_sketch *s, const kv_reads *reads, int nbands, int band, const kv_sketch *mask,
               int threshold, int consume_masked, uint64_t *n_kmers_out);
/* Re-derive n_unique_kmers exactly as a single khmer thread would have counted it over
 * `batches` consumed in this order into an initially empty sketch (needs 4 bytes of HBM
 * scratch per bin).  Used by `kevlar count` when --threads 1.                             */
int kv_unique_exact(kv_sketch *s, const kv_reads *const *batches, int n_batches, int nbands,
                    int band, const kv_sketch *mask, int threshold, int consume_masked,
                    uint64_t *n_unique_out);
/* The same figure one batch at a time: *n_new_out = the k-mers of `batch` that khmer's single thread would count as new if the batch
 * were consumed into `s` now (a bin of the k-mer is clear in the tables as they stand, and no earlier k-mer of this batch touches it
 * first).  Call it before every kv_consume of the batch with the same band / mask arguments and add the results up: the sum is the
 * "distinct k-mers stored" of kevlar/count.py:82-84 for a sample of any size, no batch kept resident.                                */
int kv_unique_new(kv_sketch *s, const kv_reads *batch, int nbands, int band, const kv_sketch *mask, int threshold,
                  int consume_masked, uint64_t *n_new_out);
/* kv_unique_new keeps its first-toucher arrays (4 bytes per bin and table, ~4.5 x the sketch, one set per stream) between the batches
 * of a sample; this gives back every set no running call holds.  The wrapper calls it when a sample is done
 * (kevlar_amd.khmer: track_exact_unique(False)); kv_scratch_trim and an out-of-memory retry inside the library do the same.          */
int kv_unique_release(void);

/* ---- point queries: .get / .add on many k-mers (kevlar/filter.py:32-34,67) ------------ */
/* hash n k-mers of length k stored back to back in `kmers` (device kernel)                */
int kv_hash_kmers(int kind, const char *kmers, int k, uint64_t n, uint64_t *hashes_out);
/* the same hashes for k-mers given by position: k-mer i is reads[ann_read[i]][ann_offset[i] : + ksize].  What
 * kevlar/filter.py:32-34,67 and kevlar/readgraph.py:60-66 obtain by slicing record.ikmerseq() and calling the
 * sketch once per k-mer; here the annotated reads stay packed in HBM and only (read, offset) pairs travel.      */
int kv_hash_positions(const kv_reads *reads, int kind, int ksize, const uint32_t *ann_read, const uint32_t *ann_offset,
                      uint64_t n, uint64_t *hashes_out);
int kv_get_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *counts_out);
/* adds in array order semantics are order-free (saturating); is_new_out may be NULL       */
int kv_add_hashes(kv_sketch *s, const uint64_t *hashes, uint64_t n, uint8_t *is_new_out);

/* ---- dist: khmer's abundance_distribution(parser, tracking) (kevlar/dist.py:47-77) ----------
 * In read order, a k-mer that `tracking` has not seen is recorded there and bumps
 * hist_out[counts.get(kmer)]; hist_out has 65536 entries.  Same result as the single-thread loop. */
int kv_abundance_distribution(kv_sketch *counts, kv_sketch *tracking, const kv_reads *const *batches,
                              int n_batches, uint64_t *hist_out);

/* ---- novel: the fused scan (kevlar/novel.py:21-53,123-169) ---------------------------- */
#define KV_BAND_NONE 0
#define KV_BAND_RANGE 1     /* keep k-mers whose hash lies in the band (count-side rule)     */
#define KV_BAND_REFQUIRK 2  /* reference literal: (h & (N-1)) != band0-1 -> skip
                               (kevlar/novel.py:144-147)        */
/* Scans reads[first_read:] ; a read shorter than k or flagged non-ACGT is skipped.
 * screen_thresh <= 0 disables --abund-screen.  If d_mask != NULL (device pointer, u32
 * words, zeroed by the caller) bit (read * mask_stride + offset) is set for every
 * interesting k-mer: the per-band mask that the multi-GPU merge all-reduces.             */
int kv_novel_scan(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                  const kv_reads *reads, uint64_t first_read, int case_min, int ctrl_max,
                  int screen_thresh, int band_mode, int nbands, int band, uint32_t *d_mask,
                  uint64_t mask_stride, kv_hits **out);
/* on != 0, for the calling host thread: a kv_novel_scan (without an abundance screen) returns as soon as its kernels are done -- the
 * sketches may be cleared and counted into again -- while the hit arrays are still being copied to the host on a stream of their own;
 * kv_hits_view / kv_hits_fetch / kv_hits_shadow / kv_hits_destroy wait for them (kv_hits_count does not need to).  A caller that scans
 * batch after batch reads the hits of one scan while the next count runs.                                                          */
int kv_hits_lazy(int on);
int kv_hits_count(const kv_hits *h, uint64_t *n_hits, uint64_t *n_discarded_reads);
/* copies hits sorted by (read, offset); abund has n_hits * (ncase+nctrl) entries          */
int kv_hits_fetch(const kv_hits *h, uint32_t *read, uint32_t *offset, uint8_t *abund,
                  uint64_t cap_hits, uint32_t *discarded_reads, uint64_t cap_discarded);
/* zero-copy alternative to kv_hits_fetch: pointers into the handle's (pinned) host arrays, valid until
 * kv_hits_destroy; sizes as reported by kv_hits_count                                              */
int kv_hits_view(const kv_hits *h, const uint32_t **read, const uint32_t **offset, const uint8_t **abund,
                 const uint32_t **discarded_reads);
/* --abund-screen only: interesting k-mers of DISCARDED reads that lie in front of the k-mer that tripped the screen.
 * The reference has already added them to its tally of unique novel k-mers when it drops the read
 * (kevlar/novel.py:152-164), so the summary line needs them although the reads are not reported.             */
int kv_hits_shadow(const kv_hits *h, const uint32_t **read, const uint32_t **offset, uint64_t *n);
int kv_hits_destroy(kv_hits *h);

/* ---- read-sharded multi-GPU count / scan (kevlar's banding, docs/banding.rst,
 *      kevlar/count.py:62-66, with the hashing done once per k-mer instead of once per band) ------
 * All buffer arguments named d_* are DEVICE pointers owned by the caller (bench.py and
 * kevlar_amd/shardrun.py hand in torch tensors so RCCL can move them); only kernels touch them.    */
/* Hash every k-mer of `reads` and group the hashes by the band (= rank) that owns them: d_out receives the
 * items of destination 0, then destination 1, ... back to back (cap_items >= the k-mers of the shard), the
 * layout an all-to-all with split sizes sends as it is; an item is a u64 hash or, with_tags, the pair
 * (hash, tag) with tag = (read_index_base + read) << 16 | offset and bit 63 set for reads the novel scan skips
 * (non-ACGT).  counts_out[ndest] (host) = items per destination.  kind = KV_COUNTTABLE ... selects the hash
 * function as in kv_hash_kmers.                                                                            */
int kv_route_hashes(const kv_reads *reads, int kind, int ksize, int ndest, uint64_t read_index_base,
                    int with_tags, void *d_out, uint64_t cap_items, uint64_t *counts_out);
/* kv_route_hashes for a count only, with the shard deduplicated first: an item is the pair (hash, occurrences
 * in this shard) and there is one per distinct k-mer of a super-k-mer bucket (so at most -- and at sequencing
 * coverage far fewer than -- one per k-mer; cap_items >= the k-mers of the shard still).  Same d_out layout
 * and counts_out as kv_route_hashes with two-word items.                                                    */
int kv_route_distinct(const kv_reads *reads, int kind, int ksize, int ndest, void *d_out,
                      uint64_t cap_items, uint64_t *counts_out);
/* count n (hash, count) items resident in HBM: each adds min(count, 255) to its bins, saturating -- the tables
 * end up as if the hash had been counted `count` times.  *n_added_out = sum of the counts.                  */
int kv_consume_hashes_weighted(kv_sketch *s, const void *d_items, uint64_t n, uint64_t *n_added_out);
/* (hash, occurrences) pairs as they travel between ranks: 9 bytes instead of 16.  A block of n pairs for one destination becomes
 * 1 + n + ceil(n / 8) 64-bit words -- the block's occurrences in all (exact: the statistic `kevlar count` prints), the n hashes, the n
 * occurrence counts as bytes saturated at 255 (no counter of any sketch kind holds more: `Hashtable::add` saturates).
 * kv_pairs_pack: d_pairs holds counts[0] pairs for destination 0, then counts[1] for destination 1, ...; d_out receives the blocks back
 * to back, words_per_dest[d] words each.  kv_pairs_unpack: d_in holds nsrc received blocks of words_per_src[s] words; d_pairs receives
 * their pairs as 16-byte (hash, count <= 255) items, pairs_per_src[s] each; *occurrences = the sum of the blocks' totals.              */
int kv_pairs_pack(const void *d_pairs, const uint64_t *counts, int ndest, void *d_out, uint64_t out_cap_words, uint64_t *words_per_dest);
int kv_pairs_unpack(const void *d_in, const uint64_t *words_per_src, int nsrc, void *d_pairs, uint64_t cap_pairs, uint64_t *pairs_per_src,
                    uint64_t *occurrences);
/* count n hashes resident in HBM, element i at ((uint64_t*)d_hashes)[i * stride_words]            */
int kv_consume_hashes(kv_sketch *s, const void *d_hashes, uint64_t n, uint32_t stride_words,
                      uint64_t *n_added_out);
/* kmer_is_interesting() over n_items (hash, tag) pairs; hits leave unordered as d_hit_tags[i] and
 * d_hit_abund[i * (ncase+nctrl) ...]                                                              */
int kv_novel_scan_hashes(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                         const void *d_items, uint64_t n_items, int case_min, int ctrl_max,
                         void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits);
/* kv_novel_scan_hashes over the (hash, occurrences) pairs a band owner received of the case sample
 * (kv_route_distinct): an interesting pair leaves as its hash (d_hit_hashes[i]) and abundances.    */
int kv_novel_scan_distinct(kv_sketch *const *cases, int ncase, kv_sketch *const *ctrls, int nctrl,
                           const void *d_items, uint64_t n_items, int case_min, int ctrl_max,
                           void *d_hit_hashes, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits);
/* the scan of a read shard against a known set of interesting k-mers: d_hashes[n] (entries ~0 are
 * padding) with d_abund[n * nsamples]; every k-mer of `reads` whose hash is in the set is a hit, with
 * the set's abundances; reads with non-ACGT are skipped as in kv_novel_scan; (read, offset) order.  */
int kv_novel_scan_set(const kv_reads *reads, int kind, int ksize, int nsamples, const void *d_hashes,
                      const void *d_abund, uint64_t n, kv_hits **out);
/* sort n_total gathered (tag, abundances) hits by tag; the n_valid smallest are real (padding has
 * tag ~0) and come back as an ordinary kv_hits in (read, offset) order                            */
int kv_hits_from_tagged(const void *d_tags, const void *d_abund, uint64_t n_total, uint64_t n_valid,
                        int nsamples, kv_hits **out);

/* ---- partition: read graph connected components (kevlar/readgraph.py:43-84,104-137) --- */
/* One annotation = one interesting k-mer occurrence (read index in `reads`, offset).
 * node_of_read maps a read to its graph node (reads sharing a name share a node).
 * K-mers are identified by their canonical (min of forward / reverse-complement) sequence;
 * a k-mer links all nodes containing it if minabund <= #nodes <= maxabund (0 = unbounded).
 * labels_out[node] = smallest node id of its component.  n_edges_out (optional) = number
 * of distinct node pairs sharing at least one retained k-mer (networkx number_of_edges).  */
int kv_readgraph_components(const kv_reads *reads, int ksize, const uint32_t *ann_read,
                            const uint32_t *ann_offset, uint64_t n_ann,
                            const uint32_t *node_of_read, uint32_t n_nodes, uint32_t minabund,
                            uint32_t maxabund, uint32_t *labels_out, uint64_t *n_edges_out);

/* ---- augmented FASTA/FASTQ text of a batch's annotated reads (kevlar/sequence.py print_augmented_fastx) ----
 * hits (read, offset, abund[nsamples]) sorted by (read, offset); the j-th distinct read among them is record
 * rec_index[j] of the text blobs (kv_fastx_batch_text's layout).  *text_out is released with kv_text_free.   */
int kv_format_augmented(const uint32_t *hit_read, const uint32_t *hit_off, const uint8_t *abund, uint64_t n_hits,
                        int nsamples, int ksize, const uint64_t *rec_index, const char *names,
                        const uint64_t *name_offs, const char *seqs, const uint64_t *seq_offs, const char *quals,
                        const uint64_t *qual_offs, const uint8_t *is_fastq, char **text_out, uint64_t *bytes_out,
                        uint64_t *n_records_out);
int kv_text_free(char *text);

/* ---- augmented FASTA/FASTQ files as flat arrays (kevlar/sequence.pyx parse_augmented_fastx / print_augmented_fastx):
 * what `kevlar filter` (filter.py:15-82) and `kevlar partition` (readgraph.py:43-84) read and write.  A loaded file
 * exposes its records as kv_fastx_batch_text-style blobs plus, per annotation, offset and abundances (record i owns
 * annotations ann_first[i] .. ann_first[i + 1]); ksize = -1 if the stream mixes k.  kv_format_records writes any
 * selection of records / annotations back as text (see kv_augfastx.hip); release it with kv_text_free.            */
typedef struct kv_augfastx kv_augfastx;
int kv_augfastx_load(const char *path, kv_augfastx **out);
int kv_augfastx_info(const kv_augfastx *a, uint64_t *n_records, uint64_t *n_annotations, int *ksize, int *nsamples,
                     uint64_t *n_mates);
int kv_augfastx_view(const kv_augfastx *a, const char **names, const uint64_t **name_offs, const char **seqs,
                     const uint64_t **seq_offs, const char **quals, const uint64_t **qual_offs,
                     const uint8_t **is_fastq, const uint64_t **ann_first, const uint32_t **ann_offset,
                     const int32_t **ann_abund, const uint32_t **mate_record, const char **mates,
                     const uint64_t **mate_offs);
int kv_augfastx_free(kv_augfastx *a);
int kv_format_records(uint64_t n_out, const uint64_t *rec_index, const uint64_t *ann_lo, const uint64_t *ann_hi,
                      const uint32_t *ann_offset, const int32_t *ann_abund, const uint8_t *keep,
                      const int32_t *case_abund, int nsamples, int ksize, const char *names,
                      const uint64_t *name_offs, const char *seqs, const uint64_t *seq_offs, const char *quals,
                      const uint64_t *qual_offs, const uint8_t *is_fastq, const char *suffix,
                      const uint64_t *suffix_offs, const uint32_t *mate_record, uint64_t n_mates, const char *mates,
                      const uint64_t *mate_offs, char **text_out, uint64_t *bytes_out);
/* kv_format_records with the text written straight to file descriptor `fd` (a regular file, a pipe): rendered a stretch of records
 * at a time on `nthreads` host threads, written in order; *bytes_out = bytes written.  What `kevlar filter` / `kevlar partition` /
 * `kevlar novel` do with the text of print_augmented_fastx (kevlar/sequence.pyx:93-126) when their output is a plain file.  On an
 * error (an annotation outside its read, no memory, a failed write) a seekable `fd` is truncated back to where the call found it and
 * *bytes_out = 0; a pipe keeps the stretches already written (*bytes_out says how much).                                          */
int kv_format_records_fd(uint64_t n_out, const uint64_t *rec_index, const uint64_t *ann_lo, const uint64_t *ann_hi,
                         const uint32_t *ann_offset, const int32_t *ann_abund, const uint8_t *keep,
                         const int32_t *case_abund, int nsamples, int ksize, const char *names,
                         const uint64_t *name_offs, const char *seqs, const uint64_t *seq_offs, const char *quals,
                         const uint64_t *qual_offs, const uint8_t *is_fastq, const char *suffix,
                         const uint64_t *suffix_offs, const uint32_t *mate_record, uint64_t n_mates, const char *mates,
                         const uint64_t *mate_offs, int fd, int nthreads, uint64_t *bytes_out);
/* the dedup key of `kevlar partition` (kevlar/partition.py:37-47: kevlar.revcommin(read.sequence)) as two 64-bit hashes per read:
 * of the sequence or its reverse complement (complement[256]: the byte table of revcom), whichever sorts first.  Host only.     */
/* flags[i] = 1 if read i (seqs[seq_offs[i] .. seq_offs[i + 1])) holds a byte other than upper-case A, C, G, T: the packed form cannot
 * hold it, and the host hashes that read's k-mers from the text (the reference hashes every k-mer from its string).  Host only. */
int kv_reads_flag_other_bytes(const char *seqs, const uint64_t *seq_offs, uint64_t n, uint8_t *flags);
int kv_canonical_read_hashes(const char *seqs, const uint64_t *seq_offs, const uint64_t *reads, uint64_t n,
                             const uint8_t *complement, uint64_t *h1, uint64_t *h2);
/* same[j] = 1 if reads a[j] and b[j] have the same canonical sequence, min(sequence, reverse complement) by byte order: what
 * kevlar/partition.py:26-33 compares; partition confirms equal-hash pairs with it.  Host only.                                   */
int kv_canonical_reads_equal(const char *seqs, const uint64_t *seq_offs, const uint64_t *a, const uint64_t *b, uint64_t n,
                             const uint8_t *complement, uint8_t *same);

/* ---- blocked gzip (BGZF) on the device (kevlar_amd/csrc/kv_inflate.hip) --------------------------------
 * Replaces the gzip stream behind khmer.ReadParser (kevlar/__init__.py:125-128) for files whose members are
 * independent: one wavefront inflates one member.  kv_fastx_open picks this path by itself; the two
 * functions below take a whole file image in host memory (tests, tools).                                   */
int kv_bgzf_text_size(const void *file, uint64_t size, uint64_t *text_bytes, uint64_t *n_members);
int kv_bgzf_inflate_host(const void *file, uint64_t size, void *out, uint64_t out_cap, double *kernel_ms);

/* ---- ordinary gzip on the device (kevlar_amd/csrc/kv_gunzip.hip) ----------------------------------------
 * The same reader (khmer.ReadParser over a *.gz, kevlar/__init__.py:125-128) for a file that is ONE deflate
 * stream (gzip, pigz): block starts are searched for in parallel, every stretch between two starts is decoded
 * by one wavefront with the unknown 32 KB in front of it as markers, the markers are resolved by pointer
 * doubling.  kv_fastx_open picks this path by itself; this function takes a whole file image in host memory
 * and returns its text (tests, tools).  segment_text: bytes of text per pass over the device (0: 1 GB);
 * stats[4]: passes, stretches decoded, stretches dropped (false starts), stretches decoded again.
 * KV_ERR_TYPE: the stream needs the host's zlib (kv_fastx_next falls back by itself).                      */
int kv_gunzip_host(const void *file, uint64_t size, void *out, uint64_t out_cap, uint64_t segment_text,
                   uint64_t *text_bytes, uint64_t *stats, double *device_ms);

/* ---- minimizer-sharded exchange (multi-GPU count; no reference counterpart beyond banding itself, docs/banding.rst) ------
 * kevlar splits a trio over workers by k-mer band, and every worker reads every read (kevlar/count.py:62-66).  On one node
 * rank r instead holds reads [r n/N, (r+1) n/N) of every sample.  kv_mex_emit cuts its shard into super-k-mer records and
 * leaves them grouped by minimizer bucket in the caller's exchange buffer (records: plan->seg_words u64 words laid out
 * [C1][nwg1][cap1][recw]; counts: plan->cnt_entries u32, [C1][nwg1]); bucket range [c_lo[d], c_lo[d+1]) is rank d's, so one
 * all-to-all with those split points delivers every occurrence of a k-mer -- whichever shard it came from -- to one rank.
 * kv_mex_route combines what n_src ranks sent (the slabs one after the other, as all_to_all_single leaves them) at the
 * sample's full coverage and writes one (hash, occurrences) pair per distinct k-mer for the hash band's owner: the output
 * of kv_route_distinct, which the band owners add with kv_consume_hashes_weighted.  Every rank derives the same plan from
 * the sample's global size.                                                                                              */
typedef struct kv_mex_plan {
    int32_t ksize, ndest;
    uint32_t C1, F2, fbits, nwg1, cap1, recw, m, read_len;
    uint64_t seg_words, cnt_entries;
    uint64_t n_kmers_global, n_reads_global;
    uint32_t c_lo[17];
    uint32_t flags;             /* bit 0: short records (kv_mex_plan_short) */
} kv_mex_plan;
int kv_mex_plan_make(int kind, int ksize, uint64_t n_reads_global, uint32_t read_len, int ndest, kv_mex_plan *plan);
/* The same exchange with 16-byte records without positions (two thirds of the bytes a rank sends) for a sample nobody asks "where" of:
 * a control.  Not for the sample kv_mex_route(keep_scan) / kv_mex_scan_set answer the scan from.  KV_ERR_NOTIMPL when the plan's shape
 * has no such records (k > 32, a window other than k = 31's, reads beyond 224 bases): keep the plan as it is. */
int kv_mex_plan_short(kv_mex_plan *plan);
int kv_mex_emit(const kv_reads *shard, const kv_mex_plan *plan, uint64_t read_base, void *d_seg, void *d_cnt);
/* kv_mex_pack: only the filled part of the segments travels -- d_out receives it, destination after destination, and
 * records_per_dest[d] says how many records rank d gets; the counts slab travels whole, so the receiver (compact != 0 in
 * kv_mex_route) knows where every segment of every source starts.                                                        */
int kv_mex_pack(const kv_mex_plan *plan, const void *d_seg, const void *d_cnt, void *d_out, uint64_t *records_per_dest);
/* keep_scan != 0 (the case sample): the combined buckets and key + hash of every distinct k-mer stay on this stream until the next
 * call that buckets anything, for kv_mex_scan_set.                                                                          */
int kv_mex_route(const kv_mex_plan *plan, int my_dest, const void *d_recv_seg, const void *d_recv_cnt, int n_src, int compact,
                 int keep_scan, void *d_out, uint64_t cap_items, uint64_t *counts_out, uint64_t *n_kmers_in);
/* kv_mex_scan_set: the scan (kevlar/novel.py:123-169) answered by the owner of the minimizer buckets.  d_hashes[n] / d_abund[n * S]:
 * every band owner's kv_novel_scan_distinct output, all-gathered (entries ~0 are padding).  Every occurrence of a member of that
 * set among this rank's buckets leaves as d_hit_tags[i] = read << 16 | offset (global read index) with its S abundances, in
 * arbitrary order: gather them and sort with kv_hits_from_tagged.  Reads the scan skips (bytes outside ACGT) are NOT left out:
 * the caller drops their hits.  KV_ERR_CAPACITY: the state kv_mex_route(keep_scan) leaves is not there (or does not suffice);
 * scan the shard against the set instead (kv_novel_scan_set).                                                                */
int kv_mex_scan_set(int kind, int ksize, int nsamples, const void *d_hashes, const void *d_abund, uint64_t n,
                    void *d_hit_tags, void *d_hit_abund, uint64_t hit_cap, uint64_t *n_hits);
/* flags[r] bit 0: read r holds a byte outside ACGT (counted with stand-in bases, skipped by the scan); host buffer of n_reads bytes */
int kv_reads_flags(const kv_reads *reads, uint8_t *flags);
/* kv_mex_emit_pack: kv_mex_emit and kv_mex_pack in one call and one stream synchronisation.  d_out holds out_cap_words u64 words;
 * *packed = 1: the filled part fitted and sits in d_out; 0: it did not (records_per_dest is right either way), nothing was
 * written to d_out and kv_mex_pack into a buffer of plan->seg_words words does it.                                          */
int kv_mex_emit_pack(const kv_reads *shard, const kv_mex_plan *plan, uint64_t read_base, void *d_seg, void *d_cnt, void *d_out,
                     uint64_t out_cap_words, uint64_t *records_per_dest, int *packed);

/* ---- argsort for the host half of partition (kevlar/readgraph.py:123-161, kevlar/partition.py:15-55: sorted(cc), sorted by
 * size; here the array form, kevlar_amd/partition.py assemble_partitions) -------------------------------------------------
 * order[i] = index of the i-th smallest key, equal keys in index order (numpy.argsort(kind='stable')); host pointers, the
 * sort runs on the device (rocPRIM radix sort of (key, index) pairs).  kv_argsort_rows orders n rows of `width` bytes like
 * byte strings (numpy 'S<width>').                                                                                       */
int kv_argsort_u64(const uint64_t *keys, uint64_t n, uint32_t *order);
int kv_argsort_rows(const void *rows, uint64_t n, uint32_t width, uint32_t *order);

#ifdef __cplusplus
}
#endif
#endif /* KVSKETCH_H */
