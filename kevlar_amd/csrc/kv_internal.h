// kv_internal.h -- shared declarations of libkvsketch_hip (gfx950 only; no CUDA path).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include "kv_knobs.h"

#include "../../include/kvsketch.h"
#include "kv_fastmod.h"

enum { ST_BYTE = 0, ST_NIBBLE = 1, ST_BIT = 2 };
enum { HF_MURMUR = 0, HF_TWOBIT = 1 };

static inline int kv_storage_of(int kind)
{
    switch (kind) {
    case KV_COUNTTABLE: case KV_COUNTGRAPH: return ST_BYTE;
    case KV_SMALLCOUNTTABLE: case KV_SMALLCOUNTGRAPH: return ST_NIBBLE;
    default: return ST_BIT;
    }
}
static inline int kv_hashfam_of(int kind) { return kind >= KV_COUNTGRAPH ? HF_TWOBIT : HF_MURMUR; }
static inline uint64_t kv_table_nbytes(int storage, uint64_t size)
{
    return storage == ST_BYTE ? size : (storage == ST_NIBBLE ? size / 2 + 1 : size / 8 + 1);
}

// Device-visible description of one sketch.  Lives in HBM (kv_sketch::d_desc) so kernels
// read it through scalar loads; also kept on the host.
struct SketchDev {
    uint64_t size[KV_MAX_TABLES];
    uint64_t magic[KV_MAX_TABLES];  // kv_fastmod_magic(size): the reciprocal fastmod() wants for this size (kv_fastmod.h)
    uint8_t *tab[KV_MAX_TABLES];
    int32_t ntables, storage, hashfam, ksize;
};

// Certificates of abundance (kv_skm.hip): the distinct k-mers ONE super-k-mer count added to a sketch at least twice, with
// how often (capped at 255), listed bucket by bucket in the bucket geometry of that count.  Counters never decrease, so
// an entry stays a valid lower bound of get(k-mer) until the sketch is cleared or overwritten; the novel scan uses the
// controls' lists to reject inherited k-mers without probing the controls' tables (Count-Min never under-counts, so
// "added c > ctrl_max times" already decides kmer_is_interesting(), kevlar/novel.py:43-50).
struct KvAbundList {
    void *mem = nullptr;
    size_t bytes = 0;
    uint64_t *keys = nullptr;      // [cap_total][kw]
    uint8_t *cnts = nullptr;       // [cap_total]
    uint32_t *bstart = nullptr;    // [n_buckets] first entry of the bucket
    uint32_t *bcount = nullptr;    // [n_buckets] entries (0: none, or the writer ran out of room: the scan then just probes)
    int k = 0, m = 0, kw = 0;
    uint32_t C1 = 0, F2 = 0, fbits = 0, n_buckets = 0, nwg = 0;
    uint64_t cap_total = 0, cap_wg = 0;
    bool valid = false;
};

struct kv_sketch {
    int kind;
    SketchDev h;          // host copy
    SketchDev *d_desc;    // device copy
    uint64_t alloc_bytes[KV_MAX_TABLES];
    uint64_t n_occupied;  // valid when !occ_dirty
    bool occ_dirty;
    uint64_t n_unique;
    uint64_t uid = 0;      // unique per allocation (pointers get recycled)
    int device = 0;        // the device the tables were allocated on (what the table cache files them under)
    uint64_t version = 0;  // bumped by everything that changes a table (invalidates cached scan verdicts)
    uint64_t *d_counters; // [0] n_kmers, [1] n_unique (device accumulators)
    bool skm_off = false;  // the last batch counted through the super-k-mer front end did not deduplicate: skip it until cleared
    uint64_t skm_off_kmers = 0;  // ... and how many k-mers that batch had.  This outlives kv_sketch_clear: a sketch that is cleared and filled
                                 // again with batches of the same size (a sample counted step after step, band after band) does not cut,
                                 // bucket and combine its first batch every time just to find the same thing (0.25 of config 4's 2.7 s per
                                 // step, and 30 GB of buckets per stream); a batch of another size is tried afresh, a success forgets it
    double skm_distinct = 0.0;   // distinct / all k-mers of that batch (0: none yet): sizes the buckets of the next one
    // the scan's own memory (kv_novel_scan, this sketch as the first case sample): the last batch it cut into super-k-mers did not
    // fit the tables -- nothing to deduplicate at that coverage -- and was scanned again tile by tile; batches that follow go
    // straight to the tile scan until the sketch is cleared.  (The count cannot say: a banded count of a sparse batch never
    // takes the super-k-mer front end, so skm_off stays unset there.)
    std::atomic<bool> skm_scan_off{false};       // (scans of several batches of one case sample run side by side: bench.py scan_batches)
    // kv_sketch_clear only notes that the tables are zero: the partitioned count's apply stage, which rewrites every
    // slice anyway, then starts from zeroed LDS instead of loading the slice (no memset, no first read of the tables);
    // every other reader or writer of the tables calls kv_sketch_ready first, which does the memset after all
    bool lazy_zero = false;
    KvAbundList abl;       // see above; valid = false whenever the tables may hold less than it says
    bool scan_hint = false; // kv_sketch_scan_hint: batches counted into this sketch are scanned next (a case sample)
    bool scan_steady = false; // ... and the process does so sample after sample: the distinct list pays from the first batch on
    std::mutex mu;
};

// One unit of work of the hashing kernels: `count` whole reads starting at read `first`, or (seg != 0) the
// segment of read `first` whose k-mers START in [seg_start, seg_start + KV_SEG_BASES): the kernel stages
// KV_SEG_BASES + k - 1 bases, so every k-mer of a chromosome-length sequence belongs to exactly one tile
// whatever k is (the tile table itself does not depend on k).
struct TileDesc {
    uint32_t first, count, seg_start, seg;
};

// Packed read batch.  Read r occupies words [woff[r], woff[r+1]) of `words`; base j sits in
// word woff[r] + j/16 at bits 2*(j%16), code A=0 C=1 G=2 T=3.
uint64_t kv_next_uid();   // process-wide, never reused (device pointers are)

struct kv_reads {
    uint64_t uid = kv_next_uid();
    uint64_t n_reads, n_bases, n_words;
    uint32_t *d_words;
    uint64_t *d_woff;   // n_reads + 1
    uint32_t *d_len;    // n_reads
    uint8_t *d_flags;   // n_reads: bit0 = contains a base outside ACGT (novel scan skips it)
    struct TileDesc *d_tile;   // n_tiles descriptors: a run of whole reads, or one segment of a long read
    uint32_t n_tiles;
    uint32_t tile_lds_bytes;  // dynamic LDS the tile kernels need for this batch (>= KV_TILE_LDS_BYTES)
    uint32_t max_len;
    uint32_t tile_max_bases = 0;    // most bases any tile stages (a segment tile: KV_SEG_BASES + KV_MAX_K); 0 = not computed
    uint32_t uni_len = 0, uni_per_tile = 0;   // all reads have this length and tile t holds reads [t * uni_per_tile, ...): the layout is
                                              // arithmetic, and a kernel can fetch the next tile's words while it works on this one
    std::vector<uint32_t> h_len;    // host copies (k-mer counting, hit bookkeeping)
    int nk_cached_k = -1;           // kv_reads_num_kmers memo (the length vector can hold 1e7+ entries)
    uint64_t nk_cached = 0;
};

// the deferred memset of kv_sketch_clear, if one is pending (s->mu held by the caller / taken here)
int kv_sketch_ready_locked(kv_sketch *s);
int kv_sketch_ready(const kv_sketch *s);

int kv_reads_from_packed_var(const uint32_t *words, const uint32_t *lens, const uint8_t *flags, uint64_t n_reads, kv_reads **out);
// the same for sequences that sit as text in HBM (kv_fastq.hip): read r = d_seq_len[r] characters at d_text + d_seq_start[r];
// `lens` is the host copy of d_seq_len; packing and flagging happen on the device
int kv_reads_from_device_text(const uint8_t *d_text, const uint64_t *d_seq_start, const uint32_t *d_seq_len, const uint32_t *lens,
                              uint64_t n_reads, kv_reads **out);
void kv_fastq_pack_launch(const uint8_t *d_text, const uint64_t *d_seq_start, const uint32_t *d_seq_len, const uint64_t *d_woff, uint64_t n_reads,
                          uint64_t n_words, uint32_t *d_words, uint32_t *d_flags32, hipStream_t st);

// hits of one scan in pinned host memory (fast DMA from the device; the Python side views it in place).
// Blocks come from a small recycling pool (kv_host.hip): hipHostMalloc costs ~1 ms per call, more than
// copying the hits of a whole scan.
void *kv_pinned_get(size_t bytes, size_t *capacity);
void kv_pinned_put(void *p, size_t capacity);

template <typename T>
struct PinnedVec {
    T *p = nullptr;
    uint64_t n = 0;
    size_t cap = 0;
    ~PinnedVec() { if (p) kv_pinned_put(p, cap); }
    hipError_t resize(uint64_t count)
    {
        if (p) { kv_pinned_put(p, cap); p = nullptr; cap = 0; }
        n = count;
        if (count == 0) return hipSuccess;
        p = (T *)kv_pinned_get(count * sizeof(T), &cap);
        return p ? hipSuccess : hipErrorOutOfMemory;
    }
    uint64_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](uint64_t i) { return p[i]; }
    const T &operator[](uint64_t i) const { return p[i]; }
    bool empty() const { return n == 0; }
};

// Small device -> host read-backs of one call gathered in one pinned block and waited for once.  (A hipMemcpyAsync into pageable
// memory -- a variable on the stack -- stalls the caller until the copy is done: every such copy is a synchronisation of its own.)
struct KvReadback {
    unsigned char *host = nullptr;
    size_t cap = 0, used = 0;
    bool pinned = true;
    KvReadback()
    {
        host = (unsigned char *)kv_pinned_get(4096, &cap);
        if (!host) { host = new unsigned char[4096]; cap = 4096; pinned = false; }
    }
    ~KvReadback() { if (pinned) kv_pinned_put(host, cap); else delete[] host; }
    KvReadback(const KvReadback &) = delete;
    KvReadback &operator=(const KvReadback &) = delete;
    // enqueue the copy of `count` values; the returned pointer is valid after wait()
    template <typename T>
    const T *add(const T *dev, size_t count, hipStream_t st, hipError_t *err)
    {
        const size_t at = (used + 7) & ~(size_t)7, bytes = count * sizeof(T);
        if (at + bytes > cap) { *err = hipErrorOutOfMemory; return nullptr; }
        const hipError_t e = hipMemcpyAsync(host + at, dev, bytes, hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) *err = e;
        used = at + bytes;
        return (const T *)(host + at);
    }
    hipError_t wait(hipStream_t st) { return hipStreamSynchronize(st); }
};

struct kv_hits {
    int nsamples;
    PinnedVec<uint32_t> read, offset;
    PinnedVec<uint8_t> abund;
    std::vector<uint32_t> discarded;
    std::vector<uint32_t> shadow_read, shadow_offset;   // interesting k-mers of discarded reads in front of the screen trip
    // kv_hits_lazy: the hit arrays are still on their way from the device (a copy stream of their own); every reader waits for this
    // event first, and so does the destructor before the pinned arrays go back to their pool
    hipEvent_t ready = nullptr;
    void wait() const { if (ready) (void)hipEventSynchronize(ready); }
    ~kv_hits() { if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); } }
};

struct HashParams {
    int k, nblocks, rem;   // k = 16*nblocks + rem
    uint64_t m1, m2;       // byte masks for the murmur tail words
    int hashfam;
};

// band + mask predicate of consume_seqfile[_banding][_with_mask] (kevlar/count.py:43-71)
struct ConsumeFilter {
    HashParams hp;
    int use_band;
    uint64_t band_lo, band_hi;
    int use_mask, threshold, consume_masked;
};

// partitioned count (kv_binned.hip)
bool kv_binned_eligible(const kv_sketch *s, const kv_reads *reads, uint64_t n_kmers, int nbands);
int kv_consume_binned(kv_sketch *s, const kv_reads *reads, const uint64_t *d_list, uint32_t list_stride,
                      const ConsumeFilter &filter, const kv_sketch *mask, uint64_t n_kmers, int nbands, uint64_t *n_added,
                      bool weighted_list = false);
double kv_estimate_distinct(uint64_t occupied, uint64_t size);

// super-k-mer front end (kv_skm.hip): the batch is deduplicated in minimizer buckets and each DISTINCT k-mer is
// hashed, filtered and counted (or evaluated by the novel scan) once
struct NovelParams;
bool kv_skm_eligible(const kv_sketch *s, const kv_reads *reads, uint64_t n_kmers, bool for_scan);
bool kv_skm_eligible_kind(int hashfam, int ksize, const kv_reads *reads, uint64_t n_kmers, bool for_scan);
int kv_consume_skm(kv_sketch *s, const kv_reads *reads, const ConsumeFilter &filter, const kv_sketch *mask,
                   uint64_t n_kmers, int nbands, uint64_t *n_added);
// sets the mask bit of every interesting k-mer of reads[first_read:] (p.mask / p.mask_stride) and p.tile_count
int kv_skm_novel_mark(const kv_reads *reads, const NovelParams &p, uint64_t n_kmers);
// hits per tile = set bits of the tile's stretch of the hit mask (p.tile_count), for scans that only mark (k_tile_hits, kv_skm.hip)
void kv_tile_hits_launch(const kv_reads *reads, const NovelParams &p, hipStream_t st);
// will a scan of `reads` go by the distinct list its count pass left (k_skm_novel_list)?  (the caller then sets up the table of
// the interesting k-mers' abundances that kernel fills)
bool kv_skm_list_ready(const kv_reads *reads, int ksize);

// Read-sharded multi-GPU count (kv_shard.hip): where routed items go.  Every writer workgroup owns one private segment
// per destination (= band = rank) and appends through a cursor in LDS; items that do not fit go to a shared overflow list.
struct KvRouteSink {
    int ndest;
    uint32_t nwg;                // writer workgroups
    uint64_t bs;                 // band width UINT64_MAX / ndest (kv_band_bounds)
    uint64_t seg_cap;            // items per private segment
    uint64_t *seg;               // [ndest][nwg][seg_cap] items of two words
    uint32_t *seg_count;         // [ndest][nwg]
    uint64_t *ovf;               // overflow items (two words each) and their destinations
    uint8_t *ovf_dest;
    uint64_t ovf_cap;
    unsigned long long *ctr;     // kv_shard.hip RouteParams::ctr: [1] overflow items, [18 + d] overflow items of destination d
};
// (hash, count) of every DISTINCT k-mer of `reads` -- deduplicated in super-k-mer buckets -- routed by band.
// `alloc(nwg, sink)` is called once the number of writers is known and must fill *sink.
int kv_skm_route_distinct(const kv_reads *reads, int ksize, uint64_t n_kmers, int ndest, int (*alloc)(void *ctx, uint32_t nwg, KvRouteSink *sink), void *ctx);
// minimizer-sharded exchange (kv_skm.hip): the plan every rank derives from the sample's global size, S1 into the caller's
// exchange buffers, S2 + distinct route over what arrived
int kv_skm_mex_plan(int ksize, uint64_t n_reads_global, uint32_t read_len, int ndest, kv_mex_plan *plan);
int kv_skm_mex_plan_short(kv_mex_plan *plan);
int kv_skm_mex_emit(const kv_reads *reads, const kv_mex_plan *plan, uint64_t read_base, uint64_t *d_seg, uint32_t *d_cnt,
                    uint64_t *d_out, uint64_t out_cap_words, uint64_t *records_per_dest, int *packed);
int kv_skm_mex_route(const kv_mex_plan *plan, int my_dest, const uint64_t *d_recv_seg, const uint32_t *d_recv_cnt, int n_src, int compact, int keep_scan,
                     int (*alloc)(void *ctx, uint32_t nwg, KvRouteSink *sink), int (*after)(void *ctx), void *ctx, uint64_t *n_kmers_in);
struct NovelParams;
int kv_skm_mex_scan_set(const NovelParams &p, int ksize, uint64_t *d_tags, uint8_t *d_abund, uint64_t cap, uint64_t *n_hits);
int kv_skm_mex_pack(const kv_mex_plan *plan, const uint64_t *d_seg, const uint32_t *d_cnt, uint64_t *d_out, uint64_t *records_per_dest);

// tile geometry of the hashing kernels
#define KV_TILE_THREADS 256
#define KV_TILE_MAX_READS 64
#define KV_TILE_LDS_BYTES 16384  // ASCII staging (forward + reverse complement) per tile: 64 reads of 100 bp
#define KV_READ_PAD 24           // over-read slack after each staged strand
#define KV_MAX_READ_LEN 0x7fffffff   // sequences longer than a tile are cut into segment tiles (reference genomes for masks)
#define KV_SEG_BASES 7680        // k-mer starts per segment tile: 2 x (7680 + KV_MAX_K - 1 + pad) bytes of ASCII fit the tile budget

// error plumbing -------------------------------------------------------------------------
void kv_set_error(const char *fmt, ...);
// the hipError_t behind the calling thread's last KV_ERR_HIP (KV_HIP notes it): callers that can do without the GPU for a step
// tell "out of memory" from a device fault by it
extern thread_local int kv_last_hip_code;
// The device of this process (kv_set_device: one process per GPU, LOCAL_RANK) made the calling host thread's current device.  HIP's
// current device is per thread and starts at 0: a worker thread of rank 3 -- a sample counted beside the others, a reader of
// `kevlar count --threads` -- would otherwise allocate and launch on GPU 0.  Cheap (an atomic load and a thread-local compare); called
// by kv_stream(), by KV_HIP in front of every wrapped call, and by the allocation helpers.
void kv_thread_device();

#define KV_HIP(call)                                                                        \
    do {                                                                                    \
        kv_thread_device();                                                                 \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            kv_last_hip_code = (int)e__;                                                    \
            kv_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,  \
                         __LINE__);                                                         \
            return KV_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)
#define KV_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                    \
        if (!(cond)) { kv_set_error(__VA_ARGS__); return (code); }                          \
    } while (0)

hipStream_t kv_stream();
// the key the per-stream scratch tables go by (kv_host.hip): a recycled slot for streams made by kv_stream_create, else the handle
hipStream_t kv_stream_key(hipStream_t st);
// table buffers of destroyed sketches kept for the next sketch (kv_host.hip) go back to the driver: called when an allocation fails
void kv_table_cache_release();
void kv_case_bits_launch(const uint8_t *d_table, uint64_t size, int case_min, uint32_t *d_bits, hipStream_t st);      // kv_skm.hip
void kv_skm_scratch_release();       // kv_skm.hip: every stream's bucket arena, distinct list and bit map
void kv_route_scratch_release();     // kv_shard.hip: every stream's pair sink
void kv_novel_scratch_release();     // kv_novel.hip: every stream's bit map of the pairs scan
void kv_bin_scratch_release();       // kv_binned.hip: every stream's staging of the partitioned add
void kv_unique_scratch_release();    // kv_graph.hip: every stream's first-toucher arrays of kv_unique_new that no running call holds
void kv_ensure_dynamic_lds(const void *kernel, size_t bytes);   // hipFuncSetAttribute once per growth

// profiling: RAII wrapper recording HIP events around a launch when enabled
struct KvProfScope {
    const char *name;
    hipEvent_t a, b;
    bool on;
    explicit KvProfScope(const char *n);
    ~KvProfScope();
};

// hipMalloc for the library: when the device is out of memory the table buffers kept for future sketches (up to KV_TABLE_CACHE_GB of
// them, kv_host.hip) are given back and the call is made once more -- every allocation of the library goes through here, so the
// cache never stands between a caller and memory that is in fact free
hipError_t kv_hip_malloc(void **p, size_t bytes);
template <typename T> static inline hipError_t kv_hip_malloc(T **p, size_t bytes) { return kv_hip_malloc((void **)p, bytes); }

// host helpers shared between files
uint64_t kv_host_murmur_lo(const void *data, int len, uint32_t seed);
uint64_t kv_host_hash(int hashfam, const char *kmer, int k, bool *ok);
int kv_sketch_alloc(int kind, int ksize, int ntables, const uint64_t *sizes, kv_sketch **out);
int kv_sketch_refresh_occupancy(kv_sketch *s);

// ---- blocked gzip (BGZF) on the device (kv_inflate.hip) ----
struct KvBgzfMember {
    uint64_t in_off;     // the member's deflate payload in the file
    uint32_t in_len;
    uint32_t isize;      // bytes it inflates to
    uint32_t crc;        // CRC-32 of them, as the member's trailer has it
    uint32_t pad;
};
struct KvArena;
int kv_bgzf_index(const uint8_t *file, uint64_t size, std::vector<KvBgzfMember> *members, int *is_bgzf);
// readable (and zeroed) bytes the caller keeps behind the compressed image it hands to kv_bgzf_inflate
#define KV_INFLATE_SLACK 2048
int kv_bgzf_inflate(const uint8_t *d_comp, uint64_t comp_base, const KvBgzfMember *members, uint64_t count, const uint64_t *text_off,
                    uint8_t *d_text, KvArena &scratch);

// CRC-32 of text on the device, a thread per range (kv_gunzip.hip); ranges of KV_CRC_SLICE bytes join fastest
#define KV_CRC_SLICE 16384u
int kv_crc32_ranges(const uint8_t *d_text, const uint64_t *start, const uint32_t *len, size_t n, uint32_t *out, KvArena &scratch);
uint32_t kv_crc32_join(uint32_t crc_a, uint32_t crc_b, uint32_t len_b);

// ---- ordinary gzip on the device (kv_gunzip.hip): a segment of the stream per decode/emit pair ----
struct KvGunzip;
struct KvGunzipArenas;                                          // its device buffers (kv_binned.h has the definition)
// NULL unless the image starts with a gzip member header; arenas: buffers to work in (kept by the caller across files), or NULL
KvGunzip *kv_gunzip_open(const uint8_t *image, uint64_t size, KvGunzipArenas *arenas);
void kv_gunzip_set_uploader(KvGunzip *g, std::function<bool(uint8_t *, uint64_t, uint64_t, hipStream_t)> upload);
void kv_gunzip_close(KvGunzip *g);
bool kv_gunzip_done(const KvGunzip *g);
// decode about want_text bytes of text; *text_bytes = how many kv_gunzip_emit will store, *last = the stream ends with them.
// KV_ERR_TYPE: not for this decoder
int kv_gunzip_decode(KvGunzip *g, uint64_t want_text, uint64_t *text_bytes, bool *last);
int kv_gunzip_emit(KvGunzip *g, uint8_t *d_text);
void kv_gunzip_stats(const KvGunzip *g, uint64_t out[4]);       // segments, stretches decoded, dropped, decoded again

// ---- FASTQ split and packed on the device (kv_fastq.hip) ----
struct KvFastqDevice;
KvFastqDevice *kv_fastq_device_open(const char *path);         // NULL unless the file is BGZF, gzip or starts with '@'
void kv_fastq_device_close(KvFastqDevice *d);
// up to max_reads records as a batch in HBM (*n_out = 0 at the end of the file); KV_ERR_TYPE = not four-line FASTQ
int kv_fastq_device_next(KvFastqDevice *d, uint64_t max_reads, kv_reads **reads_out, uint64_t *n_out);
// raw text (four lines each) of records idx[0 .. n) of the batch last returned: record i at blob[offs[i] .. offs[i + 1])
int kv_fastq_device_fetch(KvFastqDevice *d, const uint64_t *idx, uint64_t n, std::string *blob, std::vector<uint64_t> *offs);

// growing text buffer of the native writers (kv_format_augmented, kv_format_records): malloc'ed, handed to the caller as it is
struct KvTextOut {
    char *buf = nullptr;
    size_t len = 0, cap = 0;
    bool ok = true;
    ~KvTextOut() { free(buf); }
    inline bool room(size_t n)
    {
        if (len + n <= cap) return true;
        size_t want = cap ? cap : (1u << 16);
        while (want < len + n) want += want / 2 + 4096;
        char *grown = (char *)realloc(buf, want);
        if (!grown) { ok = false; return false; }
        buf = grown; cap = want;
        return true;
    }
    inline void put(char c) { if (room(1)) buf[len++] = c; }
    inline void put(const char *p, size_t n) { if (room(n)) { memcpy(buf + len, p, n); len += n; } }
    inline void fill(char c, size_t n) { if (room(n)) { memset(buf + len, c, n); len += n; } }
    inline void number(uint32_t v)
    {
        char tmp[12];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
        if (room((size_t)n)) while (n) buf[len++] = tmp[--n];
    }
    // one annotation line: `off` blanks, the k-mer, ten blanks, the abundances, '#'
    template <typename Abund>
    inline void kmer_line(const char *seq, uint32_t off, int ksize, int nsamples, Abund abund_of)
    {
        if (!room((size_t)off + (size_t)ksize + 12 + 12u * (size_t)nsamples)) return;
        memset(buf + len, ' ', off); len += off;
        memcpy(buf + len, seq + off, (size_t)ksize); len += (size_t)ksize;
        memset(buf + len, ' ', 10); len += 10;
        for (int c = 0; c < nsamples; ++c) {
            if (c) buf[len++] = ' ';
            const int64_t v = abund_of(c);
            if (v < 0) { buf[len++] = '-'; number((uint32_t)(-v)); } else number((uint32_t)v);
        }
        buf[len++] = '#'; buf[len++] = '\n';
    }
    char *release() { if (room(1)) buf[len] = 0; char *p = buf; buf = nullptr; cap = 0; return p; }
};
