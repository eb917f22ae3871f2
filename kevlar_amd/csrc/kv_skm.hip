// kv_skm.hip -- super-k-mer front end of count and novel: every DISTINCT k-mer of a batch is hashed, filtered,
// counted and evaluated once, however many reads contain it (kv_skm_device.h says why that is bit-identical).
//
//   S1  k_skm_emit    a workgroup per tile: the minimizer value of every k-mer (smallest mixed canonical m-mer
//                     among its w = k - m + 1, by log-step window minima in LDS), the reads cut into runs of
//                     consecutive k-mers of one minimizer bucket, each run stored as a record of 2-bit bases in
//                     the coarse bucket's stream (private segment per workgroup, cursor in LDS, direct stores).
//   S2  k_skm_split   every coarse stream is split into its F2 fine buckets (again private segments, direct stores);
//                     a fine bucket holds ~8 k k-mer instances, all occurrences of ~2 k distinct k-mers.
//   S3  k_skm_count   persistent workgroups take fine buckets: the canonical 2-bit k-mers are combined in an LDS hash
//                     table (key -> count), then each distinct k-mer is expanded to ASCII (both strands), hashed
//                     with the two murmurs, band/mask filtered, reduced modulo the T table sizes and appended as T
//                     WEIGHTED items to the coarse buckets of the partitioned count (kv_binned.hip stages B and C).
//   S6  k_skm_novel   same buckets of the case sample: combine, evaluate kmer_is_interesting() once per distinct
//                     k-mer, then walk the records again and set the hit-mask bit of every occurrence of an
//                     interesting k-mer; k_tile_hits + the ordinary emit kernels do the rest.
//
// Nothing here can lose a k-mer: a full segment, a full LDS table or an uncacheable key diverts single records to a
// "loose" list that is processed one occurrence at a time (k_skm_loose_*); if that list overflows too, a flag makes
#include <algorithm>
#include <cmath>

#include "kv_binned.h"
#include "kv_novel_device.h"
#include "kv_skm_device.h"

struct SkmGeom {
    int k, m, w, wpow;               // wpow: largest power of two <= w
    int kw, nbw, recw, ncap;         // key words, base words and total words of a record, most k-mers of a record
    uint32_t C1, F2, fbits;          // coarse buckets, fine buckets per coarse bucket (2^fbits)
    uint32_t nwg1, nwg2, quota1;     // writers of S1 (persistent workgroups) and S2 (workgroups per coarse bucket)
    uint32_t cap1, cap2;             // records per private segment
    uint32_t np_max;                 // most bases any tile stages
    uint64_t stride;                 // record position = read * stride + offset
    uint64_t *seg1; uint32_t *cnt1;  // [C1][nwg1][cap1] records / [C1][nwg1]
    uint64_t *seg2; uint32_t *cnt2;  // [C1 * F2][nwg2][cap2] / [C1 * F2][nwg2]
    uint64_t *loose; uint64_t loose_cap;
    unsigned long long *ctr;         // [0] loose records, [1] failure, [2] S1 tile ticket, [3] count ticket, [4] scan ticket,
                                     // [5] records emitted, [6] loose records after S1 + S2, [7] distinct k-mers the count pass found in its LDS tables,
                                     // [9] workgroups whose stretch of the distinct list ran out
    uint32_t n_buckets, quota3;
    uint32_t sbw;                    // words of record-start bits per wave in the bucket walk
    uint64_t bucket_kmers;           // average k-mers per fine bucket
    uint32_t dbg;                    // KV_SKM_DEBUG: timing experiments that skip parts of kernels (results are then wrong)
    // abundance list the count pass writes (KvAbundList; abl_keys == nullptr: off): every workgroup appends to its own
    // stretch of abl_cap_wg entries and notes where each bucket's entries start
    uint64_t *abl_keys; uint8_t *abl_cnts; uint32_t *abl_bstart, *abl_bcount; uint32_t abl_cap_wg;
    // S2 over records that several ranks emitted (minimizer-sharded exchange, kv_skm_mex_route): seg1 / cnt1 then hold n_src
    // slabs of [C1][nwg1] segments one after the other, and a coarse bucket has n_src * nwg1 segments (0 / 1: the usual one slab)
    // distinct list the count pass writes for a batch that is going to be scanned (SkmIndex::dl; dl_keys == nullptr: off): EVERY
    // distinct k-mer of a bucket with its hash, appended like the abundance list (a stretch per workgroup, start + count per bucket)
    uint64_t *dl_keys, *dl_hash; uint32_t *dl_bstart, *dl_bcount; uint32_t dl_cap_wg;
    uint32_t dl_chunk, dl_nchunks;   // k_skm_route only: != 0 -- the list is a pool of dl_nchunks chunks of dl_chunk entries that the workgroups draw from (ctr[13]) instead of one stretch each
    uint32_t n_src;
    const uint64_t *seg1_off;        // non-null: segment `slot` starts at record seg1_off[slot] (compacted records) instead of slot * cap1
    uint64_t read_base;              // global index of the batch's first read (record positions of a read shard; 0 otherwise)
    // seg1 / cnt1 laid out [nwg1][C1] instead of [C1][nwg1]: a writer's ~250 open segments then lie within a few megabytes -- two or
    // three pages of the address translation -- instead of one every nwg1 x cap1 records (11 MB: a page each).  The exchange
    // layouts keep bucket-major order (a destination's coarse buckets must be contiguous: kv_skm_mex_pack).
    uint32_t seg1_wmajor;
    // compact: the records of this batch are 16-byte records without positions (kv_skm_device.h; recw = 2) -- a count into a sketch
    // that is not going to be scanned from this batch.  Loose records keep the classic form: lrecw = 1 + nbw words each.
    uint32_t compact;
    int lrecw;
    // oriented: a record whose minimizer stands reversed in the read (strand bit of its value, skm_order_s) holds the REVERSE COMPLEMENT
    // of the read's bases (header flag), so every k-mer of every record is stored on the strand on which its minimizer is canonical --
    // and that strand is the k-mer's key in the bucket tables: the walk takes k-mers as they stand, no second strand, no comparison.
    // The exchange layouts keep classic records (canonical = the smaller strand, computed by the walk).
    uint32_t oriented;
    uint32_t dd_maxn;                // k_skm_count combines identical records first (records of up to dd_maxn k-mers; 0: it does not)
    uint32_t passes;                 // k_skm_route takes a bucket's k-mers in at least this many passes (0 / 1: one) -- buckets bigger than its LDS table
    uint32_t bpt;                    // buckets per ticket of the bucket kernels' work counter (a power of two)
};

// segment `seg` of coarse bucket c in seg1 / cnt1, counted in segments
// Phase switches of the dissection scripts (scratch/skm_phases.py, scan_phases.py) and KV_SKM_FORCE_LOOSE of the tests.  The three
// long kernels exist twice: the instance that looks at the switches is launched only when one is set (the tests in the occurrence loop
// of k_skm_count cost 0.07 ms per sample, 0.4 ms per step of config 2).
#define SKM_DBG(sg) (KNOBS ? (sg).dbg : 0u)
__device__ __forceinline__ uint64_t skm_seg1_slot(const SkmGeom &sg, uint32_t c, uint32_t seg)
{
    if (sg.seg1_wmajor) return (uint64_t)seg * sg.C1 + c;
    if (sg.n_src <= 1u) return (uint64_t)c * sg.nwg1 + seg;
    const uint32_t src = seg / sg.nwg1, w = seg - src * sg.nwg1;
    return ((uint64_t)src * sg.C1 + c) * sg.nwg1 + w;
}
__device__ __forceinline__ uint32_t skm_seg1_count(const SkmGeom &sg) { return sg.nwg1 * (sg.n_src > 1u ? sg.n_src : 1u); }
// first record of segment `slot`
__device__ __forceinline__ uint64_t skm_seg1_first(const SkmGeom &sg, uint64_t slot) { return sg.seg1_off ? sg.seg1_off[slot] : slot * sg.cap1; }

// the controls' abundance lists a scan may use (same bucket geometry as the case sample's buckets)
#define SKM_MAX_ABL 8
struct SkmAblSet {
    int n, ctrl_max;
    const uint64_t *keys[SKM_MAX_ABL];
    const uint8_t *cnts[SKM_MAX_ABL];
    const uint32_t *bstart[SKM_MAX_ABL], *bcount[SKM_MAX_ABL];
    uint32_t maxv[SKM_MAX_ABL];      // what a counter of that control can hold: an entry rejects if min(count, maxv) > ctrl_max
};

namespace {

#define SKM_THREADS1 512
#if !defined(SKM_S1_WAVE_THREADS_DEFAULT)
#define SKM_S1_WAVE_THREADS_DEFAULT 512u      // threads per workgroup of k_skm_emit_wave (1024: one workgroup per CU, see the kernel)
#endif
#define SKM_THREADS3 512
#define SKM_MAXPROBE 48
// One global counter hands out work; a returning atomic on one word saturates at ~90 per microsecond on this chip
// (MI355X_MICROARCH.md, "dequeue"), i.e. 1.3 ms for the 117 k tiles of a 7.5 M-read sample if every tile were a
// ticket.  A ticket therefore covers several units of work.
#define SKM_TILES_PER_TICKET 8u
#define SKM_BUCKETS_PER_TICKET 8u
// workgroups of the loose-list kernels (grid-stride over a list that is all but empty when the batch fits: 4096 workgroups cost
// 0.14 ms each launch just to start, find nothing and leave)
#define SKM_LOOSE_WGS 1024

// (32-byte records -- one aligned sector each, two 16-byte stores -- were measured against these 24-byte ones: WRITE_SIZE
// fell from 2.5 to 2.3 GB per sample for 1.3 / 1.7 GB stored, the readers fetched 0.4 GB more each, times did not move.)
// a record in as few store instructions as its size allows (every lane of a scattered store is its own cache line: the memory pipe
// takes an instruction's 64 lines one by one, so three 8-byte stores cost three passes where a 16-byte and an 8-byte one cost two)
__device__ __forceinline__ void skm_store_record_wide(uint64_t *dst, uint64_t hdr, const uint64_t *bw, int nbw)
{
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2), aligned(8)));
    *(u64x2 *)dst = u64x2{hdr, bw[0]};
    if (nbw == 2) dst[2] = bw[1];
    else if (nbw == 3) *(u64x2 *)(dst + 2) = u64x2{bw[1], bw[2]};
}

__device__ __forceinline__ void skm_store_record(uint64_t *dst, uint64_t hdr, const uint64_t *bw, int nbw)
{
    dst[0] = hdr;
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < nbw) dst[1 + t] = bw[t];
}

__device__ __forceinline__ void skm_loose_push(const SkmGeom &sg, uint64_t hdr, const uint64_t *bw)
{
    const unsigned long long idx = atomicAdd(&sg.ctr[0], 1ull);
    if (idx < sg.loose_cap) skm_store_record(sg.loose + idx * (uint64_t)sg.lrecw, hdr, bw, sg.nbw);
    else sg.ctr[1] = 1;
}

// ---- S1 ------------------------------------------------------------------------------------------------
struct SkmTile {
    uint32_t len[KV_TILE_MAX_READS], nk[KV_TILE_MAX_READS];
    uint32_t bpre[KV_TILE_MAX_READS + 1];      // staged-base prefix: flat position of a read's first base
    uint32_t wpre[KV_TILE_MAX_READS + 1];      // packed-word prefix inside the tile
    uint32_t cpre[KV_TILE_MAX_READS + 1];      // chunk prefix (a chunk = CH consecutive k-mer starts of one read)
    uint32_t uni_wpr, uni_cpr, uni_len;        // words / chunks / bases per read if all reads of the tile have the same length, else 0
    float inv_wpr, inv_cpr, inv_len;
    uint32_t seg_start, read0, next_tile;
    uint32_t wtot[3][SKM_THREADS1 / 64];       // run starts per (round, wave)
};

// q / d for q < 2^16 with the precomputed float reciprocal (off by at most one before the fix-up)
__device__ __forceinline__ uint32_t skm_div(uint32_t q, uint32_t d, float inv)
{
    uint32_t r = (uint32_t)((float)q * inv);
    if (r * d > q) r -= 1;
    else if ((r + 1) * d <= q) r += 1;
    return r;
}

__device__ __forceinline__ uint32_t skm_search(const uint32_t *pre, uint32_t n, uint32_t q)
{
    uint32_t lo = 0, hi = n;      // largest r with pre[r] <= q (empty entries share their successor's prefix)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] <= q) lo = mid; else hi = mid;
    }
    return lo;
}

// CH consecutive k-mer starts per thread in the window-minimum pass; needs w > CH
template <int CH>
__global__ __launch_bounds__(SKM_THREADS1, 6) void k_skm_emit(ReadsDev rd, uint32_t n_tiles, SkmGeom sg)   // 6 waves per SIMD: three workgroups per CU
{
    constexpr bool KNOBS = true;
    __shared__ SkmTile sh;
    __shared__ uint32_t cur[256];
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t nwl = sg.np_max / 16u + KV_TILE_MAX_READS + 8u;     // every read starts on a word: up to one partial word each
    uint32_t *wl = smem;                                              // the tile's packed words
    uint32_t *mh = smem + nwl;                                        // order value of the m-mer starting at every base
    uint16_t *starts = (uint16_t *)(mh + sg.np_max + 96u);            // run starts (flat positions < 2^16), in position order
    const int k = sg.k, m = sg.m, w = sg.w;
    const int lane = threadIdx.x & 63;
    for (uint32_t c = threadIdx.x; c < sg.C1; c += SKM_THREADS1) cur[c] = 0;
    if (threadIdx.x == 0) sh.next_tile = (uint32_t)atomicAdd(&sg.ctr[2], 1ull) * SKM_TILES_PER_TICKET;
    uint64_t n_rec = 0;
    // Batches of equal-length reads (rd.uni_len): the layout of a tile is arithmetic, so the packed words of the NEXT
    // tile are requested while this one is processed (two registers per thread) and the per-tile chain of dependent loads
    // -- tile descriptor, word offsets, lengths, words: half of this kernel's time went there -- disappears.
    const bool uni = rd.uni_len != 0;
    const uint32_t uni_wpr = (rd.uni_len + 15u) >> 4;
    uint32_t pf0 = 0, pf1 = 0, pf_tile = 0xffffffffu;
    auto prefetch = [&](uint32_t tile) {
        pf_tile = tile;
        if (tile >= n_tiles) return;
        const uint64_t r0 = (uint64_t)tile * rd.uni_per_tile;
        const uint32_t nr = (uint32_t)min((uint64_t)rd.uni_per_tile, rd.n_reads - r0), nw = nr * uni_wpr;
        const uint32_t *src = rd.words + r0 * uni_wpr;
        pf0 = threadIdx.x < nw ? src[threadIdx.x] : 0u;
        pf1 = threadIdx.x + SKM_THREADS1 < nw ? src[threadIdx.x + SKM_THREADS1] : 0u;
    };
    if (uni) {
        __syncthreads();
        prefetch(sh.next_tile);
    }
    for (uint32_t taken = 0; taken < sg.quota1; ++taken) {
        __syncthreads();                                   // previous tile finished, ticket visible
        const uint32_t tile = sh.next_tile;
        if (tile >= n_tiles) break;
        TileDesc td;
        if (uni) { td.first = tile * rd.uni_per_tile; td.count = (uint32_t)min((uint64_t)rd.uni_per_tile, rd.n_reads - td.first); td.seg_start = 0; td.seg = 0; }
        else td = rd.tile[tile];
        const uint32_t r0 = td.first, nr = td.count;
        const uint32_t seg_start = td.seg ? td.seg_start : 0u;
        const uint64_t w0 = uni ? (uint64_t)r0 * uni_wpr : rd.woff[r0] + (seg_start >> 4);
        uint32_t nwords;
        if (uni) {
            nwords = nr * uni_wpr;
        } else if (td.seg) {
            const uint32_t rest = rd.len[r0] - seg_start, want = (uint32_t)KV_SEG_BASES + (uint32_t)k - 1u;
            nwords = ((rest < want ? rest : want) + 15u) >> 4;
        } else {
            nwords = (uint32_t)(rd.woff[r0 + nr] - w0);
        }
        if (uni) {
            if (threadIdx.x < 64) {
                const uint32_t i0 = 2 * threadIdx.x, i1 = i0 + 1, L = rd.uni_len;
                const uint32_t kk = L >= (uint32_t)k ? L - (uint32_t)k + 1u : 0u, cc = (kk + CH - 1) / CH;
                if (i0 < nr) { sh.len[i0] = L; sh.nk[i0] = kk; sh.bpre[i0] = i0 * L; sh.cpre[i0] = i0 * cc; sh.wpre[i0] = i0 * uni_wpr; }
                if (i1 < nr) { sh.len[i1] = L; sh.nk[i1] = kk; sh.bpre[i1] = i1 * L; sh.cpre[i1] = i1 * cc; sh.wpre[i1] = i1 * uni_wpr; }
                if (threadIdx.x == 0) {
                    sh.bpre[nr] = nr * L; sh.cpre[nr] = nr * cc; sh.wpre[nr] = nr * uni_wpr; sh.seg_start = 0; sh.read0 = r0;
                    const bool okk = L >= (uint32_t)k;
                    sh.uni_wpr = okk ? uni_wpr : 0u; sh.uni_cpr = okk ? cc : 0u;
                    sh.inv_wpr = okk ? 1.0f / (float)uni_wpr : 0.0f; sh.inv_cpr = okk ? 1.0f / (float)cc : 0.0f;
                    sh.uni_len = okk ? L : 0u; sh.inv_len = okk ? 1.0f / (float)L : 0.0f;
                }
            }
        } else if (threadIdx.x < 64) {
            const uint32_t i0 = 2 * threadIdx.x, i1 = i0 + 1;
            uint32_t l0 = 0, l1 = 0;
            if (i0 < nr) {
                l0 = rd.len[r0 + i0];
                if (td.seg) {
                    const uint32_t rest = l0 - seg_start, want = (uint32_t)KV_SEG_BASES + (uint32_t)k - 1u;
                    l0 = rest < want ? rest : want;
                }
            }
            if (i1 < nr) l1 = rd.len[r0 + i1];
            const uint32_t k0 = l0 >= (uint32_t)k ? l0 - (uint32_t)k + 1u : 0u, k1 = l1 >= (uint32_t)k ? l1 - (uint32_t)k + 1u : 0u;
            const uint32_t c0 = (k0 + CH - 1) / CH, c1 = (k1 + CH - 1) / CH;
            uint32_t eb, tot, ecb, ctot;
            const uint32_t ea = wave_excl_scan2(l0, l1, eb, tot);
            const uint32_t eca = wave_excl_scan2(c0, c1, ecb, ctot);
            if (i0 < nr) { sh.len[i0] = l0; sh.nk[i0] = k0; sh.bpre[i0] = ea; sh.cpre[i0] = eca; }
            if (i1 < nr) { sh.len[i1] = l1; sh.nk[i1] = k1; sh.bpre[i1] = eb; sh.cpre[i1] = ecb; }
            if (threadIdx.x == 0) { sh.bpre[nr] = tot; sh.cpre[nr] = ctot; sh.seg_start = seg_start; sh.read0 = r0; }
            const uint32_t ref = __shfl(l0, 0);
            const bool same = (i0 >= nr || l0 == ref) && (i1 >= nr || l1 == ref);
            const bool uniform = __all(same) && ref >= (uint32_t)k;
            if (threadIdx.x == 0) {
                const uint32_t wpr = (ref + 15u) >> 4, cpr = (ref - (uint32_t)k + 1u + CH - 1) / CH;
                sh.uni_wpr = uniform ? wpr : 0u; sh.uni_cpr = uniform ? cpr : 0u;
                sh.inv_wpr = uniform ? 1.0f / (float)wpr : 0.0f; sh.inv_cpr = uniform ? 1.0f / (float)cpr : 0.0f;
                sh.uni_len = uniform ? ref : 0u; sh.inv_len = uniform ? 1.0f / (float)ref : 0.0f;
            }
            if (td.seg) {
                if (threadIdx.x == 0) { sh.wpre[0] = 0; sh.wpre[1] = (l0 + 15) >> 4; }
            } else {
                if (i0 < nr) sh.wpre[i0] = (uint32_t)(rd.woff[r0 + i0] - w0);
                if (i1 < nr) sh.wpre[i1] = (uint32_t)(rd.woff[r0 + i1] - w0);
                if (threadIdx.x == 0) sh.wpre[nr] = (uint32_t)(rd.woff[r0 + nr] - w0);
            }
        }
        if (uni && pf_tile == tile && nwords + 8u <= 2u * SKM_THREADS1) {
            wl[threadIdx.x] = pf0;                                             // (zero beyond the tile's words)
            if (threadIdx.x + SKM_THREADS1 < nwords + 8u) wl[threadIdx.x + SKM_THREADS1] = pf1;
        } else {
            for (uint32_t i = threadIdx.x; i < nwords + 8u; i += SKM_THREADS1) wl[i] = i < nwords ? rd.words[w0 + i] : 0u;
        }
        __syncthreads();
        // the next ticket is fetched while this tile is processed (everybody has read the current one by now)
        if (threadIdx.x == 0) {
            if ((tile + 1u) % SKM_TILES_PER_TICKET) sh.next_tile = tile + 1u;       // same ticket
            else sh.next_tile = taken + 1 < sg.quota1 ? (uint32_t)atomicAdd(&sg.ctr[2], 1ull) * SKM_TILES_PER_TICKET : 0xffffffffu;
        }
        const uint32_t NB = sh.bpre[nr];
        // P1: one thread per packed word: the order values of the m-mers starting at its 16 bases
        const uint32_t mmask = m == 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
        for (uint32_t wi = threadIdx.x; wi < ((SKM_DBG(sg) & 512u) ? 0u : nwords); wi += SKM_THREADS1) {
            const uint32_t r = sh.uni_wpr ? skm_div(wi, sh.uni_wpr, sh.inv_wpr) : skm_search(sh.wpre, nr, wi);
            const uint32_t j0 = (wi - sh.wpre[r]) * 16u, L = sh.len[r];
            const uint32_t q0 = sh.bpre[r] + j0;
            const uint64_t win = (uint64_t)wl[wi] | ((uint64_t)wl[wi + 1] << 32);
            // the reverse complement of all 32 bases once: base i of `win`, complemented, is base 31 - i of `rcw`, so the reverse
            // complement of the m-mer at base jj is the m-mer of rcw at base 32 - m - jj (m <= 16, jj <= 15)
            const uint64_t rcw = ~skm_rev2_64(win);
            const uint32_t rc0 = 2u * (32u - (uint32_t)m);
            // lane l starts at base l mod 16 of its word: the 64 stores of one instruction then fall into different banks
#pragma unroll
            for (int step = 0; step < 16; ++step) {
                const uint32_t jj = ((uint32_t)step + (uint32_t)lane) & 15u, j = j0 + jj;
                const uint32_t f = (uint32_t)(win >> (2u * jj)) & mmask, r = (uint32_t)(rcw >> (rc0 - 2u * jj)) & mmask;
                if (j < L) mh[q0 + jj] = j + (uint32_t)m <= L ? skm_order_s(f, r) : 0xffffffffu;
            }
        }
        if (threadIdx.x < 96) mh[NB + threadIdx.x] = 0xffffffffu;
        __syncthreads();
        if (uni) prefetch(sh.next_tile);                   // written before this barrier; lands during P2 .. P4
        // P2: one thread per chunk of CH k-mer starts: the minimum over the w m-mers of each (shared suffix of the chunk + the
        // w - 1 - CH values every window contains + a growing prefix) and where it changes.  A run is a stretch of k-mers with the
        // same minimizer VALUE (its bucket is a function of that value, taken once per run in P4; two values that share a bucket
        // -- one pair in C1 x F2 -- merely give two records where one would have done).
        const uint32_t nchunks = (SKM_DBG(sg) & 32u) ? 0u : sh.cpre[nr];             // <= 8192 / CH + 64 < 3 * SKM_THREADS1
        const uint32_t nrounds = (nchunks + SKM_THREADS1 - 1u) / SKM_THREADS1;
        uint32_t my_q[3], my_starts[3];
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            const uint32_t ci = (uint32_t)round * SKM_THREADS1 + threadIdx.x;
            uint32_t startmask = 0, q = 0;
            if (ci < nchunks) {
                const uint32_t r = sh.uni_cpr ? skm_div(ci, sh.uni_cpr, sh.inv_cpr) : skm_search(sh.cpre, nr, ci);
                const uint32_t j = (ci - sh.cpre[r]) * CH, nkr = sh.nk[r];
                q = sh.bpre[r] + j;
                uint32_t suf[CH + 1];                     // suf[i + 1] = min(mh[q + i .. q + CH - 1]), i = -1 .. CH - 1
                uint32_t run = 0xffffffffu;
#pragma unroll
                for (int i = CH - 1; i >= 0; --i) { run = min(run, mh[q + i]); suf[i + 1] = run; }
                suf[0] = j > 0 ? min(run, mh[q - 1]) : run;
                uint32_t mid = 0xffffffffu;
                for (uint32_t t = CH; t + 2 <= (uint32_t)w; ++t) mid = min(mid, mh[q + t]);     // mh[q + CH .. q + w - 2]
                uint32_t prev = min(suf[0], mid);         // minimizer of the k-mer in front of the chunk (unused when j == 0)
                uint32_t diff = j == 0 ? 1u : 0u;
                run = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    run = min(run, mh[q + (uint32_t)w - 1u + i]);
                    const uint32_t v = min(min(suf[i + 1], mid), run);
                    diff |= (v != prev ? 1u : 0u) << i;
                    prev = v;
                }
                const uint32_t nv = min((uint32_t)CH, nkr - j);          // k-mers of the read that start in this chunk (>= 1)
                startmask = diff & ((2u << (nv - 1u)) - 1u);
            }
            my_q[round] = q; my_starts[round] = startmask;
        }
        // run starts in position order (chunks are numbered in position order): prefix over lanes, waves, rounds
        uint32_t incl[3] = {0u, 0u, 0u};
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            if ((uint32_t)round < nrounds) {
                uint32_t v = (uint32_t)__popc(my_starts[round]);
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = __shfl_up(v, d);
                    if (lane >= d) v += up;
                }
                incl[round] = v;
                if (lane == 63) sh.wtot[round][threadIdx.x >> 6] = v;
            }
        }
        __syncthreads();
        uint32_t nstart = 0;
#pragma unroll
        for (int round = 0; round < 3; ++round) {
            if ((uint32_t)round < nrounds) {
                uint32_t before = nstart;
                for (uint32_t wv = 0; wv < SKM_THREADS1 / 64; ++wv) {
                    const uint32_t n = sh.wtot[round][wv];
                    if (wv < (threadIdx.x >> 6)) before += n;
                    nstart += n;
                }
                uint32_t base = before + incl[round] - (uint32_t)__popc(my_starts[round]);
                uint32_t bits = my_starts[round];
                while (bits) {
                    const int i = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    starts[base++] = (uint16_t)(my_q[round] + (uint32_t)i);
                }
            }
        }
        if (SKM_DBG(sg) & 16u) nstart = 0;
        __syncthreads();
        // P4: one thread per run: measure it, cut it into records of <= ncap k-mers, store them
        for (uint32_t i = threadIdx.x; i < nstart; i += SKM_THREADS1) {
            const uint32_t q = starts[i];
            const uint32_t r = sh.uni_len ? skm_div(q, sh.uni_len, sh.inv_len) : skm_search(sh.bpre, nr, q), j = q - sh.bpre[r];
            const uint32_t limit = q - j + sh.nk[r];          // flat position one past the read's last k-mer
            const uint32_t nxt = i + 1 < nstart ? (uint32_t)starts[i + 1] : limit;     // the next run (of this read or a later one)
            uint32_t left = (nxt < limit ? nxt : limit) - q;
            uint32_t minv = 0xffffffffu;
            for (uint32_t t = 0; t < (uint32_t)w; ++t) minv = min(minv, mh[q + t]);
            uint32_t coarse, fine;
            skm_bucket_of(minv, sg.C1, sg.fbits, coarse, fine);
            uint64_t pos = (sg.read_base + sh.read0 + r) * sg.stride + sh.seg_start + j;
            uint32_t b = sh.wpre[r] * 16u + j;                // base index inside wl
            while (left) {
                const uint32_t n = min(left, (uint32_t)sg.ncap);
                uint64_t bw[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? skm_bases32(wl, b + 32u * t) : 0ull;
                const uint32_t rev = sg.oriented ? (minv & 1u) : 0u;
                if (rev) skm_rc_bases(bw, sg.nbw, n + (uint32_t)k - 1u);
                const uint64_t hdr = skm_header(pos, n, fine, rev);
                const uint32_t p = atomicAdd(&cur[coarse], 1u);
                if (SKM_DBG(sg) & 1024u) n_rec += hdr ^ bw[0] ^ bw[1];
                else if (p < sg.cap1) skm_store_record(sg.seg1 + (skm_seg1_slot(sg, coarse, blockIdx.x) * sg.cap1 + p) * (uint64_t)sg.recw, hdr, bw, sg.nbw);
                else skm_loose_push(sg, hdr, bw);
                n_rec += 1;
                left -= n; pos += n; b += n;
            }
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < sg.C1; c += SKM_THREADS1) sg.cnt1[skm_seg1_slot(sg, c, blockIdx.x)] = min(cur[c], sg.cap1);
    n_rec = wave_sum_u64(n_rec);
    if ((threadIdx.x & 63) == 0 && n_rec) atomicAdd(&sg.ctr[5], (unsigned long long)n_rec);
}

// ---- S1, batches of equal-length reads: a wave per group of reads -----------------------------------------------
// The tile kernel above spends most of its time parked: six workgroup barriers per tile, between phases that keep 60-80 % of
// the lanes busy, and a phase is as slow as its slowest wave (dissected: skeleton 0.37 ms, order values 0.37, minima + run
// starts 0.43, records 0.30, their stores 0.43 per 7.5 M reads).  When every read of the batch has the same length the layout
// is arithmetic, so a WAVE can take R consecutive reads (R x words per read <= 64: one packed word per lane) through the same
// four phases on its own slice of LDS with no workgroup barrier at all -- only the order of its own LDS operations -- and the
// 24 waves of a CU are in different phases at any time.  The workgroup still shares the coarse cursors and its segments.
#define SKM_MT_PER_TICKET 64u     // 1 M groups per 7.5 M reads: one device-wide counter takes ~90 updates per microsecond (16 per ticket: 0.74 ms of the kernel)
#define SKM_WAVE_RUNS 128u        // runs of a group listed at a time (a group has ~55; more go round again)
// Flat position of base j of the group's read r: r * Lp + j with Lp = 16 * words per read, so a position is also the base index
// into the staged words.  Order values live at mh[p + (p >> PS)], PS = log2(CH): a padding word per chunk, so that lanes that
// walk consecutive chunks (P2) or consecutive words (P1) hit different banks and every offset inside a chunk is a constant.
__host__ __device__ inline uint32_t skm_wave_slice_words(uint32_t R, uint32_t L, uint32_t nk, int ch)
{
    const uint32_t lp = 16u * ((L + 15u) >> 4), pmax = R * lp + 96u;
    (void)nk;
    return 132u + pmax + pmax / (uint32_t)ch + 2u + SKM_WAVE_RUNS + (SKM_WAVE_RUNS + 2u) / 2u;      // words (twice), order values, minimizer + start of SKM_WAVE_RUNS runs at a time
}

// THREADS: 512 (three workgroups per CU, 768 writers) or 1024 (ONE workgroup of 16 waves per CU, 256 writers).  A writer keeps one
// open cache line per coarse bucket, and the L2 of an XCD (4 MB) merges a record's 16- and 8-byte stores into whole lines only while
// the open lines of the XCD's workgroups fit beside the rest of its traffic: 251 buckets x 96 workgroups x 128 B = 3 MB do not
// (WRITE_SIZE 2.5-2.7 GB per 1.36 GB of records, measured with 512 and with 768 writers), 251 x 32 x 128 B = 1 MB do (1.0 x).
template <int CH, bool KNOBS, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS == 1024 ? 4 : 6) void k_skm_emit_wave(ReadsDev rd, SkmGeom sg, uint32_t R, uint32_t n_mt, uint32_t quota_mt, uint32_t per_ticket)
{
    constexpr uint32_t PS = CH == 16 ? 4u : 3u;
    __shared__ uint32_t cur[256];
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    const int k = sg.k, m = sg.m, w = sg.w;
    const uint32_t L = rd.uni_len, wpr = (L + 15u) >> 4, Lp = 16u * wpr, nk = L - (uint32_t)k + 1u, cpr = (nk + CH - 1) / CH;
    const float inv_wpr = 1.0f / (float)wpr, inv_cpr = 1.0f / (float)cpr;
    const uint32_t pmax = R * Lp + 96u;
    uint32_t *wl0 = smem + wave * skm_wave_slice_words(R, L, nk, CH), *mh = wl0 + 132u;
    uint32_t *sval = mh + pmax + pmax / CH + 2u;
    uint16_t *starts = (uint16_t *)(sval + SKM_WAVE_RUNS);
    for (uint32_t c = threadIdx.x; c < sg.C1; c += THREADS) cur[c] = 0;
    if ((threadIdx.x & 63u) < 2u) { wl0[64u + (threadIdx.x & 63u)] = 0; wl0[130u + (threadIdx.x & 63u)] = 0; }
    __syncthreads();
    const uint32_t mmask = m == 16 ? 0xffffffffu : ((1u << (2 * m)) - 1u);
    const uint32_t rc0 = 2u * (32u - (uint32_t)m);
    uint64_t n_rec = 0;
    uint64_t *const my_seg = sg.seg1 + skm_seg1_slot(sg, 0u, blockIdx.x) * sg.cap1 * (uint64_t)sg.recw;       // + coarse * cstride: this workgroup's segment of a coarse bucket
    const uint64_t cstride = (sg.seg1_wmajor ? 1ull : (uint64_t)sg.nwg1) * sg.cap1 * (uint64_t)sg.recw;
    uint32_t mt = 0, mt_end = 0, taken = 0, pf = 0, pf_mt = 0xffffffffu, parity = 0;
    for (;;) {
        // (everything P1 and P2 derive from the lane number alone is the same for every group; left to itself the compiler
        // computes all of it once in front of the loop -- 16 shift pairs, 32 lane masks -- and spills it: 196 bytes of scratch
        // per lane and two reloads per order value.  An opaque copy of the lane number keeps the arithmetic where it is used.)
        uint32_t lane = threadIdx.x & 63u;
        asm volatile("" : "+v"(lane));
        auto words_of = [&](uint32_t t) -> uint32_t {
            const uint64_t r0 = (uint64_t)t * R;
            const uint32_t nr = (uint32_t)min((uint64_t)R, rd.n_reads - r0);
            return lane < nr * wpr ? rd.words[r0 * wpr + lane] : 0u;
        };
        if (mt == mt_end) {
            if (taken >= quota_mt) break;
            uint32_t t = 0;
            if (lane == 0) t = (uint32_t)atomicAdd(&sg.ctr[2], 1ull);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if ((uint64_t)t * per_ticket >= n_mt) break;
            mt = t * per_ticket;
            mt_end = min(mt + per_ticket, n_mt);
        }
        const uint64_t r0 = (uint64_t)mt * R;
        const uint32_t nr = (uint32_t)min((uint64_t)R, rd.n_reads - r0), nwords = nr * wpr;
        // The next group's words are requested now and WAITED FOR in front of P4: a wait at the top of the next round would also
        // wait for this round's record stores (one counter, in order), i.e. for a round trip to HBM per group.
        uint32_t *wl = wl0 + 66u * parity;
        uint32_t word;
        if (pf_mt == mt) word = pf;                               // (already in the other buffer's place: stored below)
        else { word = words_of(mt); wl[lane] = word; }
        const bool more = mt + 1u < mt_end;
        if (more) { pf = words_of(mt + 1u); pf_mt = mt + 1u; }
        // P1: the order values of the m-mers starting at the 16 bases of this lane's word.  (Values of m-mers that run past the
        // end of their read are garbage: no window of a k-mer of the read contains them.)
        if (lane < nwords && !(SKM_DBG(sg) & 512u)) {
            const uint32_t up = (uint32_t)__shfl_down((int)word, 1);
            const uint64_t win = (uint64_t)word | ((uint64_t)(lane == 63u ? 0u : up) << 32);
            const uint64_t rcw = ~skm_rev2_64(win);
            uint32_t *dst = mh + 16u * lane + (CH == 16 ? lane : 2u * lane);
#pragma unroll
            for (int step = 0; step < 16; ++step) {
                const uint32_t f = (uint32_t)(win >> (2 * step)) & mmask, rv = (uint32_t)(rcw >> (rc0 - 2u * (uint32_t)step)) & mmask;
                dst[step + (step >> PS)] = skm_order_s(f, rv);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // P2: a lane per chunk of CH k-mer starts: window minima (shared suffix of the chunk + the values every window of the chunk
        // contains + a growing prefix) and where they change; then every run start with its minimizer, in position order
        const uint32_t nchunks = (SKM_DBG(sg) & 32u) ? 0u : nr * cpr;
        uint32_t startmask = 0, q = 0;
        uint32_t v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) v[i] = 0;
        if (lane < nchunks) {
            const uint32_t r = skm_div(lane, cpr, inv_cpr);
            const uint32_t j = (lane - r * cpr) * CH;
            q = r * Lp + j;                                   // a multiple of CH
            const uint32_t *src = mh + q + (q >> PS);
            auto at = [&](uint32_t i) -> uint32_t { return src[i + (i >> PS)]; };       // i: constant offsets from the chunk's first value
            uint32_t suf[CH + 1];
            uint32_t run = 0xffffffffu;
#pragma unroll
            for (int i = CH - 1; i >= 0; --i) { run = min(run, at((uint32_t)i)); suf[i + 1] = run; }
            suf[0] = j > 0 ? min(run, mh[q - 1u + ((q - 1u) >> PS)]) : run;
            uint32_t mid = 0xffffffffu;
            for (uint32_t t = CH; t + 2 <= (uint32_t)w; ++t) mid = min(mid, at(t));
            uint32_t prev = min(suf[0], mid);
            uint32_t diff = j == 0 ? 1u : 0u;
            run = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                run = min(run, at((uint32_t)w - 1u + (uint32_t)i));
                v[i] = min(min(suf[i + 1], mid), run);
                diff |= (v[i] != prev ? 1u : 0u) << i;
                prev = v[i];
            }
            const uint32_t nv = min((uint32_t)CH, nk - j);
            startmask = diff & ((2u << (nv - 1u)) - 1u);
        }
        uint32_t incl = (uint32_t)__popc(startmask);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t upv = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += upv;
        }
        uint32_t nstart = (uint32_t)__shfl((int)incl, 63);
        if (SKM_DBG(sg) & 16u) nstart = 0;
        if (more) wl0[66u * (parity ^ 1u) + lane] = pf;
        for (uint32_t lo = 0; lo < nstart; lo += SKM_WAVE_RUNS) {
            // runs lo .. lo + SKM_WAVE_RUNS (one more than are processed: the end of the last one), listed by the lanes that found them
            {
                uint32_t at = incl - (uint32_t)__popc(startmask) - lo;          // wraps for runs in front of this round: fails the test below
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    if ((startmask >> i) & 1u) {
                        if (at <= SKM_WAVE_RUNS) { starts[at] = (uint16_t)(q + (uint32_t)i); if (at < SKM_WAVE_RUNS) sval[at] = v[i]; }
                        at += 1u;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // P4: a lane per run: its bucket, its records
            const uint32_t here = min(SKM_WAVE_RUNS, nstart - lo);
            for (uint32_t i = lane; i < here; i += 64u) {
                const uint32_t qs = starts[i];
                const uint32_t r = skm_div(qs >> 4, wpr, inv_wpr), j = qs - r * Lp;
                const uint32_t limit = qs - j + nk;                // position one past the read's last k-mer
                const uint32_t nxt = lo + i + 1u < nstart ? (uint32_t)starts[i + 1] : limit;
                uint32_t left = (nxt < limit ? nxt : limit) - qs;
                uint32_t coarse, fine;
                skm_bucket_of(sval[i], sg.C1, sg.fbits, coarse, fine);
                uint64_t pos = (sg.read_base + r0 + r) * sg.stride + j;
                uint32_t b = qs;
                while (left) {
                    const uint32_t n = min(left, (uint32_t)sg.ncap);
                    uint64_t bw[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? skm_bases32(wl, b + 32u * t) : 0ull;
                    const uint32_t rev = sg.oriented ? (sval[i] & 1u) : 0u;
                    if (rev) skm_rc_bases(bw, sg.nbw, n + (uint32_t)k - 1u);
                    const uint64_t hdr = skm_header(pos, n, fine, rev);
                    const uint32_t p = atomicAdd(&cur[coarse], 1u);
                    if (SKM_DBG(sg) & 1024u) n_rec += hdr ^ bw[0] ^ bw[1];
                    else if (p < sg.cap1) skm_store_record_wide(my_seg + coarse * cstride + p * (uint32_t)sg.recw, hdr, bw, sg.nbw);
                    else skm_loose_push(sg, hdr, bw);
                    n_rec += 1;
                    left -= n; pos += n; b += n;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        parity ^= 1u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        mt += 1; taken += 1;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < sg.C1; c += THREADS) sg.cnt1[skm_seg1_slot(sg, c, blockIdx.x)] = min(cur[c], sg.cap1);
    n_rec = wave_sum_u64(n_rec);
    if ((threadIdx.x & 63u) == 0 && n_rec) atomicAdd(&sg.ctr[5], (unsigned long long)n_rec);
}

// ---- S1, batches of equal-length reads: a LANE per read (round 5) ---------------------------------------------------
// The wave kernel above deals a group of 7 reads over the lanes three different ways (a lane per packed word, per chunk of k-mer
// starts, per run) with the order values, the run starts and their minimizers passing through LDS between the phases: 57
// lane-instructions per base, a third of the lanes idle in every phase (100 bases are 6.25 words; 7 reads are 49 words and 63 chunks).
// Here a lane takes ONE READ and walks its m-mers in order, all 64 lanes at the same position -- so everything that depends on the
// position alone (which words, which shifts, whether a k-mer ends here) is scalar -- and the window minimum is the streaming form of
// the block decomposition: m-mers in blocks of B = 20; the k-mer whose window ENDS at m-mer p takes min(suffix minimum of the block
// it starts in, [the whole block between, for w = 40,] prefix minimum of the block p is in); a block's values overwrite the suffix
// minima they have just been compared with, and one backward pass turns them into the next block's suffix minima.  Per m-mer:
// two extractions (forward and reverse complement from the block's 64-bit window, constant shifts), a minimum, the order multiply;
// per k-mer: three minima and a compare.  A lane whose minimizer changes appends the run it has just finished -- (minimizer, read,
// first k-mer, length) -- to the wave's list in LDS (ballot + mbcnt; one in thirteen lanes per position), and every few blocks the
// list is turned into records a lane per RUN, all lanes busy: the record code of the kernels above, unchanged, so the records are
// the same records.  No order value, window minimum or run start ever goes through LDS; the words do, once.
// Work is dealt statically (group g to wave g mod waves): a wave has ~19 groups of 64 reads per 7.5 M-read sample, tickets of any
// useful size would leave a tail of a ticket, and the device-wide counter is not in the picture.
// Instances: w = 20 (k = 31: C = 1) and w = 40 (k = 51: C = 2), m = 12, reads of up to 16 x NW bases; everything else takes the kernels above.
#define SKM_LANE_B 20
#if !defined(SKM_LANE_FULL)
#define SKM_LANE_FULL 0           // 1: blocks in the middle of a read run without the per-position scalar tests (measured: slower)
#endif
#if !defined(SKM_LANE_BRANCHFREE)
#define SKM_LANE_BRANCHFREE 0     // 1: every lane writes an entry at every k-mer, lanes without a finished run to a dummy slot (no exec regions)
#endif
#if !defined(SKM_LANE_CAP)
#define SKM_LANE_CAP 320u         // finished runs a wave lists between two flushes (64 reads x 40 k-mers bring ~250; what does not fit is written at once)
#endif
#define SKM_LANE_THREADS 512
#if !defined(SKM_LANE_WAVES)
#define SKM_LANE_WAVES 6          // waves per SIMD the kernel is compiled for (3 workgroups per CU)
#endif
// a wave's slice of LDS: the group's words, their reverse complement read by read (oriented records), the run list
__host__ __device__ inline uint32_t skm_lane_slice_words(uint32_t wpr) { return 2u * (64u * wpr + 4u) + 2u * SKM_LANE_CAP + 2u; }

// one block of B m-mers for every lane's read: Sold holds the suffix minima of the block the finishing k-mers START in (C blocks
// back) and receives this block's values; Smid (C = 2) the suffix minima of the block between, Smid[0] its minimum
// FULL: every m-mer of the block exists and a k-mer > 0 of the read ends at each of them (the blocks in the middle of a read: no
// per-position tests, which are scalar but not free -- a compare and a branch each, 13 scalar instructions per m-mer with them)
template <int C, bool FULL, typename Append>
__device__ __forceinline__ void skm_lane_block(const uint32_t *wl, uint32_t rbase, uint32_t lane, uint32_t b, uint32_t npos, uint32_t nk,
                                               uint32_t (&Sold)[SKM_LANE_B], const uint32_t (&Smid)[SKM_LANE_B], uint32_t &prev_v, uint32_t &start_prev,
                                               uint32_t &n_ent, Append append)
{
    constexpr int B = SKM_LANE_B;
    const uint32_t bit0 = 2u * B * b, w0 = bit0 >> 5, s0 = bit0 & 31u;
    const uint32_t x0 = wl[rbase + w0], x1 = wl[rbase + w0 + 1u], x2 = wl[rbase + w0 + 2u];
    const uint32_t xlo = __builtin_amdgcn_alignbit(x1, x0, s0), xhi = __builtin_amdgcn_alignbit(x2, x1, s0);
    const uint64_t X = (uint64_t)xlo | ((uint64_t)xhi << 32);          // bases 20 b .. 20 b + 31 of the read
    const uint64_t R = ~skm_rev2_64(X);                                // their reverse complement: the m-mer at base j in bits 40 - 2 j .. 63 - 2 j
    uint32_t pre = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const uint32_t p = (uint32_t)B * b + (uint32_t)j;
        if (!FULL && p >= npos) break;                                 // (the same for every lane: a scalar branch)
        const uint32_t f = (uint32_t)(X >> (2 * j)) & 0xffffffu, rv = (uint32_t)(R >> (40 - 2 * j)) & 0xffffffu;
        const uint32_t val = skm_order_s(f, rv);
        pre = min(pre, val);
        if (FULL || p + 1u >= (uint32_t)(C * B)) {
            const uint32_t i = p + 1u - (uint32_t)(C * B);             // the k-mer whose window ends at this m-mer
            if (FULL || i < nk) {
                uint32_t minv;
                if (C == 1) minv = j == B - 1 ? pre : min(Sold[j == B - 1 ? 0 : j + 1], pre);
                else minv = j == B - 1 ? min(Smid[0], pre) : min(min(Sold[j == B - 1 ? 0 : j + 1], Smid[0]), pre);
                if (!FULL && i == 0u) {
                    prev_v = minv; start_prev = lane;                 // lane | first k-mer << 6: the upper word of the run's entry, but for its end
                } else {
                    // (a lane that finds the list full keeps its run open and asks again at the next k-mer, after the flush that full
                    // list brings about: the run is then cut a few k-mers late -- k-mers in a bucket other than their minimizer's, which
                    // costs deduplication, never correctness: adds of a k-mer's occurrences compose whichever buckets they met in)
                    // (lanes without a read walk words of zeros: one minimizer from end to end, no run ever finishes)
                    const bool want = minv != prev_v;
                    const unsigned long long bal = __ballot(want);
#if SKM_LANE_BRANCHFREE
                    const bool done = append(bal, want, prev_v, start_prev, i);
                    prev_v = done ? minv : prev_v;
                    start_prev = done ? (lane | (i << 6)) : start_prev;
#else
                    if (want && append(bal, true, prev_v, start_prev, i)) { prev_v = minv; start_prev = lane | (i << 6); }
#endif
                    n_ent += (uint32_t)__popcll(bal);
                }
            }
        }
        Sold[j] = val;
    }
#pragma unroll
    for (int j = B - 2; j >= 0; --j) Sold[j] = min(Sold[j], Sold[j + 1]);
}

// (C = 2 keeps two arrays of suffix minima: 4 waves per SIMD -- two workgroups per CU, 128 registers -- where C = 1 runs 6; the kernel
// is bound by instruction issue from 4 waves per SIMD up, so the lower occupancy costs nothing: measured with KV_SKM_NWG1=512 at C = 1)
template <int C, int NW>
__global__ __launch_bounds__(SKM_LANE_THREADS, C == 2 ? 4 : SKM_LANE_WAVES) void k_skm_emit_lane(ReadsDev rd, SkmGeom sg, uint32_t n_groups, uint32_t flush_blocks)
{
    constexpr int B = SKM_LANE_B;
    __shared__ uint32_t cur[256];
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nwaves = SKM_LANE_THREADS / 64;
    const uint32_t L = rd.uni_len, wpr = (L + 15u) >> 4, nk = L - (uint32_t)sg.k + 1u, npos = L - 12u + 1u;
    uint32_t *wl = smem + wave * skm_lane_slice_words(wpr);
    uint32_t *rcl = wl + 64u * wpr + 4u;             // read r's reverse complement: base i of it = complement of the read's base L - 1 - i
    unsigned long long *ent = (unsigned long long *)(rcl + 64u * wpr + 4u);
    for (uint32_t c = threadIdx.x; c < sg.C1; c += SKM_LANE_THREADS) cur[c] = 0;
    __syncthreads();
    uint64_t n_rec = 0;
    uint64_t *const my_seg = sg.seg1 + skm_seg1_slot(sg, 0u, blockIdx.x) * sg.cap1 * (uint64_t)sg.recw;       // + coarse * cstride: this workgroup's segment of a coarse bucket
    const uint64_t cstride = (sg.seg1_wmajor ? 1ull : (uint64_t)sg.nwg1) * sg.cap1 * (uint64_t)sg.recw;
    const uint64_t n_words = rd.n_reads * (uint64_t)wpr;
    const uint32_t nblocks = (npos + B - 1u) / B;
    const uint32_t gstep = gridDim.x * nwaves;
    // words of group g as the wave loads them: word t = lane + 64 i of the group's 64 x wpr
    // (lanes and words beyond the batch's end read as zeros)
    auto fetch = [&](uint32_t g, uint32_t (&dst)[NW]) {
        const uint64_t first = (uint64_t)g * 64u * wpr;
        const uint32_t *gw = rd.words + first;
        const uint32_t have = (uint32_t)min((uint64_t)64u * wpr, n_words - first);
        uint32_t ln = lane;
        asm volatile("" : "+v"(ln));              // (or the 64-bit addresses of all NW words are computed once in front of the loop, and spilled)
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t t = ln + 64u * (uint32_t)i;
            dst[i] = t < have ? gw[t] : 0u;
        }
    };
    uint32_t pf[NW];
    uint32_t g = blockIdx.x * nwaves + wave;
    if (g < n_groups) fetch(g, pf);
    for (; g < n_groups; g += gstep) {
        const uint64_t r0 = (uint64_t)g * 64u;
        const uint32_t nr = (uint32_t)min((uint64_t)64u, rd.n_reads - r0);
        const bool active = lane < nr;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if ((uint32_t)i < wpr) wl[lane + 64u * (uint32_t)i] = pf[i];
        if (lane < 4u) { wl[64u * wpr + lane] = 0u; rcl[64u * wpr + lane] = 0u; }
        // the next group's words are requested now; they are waited for in front of the first flush (below), not at the top of the
        // next round, where the wait would also cover this round's record stores
        const bool more = g + gstep < n_groups;
        if (more) fetch(g + gstep, pf);
        bool pf_waited = !more;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (sg.oriented) {
            // every lane reverses and complements its own read once: the 16 wpr base slots of its words turned round (word order and
            // the bases inside each word), complemented, and the 16 wpr - L slots of padding that came to the front shifted out.  A
            // reversed run's bases are then read from this image exactly as a forward run's are read from the words.
            const uint32_t sh = 2u * (16u * wpr - L);                  // < 32
            uint32_t prev = ~skm_rev2_32(wl[lane * wpr + wpr - 1u]);
            for (uint32_t t = 0; t < wpr; ++t) {
                const uint32_t next = t + 1u < wpr ? ~skm_rev2_32(wl[lane * wpr + wpr - 2u - t]) : 0u;
                rcl[lane * wpr + t] = __builtin_amdgcn_alignbit(next, prev, sh);
                prev = next;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // a finished run as records (a run of more than ncap k-mers is cut)
        auto emit_run = [&](uint32_t v, uint32_t r, uint32_t j, uint32_t left) {
            uint32_t coarse, fine;
            skm_bucket_of(v, sg.C1, sg.fbits, coarse, fine);
            uint64_t pos = (sg.read_base + r0 + r) * sg.stride + j;
            const uint32_t rev = sg.oriented ? (v & 1u) : 0u;
            const uint32_t *src = rev ? rcl : wl;
            uint32_t bidx = r * wpr * 16u + j;                        // the piece's first base in the read (in wl)
            while (left) {
                const uint32_t n = min(left, (uint32_t)sg.ncap);
                // (a reversed piece: its n + k - 1 bases end at base L - j' of the reverse complement, j' its first base in the read)
                const uint32_t from = rev ? 2u * r * wpr * 16u + L - bidx - (n + (uint32_t)sg.k - 1u) : bidx;
                uint64_t bw[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? skm_bases32(src, from + 32u * t) : 0ull;
                const uint64_t hdr = skm_header(pos, n, fine, rev);
                const uint32_t p = atomicAdd(&cur[coarse], 1u);
                if (p < sg.cap1) {
                    if (sg.compact) {                                 // one aligned 16-byte store: bases, n and the fine bucket (no position)
                        typedef uint64_t u64x2a __attribute__((ext_vector_type(2), aligned(16)));
                        *(u64x2a *)(my_seg + coarse * cstride + p * 2u) = u64x2a{bw[0], skm_c_pack1(bw[1], n, fine, rev)};
                    } else {
                        skm_store_record_wide(my_seg + coarse * cstride + p * (uint32_t)sg.recw, hdr, bw, sg.nbw);
                    }
                }
                else skm_loose_push(sg, hdr, bw);
                n_rec += 1;
                left -= n; pos += n; bidx += n;
            }
        };
        uint32_t n_ent = 0;
        // an entry: the run's minimizer | (lane | first k-mer << 6 | the k-mer behind its last << 18) << 32
        auto append = [&](unsigned long long bal, bool want, uint32_t v, uint32_t lane_start, uint32_t end) -> bool {
            uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, n_ent));
#if SKM_LANE_BRANCHFREE
            const bool ok = want && slot < SKM_LANE_CAP;
            slot = ok ? slot : SKM_LANE_CAP;                          // (slot CAP: the word nobody reads)
            ent[slot] = (unsigned long long)v | ((unsigned long long)(lane_start | (end << 18)) << 32);
            return ok;
#else
            (void)want;
            if (slot >= SKM_LANE_CAP) return false;                   // (a wave of reads that change minimizer at nearly every k-mer)
            ent[slot] = (unsigned long long)v | ((unsigned long long)(lane_start | (end << 18)) << 32);
            return true;
#endif
        };
        auto flush = [&]() {
            if (!pf_waited) {
#pragma unroll
                for (int i = 0; i < NW; ++i) asm volatile("" : "+v"(pf[i]));          // the prefetched words have arrived from here on
                pf_waited = true;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n = min(n_ent, SKM_LANE_CAP);
            for (uint32_t e = lane; e < n; e += 64u) {
                const unsigned long long en = ent[e];
                const uint32_t hi = (uint32_t)(en >> 32), first = (hi >> 6) & 4095u;
                emit_run((uint32_t)en, hi & 63u, first, (hi >> 18) - first);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            n_ent = 0;
        };
        uint32_t Sa[B], Sb[B];
#pragma unroll
        for (int j = 0; j < B; ++j) { Sa[j] = 0xffffffffu; Sb[j] = 0xffffffffu; }
        uint32_t prev_v = 0, start_prev = 0;
        const uint32_t rbase = lane * wpr;
        uint32_t since = 0;
        for (uint32_t b = 0; b < nblocks; b += (C == 2 ? 2u : 1u)) {
            // block b is FULL when its first m-mer ends k-mer 1 or a later one and its last m-mer ends a k-mer of the read
            auto full = [&](uint32_t bb) { return (uint32_t)B * bb + 1u > (uint32_t)(C * B) && (uint32_t)B * bb + (uint32_t)B <= nk + (uint32_t)(C * B) - 1u; };
            if (SKM_LANE_FULL && full(b)) skm_lane_block<C, true>(wl, rbase, lane, b, npos, nk, Sa, Sb, prev_v, start_prev, n_ent, append);
            else skm_lane_block<C, false>(wl, rbase, lane, b, npos, nk, Sa, Sb, prev_v, start_prev, n_ent, append);
            if (C == 2 && b + 1u < nblocks) {
                if (SKM_LANE_FULL && full(b + 1u)) skm_lane_block<C, true>(wl, rbase, lane, b + 1u, npos, nk, Sb, Sa, prev_v, start_prev, n_ent, append);
                else skm_lane_block<C, false>(wl, rbase, lane, b + 1u, npos, nk, Sb, Sa, prev_v, start_prev, n_ent, append);
            }
            since += (C == 2 ? 2u : 1u);
            // (the last block's runs leave with the final ones; a list that is filling up is flushed whatever the count says)
            if ((since >= flush_blocks || n_ent + 192u > SKM_LANE_CAP) && n_ent && b + (C == 2 ? 2u : 1u) < nblocks) { flush(); since = 0; }
        }
        // the run every read ends in (room for all 64: the list is emptied first if need be)
        if (n_ent + 64u > SKM_LANE_CAP) flush();
        {
            const unsigned long long bal = __ballot(active);
            if (active) (void)append(bal, true, prev_v, start_prev, nk);
            n_ent += (uint32_t)__popcll(bal);
        }
        flush();
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < sg.C1; c += SKM_LANE_THREADS) sg.cnt1[skm_seg1_slot(sg, c, blockIdx.x)] = min(cur[c], sg.cap1);
    n_rec = wave_sum_u64(n_rec);
    if ((threadIdx.x & 63u) == 0 && n_rec) atomicAdd(&sg.ctr[5], (unsigned long long)n_rec);
}

// ---- S2 ------------------------------------------------------------------------------------------------
#define SKM_THREADS2 512
#define SKM_MAX_F2 4096u          // fine buckets per coarse bucket (12 bits of a k-mer's bucket id in S1; 16 in a record header)
// COMPACT: 16-byte records (kv_skm_device.h: the fine bucket sits in the second word; one 16-byte load and store each)
template <bool COMPACT>
__global__ __launch_bounds__(SKM_THREADS2) void k_skm_split(SkmGeom sg)
{
    __shared__ uint32_t cur[SKM_MAX_F2];
    __shared__ uint32_t spre[769];                    // record prefix over the coarse segments this workgroup drains
    const uint32_t c = blockIdx.y;
    for (uint32_t f = threadIdx.x; f < sg.F2; f += SKM_THREADS2) cur[f] = 0;
    // segments seg = blockIdx.x, blockIdx.x + nwg2, ...: their records are enumerated flat, so every thread has work
    const uint32_t nmine = (skm_seg1_count(sg) - blockIdx.x + sg.nwg2 - 1) / sg.nwg2;       // <= 768
    for (uint32_t i = threadIdx.x; i < nmine; i += SKM_THREADS2) spre[i + 1] = sg.cnt1[skm_seg1_slot(sg, c, blockIdx.x + i * sg.nwg2)];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < nmine; ++i) { const uint32_t n = spre[i + 1]; spre[i] = acc; acc += n; }
        spre[nmine] = acc;
    }
    __syncthreads();
    const uint32_t total = spre[nmine];
    const int recw = COMPACT ? 2 : sg.recw;
    for (uint32_t i = threadIdx.x; i < total; i += SKM_THREADS2) {
        const uint32_t si = skm_search(spre, nmine, i);
        const uint32_t seg = blockIdx.x + si * sg.nwg2;
        const uint64_t *rec = sg.seg1 + (skm_seg1_first(sg, skm_seg1_slot(sg, c, seg)) + (i - spre[si])) * (uint64_t)recw;
        if (COMPACT) {
            typedef uint64_t u64x2a __attribute__((ext_vector_type(2), aligned(16)));
            const u64x2a both = *(const u64x2a *)rec;
            const uint32_t fine = skm_c_fine(both.y);
            const uint32_t p = atomicAdd(&cur[fine], 1u);
            if (p < sg.cap2) {
                *(u64x2a *)(sg.seg2 + ((((uint64_t)c * sg.F2 + fine) * sg.nwg2 + blockIdx.x) * sg.cap2 + p) * 2ull) = both;
            } else {                                         // (leaves in the loose list's classic form, position unknown)
                uint64_t lw[3] = {both.x, skm_c_b1(both.y), 0ull};
                skm_loose_push(sg, skm_header(0ull, skm_c_n(both.y), fine, skm_c_rev(both.y)), lw);
            }
            continue;
        }
        const uint64_t hdr = rec[0];
        uint64_t bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? rec[1 + t] : 0ull;
        const uint32_t fine = skm_hdr_fine(hdr);
        const uint32_t p = atomicAdd(&cur[fine], 1u);
        if (p < sg.cap2)
            skm_store_record(sg.seg2 + ((((uint64_t)c * sg.F2 + fine) * sg.nwg2 + blockIdx.x) * sg.cap2 + p) * (uint64_t)recw, hdr, bw, sg.nbw);
        else skm_loose_push(sg, hdr, bw);
    }
    __syncthreads();
    for (uint32_t f = threadIdx.x; f < sg.F2; f += SKM_THREADS2)
        sg.cnt2[((uint64_t)c * sg.F2 + f) * sg.nwg2 + blockIdx.x] = min(cur[f], sg.cap2);
}

// S2 with the scatter sorted in LDS first.  The plain kernel above stores every record where it falls: 64 lanes, 64
// different cache lines per store instruction, 8-byte pieces of 24-byte records that straddle 32-byte sectors -- PMC:
// 2.5 GB written for 1.36 GB of records, 68 % of the wave cycles stalled on the memory pipe.  Here a workgroup takes
// SKM_S2_CHUNK records at a time, ranks them by fine bucket (LDS atomics: rank inside the chunk's share of the bucket),
// lays them out bucket by bucket in LDS and copies that image out with consecutive lanes on consecutive words: a bucket's
// records of one chunk leave as one contiguous run.  Used while a chunk holds at least ~2 records per bucket.
#if defined(SKM_S2_CHUNK_OVERRIDE)
#define SKM_S2_CHUNK SKM_S2_CHUNK_OVERRIDE
#else
#define SKM_S2_CHUNK 2048u
#endif
#define SKM_S2_MAXF 1024u
template <int RECW>
__global__ __launch_bounds__(SKM_THREADS2) void k_skm_split_sorted(SkmGeom sg)
{
    __shared__ uint32_t cur[SKM_S2_MAXF];            // records this workgroup has stored per fine bucket
    __shared__ uint32_t hist[SKM_S2_MAXF];           // the chunk's records per bucket, then slot of sorted position 0 minus that position
    __shared__ uint32_t off[SKM_S2_MAXF];            // sorted position of the bucket's first record of the chunk
    __shared__ uint32_t spre[769];
    __shared__ uint32_t wsum[SKM_THREADS2 / 64];
    extern __shared__ __attribute__((aligned(16))) uint64_t img[];     // [SKM_S2_CHUNK][RECW]
    const uint32_t c = blockIdx.y, F2 = sg.F2;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t f = threadIdx.x; f < F2; f += SKM_THREADS2) { cur[f] = 0; hist[f] = 0; }
    const uint32_t nmine = (skm_seg1_count(sg) - blockIdx.x + sg.nwg2 - 1) / sg.nwg2;       // <= 768
    for (uint32_t i = threadIdx.x; i < nmine; i += SKM_THREADS2) spre[i + 1] = sg.cnt1[skm_seg1_slot(sg, c, blockIdx.x + i * sg.nwg2)];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < nmine; ++i) { const uint32_t n = spre[i + 1]; spre[i] = acc; acc += n; }
        spre[nmine] = acc;
    }
    __syncthreads();
    const uint32_t total = spre[nmine];
    constexpr uint32_t PER = SKM_S2_CHUNK / SKM_THREADS2;             // records per thread and chunk
    const uint32_t fper = (F2 + SKM_THREADS2 - 1) / SKM_THREADS2;     // buckets per thread in the scan (1 or 2)
    uint64_t *const out = sg.seg2 + ((uint64_t)c * F2 * sg.nwg2 + blockIdx.x) * sg.cap2 * RECW;      // + fine * nwg2 * cap2 * RECW
    const uint64_t fstride = (uint64_t)sg.nwg2 * sg.cap2 * RECW;
    // the records of the NEXT chunk are requested before this chunk is ranked, laid out and stored: a workgroup's chunk used to begin
    // with a round trip to HBM that nothing covered (two workgroups per CU, five barriers per chunk)
    uint64_t nh[PER], n0[PER], n1[PER], n2[PER];
    auto request = [&](uint32_t c0) {
        const uint32_t n = c0 < total ? min(SKM_S2_CHUNK, total - c0) : 0u;
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            const uint32_t i = r * SKM_THREADS2 + threadIdx.x;
            nh[r] = 0; n0[r] = 0; n1[r] = 0; n2[r] = 0;
            if (i < n) {
                const uint32_t gi = c0 + i, si = skm_search(spre, nmine, gi);
                const uint32_t seg = blockIdx.x + si * sg.nwg2;
                const uint64_t *rec = sg.seg1 + (skm_seg1_first(sg, skm_seg1_slot(sg, c, seg)) + (gi - spre[si])) * (uint64_t)RECW;
                if (RECW == 2) {                                      // compact: the word that says the bucket is the second one
                    typedef uint64_t u64x2a __attribute__((ext_vector_type(2), aligned(16)));
                    const u64x2a both = *(const u64x2a *)rec;
                    n0[r] = both.x; nh[r] = both.y;
                } else {
                    nh[r] = rec[0]; n0[r] = rec[1]; n1[r] = rec[2];
                    if (RECW == 4) n2[r] = rec[3];
                }
            }
        }
    };
    request(0);
    for (uint32_t c0 = 0; c0 < total; c0 += SKM_S2_CHUNK) {
        const uint32_t n = min(SKM_S2_CHUNK, total - c0);
        uint64_t hdr[PER], w0[PER], w1[PER], w2[PER];
        uint32_t rank[PER];
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) { hdr[r] = nh[r]; w0[r] = n0[r]; w1[r] = n1[r]; w2[r] = n2[r]; rank[r] = 0; }
        request(c0 + SKM_S2_CHUNK);
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r)
            if (r * SKM_THREADS2 + threadIdx.x < n) rank[r] = atomicAdd(&hist[RECW == 2 ? skm_c_fine(hdr[r]) : skm_hdr_fine(hdr[r])], 1u);
        __syncthreads();
        // exclusive scan of the chunk's histogram; the bucket's run then starts at slot cur[f] of the private segment
        {
            uint32_t h[2] = {0, 0}, sum = 0;
            for (uint32_t j = 0; j < fper; ++j) {
                const uint32_t f = threadIdx.x * fper + j;
                h[j] = f < F2 ? hist[f] : 0u;
                sum += h[j];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d);
                if (lane >= (uint32_t)d) incl += up;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = incl - sum;
            for (uint32_t wv = 0; wv < wave; ++wv) before += wsum[wv];
            for (uint32_t j = 0; j < fper; ++j) {
                const uint32_t f = threadIdx.x * fper + j;
                if (f < F2) {
                    off[f] = before;
                    const uint32_t at = cur[f];
                    cur[f] = at + h[j];
                    hist[f] = at - before;              // slot = sorted position + this (mod 2^32)
                    before += h[j];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < PER; ++r) {
            if (r * SKM_THREADS2 + threadIdx.x < n) {
                uint64_t *dst = img + (uint64_t)(off[RECW == 2 ? skm_c_fine(hdr[r]) : skm_hdr_fine(hdr[r])] + rank[r]) * RECW;
                if (RECW == 2) { dst[0] = w0[r]; dst[1] = hdr[r]; }
                else { dst[0] = hdr[r]; dst[1] = w0[r]; dst[2] = w1[r]; }
                if (RECW == 4) dst[3] = w2[r];
            }
        }
        __syncthreads();
        for (uint32_t wd = threadIdx.x; wd < n * RECW; wd += SKM_THREADS2) {
            const uint32_t q = wd / RECW, part = wd - q * RECW;
            const uint64_t h0 = img[(uint64_t)q * RECW + (RECW == 2 ? 1 : 0)];
            const uint32_t f = RECW == 2 ? skm_c_fine(h0) : skm_hdr_fine(h0);
            const uint32_t slot = q + hist[f];
            if (slot < sg.cap2) out[(uint64_t)f * fstride + (uint64_t)slot * RECW + part] = img[wd];
            else if (part == 0) {
                if (RECW == 2) {                                      // a compact record leaves in the loose list's (classic) form, position unknown
                    uint64_t bw[3] = {img[(uint64_t)q * RECW], skm_c_b1(h0), 0ull};
                    skm_loose_push(sg, skm_header(0ull, skm_c_n(h0), f, skm_c_rev(h0)), bw);
                } else {
                    uint64_t bw[3] = {img[(uint64_t)q * RECW + 1], img[(uint64_t)q * RECW + 2], RECW == 4 ? img[(uint64_t)q * RECW + 3] : 0ull};
                    skm_loose_push(sg, h0, bw);
                }
            }
        }
        __syncthreads();
        for (uint32_t f = threadIdx.x; f < F2; f += SKM_THREADS2) hist[f] = 0;
        // (the next chunk's atomics on hist come after its loads and after the barrier below at the latest: the loop's
        // first barrier orders them behind this reset only if every thread passes it, so reset before a barrier)
        __syncthreads();
    }
    for (uint32_t f = threadIdx.x; f < F2; f += SKM_THREADS2)
        sg.cnt2[((uint64_t)c * F2 + f) * sg.nwg2 + blockIdx.x] = min(cur[f], sg.cap2);
}

// ---- LDS combining table -------------------------------------------------------------------------------
// the step from a slot to the next one tried: 1 (linear probing), or an odd number taken from the key's hash (double hashing: two keys
// that meet in one slot part ways at once, no clusters -- the wave waits for the lane with the longest chain)
// (k_skm_count 3.23 -> 3.04 ms per sample of config 2; -DSKM_LINEAR_PROBE keeps the old order for A/B builds)
#if defined(SKM_LINEAR_PROBE)
#define SKM_PROBE_STEP(h) 1u
#else
#if !defined(SKM_STEP_MASK)
#define SKM_STEP_MASK 62u
#endif
#define SKM_PROBE_STEP(h) ((((h) >> 20) & SKM_STEP_MASK) | 1u)
#endif
// slot a hash starts at / the slot `step` further on, for tables of 2^n slots and of any other size (3072: the count's instances that
// share their LDS with a table of records) -- there the top bits of the hash pick the slot, so the step comes from the low ones
template <int TS>
__device__ __forceinline__ uint32_t skm_slot0(uint32_t sh) { return (TS & (TS - 1)) == 0 ? sh & (uint32_t)(TS - 1) : __umulhi(sh, (uint32_t)TS); }
template <int TS>
__device__ __forceinline__ uint32_t skm_step_of(uint32_t sh) { return (TS & (TS - 1)) == 0 ? SKM_PROBE_STEP(sh) : (((sh >> 2) & 62u) | 1u); }
template <int TS>
__device__ __forceinline__ uint32_t skm_slot_next(uint32_t slot, uint32_t step)
{
    if ((TS & (TS - 1)) == 0) return (slot + step) & (uint32_t)(TS - 1);
    slot += step;
    return slot >= (uint32_t)TS ? slot - (uint32_t)TS : slot;
}
template <int KW, int TS>
struct SkmTable {
    unsigned long long key[KW][TS];
};

template <int KW, int TS>
__device__ __forceinline__ void skm_table_clear(SkmTable<KW, TS> &tb)
{
    for (uint32_t i = threadIdx.x; i < TS; i += blockDim.x) {
        tb.key[0][i] = SKM_EMPTY;
        if (KW == 2) tb.key[KW - 1][i] = SKM_EMPTY;
    }
}

// slot of the key, claiming a free slot if it is new; -1 when SKM_MAXPROBE slots are all taken by other keys.  With
// two key words a slot is identified word by word: whoever sets word 0 (or finds it equal) goes on to set or
// compare word 1, and moves to the next slot if another k-mer with the same first 32 bases got there first.
template <int KW, int TS>
__device__ __forceinline__ int skm_table_insert(SkmTable<KW, TS> &tb, const SkmKey<KW> &c)
{
    const uint32_t sh = skm_slot_hash<KW>(c);
    uint32_t slot = skm_slot0<TS>(sh);
    const uint32_t step = skm_step_of<TS>(sh);
    for (int probe = 0; probe < SKM_MAXPROBE; ++probe) {
        // (reading the slot first and swapping only into an empty one was measured 5 % slower: the read does not save
        // the swap's round trip, it adds one for every new key)
        const unsigned long long old0 = atomicCAS(&tb.key[0][slot], SKM_EMPTY, (unsigned long long)c.w[0]);
        if (old0 == SKM_EMPTY || old0 == c.w[0]) {
            if (KW == 1) return (int)slot;
            const unsigned long long old1 = atomicCAS(&tb.key[KW - 1][slot], SKM_EMPTY, (unsigned long long)c.w[KW - 1]);
            if (old1 == SKM_EMPTY || old1 == c.w[KW - 1]) return (int)slot;
        }
        slot = skm_slot_next<TS>(slot, step);
    }
    return -1;
}

template <int KW, int TS>
__device__ __forceinline__ int skm_table_find(const SkmTable<KW, TS> &tb, const SkmKey<KW> &c)
{
    const uint32_t sh = skm_slot_hash<KW>(c);
    uint32_t slot = skm_slot0<TS>(sh);
    const uint32_t step = skm_step_of<TS>(sh);
    for (int probe = 0; probe < SKM_MAXPROBE; ++probe) {
        const unsigned long long k0 = tb.key[0][slot];
        if (k0 == SKM_EMPTY) return -1;
        if (k0 == c.w[0] && (KW == 1 || tb.key[KW - 1][slot] == c.w[KW - 1])) return (int)slot;
        slot = skm_slot_next<TS>(slot, step);
    }
    return -1;
}

// count side of one distinct k-mer seen `count` times: filter, then T weighted items through `emit(table, bin, weight)`
template <typename Emit>
__device__ __forceinline__ uint32_t skm_count_kmer(uint64_t h, uint32_t count, const SketchDev *__restrict__ sk,
                                                   const SketchDev *__restrict__ mask, const ConsumeFilter &f, int T, Emit emit)
{
    if (!consume_filter_pass(f, mask, h)) return 0;
    uint32_t left = min(count, 255u);                  // no counter holds more than 255
    // (tried: lane l starting with table l mod 4, so that one emit instruction spreads over the cursors of all tables instead
    // of 64 lanes sharing the ~20 cursors of one -- 3.6 -> 4.2 ms: what costs is the number of distinct segments, i.e. cache
    // lines, one store instruction touches, not the same-address LDS atomics; more coarse buckets cost the same way)
    while (left) {
        const uint32_t wgt = min(left, BIN_W_MAX);
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t)
            if (t < T) emit(t, fastmod(h, sk->size[t], sk->magic[t]), wgt);
        left -= wgt;
    }
    return count;
}

// every wave walks its share of the table and hands the occupied slots to `body` 64 at a time (all lanes busy):
// occupied slots are queued in LDS and drained whenever a full wave of them is ready
template <int TS, typename Body>
__device__ __forceinline__ void skm_for_occupied(const unsigned long long *key0, uint16_t *queue_all, uint32_t queue_stride, Body body, const uint32_t *skip = nullptr)
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    uint16_t *queue = queue_all + wave * queue_stride;               // >= 128 entries; wave-private, ordered by the fences below
    const uint32_t per_wave = TS / nwaves;
    const uint32_t s_end = (wave + 1) * per_wave;
    uint32_t qn = 0;                                    // < 64 between iterations
    for (uint32_t s0 = wave * per_wave;; s0 += 64) {
        const bool last = s0 >= s_end;
        if (!last) {
            const uint32_t slot = s0 + lane;
            const bool occ = key0[slot] != SKM_EMPTY && !(skip && ((skip[slot >> 5] >> (slot & 31u)) & 1u));      // skip: a bit per slot
            const unsigned long long ballot = __ballot(occ);
            if (occ) __hip_atomic_store(&queue[qn + (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull))], (uint16_t)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            qn += (uint32_t)__popcll(ballot);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (qn >= 64 || (last && qn > 0)) {
            const uint32_t take = min(qn, 64u);
            qn -= take;
            if (lane < take) body((uint32_t)__hip_atomic_load(&queue[qn + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (last) break;
    }
}

// ---- balanced walk over the k-mer occurrences of a fine bucket ------------------------------------------------
// Records hold 1..ncap k-mers, so a thread per record would leave most lanes waiting for the longest one, and a
// thread per occurrence cuts every k-mer out of its record from scratch (round 2: 115 VALU instructions per
// occurrence, the largest item of the count kernel).  The walk therefore deals UNITS: SKM_UNIT consecutive k-mers
// of one record.  Each wave takes 64 records (one per lane, in registers), numbers their units with a prefix sum and
// walks them 64 at a time: lane p of block t0 handles unit t0 + p, finds the record that owns it from a bit mask of
// record starts (one LDS word pair per block, popcounts), fetches that record's words from the owning lane with
// ds_bpermute, cuts the unit's first k-mer out, takes its reverse complement once and ROLLS both strands through
// the unit's other k-mers (two shifts each).  Records average ~9 k-mers: units of 4 are 85 % full.
#define SKM_UNIT 4

__device__ __forceinline__ uint64_t skm_shfl64(uint64_t v, uint32_t src)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// bases j .. j + k - 1 of a record's base words
template <int KW>
__device__ __forceinline__ SkmKey<KW> skm_kmer_at(uint64_t b0, uint64_t b1, uint64_t b2, uint32_t j, int k)
{
    return skm_kmer_of<KW>(b0, b1, b2, j, k);
}

// occurrences that found no room in the LDS table leave as one-k-mer records (either strand of the k-mer): one atomic per wave
template <int KW, bool WANT_POS>
__device__ __forceinline__ void skm_walk_loose(const SkmGeom &sg, bool alone, const SkmKey<KW> &kmer, uint64_t hdr, uint32_t owner,
                                               uint64_t pos0, uint32_t j0, uint32_t u, uint32_t lane, int recw)
{
    const unsigned long long need = __ballot(alone);
    if (!need) return;
    // (the count pass does not fetch the records' headers for the walk; here -- a wave in a few hundred -- it does,
    // so that the loose record carries the occurrence's real position: a scan that goes by the count pass's
    // distinct list evaluates these records as they are)
    // (WANT_POS: pos0 is this occurrence's position already; else it is derived here from the owner's header, orientation included)
    const uint64_t posu = WANT_POS ? pos0 : skm_hdr_pos_of(skm_shfl64(hdr, owner), j0 + u);
    unsigned long long first = 0;
    if (lane == 0) first = atomicAdd(&sg.ctr[0], (unsigned long long)__popcll(need));
    first = skm_shfl64(first, 0);
    if (alone) {
        const unsigned long long idx = first + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
        uint64_t one[3] = {kmer.w[0], KW == 2 ? kmer.w[KW - 1] : 0ull, 0ull};
        if (idx < sg.loose_cap) skm_store_record(sg.loose + idx * (uint64_t)recw, skm_header(posu, 1u, 0u), one, sg.nbw);
        else sg.ctr[1] = 1;
    }
}

// body(canonical k-mer, forward k-mer, position of the occurrence) -> true if the occurrence could not be combined and
// must travel alone through the loose list; WANT_POS = false skips fetching the header (count pass)
// ORI: the bucket's records are oriented (SkmGeom::oriented): a k-mer's key is the k-mer as the record holds it
template <int KW, bool WANT_POS, int FK = 0, bool COMPACT = false, bool ORI = false, typename Body>
__device__ __forceinline__ void skm_walk_bucket(const SkmGeom &sg, uint32_t b, uint32_t *sbits_all, Body body)
{
    static_assert(!COMPACT || (KW == 1 && !WANT_POS), "compact records: one-word keys, no positions");
    constexpr uint32_t G = SKM_UNIT;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t *sbits = sbits_all + wave * sg.sbw;       // sbw words: 64 records x ncap k-mers (units need fewer), + the word a 64-bit read may straddle into
    const int k = FK ? FK : sg.k, recw = COMPACT ? 2 : (FK ? 1 + KW + 1 : sg.recw);        // FK: k (and with it the record width) known when compiled
    // a compact record (16 bytes, one load) is taken apart into what the code below knows: a header without a position, two base words
    auto load = [&](const uint64_t *rec, uint64_t &hdr, uint64_t &b0, uint64_t &b1, uint64_t &b2) {
        if (COMPACT) {
            typedef uint64_t u64x2a __attribute__((ext_vector_type(2), aligned(16)));
            const u64x2a both = *(const u64x2a *)rec;
            b0 = both.x; b1 = skm_c_b1(both.y); hdr = skm_header(0ull, skm_c_n(both.y), 0u, skm_c_rev(both.y));
        } else {
            hdr = rec[0]; b0 = rec[1]; b1 = rec[2];
            if (KW == 2) b2 = rec[3];
        }
    };
    // Which records a wave takes does not depend on how full the bucket's segments are: pair p = wave, wave + nwaves, ...
    // is records [64 g, 64 g + 64) of segment s = p mod nwg2, g = p / nwg2.  The records are therefore requested together
    // with the segment counts (which only mask them afterwards) instead of behind them, and the same addresses of the
    const uint32_t *cnt2 = sg.cnt2 + (uint64_t)b * sg.nwg2;
    uint32_t p = wave;
    uint32_t g = p / sg.nwg2, sgm = p - g * sg.nwg2;
    uint64_t hdr = 0, b0 = 0, b1 = 0, b2 = 0;
    {
        const uint32_t at = min(g * 64u + lane, sg.cap2 - 1u);
        const uint64_t *rec = sg.seg2 + (((uint64_t)b * sg.nwg2 + sgm) * sg.cap2 + at) * (uint64_t)recw;
        load(rec, hdr, b0, b1, b2);
    }
    uint32_t maxc = 0;
    for (uint32_t s2 = 0; s2 < sg.nwg2; ++s2) maxc = max(maxc, cnt2[s2]);
    for (;;) {
        if (g * 64u >= maxc) break;
        const uint32_t NR = cnt2[sgm], i = g * 64u + lane;      // (the name the code below knows the bound by)
        const uint32_t nk = i < NR ? skm_hdr_n(hdr) : 0u;
        const uint32_t nu = (nk + G - 1u) / G;          // units of this record
        uint32_t incl = nu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63), excl = incl - nu;
        const uint32_t exnk = excl | (nk << 16);        // < 64 * ncap units: 16 bits are plenty
        const uint32_t nwords = (total >> 5) + 2u;
        for (uint32_t wd = lane; wd < nwords; wd += 64) sbits[wd] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (nu) atomicOr(&sbits[excl >> 5], 1u << (excl & 31));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t before = 0;                            // records that start in front of this block
        for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            // (relaxed wave-scope loads: a volatile access through the generic pointer became a system-scope FLAT load
            // with its own s_waitcnt vmcnt(0), two in a row per block)
            const uint64_t starts = (uint64_t)__hip_atomic_load(&sbits[t0 >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) |
                                    ((uint64_t)__hip_atomic_load(&sbits[(t0 >> 5) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) << 32);
            const uint32_t t = t0 + lane;
            const uint32_t owner = (before + (uint32_t)__popcll(starts & ((2ull << lane) - 1ull)) - 1u) & 63u;
            before += (uint32_t)__popcll(starts);
            const uint32_t oe = (uint32_t)__shfl((int)exnk, (int)owner);
            const uint64_t o0 = skm_shfl64(b0, owner), o1 = skm_shfl64(b1, owner);
            const uint64_t o2 = KW == 2 ? skm_shfl64(b2, owner) : 0ull;
            const uint64_t oh = WANT_POS ? skm_shfl64(hdr, owner) : 0ull;
            const uint32_t j0 = t < total ? (t - (oe & 0xffffu)) * G : 0u;
            const uint32_t cnt = t < total ? min(G, (oe >> 16) - j0) : 0u;     // k-mers of this unit: 1..G (0: no unit)
            SkmKey<KW> fw = skm_kmer_of<KW>(o0, o1, o2, j0, k);
            SkmKey<KW> rc = fw;
            if (!ORI) rc = skm_revcomp<KW>(fw, k);
            const uint32_t tail = (uint32_t)skm_window64(o0, o1, o2, j0 + (uint32_t)k);   // the bases that enter k-mers 1 .. G - 1
            // read position of the unit's first k-mer and the step to the next: a reversed record's k-mer j stands at pos + n - 1 - j
            uint64_t pos0 = 0;
            int step = 1;
            if (WANT_POS) {
                const bool rv = skm_hdr_rev(oh) != 0u;
                pos0 = skm_hdr_pos(oh) + (rv ? skm_hdr_n(oh) - 1u - j0 : j0);
                step = rv ? -1 : 1;
            }
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {
                if (u) {
                    if (ORI) { SkmKey<KW> other = fw; skm_roll<KW>(fw, other, (tail >> (2u * (u - 1u))) & 3u, k); }      // (the other strand is dead code here)
                    else skm_roll<KW>(fw, rc, (tail >> (2u * (u - 1u))) & 3u, k);
                }
                const uint64_t posu = pos0 + (uint64_t)(int64_t)(step * (int)u);
                bool alone = false;
                if (u < cnt) alone = body(ORI ? fw : skm_canonical<KW>(fw, rc), fw, posu);
                skm_walk_loose<KW, WANT_POS>(sg, alone, fw, hdr, owner, posu, j0, u, lane, sg.lrecw);
            }
        }
        __builtin_amdgcn_wave_barrier();                // the next group clears the mask
        p += nwaves;
        g = p / sg.nwg2; sgm = p - g * sg.nwg2;
        if (g * 64u >= maxc) break;
        if (g * 64u + lane < cnt2[sgm]) {
            const uint64_t *rec = sg.seg2 + (((uint64_t)b * sg.nwg2 + sgm) * sg.cap2 + g * 64u + lane) * (uint64_t)recw;
            load(rec, hdr, b0, b1, b2);
        }
    }
}

// ---- identical records first (the k = 31 instances of the count) ------------------------------------------------------
// (KV_SKM_DEDUP=1 only -- an experiment that did not pay, kept for the next attempt: see the end of this comment.)
// Reads that cover the same stretch of the genome cut the same super-k-mers out of it: at 30 x, 61 % of a bucket's records repeat
// another record of the bucket base for base (measured on the synthetic trio; the rest were cut short by a read's end or hold an
// error), and they hold 60 % of the k-mer occurrences.  So the count first puts the bucket's RECORDS into a small LDS table (key = the
// bases the record's k-mers use + their number; value = how many records said the same), and only then walks every distinct record
// once, adding its weight to each of its k-mers: 2.5 x fewer cut-outs and table inserts for one insert per record.
// Anything irregular -- no room in the record table, a record longer than dd_maxn k-mers, a k-mer that finds the k-mer table full --
// only raises a flag: the workgroup then empties its tables and walks the bucket the plain way (skm_walk_bucket), whose loose
// records carry the positions of the occurrences themselves.  (Saturating adds compose in any grouping: weights are exact.)
// Measured (config 2, one stream, ms per step for the three samples): k_skm_count 9.25 without, 10.2 with 512 record slots + 3072 k-mer
// slots (114 k buckets instead of 64 k: S2 2.24 -> 2.65 as well), 10.15 with 1024 + 4096 slots at two workgroups per CU (64 k buckets).
// The walk and the inserts it saves (1.5 of 3.1 ms per sample, 60 % of them) are outweighed by what it adds per bucket: a phase that
// only waits for the records and three dependent LDS atomics, a second barrier, and a walk whose groups of 64 slots hold ~22 records.
#define SKM_REC_MAXPROBE 16
template <int RS>
struct SkmRecTable {
    unsigned long long k0[RS], k1[RS];
    uint32_t w[RS];
};

template <int RS>
__device__ __forceinline__ void skm_rec_table_clear(SkmRecTable<RS> &rt)
{
    for (uint32_t i = threadIdx.x; i < RS; i += blockDim.x) { rt.k0[i] = SKM_EMPTY; rt.k1[i] = SKM_EMPTY; rt.w[i] = 0u; }
}

// every record of bucket b into the table; true if one of this thread's records could not be put there
template <int RS, bool COMPACT>
__device__ __forceinline__ bool skm_rec_combine(const SkmGeom &sg, uint32_t b, SkmRecTable<RS> &rt, int k)
{
    static_assert((RS & (RS - 1)) == 0, "record table: 2^n slots");
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    constexpr uint32_t recw = COMPACT ? 2u : 3u;
    const uint32_t *cnt2 = sg.cnt2 + (uint64_t)b * sg.nwg2;
    bool failed = false;
    uint32_t p = wave;
    uint32_t g = p / sg.nwg2, sgm = p - g * sg.nwg2;
    uint64_t b0 = 0, w1 = 0;
    auto load = [&](const uint64_t *rec) {
        if (COMPACT) {
            typedef uint64_t u64x2a __attribute__((ext_vector_type(2), aligned(16)));
            const u64x2a both = *(const u64x2a *)rec;
            b0 = both.x; w1 = skm_c_b1(both.y) | ((uint64_t)skm_c_n(both.y) << 40);
        } else {
            const uint32_t n = skm_hdr_n(rec[0]);
            b0 = rec[1];
            // (more than 20 bases in the second word: not a record this table takes -- the number of k-mers says so below)
            w1 = (rec[2] & ((1ull << 40) - 1ull)) | ((uint64_t)n << 40);
        }
    };
    {   // (as in skm_walk_bucket: the first records are requested together with the segment counts)
        const uint32_t at = min(g * 64u + lane, sg.cap2 - 1u);
        load(sg.seg2 + (((uint64_t)b * sg.nwg2 + sgm) * sg.cap2 + at) * (uint64_t)recw);
    }
    uint32_t maxc = 0;
    for (uint32_t s2 = 0; s2 < sg.nwg2; ++s2) maxc = max(maxc, cnt2[s2]);
    for (;;) {
        if (g * 64u >= maxc) break;
        if (g * 64u + lane < cnt2[sgm]) {
            const uint32_t n = (uint32_t)(w1 >> 40), nb = n + (uint32_t)k - 1u;          // the bases its k-mers use: k .. 52
            if (n == 0u || n > sg.dd_maxn || nb > SKM_C_BASES) {
                failed = true;
            } else {
                // bases behind the last k-mer are whatever followed in the read: not part of the key
                const uint64_t key0 = nb >= 32u ? b0 : b0 & ((1ull << (2u * nb)) - 1ull);
                const uint64_t key1 = (nb > 32u ? w1 & ((1ull << (2u * (nb - 32u))) - 1ull) : 0ull) | ((uint64_t)n << 40);
                uint32_t y = (uint32_t)key0 * 0x9e3779b1u ^ (uint32_t)(key0 >> 32) * 0x85ebca6bu ^ (uint32_t)key1 * 0xc2b2ae35u ^ (uint32_t)(key1 >> 32) * 0x27d4eb2fu;
                y ^= y >> 15;
                y *= 0x2c1b3c6du;
                uint32_t slot = y >> (32 - __builtin_ctz(RS));
                const uint32_t step = ((y >> 3) & 30u) | 1u;
                bool placed = false;
                if (key0 != SKM_EMPTY) {
                    for (int probe = 0; probe < SKM_REC_MAXPROBE; ++probe) {
                        const unsigned long long old0 = atomicCAS(&rt.k0[slot], SKM_EMPTY, (unsigned long long)key0);
                        if (old0 == SKM_EMPTY || old0 == key0) {
                            const unsigned long long old1 = atomicCAS(&rt.k1[slot], SKM_EMPTY, (unsigned long long)key1);
                            if (old1 == SKM_EMPTY || old1 == key1) { atomicAdd(&rt.w[slot], 1u); placed = true; break; }
                        }
                        slot = (slot + step) & (uint32_t)(RS - 1);
                    }
                }
                failed = failed || !placed;
            }
        }
        p += nwaves;
        g = p / sg.nwg2; sgm = p - g * sg.nwg2;
        if (g * 64u >= maxc) break;
        if (g * 64u + lane < cnt2[sgm])
            load(sg.seg2 + (((uint64_t)b * sg.nwg2 + sgm) * sg.cap2 + g * 64u + lane) * (uint64_t)recw);
    }
    return failed;
}

// every distinct record of the table once (the table is emptied on the way): body(k-mer as the record holds it, weight) for each of
// its k-mers, dealt to the lanes in units of SKM_UNIT k-mers like skm_walk_bucket deals them
template <int RS, typename Body>
__device__ __forceinline__ void skm_rec_walk(const SkmGeom &sg, SkmRecTable<RS> &rt, uint32_t *sbits_all, int k, Body body)
{
    constexpr uint32_t G = SKM_UNIT;
    __shared__ uint8_t lane_of_all[SKM_THREADS3];       // the wave's occupied slots in order: the r-th of them sits in lane lane_of[r]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    uint32_t *sbits = sbits_all + wave * sg.sbw;
    uint8_t *lane_of = lane_of_all + wave * 64u;
    for (uint32_t s0 = wave * 64u; s0 < (uint32_t)RS; s0 += nwaves * 64u) {
        const uint32_t s = s0 + lane;
        const uint64_t b0 = rt.k0[s];
        uint64_t b1 = 0;
        uint32_t nk = 0, wgt = 0;
        if (b0 != SKM_EMPTY) {
            const uint64_t w1 = rt.k1[s];
            wgt = rt.w[s];
            rt.k0[s] = SKM_EMPTY; rt.k1[s] = SKM_EMPTY; rt.w[s] = 0u;
            nk = (uint32_t)(w1 >> 40);
            b1 = w1 & ((1ull << 40) - 1ull);
        }
        const unsigned long long occ = __ballot(nk != 0u);
        if (!occ) continue;
        if (nk) lane_of[__popcll(occ & ((1ull << lane) - 1ull))] = (uint8_t)lane;      // (ordered with the reads below by the fences around the start bits)
        const uint32_t nu = (nk + G - 1u) / G;
        uint32_t incl = nu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63), excl = incl - nu;
        const uint32_t exnk = excl | (nk << 16);
        const uint32_t nwords = (total >> 5) + 2u;
        for (uint32_t wd = lane; wd < nwords; wd += 64) sbits[wd] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (nu) atomicOr(&sbits[excl >> 5], 1u << (excl & 31));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t before = 0;
        for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            const uint64_t starts = (uint64_t)__hip_atomic_load(&sbits[t0 >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) |
                                    ((uint64_t)__hip_atomic_load(&sbits[(t0 >> 5) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) << 32);
            const uint32_t t = t0 + lane;
            // (the start bits number the records that have units: empty slots lie between them here, hence the look-up)
            const uint32_t owner = lane_of[(before + (uint32_t)__popcll(starts & ((2ull << lane) - 1ull)) - 1u) & 63u];
            before += (uint32_t)__popcll(starts);
            const uint32_t oe = (uint32_t)__shfl((int)exnk, (int)owner);
            const uint32_t ow = (uint32_t)__shfl((int)wgt, (int)owner);
            const uint64_t o0 = skm_shfl64(b0, owner), o1 = skm_shfl64(b1, owner);
            const uint32_t j0 = t < total ? (t - (oe & 0xffffu)) * G : 0u;
            const uint32_t cnt = t < total ? min(G, (oe >> 16) - j0) : 0u;
            SkmKey<1> fw = skm_kmer_of<1>(o0, o1, 0ull, j0, k);
            const uint32_t tail = (uint32_t)skm_window64(o0, o1, 0ull, j0 + (uint32_t)k);
#pragma unroll
            for (uint32_t u = 0; u < G; ++u) {
                if (u) { SkmKey<1> other = fw; skm_roll<1>(fw, other, (tail >> (2u * (u - 1u))) & 3u, k); }
                if (u < cnt) body(fw, ow);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- S3: count ------------------------------------------------------------------------------------------
// dynamic LDS of the bucket kernels: [256] byte -> ASCII table, [T * C] bin cursors (count only), then the per-wave
// scratch: record-start bits of the walk, reused as the queue of occupied slots
__host__ __device__ inline uint32_t skm_wave_scratch_words(uint32_t sbw) { return sbw > 64u ? sbw : 64u; }

// FK: the instance for the k everybody runs (kevlar's default, 31; the host checks k and the record width): shifts, masks and the
// murmur tail become constants -- 3 % of the kernel.  FK = 0 reads k from the geometry.
// waves per SIMD the two-word-key instances are compiled for (6: three workgroups per CU at 80 VGPRs, 13-19 of them spilled; 4: two
// workgroups at up to 128 VGPRs, nothing spilled)
#if !defined(SKM_K2_WAVES)
#define SKM_K2_WAVES 6
#endif
// RS: slots of the table of records (above) the instance combines identical records in before it walks them; 0: it walks them all
// The instances for a fixed k (FK != 0, no record table) take the two murmurs from the product tables (skm_key_hash_pl: P1 / P2, 4 KB of
// dynamic LDS in place of the 1 KB ASCII table) and keep their occurrence counters as 16-bit halves of a word -- that is where the
// 3 KB come from with three workgroups on a CU; the host launches them only for buckets that cannot hold 65536 occurrences
// (skm_count_pl_fits), the instances with k at run time count in 32 bits.  -DSKM_PL=0: A/B builds without either.
#if !defined(SKM_PL)
#define SKM_PL 1
#endif
// slots of the k = 51 instances' table: their buckets are sized for 2048 slots (the scan kernels' tables), the count's own table may be
// roomier -- fewer occurrences on the loose list, shorter probe chains -- as long as three workgroups fit a CU (2560: 52.7 KB each)
#if !defined(SKM_TS51)
#define SKM_TS51 2560
#endif
// (k = 31: 4608 slots measured 1 % faster than 4096 -- and let a bucket's distinct list outgrow what the list scan takes, SKM_LIST_MAX:
// scratch/fuzz_list.py seed 702, trial 149 lost 8 % of its hits; the table stays at 4096, the assertion below holds every instance to it)
#if !defined(SKM_TS31)
#define SKM_TS31 4096
#endif
// entries of one bucket the scan from the distinct list takes (k_skm_novel_list): a bucket's list has at most as many entries as the
// count kernel's LDS table has slots
#define SKM_LIST_MAX 4096u
template <int KW, int TS, bool KNOBS, int FK, bool COMPACT = false, bool ORI = false, int RS = 0>
__global__ __launch_bounds__(SKM_THREADS3, KW == 2 ? SKM_K2_WAVES : (RS >= 1024 ? 4 : 6)) void k_skm_count(SkmGeom sg, const SketchDev *__restrict__ sk,
                                                           const SketchDev *__restrict__ mask, ConsumeFilter f, BinGeom g)
{
    static_assert(RS == 0 || (KW == 1 && ORI && !KNOBS), "records are combined in the oriented one-word instances");
    static_assert((uint32_t)TS <= SKM_LIST_MAX, "a bucket's distinct list (one entry per occupied slot) must fit what k_skm_novel_list takes");
    constexpr bool PL = SKM_PL && FK != 0 && RS == 0;
    __shared__ SkmTable<KW, TS> tb;
    __shared__ uint32_t cnt[PL ? TS / 2 : TS];   // occurrences of the key in the same slot (PL: slot s in half s & 1 of word s >> 1)
    __shared__ SkmRecTable<RS ? RS : 1> rt;
    __shared__ uint32_t next_bucket;
    __shared__ uint32_t abl_cur, abl_b0, abl_prev;      // abundance list: entries appended so far, ... when the bucket began, the bucket
    __shared__ uint32_t dl_cur, dl_b0;                  // distinct list: the same two
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    const uint32_t ns = (uint32_t)(g.T * g.C);
    uint32_t *lut = dyn, *cur = dyn + (PL ? 1024 : 256), *scratch = cur + ((ns + 3u) & ~3u);
    uint64_t *P1 = (uint64_t *)dyn, *P2 = P1 + 256;
    for (uint32_t s = threadIdx.x; s < ns; s += SKM_THREADS3) cur[s] = 0;
    if (threadIdx.x < 256) {
        if (PL) { P1[threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C1); P2[threadIdx.x] = skm_ascii4_times(threadIdx.x, MM_C2); }
        else lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    }
    const int k = FK ? FK : sg.k;
    const HashParams hp = FK ? make_hash_params(FK, f.hp.hashfam) : f.hp;
    uint64_t n_added = 0, n_distinct = 0;
    const uint32_t cap1 = (uint32_t)g.cap1;
    uint32_t *my_seg = g.gbuf1 + (uint64_t)blockIdx.x * g.cap1;          // + stream * seg_stride: this workgroup's segment of a stream
    const uint64_t seg_stride = (uint64_t)g.nwgA * g.cap1;
    auto emit = [&](int t, uint64_t bin, uint32_t wgt) {
        const uint32_t slice = (uint32_t)(bin >> g.sbits);
        const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
        const uint32_t sidx = (uint32_t)t * (uint32_t)g.C + c;
        const uint32_t item = ((slice - c * (uint32_t)g.F) << g.sbits) | ((uint32_t)bin & ((1u << g.sbits) - 1u)) | ((wgt - 1u) << BIN_W_SHIFT);
        const uint32_t pos = atomicAdd(&cur[sidx], 1u);
        if (pos < cap1) my_seg[sidx * seg_stride + pos] = item;
        else spill_item(g, t, bin, wgt);
    };
    // the table is emptied as it is read (below), so it is cleared only once; the next bucket's ticket is fetched
    // while the current bucket is processed
    skm_table_clear(tb);
    for (uint32_t i = threadIdx.x; i < (PL ? TS / 2 : TS); i += SKM_THREADS3) cnt[i] = 0;
    if constexpr (RS != 0) skm_rec_table_clear(rt);
    if (threadIdx.x == 0) { next_bucket = (uint32_t)atomicAdd(&sg.ctr[3], 1ull) * sg.bpt; abl_cur = 0; abl_b0 = 0; abl_prev = 0xffffffffu; dl_cur = 0; dl_b0 = 0; }
    // where the finished bucket's entries of the abundance list lie (nothing if the workgroup's stretch ran out)
    auto abl_close = [&]() {
        if (sg.abl_keys && abl_prev != 0xffffffffu) {
            const bool fits = abl_cur <= sg.abl_cap_wg;
            sg.abl_bstart[abl_prev] = blockIdx.x * sg.abl_cap_wg + abl_b0;
            sg.abl_bcount[abl_prev] = fits ? abl_cur - abl_b0 : 0u;
        }
        if (sg.dl_keys && abl_prev != 0xffffffffu) {
            sg.dl_bstart[abl_prev] = blockIdx.x * sg.dl_cap_wg + dl_b0;
            sg.dl_bcount[abl_prev] = dl_cur - dl_b0;                // (a stretch that ran out is reported through ctr[9]: the whole list is then dropped)
        }
    };
    for (uint32_t taken = 0; taken < sg.quota3; ++taken) {
        __syncthreads();
        const uint32_t b = next_bucket;
        if (b >= sg.n_buckets) break;
        __syncthreads();
        if (threadIdx.x == 0) {
            abl_close();
            abl_prev = b; abl_b0 = abl_cur; dl_b0 = dl_cur;
            if ((b + 1u) & (sg.bpt - 1u)) next_bucket = b + 1u;
            else if (taken + 1 >= sg.quota3 || __hip_atomic_load(&sg.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) next_bucket = 0xffffffffu;
            else {
                const unsigned long long ticket = atomicAdd(&sg.ctr[3], 1ull);
                next_bucket = (uint32_t)ticket * sg.bpt;     // (a raised flag ends the pass: the caller redoes the batch)
                // a batch that does not fit the LDS tables (low coverage per batch: nearly every k-mer distinct) is given up
                // early: once 2 % of the buckets are done, more than 4 % of their k-mers outside the tables raise the flag
                const unsigned long long done = ticket * sg.bpt;
                if (done * 50ull >= sg.n_buckets) {
                    const unsigned long long now = __hip_atomic_load(&sg.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), start = sg.ctr[6];
                    if (now > start && (now - start) * 25ull > done * sg.bucket_kmers) sg.ctr[1] = 1;
                }
            }
        }
        // combine the occurrences of the bucket: identical records first where the instance has the table for them
        bool plain = true;
        if constexpr (RS != 0) {
            bool failed = skm_rec_combine<RS, COMPACT>(sg, b, rt, k);
            __syncthreads();
            skm_rec_walk<RS>(sg, rt, scratch, k, [&](const SkmKey<1> &c1, uint32_t wgt) {
                SkmKey<KW> c;
                c.w[0] = c1.w[0];
                const int slot = skm_table_insert(tb, c);
                if (slot >= 0) atomicAdd(&cnt[slot], wgt);
                else failed = true;
            });
            plain = __syncthreads_or(failed ? 1 : 0) != 0;
            if (plain) {                         // (a bucket in thousands: too many distinct records or k-mers for the tables) start over
                skm_table_clear(tb);
                for (uint32_t i = threadIdx.x; i < TS; i += SKM_THREADS3) cnt[i] = 0;
                if (threadIdx.x == 0) atomicAdd(&sg.ctr[12], 1ull);
                __syncthreads();
            }
        }
        if (plain && !(SKM_DBG(sg) & 2u)) skm_walk_bucket<KW, false, FK, COMPACT, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t) {
            if (SKM_DBG(sg) & 128u) { n_added += c.w[0] & 1; return false; }
            // (KV_SKM_FORCE_LOOSE: one key in 64 is treated like a key that found its table full -- every occurrence travels alone;
            // results stay exact, tests use it to put single k-mers on the loose list of a batch that otherwise fits)
            const bool forced = (SKM_DBG(sg) & 4096u) && ((c.w[0] * 0x9e3779b97f4a7c15ull) >> 58) == 0;
            const int slot = skm_cacheable<KW>(c) && !forced ? skm_table_insert(tb, c) : -1;
            if (slot >= 0 && !(SKM_DBG(sg) & 256u)) {
                if (PL) atomicAdd(&cnt[(uint32_t)slot >> 1], 1u << (((uint32_t)slot & 1u) * 16u));
                else atomicAdd(&cnt[slot], 1u);
            }
            return slot < 0;         // table region full (or unstorable key): this occurrence travels alone
        });
        __syncthreads();
        // every distinct k-mer once
        skm_for_occupied<TS>(tb.key[0], (uint16_t *)scratch, skm_wave_scratch_words(sg.sbw) * 2u, [&](uint32_t slot) {
            SkmKey<KW> c;
            c.w[0] = tb.key[0][slot];
            tb.key[0][slot] = SKM_EMPTY;
            if (KW == 2) { c.w[KW - 1] = tb.key[KW - 1][slot]; tb.key[KW - 1][slot] = SKM_EMPTY; }
            uint32_t seen;
            if (PL) {               // (the other half of the word is another lane's, maybe at this very moment)
                const uint32_t sh = (slot & 1u) * 16u;
                seen = (cnt[slot >> 1] >> sh) & 0xffffu;
                atomicSub(&cnt[slot >> 1], seen << sh);
            } else {
                seen = cnt[slot];
                cnt[slot] = 0;
            }
            n_distinct += 1;
            if (SKM_DBG(sg) & 1u) return;
            uint64_t h;
            if constexpr (PL) h = skm_key_hash_pl<KW, FK>(c, P1, P2);
            else h = skm_key_hash<KW>(c, lut, hp);
            if (SKM_DBG(sg) & 64u) { n_added += h & 1; return; }
            // distinct list: key and hash of every k-mer in here (the scan of this batch then neither combines nor hashes again)
            if (sg.dl_keys) {
                const unsigned long long here = __ballot(true);
                const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)here) - 1u;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&dl_cur, (uint32_t)__popcll(here));
                base = (uint32_t)__shfl((int)base, (int)leader);
                const uint32_t at = base + (uint32_t)__popcll(here & ((1ull << lane) - 1ull));
                if (at < sg.dl_cap_wg) {
                    const uint64_t e = (uint64_t)blockIdx.x * sg.dl_cap_wg + at;
                    sg.dl_keys[e * KW] = c.w[0];
                    if (KW == 2) sg.dl_keys[e * KW + (KW - 1)] = c.w[KW - 1];
                    sg.dl_hash[e] = h;
                }
            }
            uint32_t added;
            if (g.fast4 && !f.use_mask) {
                // four tables below 2^31 bins: the quotient of h / size from the FP64 pipe (kv_fastmod.h), the remainder in 32 bits --
                // h - q size lies in [-size, size), so its low word is the remainder or the remainder + 2^32 - size, told apart by bit 31 --
                // and the items appended with 32-bit index arithmetic: ~95 lane-instructions per k-mer for the 222 of the general form
                // (64-bit remainders, the sizes fetched from the sketch's descriptor, scalar registers spilled around them)
                const bool pass = !f.use_band || (h >= f.band_lo && h < f.band_hi);
                added = pass ? seen : 0u;
                if (pass) {
                    const double hd = fma((double)(uint32_t)(h >> 32), 4294967296.0, (double)(uint32_t)h);
                    uint32_t bin[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double q = fma(hd, __longlong_as_double((long long)g.tmagic[t]), 4503599627370496.0);
                        const uint32_t psz = (uint32_t)g.tsize[t];
                        uint32_t r = (uint32_t)h - (uint32_t)__double_as_longlong(q) * psz;
                        r += (uint32_t)((int32_t)r >> 31) & psz;
                        bin[t] = r;
                    }
                    uint32_t left = min(seen, 255u);
                    const uint32_t omask = (1u << g.sbits) - 1u, stride32 = (uint32_t)seg_stride;
                    while (left) {
                        const uint32_t wgt = min(left, BIN_W_MAX), wbits = (wgt - 1u) << BIN_W_SHIFT;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const uint32_t slice = bin[t] >> g.sbits;
                            const uint32_t c = g.F == 1 ? slice : __umulhi(slice, g.recipF);
                            const uint32_t sidx = (uint32_t)t * (uint32_t)g.C + c;
                            const uint32_t item = ((slice - c * (uint32_t)g.F) << g.sbits) | (bin[t] & omask) | wbits;
                            const uint32_t pos = atomicAdd(&cur[sidx], 1u);
                            if (pos < cap1) my_seg[sidx * stride32 + pos] = item;
                            else spill_item(g, t, (uint64_t)bin[t], wgt);
                        }
                        left -= wgt;
                    }
                }
            } else {
                added = skm_count_kmer(h, seen, sk, mask, f, g.T, emit);
            }
            n_added += added;
            // abundance list: a k-mer this batch adds at least twice (the lanes of the wave that are in here vote)
            if (sg.abl_keys) {
                const bool want = added >= 2u;
                const unsigned long long here = __ballot(true), vote = __ballot(want);
                if (vote) {
                    const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)here) - 1u;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(&abl_cur, (uint32_t)__popcll(vote));
                    base = (uint32_t)__shfl((int)base, (int)leader);
                    const uint32_t at = base + (uint32_t)__popcll(vote & ((1ull << lane) - 1ull));
                    if (want && at < sg.abl_cap_wg) {
                        const uint64_t e = (uint64_t)blockIdx.x * sg.abl_cap_wg + at;
                        sg.abl_keys[e * KW] = c.w[0];
                        if (KW == 2) sg.abl_keys[e * KW + (KW - 1)] = c.w[KW - 1];
                        sg.abl_cnts[e] = (uint8_t)min(added, 255u);
                    }
                }
            }
        });
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        abl_close();
        if (sg.dl_keys && dl_cur > sg.dl_cap_wg) atomicAdd(&sg.ctr[9], 1ull);
    }
    for (uint32_t s = threadIdx.x; s < ns; s += SKM_THREADS3)
        g.gcnt1[(uint64_t)s * g.nwgA + blockIdx.x] = (uint32_t)min((uint64_t)cur[s], g.cap1);
    n_added = wave_sum_u64(n_added);
    n_distinct = wave_sum_u64(n_distinct);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
    if ((threadIdx.x & 63) == 0 && n_distinct) atomicAdd(&sg.ctr[7], (unsigned long long)n_distinct);
}

// loose records: every k-mer occurrence on its own, increments through the spill list (global atomics)
template <int KW>
__global__ __launch_bounds__(256) void k_skm_loose_count(SkmGeom sg, const SketchDev *__restrict__ sk,
                                                         const SketchDev *__restrict__ mask, ConsumeFilter f, BinGeom g)
{
    __shared__ uint32_t lut[256];
    if (sg.ctr[1] != 0) return;                  // the pass was given up: the caller redoes the batch another way
    lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    __syncthreads();
    unsigned long long n = sg.ctr[0];
    if (n > sg.loose_cap) n = sg.loose_cap;
    const int k = sg.k, recw = sg.lrecw;
    uint64_t n_added = 0;
    // wave-uniform loops (every lane votes in spill_items_wave).  A wave takes 64 records at a time: the one-k-mer records
    // (occurrences that missed a full LDS table: nearly all of them) are handled one per lane; a record with several k-mers
    // (it missed its segment in S1 / S2) is spread over the lanes, one k-mer each, instead of holding 63 lanes up while one
    // lane rolls through it
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u;
    auto count_one = [&](const SkmKey<KW> &fw, bool live) {
        uint64_t h = 0;
        bool pass = false;
        if (live) {
            h = skm_key_hash<KW>(skm_canonical<KW>(fw, skm_revcomp<KW>(fw, k)), lut, f.hp);
            pass = consume_filter_pass(f, mask, h);
        }
        n_added += pass ? 1 : 0;
        uint64_t bins[BIN_MAX_T];
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t) bins[t] = (pass && t < g.T) ? fastmod(h, sk->size[t], sk->magic[t]) : 0ull;
        spill_items_wave(g, bins, 1u, pass);
    };
    // records a wave takes at a time: the records with several k-mers are handled one after the other (each a round trip to the spill
    // counter), so a short list -- 40 k records at config 2 -- is dealt a few records to every wave, not 64 to one wave in seven
    // (0.135 -> 0.0x ms per launch; a long list -- k = 51: millions of one-k-mer records -- keeps 64)
    const uint64_t total_waves = (uint64_t)gridDim.x * (blockDim.x >> 6), wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t per = (uint32_t)min((unsigned long long)64, max((unsigned long long)1, (n + total_waves - 1) / total_waves));
    (void)stride;
    for (uint64_t i0 = wave_id * per; i0 < n; i0 += total_waves * per) {
        const uint64_t i = i0 + lane;
        const bool have = lane < per && i < n;
        const uint64_t *rec = sg.loose + (have ? i : 0) * (uint64_t)recw;
        uint64_t bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = (have && t < sg.nbw) ? rec[1 + t] : 0ull;
        const uint32_t nk = have ? skm_hdr_n(rec[0]) : 0u;
        count_one(skm_first_kmer<KW>(bw, k), nk == 1);
        unsigned long long longer = __ballot(nk > 1);
        while (longer) {
            const uint32_t src = (uint32_t)__ffsll((long long)longer) - 1u;
            longer &= longer - 1ull;
            const uint64_t b0 = skm_shfl64(bw[0], src), b1 = skm_shfl64(bw[1], src), b2 = skm_shfl64(bw[2], src);
            const uint32_t nk_src = (uint32_t)__shfl((int)nk, (int)src);
            for (uint32_t j0 = 0; j0 < nk_src; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool live = j < nk_src;
                count_one(skm_kmer_at<KW>(b0, b1, b2, live ? j : 0u, k), live);
            }
        }
    }
    n_added = wave_sum_u64(n_added);
    if ((threadIdx.x & 63) == 0 && n_added) atomicAdd(&g.ctr[2], (unsigned long long)n_added);
}

// ---- S3': route (read-sharded multi-GPU count) ---------------------------------------------------------------
// Same walk as k_skm_count, but a distinct k-mer leaves as one (hash, count) item for the rank that owns the hash's
// band instead of T bin items: what crosses xGMI shrinks by the shard's own coverage, and the owner adds each item
// with one weighted saturating add per table (kv_consume_hashes_weighted).
#define SKM_ROUTE_MAX_DEST 16
// (out of line: a pair that misses its segment is rare in the kernels that call this from their drain, whose registers it must not cost)
struct SkmOverflowSink { unsigned long long *ctr; uint64_t *ovf; uint8_t *ovf_dest; uint64_t ovf_cap; int ndest; };
__device__ __attribute__((noinline)) void skm_route_overflow(SkmOverflowSink rs, uint32_t d, uint64_t h, uint64_t count)
{
    {
        // the overflow list: one returning atomic for the lanes of the wave that are here together, not one each (the loose kernel sends
        // every item this way: 4.5 M of them took 25 ms one by one at ~180 per microsecond), and one add per destination among them
        const unsigned long long here = __ballot(true);
        const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)here) - 1u;
        unsigned long long o = 0;
        if (lane == leader) o = atomicAdd(&rs.ctr[1], (unsigned long long)__popcll(here));
        o = skm_shfl64(o, leader) + (unsigned long long)__popcll(here & ((1ull << lane) - 1ull));
        const bool fits = o < rs.ovf_cap;
        for (int dd = 0; dd < rs.ndest; ++dd) {
            const unsigned long long same = __ballot(fits && d == (uint32_t)dd);
            if (same && lane == (uint32_t)__ffsll((long long)same) - 1u) atomicAdd(&rs.ctr[18 + dd], (unsigned long long)__popcll(same));
        }
        if (fits) {
            *(ulonglong2 *)(rs.ovf + 2 * o) = make_ulonglong2(h, count);
            rs.ovf_dest[o] = (uint8_t)d;
        }
    }
}

__device__ __forceinline__ void skm_route_item(const KvRouteSink &rs, const uint64_t *lo, uint32_t *cur, uint64_t h, uint64_t count)
{
    if (h == UINT64_MAX) return;                 // the top hash value belongs to no band (kv_shard.hip)
    uint32_t d = 0;
    for (int b = 1; b < rs.ndest; ++b) d += h >= lo[b] ? 1u : 0u;
    const uint32_t pos = cur ? atomicAdd(&cur[d], 1u) : 0xffffffffu;
    if (pos < rs.seg_cap) *(ulonglong2 *)(rs.seg + (((uint64_t)d * rs.nwg + blockIdx.x) * rs.seg_cap + pos) * 2) = make_ulonglong2(h, count);
    else skm_route_overflow(SkmOverflowSink{rs.ctr, rs.ovf, rs.ovf_dest, rs.ovf_cap, rs.ndest}, d, h, count);
}

template <int KW, int TS, bool ORI = false, bool COMPACT = false>
__global__ __launch_bounds__(SKM_THREADS3, 6) void k_skm_route(SkmGeom sg, HashParams hp, KvRouteSink rs)
{
    __shared__ SkmTable<KW, TS> tb;
    __shared__ uint32_t cnt[TS];
    __shared__ uint32_t next_bucket;
    __shared__ uint32_t cur[SKM_ROUTE_MAX_DEST];
    __shared__ uint64_t lo[SKM_ROUTE_MAX_DEST];
    __shared__ uint32_t dl_cur, dl_b0, dl_prev;         // distinct list (sg.dl_keys != NULL: the owner will answer the scan from it, kv_skm_mex_scan_set)
    __shared__ uint32_t dl_org, dl_lim;                 // where the workgroup's entries start and how many fit: its own stretch, or (sg.dl_chunk) the chunk it drew last
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *lut = dyn, *scratch = dyn + 256;
    if (threadIdx.x < 256) lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    if (threadIdx.x < (uint32_t)rs.ndest) { cur[threadIdx.x] = 0; lo[threadIdx.x] = rs.bs * (uint64_t)threadIdx.x; }
    const int k = sg.k;
    uint64_t n_distinct = 0;
    skm_table_clear(tb);
    for (uint32_t i = threadIdx.x; i < TS; i += SKM_THREADS3) cnt[i] = 0;
    if (threadIdx.x == 0) {
        next_bucket = (uint32_t)atomicAdd(&sg.ctr[3], 1ull) * sg.bpt; dl_cur = 0; dl_b0 = 0; dl_prev = 0xffffffffu;
        dl_org = sg.dl_chunk ? 0u : blockIdx.x * sg.dl_cap_wg; dl_lim = sg.dl_chunk ? 0u : sg.dl_cap_wg;
    }
    auto dl_close = [&]() {
        if (sg.dl_keys && dl_prev != 0xffffffffu) {
            sg.dl_bstart[dl_prev] = dl_org + dl_b0;
            sg.dl_bcount[dl_prev] = dl_cur - dl_b0;              // (a stretch that ran out is reported through ctr[9]: the whole list is then dropped)
            if (sg.dl_chunk && dl_cur > dl_lim) atomicAdd(&sg.ctr[9], 1ull);
        }
    };
    // passes a bucket is combined in (see below), from its record count
    auto passes_of = [&](uint32_t b) {
        const uint32_t least = sg.passes > 1u ? sg.passes : 1u;
        const uint32_t *cnt2 = sg.cnt2 + (uint64_t)b * sg.nwg2;
        uint32_t nrec = 0;
        for (uint32_t s2 = 0; s2 < sg.nwg2; ++s2) nrec += cnt2[s2];
        // ~9.5 k-mers a record, a fifth of them distinct at sequencing coverage: 1.9 distinct k-mers per record, a pass's share under
        // 0.65 of the table.  Any number of passes (the class of a k-mer is a range of its hash, not a bit field): powers of two made the
        // average bucket of config 4 take 8 where 5 do.
        const uint32_t per_pass = (uint32_t)TS * 13u / 20u;
        const uint32_t want = (19u * nrec / 10u + per_pass - 1u) / per_pass;
        return min(64u, max(least, want));
    };
    for (uint32_t taken = 0; taken < sg.quota3; ++taken) {
        __syncthreads();
        const uint32_t b = next_bucket;
        if (b >= sg.n_buckets) break;
        __syncthreads();
        if (threadIdx.x == 0) {
            dl_close();
            if (sg.dl_keys && sg.dl_chunk) {
                // The pool (an owner whose share is too big for a stretch per workgroup with slack for the unevenness of their shares:
                // config 4's 7.9 G occurrences, 1.5 G of them distinct).  A bucket's list is one stretch, and a pass leaves at most a table's
                // worth of entries: a chunk that may not hold this bucket's is left as it is and the next one drawn.
                const uint32_t need = passes_of(b) * (uint32_t)TS;
                if (dl_cur + need > dl_lim) {
                    const unsigned long long c = atomicAdd(&sg.ctr[13], 1ull);
                    if (c < sg.dl_nchunks && need <= sg.dl_chunk) { dl_org = (uint32_t)c * sg.dl_chunk; dl_lim = sg.dl_chunk; }
                    else { dl_org = 0; dl_lim = 0; atomicAdd(&sg.ctr[9], 1ull); }          // the pool is empty (or the bucket beyond any chunk): no list
                    dl_cur = 0;
                }
            }
            dl_prev = b; dl_b0 = dl_cur;
            if ((b + 1u) & (sg.bpt - 1u)) next_bucket = b + 1u;
            else if (taken + 1 >= sg.quota3 || __hip_atomic_load(&sg.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) next_bucket = 0xffffffffu;
            else next_bucket = (uint32_t)atomicAdd(&sg.ctr[3], 1ull) * sg.bpt;
        }
        // A sample too big for the bucket geometry (255 x 4096 buckets: beyond 8.5 G k-mers a bucket holds more distinct k-mers than the
        // table has slots) is combined in passes: pass p walks the whole bucket and takes the k-mers whose hash says p -- the walk is paid
        // once per pass, the inserts, the hashes and the pairs once in all (kv_skm_mex_route sets sg.passes)
        // (how many: by the bucket's own size -- with 12-base minimizers a million buckets are far from even, a popular minimizer makes
        // its bucket several times the average -- from its record count: ~9.5 k-mers a record, a fifth of them distinct at sequencing
        // coverage, a pass's share under 0.7 of the table; sg.passes, the estimate from the average, is the least)
        const uint32_t passes = passes_of(b);
        for (uint32_t pass = 0; pass < passes; ++pass) {
        if (pass) __syncthreads();                       // (the drain of the pass before empties the table)
        skm_walk_bucket<KW, false, 0, COMPACT, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t) {
            if (!skm_cacheable<KW>(c)) return pass == 0u;                                           // (travels alone, once)
            if (passes > 1u && __umulhi(skm_slot_hash<KW>(c) * 0x9E3779B1u, passes) != pass) return false;       // (bits the slot and the probe step do not come from)
            const int slot = skm_table_insert(tb, c);
            if (slot >= 0) atomicAdd(&cnt[slot], 1u);
            return slot < 0;
        });
        __syncthreads();
        skm_for_occupied<TS>(tb.key[0], (uint16_t *)scratch, skm_wave_scratch_words(sg.sbw) * 2u, [&](uint32_t slot) {
            SkmKey<KW> c;
            c.w[0] = tb.key[0][slot];
            tb.key[0][slot] = SKM_EMPTY;
            if (KW == 2) { c.w[KW - 1] = tb.key[KW - 1][slot]; tb.key[KW - 1][slot] = SKM_EMPTY; }
            const uint32_t seen = cnt[slot];
            cnt[slot] = 0;
            n_distinct += 1;
            const uint64_t h = skm_key_hash<KW>(c, lut, hp);
            if (sg.dl_keys) {
                const unsigned long long here = __ballot(true);
                const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)__ffsll((long long)here) - 1u;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&dl_cur, (uint32_t)__popcll(here));
                base = (uint32_t)__shfl((int)base, (int)leader);
                const uint32_t at = base + (uint32_t)__popcll(here & ((1ull << lane) - 1ull));
                if (at < dl_lim) {
                    const uint64_t e = (uint64_t)dl_org + at;
                    sg.dl_keys[e * KW] = c.w[0];
                    if (KW == 2) sg.dl_keys[e * KW + (KW - 1)] = c.w[KW - 1];
                    sg.dl_hash[e] = h;
                }
            }
            skm_route_item(rs, lo, cur, h, seen);
        });
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        dl_close();
        if (sg.dl_keys && !sg.dl_chunk && dl_cur > sg.dl_cap_wg) atomicAdd(&sg.ctr[9], 1ull);
    }
    if (threadIdx.x < (uint32_t)rs.ndest)
        rs.seg_count[(uint64_t)threadIdx.x * rs.nwg + blockIdx.x] = (uint32_t)min((uint64_t)cur[threadIdx.x], rs.seg_cap);
    n_distinct = wave_sum_u64(n_distinct);
    if ((threadIdx.x & 63) == 0 && n_distinct) atomicAdd(&sg.ctr[7], (unsigned long long)n_distinct);
}

// loose records: one item of count 1 per k-mer occurrence, through the overflow list
template <int KW>
__global__ __launch_bounds__(256) void k_skm_loose_route(SkmGeom sg, HashParams hp, KvRouteSink rs)
{
    __shared__ uint32_t lut[256];
    __shared__ uint64_t lo[SKM_ROUTE_MAX_DEST];
    __shared__ uint32_t dcount[SKM_ROUTE_MAX_DEST];      // items of this workgroup per destination
    __shared__ uint32_t wg_items, wg_cur;
    __shared__ unsigned long long wg_base;
    if (sg.ctr[1] != 0) return;
    lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    if (threadIdx.x < SKM_ROUTE_MAX_DEST) { lo[threadIdx.x] = rs.bs * (uint64_t)threadIdx.x; dcount[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { wg_items = 0; wg_cur = 0; }
    __syncthreads();
    unsigned long long n = sg.ctr[0];
    if (n > sg.loose_cap) n = sg.loose_cap;
    const int k = sg.k, recw = sg.lrecw;
    // Every item here goes to the sink's overflow list.  Its slots are reserved ONCE per workgroup (the k-mers of the workgroup's records
    // are counted first: one header each) and dealt out through an LDS cursor, the per-destination totals are kept in LDS and added once
    // at the end: a returning device atomic per item -- then per wave -- made this kernel 25 and 10 ms for the 4.5 M loose k-mers of a
    // 75 M-read sample's owner, against 11 ms for the combine itself.
    {
        uint32_t mine = 0;
        for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
            mine += skm_hdr_n(sg.loose[i * (uint64_t)recw]);
        if (mine) atomicAdd(&wg_items, mine);
    }
    __syncthreads();
    if (threadIdx.x == 0) wg_base = wg_items ? atomicAdd(&rs.ctr[1], (unsigned long long)wg_items) : 0ull;
    __syncthreads();
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t *rec = sg.loose + i * (uint64_t)recw;
        uint64_t bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? rec[1 + t] : 0ull;
        const uint32_t nk = skm_hdr_n(rec[0]);
        SkmKey<KW> fw = skm_first_kmer<KW>(bw, k);
        SkmKey<KW> rc = skm_revcomp<KW>(fw, k);
        for (uint32_t j = 0; j < nk; ++j) {
            if (j) skm_roll<KW>(fw, rc, skm_base_at(bw, j + (uint32_t)k - 1u), k);
            const uint64_t h = skm_key_hash<KW>(skm_canonical<KW>(fw, rc), lut, hp);
            const unsigned long long o = wg_base + atomicAdd(&wg_cur, 1u);
            if (o >= rs.ovf_cap) continue;
            uint32_t d = 0;
            for (int b = 1; b < rs.ndest; ++b) d += h >= lo[b] ? 1u : 0u;
            if (h == UINT64_MAX) d = 0xffu;             // (the top hash value belongs to no band: the slot stays, marked -- k_route_tail skips it)
            else atomicAdd(&dcount[d], 1u);
            *(ulonglong2 *)(rs.ovf + 2 * o) = make_ulonglong2(h, 1ull);
            rs.ovf_dest[o] = (uint8_t)d;
        }
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)rs.ndest && dcount[threadIdx.x]) atomicAdd(&rs.ctr[18 + threadIdx.x], (unsigned long long)dcount[threadIdx.x]);
}

// ---- S6: novel ------------------------------------------------------------------------------------------
__device__ __forceinline__ void skm_mark(const NovelParams &p, const ReadsDev &rd, uint64_t pos, uint64_t stride)
{
    const uint64_t read = pos / stride;
    const uint64_t off = pos - read * stride;
    if ((rd.flags[read] & 1) || read < p.first_read) return;     // the scan skips these reads (kevlar/novel.py:134-139)
    const uint64_t bit = read * p.mask_stride + off;
    atomicOr(&p.mask[bit >> 5], 1u << (bit & 31));
}

template <int KW, int TS, bool ORI = false>
__global__ __launch_bounds__(SKM_THREADS3, 6) void k_skm_novel(SkmGeom sg, ReadsDev rd, NovelParams p, SkmAblSet abls)
{
    constexpr bool KNOBS = true;
    __shared__ SkmTable<KW, TS> tb;
    __shared__ uint32_t flag[TS / 32];           // bit per slot: the key is interesting
    __shared__ uint32_t rej[TS / 32];            // bit per slot: a control's abundance list rejects the key (no probe needed)
    __shared__ NovelShared ns;
    __shared__ uint32_t next_bucket, any_hit;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *lut = dyn, *scratch = dyn + 256;
    if (threadIdx.x < 256) lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    load_descs(ns, p);
    const int k = sg.k;
    if (threadIdx.x == 0) next_bucket = (uint32_t)atomicAdd(&sg.ctr[4], 1ull) * sg.bpt;
    for (uint32_t taken = 0; taken < sg.quota3; ++taken) {
        __syncthreads();
        const uint32_t b = next_bucket;
        if (b >= sg.n_buckets) break;
        skm_table_clear(tb);
        if (threadIdx.x < TS / 32) { flag[threadIdx.x] = 0; rej[threadIdx.x] = 0; }
        __syncthreads();
        if (threadIdx.x == 0) {
            if ((b + 1u) & (sg.bpt - 1u)) next_bucket = b + 1u;
            else if (taken + 1 >= sg.quota3 || __hip_atomic_load(&sg.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) next_bucket = 0xffffffffu;
            else {
                const unsigned long long ticket = atomicAdd(&sg.ctr[4], 1ull);
                next_bucket = (uint32_t)ticket * sg.bpt;
                // the count pass's early exit, for a scan that cut the batch itself: once 2 % of the buckets are done, more than
                // 4 % of their k-mers outside the tables (a batch of low coverage: nearly every k-mer distinct) raise the flag --
                // the caller then scans tile by tile, and remembers -- instead of pushing the whole batch through the loose list
                const unsigned long long done = ticket * sg.bpt;
                if (done * 50ull >= sg.n_buckets) {
                    const unsigned long long now = __hip_atomic_load(&sg.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), start = sg.ctr[6];
                    if (now > start && (now - start) * 25ull > done * sg.bucket_kmers) sg.ctr[1] = 1;
                }
            }
            any_hit = 0;
        }
        // collect the distinct k-mers
        skm_walk_bucket<KW, true, 0, false, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t) {
            const int slot = skm_cacheable<KW>(c) ? skm_table_insert(tb, c) : -1;
            return slot < 0;
        });
        // the k-mers of this bucket that a control is known to hold more than ctrl_max times (KvAbundList): they go into the
        // same table, marked; whichever of them the case sample has too is then skipped by the evaluation.  A full table
        // only costs the shortcut for that key.
        for (int a = 0; a < abls.n; ++a) {
            const uint32_t e0 = abls.bstart[a][b], en = abls.bcount[a][b];
            for (uint32_t i = threadIdx.x; i < en; i += SKM_THREADS3) {
                if ((int)min((uint32_t)abls.cnts[a][e0 + i], abls.maxv[a]) <= abls.ctrl_max) continue;
                SkmKey<KW> c;
                c.w[0] = abls.keys[a][(uint64_t)(e0 + i) * KW];
                if (KW == 2) c.w[KW - 1] = abls.keys[a][(uint64_t)(e0 + i) * KW + (KW - 1)];
                const int slot = skm_table_insert(tb, c);
                if (slot >= 0) atomicOr(&rej[slot >> 5], 1u << (slot & 31));
            }
        }
        __syncthreads();
        // evaluate each of them once
        if (!(SKM_DBG(sg) & 4u)) skm_for_occupied<TS>(tb.key[0], (uint16_t *)scratch, skm_wave_scratch_words(sg.sbw) * 2u, [&](uint32_t slot) {
            SkmKey<KW> c;
            c.w[0] = tb.key[0][slot];
            if (KW == 2) c.w[KW - 1] = tb.key[KW - 1][slot];
            const uint64_t h = skm_key_hash<KW>(c, lut, p.hp);
            if (band_pass(p, h) && novel_test_fast(ns, p, h, nullptr, 0ull)) { atomicOr(&flag[slot >> 5], 1u << (slot & 31)); any_hit = 1; }
        }, rej);            // (keys a control's list rejects are not queued: 25 M of 106 M at config 2, lanes that used to sit out their wave's hashes)
        __syncthreads();
        if (any_hit == 0 || (SKM_DBG(sg) & 8u)) continue;
        // mark every occurrence of an interesting k-mer (an occurrence whose key is absent went to the loose list)
        skm_walk_bucket<KW, true, 0, false, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t pos) {
            if (!skm_cacheable<KW>(c)) return false;
            const int slot = skm_table_find(tb, c);
            if (slot >= 0 && ((flag[slot >> 5] >> (slot & 31)) & 1u)) skm_mark(p, rd, pos, sg.stride);
            return false;
        });
    }
}

// bits[w] bit j = (tab[32 w + j] >= case_min), byte counters: the first probe of the list scan as a bit map (NovelParams::case0_bits)
__global__ __launch_bounds__(256) void k_case_bits(const uint8_t *__restrict__ tab, uint64_t size, int case_min, uint32_t *__restrict__ bits)
{
    const uint64_t n_words = (size + 31) >> 5;
    for (uint64_t w = blockIdx.x * 256ull + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * 256ull) {
        uint32_t out = 0;
        if (w * 32 + 32 <= size) {
            const uint4 a = ((const uint4 *)tab)[2 * w], b = ((const uint4 *)tab)[2 * w + 1];
            const uint32_t q[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) out |= ((int)((q[i] >> (8 * j)) & 0xffu) >= case_min ? 1u : 0u) << (4 * i + j);
        } else {
            for (uint64_t j = w * 32; j < size; ++j) out |= ((int)tab[j] >= case_min ? 1u : 0u) << (uint32_t)(j - w * 32);
        }
        bits[w] = out;
    }
}

// The scan of a batch whose count pass left a distinct list (SkmGeom::dl_*): nothing is combined and nothing is hashed again.
// Per bucket: (1) the controls' abundance-list entries that exceed ctrl_max go into one small LDS table (the k-mers no probe is
// needed for); (2) every entry of the distinct list that is not in there gets ONE probe -- table 0 of the first case sample, where
// a sequencing-error k-mer ends -- four entries per thread in flight and no barrier anywhere, and the few that pass are queued;
// (3) the queue is dealt one candidate per thread for the rest of kmer_is_interesting() (a dozen dependent probes: paid once per
// bucket, side by side, instead of once per wave and round); the interesting ones go into a second table, and (4) only if there
// are any are the bucket's records walked, to mark their occurrences.  Occurrences that missed the count pass's LDS tables are in
// the loose list already (k_skm_loose_novel evaluates them one by one).  A bucket's list has at most as many entries as the count
// kernel's LDS table has slots.
template <int KW, int TSM, bool KNOBS, bool ORI = false>
__global__ __launch_bounds__(SKM_THREADS3, 6) void k_skm_novel_list(SkmGeom sg, ReadsDev rd, NovelParams p, SkmAblSet abls)
{
#if defined(SKM_LIST_E)
    constexpr uint32_t E = SKM_LIST_E;
#else
    constexpr uint32_t E = 4;                    // entries a thread has in flight (2 / 4 / 6 measured: 2.64 / 2.6-2.8 / 2.52 ms, within the spread between runs)
#endif
    __shared__ SkmTable<KW, TSM> rtb;            // rejected by a control's list
    __shared__ SkmTable<KW, TSM> itb;            // interesting
    __shared__ NovelShared ns;
    __shared__ uint16_t cand[SKM_LIST_MAX];      // entries (relative to the bucket's first) that passed the first probe
    __shared__ uint32_t next_bucket, n_int, n_cand;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *scratch = dyn;
    load_descs(ns, p);
    skm_table_clear(itb);
    if (threadIdx.x == 0) { next_bucket = (uint32_t)atomicAdd(&sg.ctr[4], 1ull) * sg.bpt; n_int = 0; n_cand = 0; }
    auto mark_pass = [&](uint32_t b) {
        skm_walk_bucket<KW, true, 0, false, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t pos) {
            if (skm_cacheable<KW>(c) && skm_table_find(itb, c) >= 0) skm_mark(p, rd, pos, sg.stride);
            return false;
        });
        __syncthreads();
        skm_table_clear(itb);
        if (threadIdx.x == 0) n_int = 0;
        __syncthreads();
    };
    for (uint32_t taken = 0; taken < sg.quota3; ++taken) {
        __syncthreads();
        const uint32_t b = next_bucket;
        if (b >= sg.n_buckets) break;
        skm_table_clear(rtb);
        __syncthreads();
        if (threadIdx.x == 0) {
            if ((b + 1u) & (sg.bpt - 1u)) next_bucket = b + 1u;
            else if (taken + 1 >= sg.quota3 || __hip_atomic_load(&sg.ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) next_bucket = 0xffffffffu;
            else next_bucket = (uint32_t)atomicAdd(&sg.ctr[4], 1ull) * sg.bpt;
            n_cand = 0;
        }
        // the first list entries are requested before the controls' lists are worked in: they arrive meanwhile
        const uint32_t e0 = sg.dl_bstart[b], en = min(sg.dl_bcount[b], SKM_LIST_MAX);
        SkmKey<KW> c[E];
        uint64_t h[E];
        auto request = [&](uint32_t base) {
#pragma unroll
            for (uint32_t u = 0; u < E; ++u) {
                const uint32_t i = base + u * SKM_THREADS3 + threadIdx.x;
                const bool in = i < en;
                c[u].w[0] = in ? sg.dl_keys[(uint64_t)(e0 + i) * KW] : SKM_EMPTY;
                if (KW == 2) c[u].w[KW - 1] = in ? sg.dl_keys[(uint64_t)(e0 + i) * KW + (KW - 1)] : SKM_EMPTY;
                h[u] = in ? sg.dl_hash[e0 + i] : 0ull;
            }
        };
        request(0);
        for (int a = 0; a < abls.n; ++a) {
            const uint32_t a0 = abls.bstart[a][b], an = abls.bcount[a][b];
            for (uint32_t i = threadIdx.x; i < an; i += SKM_THREADS3) {
                if ((int)min((uint32_t)abls.cnts[a][a0 + i], abls.maxv[a]) <= abls.ctrl_max) continue;
                SkmKey<KW> c;
                c.w[0] = abls.keys[a][(uint64_t)(a0 + i) * KW];
                if (KW == 2) c.w[KW - 1] = abls.keys[a][(uint64_t)(a0 + i) * KW + (KW - 1)];
                (void)skm_table_insert(rtb, c);              // (a full table only costs the shortcut for that key)
            }
        }
        __syncthreads();
        for (uint32_t base = 0; base < ((SKM_DBG(sg) & 4u) ? 0u : en); base += E * SKM_THREADS3) {
            if (base) request(base);
            bool live[E];
            uint32_t v[E];
#pragma unroll
            for (uint32_t u = 0; u < E; ++u) {
                live[u] = base + u * SKM_THREADS3 + threadIdx.x < en && skm_table_find(rtb, c[u]) < 0 && band_pass(p, h[u]);
                if (p.case0_bits) {
                    const uint64_t bin = fastmod(h[u], ns.d[0].size, ns.d[0].magic);
                    const __attribute__((address_space(1))) uint32_t *bits = (const __attribute__((address_space(1))) uint32_t *)p.case0_bits;
                    v[u] = live[u] && ((bits[bin >> 5] >> (uint32_t)(bin & 31u)) & 1u) ? (uint32_t)p.case_min : 0u;
                } else {
                    v[u] = live[u] ? probe(ns, 0, 0, h[u]) : 0u;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < E; ++u)
                if (live[u] && (int)v[u] >= p.case_min) cand[atomicAdd(&n_cand, 1u)] = (uint16_t)(base + u * SKM_THREADS3 + threadIdx.x);
        }
        __syncthreads();
        const uint32_t nc = n_cand;
        for (uint32_t j0 = 0; j0 < nc; j0 += SKM_THREADS3) {
            const uint32_t j = j0 + threadIdx.x;
            if (j < nc) {
                const uint32_t i = cand[j];
                SkmKey<KW> c;
                c.w[0] = sg.dl_keys[(uint64_t)(e0 + i) * KW];
                if (KW == 2) c.w[KW - 1] = sg.dl_keys[(uint64_t)(e0 + i) * KW + (KW - 1)];
                const uint64_t h = sg.dl_hash[e0 + i];
                if (novel_test_wide(ns, p, h)) {
                    if (skm_table_insert(itb, c) < 0) sg.ctr[1] = 1;          // (cannot happen below half full; the caller then redoes the scan the other way)
                    atomicAdd(&n_int, 1u);
                    if (p.ab_keys) (void)ab_claim(p, h, p.ncase + p.nctrl);      // a place for its abundances (k_ab_fill), for the kernel that reports the hits
                }
            }
            __syncthreads();
            // a bucket with more interesting k-mers than a quarter of the table (a case sample without controls to speak of)
            // marks in instalments
            if (n_int > TSM / 4 && j0 + SKM_THREADS3 < nc) mark_pass(b);
        }
        if (n_int != 0 && !(SKM_DBG(sg) & 8u)) mark_pass(b);
    }
}

// the abundances of the k-mers k_skm_novel_list claimed a slot for (a dozen probes each, side by side, off the scan's critical path:
// inside the scan they were a chain of round trips in front of a workgroup barrier, 0.25 ms)
__global__ __launch_bounds__(256) void k_ab_fill(NovelParams p)
{
    __shared__ NovelShared ns;
    load_descs(ns, p);
    __syncthreads();
    const int S = p.ncase + p.nctrl;
    // the slots that were claimed are on a list (ab_claim) -- some 10^5 of the table's 8 M at config 2, whose walk was 0.11 ms; a list
    // that ran out of room sends the kernel over the whole table as before
    const unsigned int listed = p.ab_list ? *p.ab_count : 0xffffffffu;
    if (listed <= p.ab_list_cap) {
        for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < listed; i += (uint64_t)gridDim.x * 256ull) {
            const uint64_t slot = p.ab_list[i];
            hit_abundances(ns, p, (uint64_t)p.ab_keys[slot], p.ab_vals + slot * (uint64_t)S);
        }
        return;
    }
    for (uint64_t slot = blockIdx.x * 256ull + threadIdx.x; slot <= p.ab_mask; slot += (uint64_t)gridDim.x * 256ull) {
        const unsigned long long h = p.ab_keys[slot];
        if (h != 0ull) hit_abundances(ns, p, (uint64_t)h, p.ab_vals + slot * (uint64_t)S);
    }
}

template <int KW>
__global__ __launch_bounds__(256) void k_skm_loose_novel(SkmGeom sg, ReadsDev rd, NovelParams p)
{
    __shared__ uint32_t lut[256];
    __shared__ NovelShared ns;
    lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    load_descs(ns, p);
    __syncthreads();
    unsigned long long n = sg.ctr[0];
    if (n > sg.loose_cap) n = sg.loose_cap;
    const int k = sg.k, recw = sg.lrecw;
    const uint32_t lane = threadIdx.x & 63u;
    auto test_one = [&](const SkmKey<KW> &fw, bool live, uint64_t pos) {
        if (!live) return;
        const uint64_t h = skm_key_hash<KW>(skm_canonical<KW>(fw, skm_revcomp<KW>(fw, k)), lut, p.hp);
        if (band_pass(p, h) && novel_test_fast(ns, p, h, nullptr, 0ull)) skm_mark(p, rd, pos, sg.stride);
    };
    // as k_skm_loose_count deals the list: a few records to every wave, a one-k-mer record a lane, a longer record spread over the
    // lanes k-mer by k-mer (a thread per record rolled through up to ncap k-mers, each a chain of table probes: 0.167 ms for the
    // handful of records a config-2 scan finds here)
    const uint64_t total_waves = (uint64_t)gridDim.x * (blockDim.x >> 6), wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t per = (uint32_t)min((unsigned long long)64, max((unsigned long long)1, (n + total_waves - 1) / total_waves));
    for (uint64_t i0 = wave_id * per; i0 < n; i0 += total_waves * per) {
        const uint64_t i = i0 + lane;
        const bool have = lane < per && i < n;
        const uint64_t *rec = sg.loose + (have ? i : 0) * (uint64_t)recw;
        uint64_t bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = (have && t < sg.nbw) ? rec[1 + t] : 0ull;
        const uint64_t hdr = have ? rec[0] : 0ull;
        const uint32_t nk = have ? skm_hdr_n(hdr) : 0u;
        test_one(skm_first_kmer<KW>(bw, k), nk == 1, skm_hdr_pos_of(hdr, 0u));
        unsigned long long longer = __ballot(nk > 1);
        while (longer) {
            const uint32_t src = (uint32_t)__ffsll((long long)longer) - 1u;
            longer &= longer - 1ull;
            const uint64_t b0 = skm_shfl64(bw[0], src), b1 = skm_shfl64(bw[1], src), b2 = skm_shfl64(bw[2], src), h_src = skm_shfl64(hdr, src);
            const uint32_t nk_src = (uint32_t)__shfl((int)nk, (int)src);
            for (uint32_t j0 = 0; j0 < nk_src; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool live = j < nk_src;
                test_one(skm_kmer_at<KW>(b0, b1, b2, live ? j : 0u, k), live, skm_hdr_pos_of(h_src, live ? j : 0u));
            }
        }
    }
}

// ---- the scan answered by the owner of the minimizer buckets (minimizer-sharded exchange, kv_skm_mex_scan_set) ----------------
// After kv_skm_mex_route(keep_scan) this rank holds every occurrence of the k-mers of its buckets -- from whichever shard, with the
// global (read, offset) of each in the record headers -- and key + hash of every distinct one (the distinct list k_skm_route left).
// Once the band owners' verdicts have been gathered into the set of interesting hashes, the hits are these: per bucket, the list
// entries that are members go into a small LDS table with their slot of the set; only if there are any is the bucket walked, and every
// occurrence of a member leaves as (read << 16 | offset, the S abundances the set carries) -- the form kv_hits_from_tagged sorts.
// Hits are staged in LDS and appended with one update of the device-wide counter per flush (that counter takes ~90 updates per
// microsecond however they are issued); a bucket with more hits than the stage holds appends the rest one by one.
struct SetHitSink {
    unsigned long long *tags;
    uint8_t *abund;
    unsigned long long *count;
    uint64_t cap;
};
__device__ __forceinline__ void set_hit_store(const NovelParams &p, const SetHitSink &out, uint64_t at, unsigned long long tag, uint64_t slot)
{
    if (at >= out.cap) return;
    const int S = p.ncase + p.nctrl;
    out.tags[at] = tag;
    for (int c = 0; c < S; ++c) out.abund[at * (uint64_t)S + c] = p.set_abund[slot * (uint64_t)S + c];
}
#define SKM_HIT_STAGE 1024u
template <int KW, int TSM, bool ORI = false>
__global__ __launch_bounds__(SKM_THREADS3, 6) void k_skm_set_hits(SkmGeom sg, NovelParams p, SetHitSink out)
{
    __shared__ SkmTable<KW, TSM> itb;            // the bucket's members of the set
    __shared__ uint32_t islot[TSM];              // their slots of the set
    __shared__ unsigned long long htag[SKM_HIT_STAGE];
    __shared__ uint32_t hslot[SKM_HIT_STAGE];
    __shared__ uint32_t next_bucket, n_int, n_hit;
    __shared__ unsigned long long flush_base;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    uint32_t *scratch = dyn;
    skm_table_clear(itb);
    if (threadIdx.x == 0) { next_bucket = (uint32_t)atomicAdd(&sg.ctr[4], 1ull) * sg.bpt; n_int = 0; n_hit = 0; }
    for (uint32_t taken = 0; taken < sg.quota3; ++taken) {
        __syncthreads();
        const uint32_t b = next_bucket;
        if (b >= sg.n_buckets) break;
        __syncthreads();
        if (threadIdx.x == 0) {
            if ((b + 1u) & (sg.bpt - 1u)) next_bucket = b + 1u;
            else if (taken + 1 >= sg.quota3) next_bucket = 0xffffffffu;
            else next_bucket = (uint32_t)atomicAdd(&sg.ctr[4], 1ull) * sg.bpt;
        }
        const uint32_t e0 = sg.dl_bstart[b], en = sg.dl_bcount[b];
        for (uint32_t i = threadIdx.x; i < en; i += SKM_THREADS3) {
            const uint64_t h = sg.dl_hash[e0 + i];
            const uint64_t slot = set_find(p, h);
            if (slot == KV_SET_NONE) continue;
            SkmKey<KW> c;
            c.w[0] = sg.dl_keys[(uint64_t)(e0 + i) * KW];
            if (KW == 2) c.w[KW - 1] = sg.dl_keys[(uint64_t)(e0 + i) * KW + (KW - 1)];
            const int at = skm_table_insert(itb, c);
            if (at < 0) { sg.ctr[1] = 1; continue; }            // more members than the table takes: the caller scans the other way
            islot[at] = (uint32_t)slot;
            atomicAdd(&n_int, 1u);
        }
        __syncthreads();
        if (n_int == 0) continue;
        skm_walk_bucket<KW, true, 0, false, ORI>(sg, b, scratch, [&](const SkmKey<KW> &c, const SkmKey<KW> &, uint64_t pos) {
            if (!skm_cacheable<KW>(c)) return false;
            const int at = skm_table_find(itb, c);
            if (at < 0) return false;
            const uint64_t read = pos / sg.stride;
            const unsigned long long tag = ((unsigned long long)read << 16) | (unsigned long long)(pos - read * sg.stride);
            const uint32_t i = atomicAdd(&n_hit, 1u);
            if (i < SKM_HIT_STAGE) { htag[i] = tag; hslot[i] = islot[at]; }
            else set_hit_store(p, out, atomicAdd(out.count, 1ull), tag, islot[at]);
            return false;
        });
        __syncthreads();
        const uint32_t staged = min(n_hit, SKM_HIT_STAGE);
        if (threadIdx.x == 0) flush_base = atomicAdd(out.count, (unsigned long long)staged);
        skm_table_clear(itb);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < staged; i += SKM_THREADS3) set_hit_store(p, out, flush_base + i, htag[i], hslot[i]);
        if (threadIdx.x == 0) { n_int = 0; n_hit = 0; }
    }
}

// the occurrences that travelled alone (records of the loose list: one k-mer each from the combine, whole records from the split)
template <int KW>
__global__ __launch_bounds__(256) void k_skm_loose_set_hits(SkmGeom sg, NovelParams p, SetHitSink out)
{
    __shared__ uint32_t lut[256];
    lut[threadIdx.x] = skm_ascii4(threadIdx.x);
    __syncthreads();
    unsigned long long n = sg.ctr[0];
    if (n > sg.loose_cap) n = sg.loose_cap;
    const int k = sg.k, recw = sg.lrecw;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t *rec = sg.loose + i * (uint64_t)recw;
        uint64_t bw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bw[t] = t < sg.nbw ? rec[1 + t] : 0ull;
        const uint32_t nk = skm_hdr_n(rec[0]);
        SkmKey<KW> fw = skm_first_kmer<KW>(bw, k);
        SkmKey<KW> rc = skm_revcomp<KW>(fw, k);
        for (uint32_t j = 0; j < nk; ++j) {
            if (j) skm_roll<KW>(fw, rc, skm_base_at(bw, j + (uint32_t)k - 1u), k);
            const uint64_t slot = set_find(p, skm_key_hash<KW>(skm_canonical<KW>(fw, rc), lut, p.hp));
            if (slot == KV_SET_NONE) continue;
            const uint64_t pos = skm_hdr_pos_of(rec[0], j), read = pos / sg.stride;
            set_hit_store(p, out, atomicAdd(out.count, 1ull), ((unsigned long long)read << 16) | (unsigned long long)(pos - read * sg.stride), slot);
        }
    }
}

__global__ void k_skm_forward_flag(const unsigned long long *skm_ctr, unsigned long long *bin_ctr)
{
    if (skm_ctr[1] != 0) bin_ctr[1] = 1;
}

// hits per tile = set bits of the tile's range of the mask (what k_novel_mark counts while it marks)
__global__ __launch_bounds__(64) void k_tile_hits(ReadsDev rd, NovelParams p)
{
    const TileDesc td = rd.tile[blockIdx.x];
    uint64_t b0, b1;
    if (td.seg) {
        b0 = (uint64_t)td.first * p.mask_stride + td.seg_start;
        b1 = min(b0 + (uint64_t)KV_SEG_BASES, ((uint64_t)td.first + 1) * p.mask_stride);
    } else {
        b0 = (uint64_t)td.first * p.mask_stride;
        b1 = ((uint64_t)td.first + td.count) * p.mask_stride;
    }
    uint64_t n = 0;
    for (uint64_t w = (b0 >> 5) + threadIdx.x; w < ((b1 + 31) >> 5); w += 64) {
        uint32_t bits = p.mask[w];
        const uint64_t wlo = w << 5;
        if (wlo < b0) bits &= ~0u << (uint32_t)(b0 - wlo);
        if (wlo + 32 > b1) bits &= b1 > wlo ? (~0u >> (uint32_t)(wlo + 32 - b1)) : 0u;
        n += (uint32_t)__popc(bits);
    }
    n = wave_sum_u64(n);
    if (threadIdx.x == 0) p.tile_count[blockIdx.x] = (uint32_t)n;
}

// ---- host ----------------------------------------------------------------------------------------------
// The bucketed form of the last batch consumed on a stream stays in that stream's arena, so a scan of the same
// batch (the case sample: count, then novel) does not cut its reads again.
struct SkmIndex {
    KvArena arena;
    uint64_t reads_uid = 0;
    int k = 0;
    bool valid = false;
    double distinct_hint = 0.0;     // distinct / all k-mers of the last batch routed on this stream (kv_skm_route_distinct)
    SkmGeom g;
    // distinct list of the batch (kv_sketch::scan_hint): key + hash of every distinct k-mer the count pass drained, bucket by bucket
    KvArena dl;
    size_t dl_refused = SIZE_MAX;    // the smallest list kv_skm_mex_route asked for and did not get since the arena was last empty (a refused
                                     // allocation of tens of gigabytes takes a third of a second, and the arena is given up for the attempt)
    uint64_t *dl_keys = nullptr, *dl_hash = nullptr;
    uint32_t *dl_bstart = nullptr, *dl_bcount = nullptr;
    uint32_t dl_cap_wg = 0;
    bool dl_valid = false;
    bool mex_scan_ready = false;     // the arena holds an owner's combined buckets with their distinct list (kv_skm_mex_route, keep_scan): kv_skm_mex_scan_set
    uint64_t builds = 0;             // batches bucketed on this stream so far
    KvArena bits;                    // NovelParams::case0_bits of the scan in flight
    std::mutex mu;
};
std::map<hipStream_t, SkmIndex> g_skm;
std::mutex g_skm_mu;
int g_skm_pref_k = 0;                    // bucket count of the last build that chose its own (guarded by g_skm_mu), see skm_build
uint64_t g_skm_pref_nfine = 0;
double g_skm_last_distinct = 0.0;        // distinct share of the batch counted last on any stream (guarded by g_skm_mu): a scan
                                          // that has to bucket its batch itself sizes the buckets by it

SkmIndex &skm_index_for(hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_skm_mu);
    return g_skm[kv_stream_key(st)];
}
}
void kv_skm_scratch_release()
{
    std::lock_guard<std::mutex> lk(g_skm_mu);
    for (auto &kv : g_skm) {
        SkmIndex &idx = kv.second;
        std::lock_guard<std::mutex> ilk(idx.mu);
        idx.valid = false; idx.dl_valid = false; idx.mex_scan_ready = false;
        idx.dl_keys = nullptr; idx.dl_hash = nullptr; idx.dl_bstart = nullptr; idx.dl_bcount = nullptr;
        idx.arena.release(); idx.dl.release(); idx.bits.release();
    }
}
namespace {

inline uint32_t skm_nwg3(const SkmGeom &g)
{
    // persistent workgroups of the bucket kernels: three per CU fill its LDS (KV_SKM_WG3_PER_CU=2 leaves a third of it -- and of the wave
    // slots -- to whatever another stream has queued: the experiment behind DESIGN.md section 4.1, "samples on separate streams")
    static const uint32_t per_cu = [] { const char *e = kv_knob("KV_SKM_WG3_PER_CU"); const int v = e ? atoi(e) : 3; return (uint32_t)(v >= 1 && v <= 3 ? v : 3); }();
    return (uint32_t)std::min<uint64_t>((g.n_buckets + g.bpt - 1) / g.bpt, per_cu * (uint32_t)kv_device_cus());
}

static uint32_t skm_default_bpt()
{
    if (const char *e = kv_knob("KV_SKM_BPT")) { const int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) return (uint32_t)v; }
    return SKM_BUCKETS_PER_TICKET;
}
// Buckets per ticket of the work counter.  A ticket costs a returning atomic on a word every workgroup asks for -- ~10 per microsecond
// device-wide, measured: a sample's 64 k buckets one per ticket took k_skm_count from 3.1 to 6.9 ms, two per ticket to 3.7 -- and a big
// ticket leaves a tail (16: +0.08 ms, 32: +0.13).  8 for a whole sample; fewer where the buckets are few (a rank's share of the exchange:
// 7 936 buckets for 768 workgroups -- tickets of 8 give a quarter of them two and the rest one), down to 2.  KV_SKM_BPT overrides.
static void skm_pick_bpt(SkmGeom &g)
{
    g.bpt = SKM_BUCKETS_PER_TICKET;
    while (g.bpt > 2u && g.n_buckets / g.bpt < 4u * 768u) g.bpt /= 2u;
    if (kv_knob("KV_SKM_BPT")) g.bpt = skm_default_bpt();
}

inline uint32_t pow2_ceil(uint64_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

int skm_minimizer_len(int k) { return k >= 24 ? 12 : k / 2; }

// reads per wave and chunk size of the wave kernel for batches of equal-length reads (0: use the tile kernel).  One packed
// word and one chunk per lane, and about as many runs as lanes: a run per (w + 1) / 2 k-mers, one more per read.
// threads per workgroup of the wave kernel (KV_SKM_S1_THREADS=512|1024)
static thread_local uint32_t tl_s1_threads = 0;       // a caller's choice for the launch it is about to make (kv_skm_mex_emit); the variable in the environment wins
static uint32_t skm_wave_threads()
{
    if (const char *e = kv_knob("KV_SKM_S1_THREADS")) return atoi(e) == 512 ? 512u : 1024u;
    return tl_s1_threads ? tl_s1_threads : SKM_S1_WAVE_THREADS_DEFAULT;
}

static uint32_t skm_wave_plan(const SkmGeom &g, const kv_reads *reads, int *ch)
{
    const uint32_t L = reads->uni_len;
    if (!L || L < (uint32_t)g.k || g.w <= 8) return 0;
    if (const char *e = kv_knob("KV_SKM_S1")) if (!strcmp(e, "tile")) return 0;
    const uint32_t wpr = (L + 15) / 16, nk = L - (uint32_t)g.k + 1;
    const double runs = 1.0 + (nk - 1) * 2.0 / (g.w + 1) + (double)nk / g.ncap * 0.5;
    uint32_t best = 0;
    for (int c : {16, 8}) {
        if (g.w <= c) continue;
        const uint32_t cpr = (nk + c - 1) / c;
        uint32_t R = std::min<uint32_t>(64 / wpr, 64 / cpr);
        R = std::min<uint32_t>(R, (uint32_t)(58.0 / runs));
        if (const char *e = kv_knob("KV_SKM_R")) R = std::min<uint32_t>(std::min<uint32_t>(64 / wpr, 64 / cpr), (uint32_t)atoi(e));
        if (R >= best && R > 0) { best = R; *ch = c; }       // equal R: the smaller chunk keeps more lanes busy
    }
    if (const char *e = kv_knob("KV_SKM_CH")) { const int c = atoi(e); if ((c == 8 || c == 16) && g.w > c) { *ch = c; best = std::min<uint32_t>(best, 64 / ((nk + c - 1) / c)); } }
    if (best < 2) return 0;
    const uint32_t threads = skm_wave_threads();
    // three workgroups of 512 per CU, or one of 1024 -- or the tile kernel
    if ((size_t)skm_wave_slice_words(best, L, nk, *ch) * 4 * (threads / 64) + 2048 > (threads == 1024 ? 160000u : 54000u)) return 0;
    return best;
}

// the lane-per-read kernel takes batches of equal-length reads with w = 20 or 40 (k = 31, 51: m = 12) and up to 256 bases;
// KV_SKM_S1=wave|tile keeps the older kernels
// workgroups of the lane-per-read cut a CU holds at once, by their LDS (0: not even two -- other kernels cut such reads)
static uint32_t skm_lane_wgs_per_cu(uint32_t L)
{
    const size_t need = (size_t)skm_lane_slice_words((L + 15u) / 16u) * 4 * (SKM_LANE_THREADS / 64) + 1200;
    const char *e = kv_knob("KV_SKM_LANE_MAXWG");                // (A/B: 3 keeps the cut to the lengths that fit three times)
    const uint32_t floor_wgs = e ? (uint32_t)std::max(2, std::min(3, atoi(e))) : 2u;
    if (need <= 160000u / (SKM_LANE_WAVES / 2)) return 3u;
    return (floor_wgs <= 2u && need <= 160000u / 2u) ? 2u : 0u;
}
// (by the read length alone: kv_mex_plan_short, which sees no reads, asks this way)
static bool skm_lane_fits_len(const SkmGeom &g, uint32_t L)
{
    if (!L || L < (uint32_t)g.k || g.m != 12 || (g.w != SKM_LANE_B && g.w != 2 * SKM_LANE_B) || L > 256u) return false;
    const char *e1 = kv_knob("KV_SKM_S1");
    if (e1 && strcmp(e1, "lane") != 0) return false;
    // (w = 40, k = 51: two arrays of suffix minima; that instance is compiled for 4 waves per SIMD and skm_build starts two workgroups
    // per CU for it -- at six waves it spilled and measured slower than the wave kernel, 5.5 against 4.1 ms per step of config 5)
    if (g.w != SKM_LANE_B && e1 && !strcmp(e1, "lane6")) return false;
    if (g.dbg & ~4096u) return false;                           // the phase switches of the dissection scripts live in the older kernels
    // three workgroups per CU for reads of up to 112 bases (seven packed words and their reverse complement per lane), two up to 224
    return skm_lane_wgs_per_cu(L) >= 2u;
}
static bool skm_lane_fits(const SkmGeom &g, const kv_reads *reads) { return skm_lane_fits_len(g, reads->uni_len); }

void skm_launch_emit(const SkmGeom &g, const kv_reads *reads, hipStream_t st)
{
    KvProfScope prof("k_skm_emit");
    int ch = 16;
    if (skm_lane_fits(g, reads)) {
        const uint32_t wpr = (reads->uni_len + 15u) / 16u;
        const size_t lds = (size_t)skm_lane_slice_words(wpr) * 4 * (SKM_LANE_THREADS / 64);
        const uint64_t n_groups = (reads->n_reads + 63) / 64;
        uint32_t flush_blocks = 2;
        if (const char *e = kv_knob("KV_SKM_LANE_FLUSH")) flush_blocks = std::max(1, atoi(e));
        void (*kernel)(ReadsDev, SkmGeom, uint32_t, uint32_t);
        if (g.w == SKM_LANE_B) kernel = wpr <= 8 ? k_skm_emit_lane<1, 8> : k_skm_emit_lane<1, 16>;
        else kernel = wpr <= 8 ? k_skm_emit_lane<2, 8> : k_skm_emit_lane<2, 16>;
        kv_ensure_dynamic_lds((const void *)kernel, lds);
        hipLaunchKernelGGL(kernel, dim3(g.nwg1), dim3(SKM_LANE_THREADS), lds, st, reads_dev(reads), g, (uint32_t)n_groups, flush_blocks);
        return;
    }
    if (const uint32_t R = skm_wave_plan(g, reads, &ch)) {
        const uint32_t L = reads->uni_len, nk = L - (uint32_t)g.k + 1;
        const uint32_t threads = skm_wave_threads(), nwaves = threads / 64;
        const size_t lds = (size_t)skm_wave_slice_words(R, L, nk, ch) * 4 * nwaves;
        const uint64_t n_mt = (reads->n_reads + R - 1) / R;
        // the same share of the batch per workgroup as the tile kernel's quota, dealt to its waves
        // a ticket is SKM_MT_PER_TICKET groups for a whole sample; small batches (a shard of a sample, a test) get smaller
        // tickets so that every wave of the grid still finds two
        const uint64_t waves = (uint64_t)g.nwg1 * nwaves;
        const uint32_t per_ticket = (uint32_t)std::max<uint64_t>(4, std::min<uint64_t>(SKM_MT_PER_TICKET, n_mt / (2 * waves)));
        const uint64_t per_wave = (uint64_t)g.quota1 * reads->uni_per_tile / R / nwaves + per_ticket;
        const uint32_t quota_mt = (uint32_t)std::min<uint64_t>(kv_round_up(per_wave, per_ticket), 0xfffffff0ull);
        void (*kernel)(ReadsDev, SkmGeom, uint32_t, uint32_t, uint32_t, uint32_t);
        if (threads == 1024)
            kernel = ch == 16 ? (g.dbg ? k_skm_emit_wave<16, true, 1024> : k_skm_emit_wave<16, false, 1024>) : (g.dbg ? k_skm_emit_wave<8, true, 1024> : k_skm_emit_wave<8, false, 1024>);
        else
            kernel = ch == 16 ? (g.dbg ? k_skm_emit_wave<16, true, 512> : k_skm_emit_wave<16, false, 512>) : (g.dbg ? k_skm_emit_wave<8, true, 512> : k_skm_emit_wave<8, false, 512>);
        kv_ensure_dynamic_lds((const void *)kernel, lds);
        hipLaunchKernelGGL(kernel, dim3(g.nwg1), dim3(threads), lds, st, reads_dev(reads), g, R, (uint32_t)n_mt, quota_mt, per_ticket);
        return;
    }
    const size_t lds = ((size_t)g.np_max / 16 + KV_TILE_MAX_READS + 8 + (size_t)g.np_max + 96) * 4 + (((size_t)g.np_max + 96 + 1) & ~(size_t)1) * 2;
    if (g.w > 16) {
        kv_ensure_dynamic_lds((const void *)k_skm_emit<16>, lds);
        hipLaunchKernelGGL(k_skm_emit<16>, dim3(g.nwg1), dim3(SKM_THREADS1), lds, st, reads_dev(reads), reads->n_tiles, g);
    } else {
        kv_ensure_dynamic_lds((const void *)k_skm_emit<8>, lds);
        hipLaunchKernelGGL(k_skm_emit<8>, dim3(g.nwg1), dim3(SKM_THREADS1), lds, st, reads_dev(reads), reads->n_tiles, g);
    }
}

void skm_launch_split(const SkmGeom &g, hipStream_t st)
{
    KvProfScope prof("k_skm_split");
    // sorted scatter while a chunk holds ~2 records per fine bucket or more (KV_SKM_S2=plain|sorted overrides)
    const char *s2 = kv_knob("KV_SKM_S2");
    const bool sorted = s2 ? strcmp(s2, "sorted") == 0 && g.F2 <= SKM_S2_MAXF : g.F2 <= SKM_S2_MAXF;
    if (sorted) {
        const size_t lds = (size_t)SKM_S2_CHUNK * g.recw * 8;
        if (g.recw == 2) {
            kv_ensure_dynamic_lds((const void *)k_skm_split_sorted<2>, lds);
            hipLaunchKernelGGL(k_skm_split_sorted<2>, dim3(g.nwg2, g.C1), dim3(SKM_THREADS2), lds, st, g);
        } else if (g.recw == 3) {
            kv_ensure_dynamic_lds((const void *)k_skm_split_sorted<3>, lds);
            hipLaunchKernelGGL(k_skm_split_sorted<3>, dim3(g.nwg2, g.C1), dim3(SKM_THREADS2), lds, st, g);
        } else {
            kv_ensure_dynamic_lds((const void *)k_skm_split_sorted<4>, lds);
            hipLaunchKernelGGL(k_skm_split_sorted<4>, dim3(g.nwg2, g.C1), dim3(SKM_THREADS2), lds, st, g);
        }
    } else {
        if (g.recw == 2) hipLaunchKernelGGL(k_skm_split<true>, dim3(g.nwg2, g.C1), dim3(SKM_THREADS2), 0, st, g);
        else hipLaunchKernelGGL(k_skm_split<false>, dim3(g.nwg2, g.C1), dim3(SKM_THREADS2), 0, st, g);
    }
}

// Exclusive prefix of n counts, off[n] = total -- n is the number of exchange segments, a couple of hundred thousand: chunks of 1024
// scanned side by side (k_mex_scan_chunks), the chunk totals by one workgroup (k_mex_scan_parts), the bases added (k_mex_scan_add).
// (One workgroup walking all of it took 0.15 ms, a third of a rank's packing at N = 8.)  `part` holds n / 1024 + 2 values.
__global__ __launch_bounds__(1024) void k_mex_scan_chunks(const uint32_t *cnt, uint64_t n, uint32_t cap1, uint64_t *off, uint64_t *part)
{
    __shared__ uint64_t wsum[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * 1024u + threadIdx.x;
    const uint64_t v = i < n ? (uint64_t)min(cnt[i], cap1) : 0ull;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d);
        if (lane >= (uint32_t)d) incl += (uint64_t)lo | ((uint64_t)hi << 32);
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
    if (i < n) off[i] = before + incl - v;
    if (threadIdx.x == 1023) part[blockIdx.x] = before + incl;
}

__global__ __launch_bounds__(1024) void k_mex_scan_parts(uint64_t *part, uint64_t n_parts)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry_sh;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint64_t base = 0; base < n_parts; base += 1024) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < n_parts ? part[i] : 0ull;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d);
            if (lane >= (uint32_t)d) incl += (uint64_t)lo | ((uint64_t)hi << 32);
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = carry_sh;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (i < n_parts) part[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_sh = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) part[n_parts] = carry_sh;
}

__global__ __launch_bounds__(1024) void k_mex_scan_add(uint64_t *off, uint64_t n, const uint64_t *part, uint64_t n_parts)
{
    const uint64_t i = (uint64_t)blockIdx.x * 1024u + threadIdx.x;
    if (i < n) off[i] += part[blockIdx.x];
    if (i == 0) off[n] = part[n_parts];
}

static inline size_t mex_scan_bytes(uint64_t n) { return kv_round_up((n + 1 + n / 1024 + 3) * 8, 256); }     // off[n + 1] and the chunk totals behind it
static void mex_scan_launch(const uint32_t *cnt, uint64_t n, uint32_t cap1, uint64_t *off, hipStream_t st)
{
    const uint64_t n_parts = (n + 1023) / 1024;
    uint64_t *part = off + n + 1;
    if (n_parts) hipLaunchKernelGGL(k_mex_scan_chunks, dim3((unsigned)n_parts), dim3(1024), 0, st, cnt, n, cap1, off, part);
    hipLaunchKernelGGL(k_mex_scan_parts, dim3(1), dim3(1024), 0, st, part, n_parts);
    hipLaunchKernelGGL(k_mex_scan_add, dim3((unsigned)std::max<uint64_t>(n_parts, 1)), dim3(1024), 0, st, off, n, (const uint64_t *)part, n_parts);
}

// the filled part of every segment, one after the other: a wavefront per segment (a segment of a shard at N = 8 holds ~40 records)
__global__ __launch_bounds__(256) void k_mex_gather(const uint64_t *seg, const uint32_t *cnt, const uint64_t *off, uint64_t n_segments, uint32_t cap1,
                                                    uint32_t recw, uint64_t *out, uint64_t out_cap_records)
{
    if (off[n_segments] > out_cap_records) return;            // does not fit: nothing is written, the caller sees the total and packs into a bigger buffer
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t sgi = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); sgi < n_segments; sgi += (uint64_t)gridDim.x * 4u) {
        const uint64_t words = (uint64_t)min(cnt[sgi], cap1) * recw;
        const uint64_t *src = seg + sgi * cap1 * recw;
        uint64_t *dst = out + off[sgi] * recw;
        for (uint64_t i = lane; i < words; i += 64) dst[i] = src[i];
    }
}

// k-dependent part of the geometry
void skm_geom_k(SkmGeom &g, int k)
{
    memset(&g, 0, sizeof(g));
    g.k = k;
    g.m = skm_minimizer_len(k);
    g.w = k - g.m + 1;
    g.wpow = 1;
    while (g.wpow * 2 <= g.w) g.wpow *= 2;
    g.kw = k <= 32 ? 1 : 2;
    g.nbw = g.kw + 1;
    g.recw = 1 + g.nbw;
    g.lrecw = g.recw;
    g.ncap = 32 * g.nbw - k + 1;
    g.sbw = ((64u * (uint32_t)g.ncap) >> 5) + 2u;
    g.dbg = kv_knob("KV_SKM_DEBUG") ? (uint32_t)atoi(kv_knob("KV_SKM_DEBUG")) : 0u;
    g.bpt = skm_default_bpt();
}

__global__ void k_mex_sum_kmers(const uint64_t *seg, const uint32_t *cnt, const uint64_t *off, uint64_t n_segments, uint32_t cap1, uint32_t recw,
                                unsigned long long *out)
{
    unsigned long long mine = 0;
    for (uint64_t sgi = blockIdx.x; sgi < n_segments; sgi += gridDim.x) {
        const uint32_t n = min(cnt[sgi], cap1);
        const uint64_t first = off ? off[sgi] : sgi * cap1;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) mine += recw == 2u ? skm_c_n(seg[(first + i) * 2u + 1u]) : skm_hdr_n(seg[(first + i) * recw]);
    }
    mine = wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(out, mine);
}


// cut `reads` into super-k-mers and bucket them (S1 + S2); idx.mu held by the caller
// distinct_frac: the share of distinct k-mers the caller expects (0: sequencing coverage of a whole sample, ~0.3)
int skm_build(SkmIndex &idx, const kv_reads *reads, int k, uint64_t n_kmers, hipStream_t st, double distinct_frac = 0.0, bool want_pos = true)
{
    SkmGeom &g = idx.g;
    memset(&g, 0, sizeof(g));
    idx.valid = false;
    idx.mex_scan_ready = false;
    idx.dl_valid = false;
    idx.builds += 1;
    g.k = k;
    g.m = skm_minimizer_len(k);
    g.w = k - g.m + 1;
    g.wpow = 1;
    while (g.wpow * 2 <= g.w) g.wpow *= 2;
    g.kw = k <= 32 ? 1 : 2;
    g.nbw = g.kw + 1;
    g.recw = 1 + g.nbw;
    g.lrecw = g.recw;
    g.ncap = 32 * g.nbw - k + 1;
    g.dbg = kv_knob("KV_SKM_DEBUG") ? (uint32_t)atoi(kv_knob("KV_SKM_DEBUG")) : 0u;
    if (kv_knob("KV_SKM_FORCE_LOOSE")) g.dbg |= 4096u;
    g.bpt = skm_default_bpt();
    // KV_SKM_DEDUP=1 (k = 31, oriented records): k_skm_count combines identical records before their k-mers (skm_rec_combine): a 3072-slot
    // k-mer table beside a 512-slot table of records, or KV_SKM_DEDUP_RS=1024: 4096 beside 1024 at two workgroups per CU.  Off unless asked
    // for: measured SLOWER (k_skm_count 9.25 -> 10.2 ms per step of config 2 either way, profiles/README.md round 5).
    // KV_SKM_DEDUP_MAXN=n: records of more than n k-mers send their bucket down the plain walk (tests: n = 5 sends nearly every bucket there)
    {
        const char *eo = kv_knob("KV_SKM_ORIENT"), *ed = kv_knob("KV_SKM_DEDUP"), *em = kv_knob("KV_SKM_DEDUP_MAXN");
        const bool dd = k == 31 && !(eo && atoi(eo) == 0) && (ed && atoi(ed) == 1) && !g.dbg && !kv_knob("KV_SKM_ANY_K");
        g.dd_maxn = dd ? (uint32_t)(SKM_C_BASES + 1 - k) : 0u;
        if (dd && em) g.dd_maxn = (uint32_t)std::max(0, std::min(atoi(em), SKM_C_BASES + 1 - k));
    }
    const bool dd_big = g.dd_maxn && kv_knob("KV_SKM_DEDUP_RS") && atoi(kv_knob("KV_SKM_DEDUP_RS")) >= 1024;      // (experiment: 1024 record slots beside 4096 k-mer slots, two workgroups per CU)
    const uint32_t table_slots = g.kw == 1 ? (g.dd_maxn && !dd_big ? 3072u : 4096u) : 2048u;
    const char *tgt_env = kv_knob("KV_SKM_BUCKET_KMERS");      // tests shrink the buckets to exercise many of them on small inputs
    // k-mers per fine bucket: as many as leave the LDS table ~0.4 full (0.29 for two-word keys, whose longer windows put
    // fewer, bigger minimizer loci into a bucket: more variance) given the share of distinct k-mers the previous batch
    // into the same sketch -- or routed on the same stream -- showed; without one, a whole 30x sample is assumed
    // (~0.2 distinct).  Fuller tables overflow in the larger buckets: 34 % distinct k-mers in 8192-k-mer buckets (a
    // quarter of a 30x sample per batch) put 2.1 % of the occurrences on the slow path.
    const double frac = std::min(1.0, std::max(0.05, distinct_frac > 0.0 ? distinct_frac : 0.2));
    // (two-word keys at 0.35: 2.2 M occurrences per 30x sample missed the tables and took the spill path, 67 ms per step of config 5; 0.29: 62 ms)
    uint64_t target = (uint64_t)((g.kw == 1 ? 0.4 : 0.29) * table_slots / frac);
    target = std::max<uint64_t>(table_slots / 2, std::min<uint64_t>(target, g.kw == 1 ? 2ull * table_slots : table_slots + table_slots / 2));
    // (the table of records: ~4 % of a 30x bucket's occurrences are distinct records -- 195 of 512 slots at 4608 occurrences)
    if (g.dd_maxn && !dd_big) target = std::min<uint64_t>(target, 4608);
    if (tgt_env) target = std::max<uint64_t>(64, strtoull(tgt_env, nullptr, 10));
    uint64_t nfine = std::max<uint64_t>(1, (n_kmers + target - 1) / target);
    {
        // samples of one family are about the same size but not exactly: a batch whose own bucket count lies within a
        // quarter of the previous build's takes that one, so that the buckets of the controls and of the case sample
        // are the same buckets (the scan can then use the controls' abundance lists, KvAbundList)
        std::lock_guard<std::mutex> glk(g_skm_mu);
        if (g_skm_pref_k == k && g_skm_pref_nfine && nfine * 4 >= g_skm_pref_nfine * 3 && nfine * 4 <= g_skm_pref_nfine * 5) nfine = g_skm_pref_nfine;
        else { g_skm_pref_k = k; g_skm_pref_nfine = nfine; }
    }
    // at most 255 x 4096 buckets (a bucket id travels as 20 bits through S1): 8.5 G k-mers at 8192 per bucket; bigger
    // batches get bigger buckets, which only costs deduplication efficiency
    g.F2 = std::min<uint32_t>(512u, pow2_ceil((uint64_t)std::ceil(std::sqrt((double)nfine))));
    if (nfine > 255ull * g.F2) g.F2 = std::min<uint32_t>(SKM_MAX_F2, pow2_ceil((nfine + 254) / 255));
    g.fbits = 0;
    while ((1u << g.fbits) < g.F2) ++g.fbits;
    g.C1 = (uint32_t)std::min<uint64_t>(255, std::max<uint64_t>(1, (nfine + g.F2 - 1) / g.F2));
    g.n_buckets = g.C1 * g.F2;
    // 16-byte records without positions (kv_skm_device.h) when nobody will ask where a k-mer was: the count of a sample that is not
    // scanned from this very batch.  The lane-per-read S1 writes them, the sorted S2 moves them, k_skm_count reads them; KV_SKM_COMPACT=0: never
    {
        const char *e = kv_knob("KV_SKM_COMPACT"), *s2 = kv_knob("KV_SKM_S2");
        g.compact = (!want_pos && g.kw == 1 && k + 1 <= SKM_C_BASES && skm_lane_fits(g, reads) && g.F2 <= SKM_S2_MAXF && !(s2 && strcmp(s2, "sorted") != 0) &&
                     !(e && atoi(e) == 0)) ? 1u : 0u;
        if (g.compact) { g.recw = 2; g.ncap = std::min(g.ncap, SKM_C_BASES + 1 - k); }
    }
    const int cus = kv_device_cus();
    // at least one whole ticket per workgroup: the per-writer capacities below assume even shares
    g.nwg1 = (uint32_t)std::min<uint64_t>((std::max<uint32_t>(reads->n_tiles, 1u) + SKM_TILES_PER_TICKET - 1) / SKM_TILES_PER_TICKET,
                                          std::min<uint32_t>(768u, 3u * (uint32_t)cus));
    // (the w = 40 lane kernel, and reads whose LDS slices fit a CU only twice: two workgroups per CU)
    if (skm_lane_fits(g, reads) && (g.w == 2 * SKM_LANE_B || skm_lane_wgs_per_cu(reads->uni_len) < 3u)) g.nwg1 = std::min<uint32_t>(g.nwg1, 2u * (uint32_t)cus);
    {
        // the wave kernel with 1024-thread workgroups runs one workgroup per CU (fewer writers: see k_skm_emit_wave)
        int ch_unused = 16;
        if (skm_wave_threads() == 1024 && skm_wave_plan(g, reads, &ch_unused)) g.nwg1 = std::min<uint32_t>(g.nwg1, (uint32_t)cus);
    }
    {
        const uint64_t avg = (reads->n_tiles + g.nwg1 - 1) / g.nwg1;
        g.quota1 = (uint32_t)std::min<uint64_t>(kv_round_up(avg + avg / 2 + 1, SKM_TILES_PER_TICKET), 0xfffffff0ull);
    }
    g.nwg2 = std::max<uint32_t>(1u, std::min<uint32_t>(16u, 1024u / g.C1));
    if (const char *e = kv_knob("KV_SKM_NWG1")) g.nwg1 = std::max<uint32_t>(1u, std::min<uint32_t>(g.nwg1, (uint32_t)atoi(e)));
    if (const char *e = kv_knob("KV_SKM_NWG2")) g.nwg2 = std::max<uint32_t>(1u, std::min<uint32_t>(16u, (uint32_t)atoi(e)));
    g.nwg2 = std::min<uint32_t>(g.nwg2, g.nwg1);
    g.np_max = std::max<uint32_t>(reads->tile_max_bases, 64u);
    const uint64_t min_stride = reads->max_len >= (uint32_t)k ? reads->max_len - (uint32_t)k + 1 : 1;
    g.stride = min_stride;
    // records: one run per ~ (w + 1) / 2 k-mers, one more per read, a few cuts at ncap
    const double rec_est = (double)n_kmers * 2.2 / (double)(g.w + 1) + (double)reads->n_reads + 1024.0;
    const double m1 = rec_est / ((double)g.C1 * g.nwg1), m2 = rec_est / ((double)g.n_buckets * g.nwg2);
    g.cap1 = (uint32_t)kv_round_up((uint64_t)(m1 * 1.5 + 8.0 * std::sqrt(m1)) + 64, 16);
    g.cap2 = (uint32_t)kv_round_up((uint64_t)(m2 * 1.3 + 8.0 * std::sqrt(m2)) + 32, 16);
    // loose records: segment overflow (whole records) and occurrences that missed a full LDS table (one k-mer each)
    g.loose_cap = (uint64_t)(rec_est / 8.0) + n_kmers / 16 + (1u << 20);
    if (const char *pct = kv_knob("KV_SKM_CAP_PCT")) {       // tests: undersized segments push records through the loose list
        g.cap1 = std::max<uint32_t>(16u, (uint32_t)((uint64_t)g.cap1 * (uint64_t)atoi(pct) / 100));
        g.cap2 = std::max<uint32_t>(16u, (uint32_t)((uint64_t)g.cap2 * (uint64_t)atoi(pct) / 100));
    }
    if (const char *lc = kv_knob("KV_SKM_LOOSE_CAP")) g.loose_cap = strtoull(lc, nullptr, 10);
    const size_t rb = (size_t)g.recw * 8;
    const size_t b_seg1 = kv_round_up((uint64_t)g.C1 * g.nwg1 * g.cap1 * rb, 256), b_cnt1 = kv_round_up((uint64_t)g.C1 * g.nwg1 * 4, 256);
    const size_t b_seg2 = kv_round_up((uint64_t)g.n_buckets * g.nwg2 * g.cap2 * rb, 256), b_cnt2 = kv_round_up((uint64_t)g.n_buckets * g.nwg2 * 4, 256);
    const size_t b_loose = kv_round_up(g.loose_cap * (size_t)g.lrecw * 8, 256), b_ctr = 256;
    // (no memory for the buckets -- three samples cut side by side in batches of tens of millions of reads -- is a CAPACITY answer: the batch
    // takes the plain partition, or the tile scan, as it does when a buffer runs over)
    if (idx.arena.need(b_seg1 + b_cnt1 + b_seg2 + b_cnt2 + b_loose + b_ctr) != hipSuccess) {
        (void)hipGetLastError();
        idx.valid = false; idx.mex_scan_ready = false; idx.dl_valid = false;
        kv_set_error("no device memory for the super-k-mer buckets (%.1f GB)", (double)(b_seg1 + b_cnt1 + b_seg2 + b_cnt2 + b_loose + b_ctr) / 1e9);
        return KV_ERR_CAPACITY;
    }
    unsigned char *base = (unsigned char *)idx.arena.p;
    g.seg1 = (uint64_t *)base; base += b_seg1;
    g.cnt1 = (uint32_t *)base; base += b_cnt1;
    { const char *e = kv_knob("KV_SKM_SEG1"); g.seg1_wmajor = (e && !strcmp(e, "bucket")) ? 0u : 1u; }
    { const char *e = kv_knob("KV_SKM_ORIENT"); g.oriented = (e && atoi(e) == 0) ? 0u : 1u; }
    g.seg2 = (uint64_t *)base; base += b_seg2;
    g.cnt2 = (uint32_t *)base; base += b_cnt2;
    g.loose = (uint64_t *)base; base += b_loose;
    g.ctr = (unsigned long long *)base;
    KV_HIP(hipMemsetAsync(g.ctr, 0, b_ctr, st));
    g.sbw = ((64u * (uint32_t)g.ncap) >> 5) + 2u;
    g.bucket_kmers = std::max<uint64_t>(1, n_kmers / g.n_buckets);
    skm_launch_emit(g, reads, st);
    skm_launch_split(g, st);
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(&g.ctr[6], &g.ctr[0], 8, hipMemcpyDeviceToDevice, st));   // loose records S1/S2 left behind
    skm_pick_bpt(g);
    const uint32_t nwg3 = skm_nwg3(g);
    {
        const uint64_t avg = (g.n_buckets + nwg3 - 1) / nwg3;
        g.quota3 = (uint32_t)kv_round_up(avg + avg / 2 + 1, g.bpt);
    }
    idx.reads_uid = reads->uid;
    idx.k = k;
    idx.valid = true;
    return KV_OK;
}


}  // namespace

// can (and should) the super-k-mer front end take this batch?  -1 no; 0 yes if the sketch agrees; 1 yes, asked for by name
static int skm_fits(int hashfam, int ksize, const kv_reads *reads, uint64_t n_kmers, bool for_scan)
{
    const char *force = kv_knob(for_scan ? "KV_NOVEL_PATH" : "KV_COUNT_PATH");
    if (force && strcmp(force, "skm") != 0) return -1;        // another path was asked for by name
    if (hashfam != HF_MURMUR || ksize < SKM_MIN_K || ksize > SKM_MAX_K) return -1;
    if (reads->n_tiles == 0 || n_kmers == 0) return -1;
    const uint64_t min_stride = reads->max_len >= (uint32_t)ksize ? reads->max_len - (uint32_t)ksize + 1 : 1;
    if ((double)reads->n_reads * (double)min_stride >= (double)(1ull << SKM_POS_BITS)) return -1;
    if (reads->tile_max_bases == 0 || reads->tile_max_bases > 8192u) return -1;
    if (force) return 1;
    return n_kmers >= (1ull << 22) ? 0 : -1;
}

bool kv_skm_eligible(const kv_sketch *s, const kv_reads *reads, uint64_t n_kmers, bool for_scan)
{
    const int fits = skm_fits(s->h.hashfam, s->h.ksize, reads, n_kmers, for_scan);
    // skm_off: the previous batch into this sketch did not deduplicate (kv_consume_skm); skm_off_kmers: nor did a batch of this size
    // before the sketch was cleared (within a tenth)
    const bool same_shape = !for_scan && s->skm_off_kmers != 0 && n_kmers * 10 >= s->skm_off_kmers * 9 && n_kmers * 10 <= s->skm_off_kmers * 11;
    return fits > 0 || (fits == 0 && !s->skm_off && !same_shape);
}

bool kv_skm_eligible_kind(int hashfam, int ksize, const kv_reads *reads, uint64_t n_kmers, bool for_scan)
{
    return skm_fits(hashfam, ksize, reads, n_kmers, for_scan) >= 0;
}

int kv_consume_skm(kv_sketch *s, const kv_reads *reads, const ConsumeFilter &filter, const kv_sketch *mask,
                   uint64_t n_kmers, int nbands, uint64_t *n_added)
{
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    const int k = s->h.ksize;
    // (a sketch with the scan hint keeps positions: the scan that follows marks occurrences from these very buckets)
    // no memory for the buckets, or for the staging of the bin stages beside them (three samples cut side by side in batches of tens of
    // millions of reads): the batch takes the plain partition, this sketch does not ask again -- nor after a clear, for a batch of this
    // size -- and what the attempt holds goes back at once so that the plain partition finds room for its own staging
    auto no_memory = [&](int rc) {
        s->skm_off = true;
        s->skm_off_kmers = n_kmers;
        idx.valid = false; idx.mex_scan_ready = false; idx.dl_valid = false;
        idx.arena.release();
        idx.dl.release();
        idx.dl_keys = nullptr; idx.dl_hash = nullptr; idx.dl_bstart = nullptr; idx.dl_bcount = nullptr;
        if (kv_knob("KV_SKM_VERBOSE")) fprintf(stderr, "[kv_skm] no memory for the bucketed count of this batch: the plain partition, and the stream's bucket buffers are given back\n");
        return rc;
    };
    {
        const int rc = skm_build(idx, reads, k, n_kmers, st, s->skm_distinct, s->scan_hint != 0);
        if (rc == KV_ERR_CAPACITY) return no_memory(rc);
        if (rc != KV_OK) return rc;
    }
    SkmGeom &sg = idx.g;
    const uint32_t nwg3 = skm_nwg3(sg);
    BinPlan plan;
    // the bin stages receive one item per DISTINCT k-mer and table: their buffers are sized for 60 % of the k-mers being
    // distinct (sequencing coverage leaves 20-35 %); a batch with more spills over, raises the flag, and is redone by
    // the one-item-per-k-mer partition -- which is where a batch that does not deduplicate belongs anyway
    const uint64_t n_items = std::max<uint64_t>(n_kmers * 3 / 5, std::min<uint64_t>(n_kmers, 1u << 22));
    {
        const int rc = kv_bin_plan(s, n_items, nbands, filter.use_mask != 0, sg.n_buckets, 0u, nwg3, true, &plan);
        if (rc == KV_ERR_CAPACITY) return no_memory(rc);
        if (rc != KV_OK) return rc;
    }
    const SketchDev *d_mask = mask ? mask->d_desc : nullptr;
    // abundance list of this batch (KvAbundList): the first super-k-mer count after a clear writes one, unless KV_SKM_ABL=0
    bool abl_new = false;
    {
        const char *e = kv_knob("KV_SKM_ABL");
        KvAbundList &al = s->abl;
        if (!(e && atoi(e) == 0) && !al.valid && !s->scan_hint) {            // (a case sample's list would never be asked for: it is scanned, not scanned against)
            const uint64_t cap_total = std::min<uint64_t>(std::max<uint64_t>(n_kmers / 8, 1u << 16), 0xfffffff0ull);
            const uint64_t cap_wg = std::max<uint64_t>(64, cap_total / nwg3);
            const size_t b_keys = kv_round_up(cap_wg * nwg3 * 8 * sg.kw, 256), b_cnts = kv_round_up(cap_wg * nwg3, 256);
            const size_t b_idx = kv_round_up((uint64_t)sg.n_buckets * 4, 256);
            const size_t need = b_keys + b_cnts + 2 * b_idx;
            if (al.bytes < need) {
                if (al.mem) (void)hipFree(al.mem);
                al.mem = nullptr; al.bytes = 0;
                if (kv_hip_malloc(&al.mem, need) == hipSuccess) al.bytes = need;
                else (void)hipGetLastError();              // no room: no list, nothing else changes
            }
            if (al.mem) {
                unsigned char *base = (unsigned char *)al.mem;
                al.keys = (uint64_t *)base; base += b_keys;
                al.cnts = (uint8_t *)base; base += b_cnts;
                al.bstart = (uint32_t *)base; base += b_idx;
                al.bcount = (uint32_t *)base;
                KV_HIP(hipMemsetAsync(al.bstart, 0, 2 * b_idx, st));
                al.k = k; al.m = sg.m; al.kw = sg.kw; al.C1 = sg.C1; al.F2 = sg.F2; al.fbits = sg.fbits; al.n_buckets = sg.n_buckets;
                al.nwg = nwg3; al.cap_wg = cap_wg; al.cap_total = cap_wg * nwg3;
                sg.abl_keys = al.keys; sg.abl_cnts = al.cnts; sg.abl_bstart = al.bstart; sg.abl_bcount = al.bcount; sg.abl_cap_wg = (uint32_t)cap_wg;
                abl_new = true;
            }
        }
    }
    // distinct list (kv_sketch_scan_hint): the batch is a case sample's and will be scanned next.  A workgroup takes up to
    // quota3 of the buckets, so its stretch holds one and a half average shares of the distinct k-mers the batch is expected
    // to have (what the previous batch showed, or 30 %); a stretch that runs out drops the list (ctr[9]), nothing else.
    // The first batch a stream ever buckets gets no list unless the room is there already: allocating a gigabyte or two costs
    // tens of milliseconds, more than the list saves once -- a one-shot `kevlar novel` is exactly that case -- while a process
    // that counts and scans sample after sample pays it once (KV_SKM_DL=1: always, =0: never).
    bool dl_new = false;
    const char *dl_env = kv_knob("KV_SKM_DL");
    if (s->scan_hint && !(dl_env && atoi(dl_env) == 0)) {
        const double frac = std::min(1.0, std::max(0.3, s->skm_distinct * 1.15));
        const uint64_t cap_wg = std::min<uint64_t>((uint64_t)((double)n_kmers * frac * 1.6 / nwg3) + 4096, 0xfffffff0ull / nwg3);
        const size_t b_keys = kv_round_up(cap_wg * nwg3 * 8 * sg.kw, 256), b_hash = kv_round_up(cap_wg * nwg3 * 8, 256);
        const size_t b_idx = kv_round_up((uint64_t)sg.n_buckets * 4, 256);
        const bool worth = (dl_env && atoi(dl_env) == 1) || s->scan_steady || idx.builds > 1 || idx.dl.bytes >= b_keys + b_hash + 2 * b_idx;
        if (worth && idx.dl.need(b_keys + b_hash + 2 * b_idx) == hipSuccess) {
            unsigned char *base = (unsigned char *)idx.dl.p;
            idx.dl_keys = (uint64_t *)base; base += b_keys;
            idx.dl_hash = (uint64_t *)base; base += b_hash;
            idx.dl_bstart = (uint32_t *)base; base += b_idx;
            idx.dl_bcount = (uint32_t *)base;
            idx.dl_cap_wg = (uint32_t)cap_wg;
            KV_HIP(hipMemsetAsync(idx.dl_bstart, 0, 2 * b_idx, st));
            sg.dl_keys = idx.dl_keys; sg.dl_hash = idx.dl_hash; sg.dl_bstart = idx.dl_bstart; sg.dl_bcount = idx.dl_bcount; sg.dl_cap_wg = idx.dl_cap_wg;
            dl_new = true;
        } else {
            (void)hipGetLastError();                    // no room: no list
        }
    }
    {
        KvProfScope prof("k_skm_count");
        const uint32_t ns = (uint32_t)(plan.g.T * plan.g.C);
        // the fixed-k instances count a key's occurrences in 16 bits (k_skm_count, PL): only for buckets that cannot hold 65536 of them
        const bool fixed_k = !sg.dbg && !kv_knob("KV_SKM_ANY_K") && (!SKM_PL || (uint64_t)sg.nwg2 * sg.cap2 * (uint64_t)sg.ncap < 65536ull);
        void (*kernel)(SkmGeom, const SketchDev *, const SketchDev *, ConsumeFilter, BinGeom) =
            sg.kw == 1 ? (sg.dbg ? k_skm_count<1, 4096, true, 0> : k_skm_count<1, 4096, false, 0>) : (sg.dbg ? k_skm_count<2, 2048, true, 0> : k_skm_count<2, 2048, false, 0>);
        bool pl = false;         // a fixed-k instance: 4 KB of product tables where the others keep 1 KB of ASCII
        if (sg.k == 31 && sg.recw == 3 && fixed_k) { kernel = k_skm_count<1, SKM_TS31, false, 31>; pl = true; }
        if (sg.compact) {       // 16-byte records (sg.recw == 2): their own instances
            kernel = sg.dbg ? k_skm_count<1, 4096, true, 0, true> : k_skm_count<1, 4096, false, 0, true>;
            pl = false;
            if (sg.k == 31 && fixed_k) { kernel = k_skm_count<1, SKM_TS31, false, 31, true>; pl = true; }
        }
        // (BASELINE.json configs[4]: k = 51 -- two-word keys, 128-bit reverse complement, three murmur blocks + a 3-byte tail)
        if (sg.k == 51 && sg.recw == 4 && fixed_k) { kernel = k_skm_count<2, SKM_TS51, false, 51>; pl = true; }
        if (sg.oriented) {      // oriented records: the same instances with the walk that takes k-mers as they stand
            kernel = sg.kw == 1 ? (sg.dbg ? k_skm_count<1, 4096, true, 0, false, true> : k_skm_count<1, 4096, false, 0, false, true>)
                                : (sg.dbg ? k_skm_count<2, 2048, true, 0, false, true> : k_skm_count<2, 2048, false, 0, false, true>);
            pl = false;
            if (sg.k == 31 && sg.recw == 3 && fixed_k) { kernel = k_skm_count<1, SKM_TS31, false, 31, false, true>; pl = true; }
            if (sg.compact) {
                kernel = sg.dbg ? k_skm_count<1, 4096, true, 0, true, true> : k_skm_count<1, 4096, false, 0, true, true>;
                pl = false;
                if (sg.k == 31 && fixed_k) { kernel = k_skm_count<1, SKM_TS31, false, 31, true, true>; pl = true; }
            }
            if (sg.k == 51 && sg.recw == 4 && fixed_k) { kernel = k_skm_count<2, SKM_TS51, false, 51, false, true>; pl = true; }
            // k = 31: identical records are combined before their k-mers are (skm_rec_combine; skm_build sized the buckets for it)
            if (sg.dd_maxn && sg.k == 31 && (sg.compact || sg.recw == 3) && !sg.dbg) {
                pl = false;
                kernel = sg.compact ? k_skm_count<1, 3072, false, 31, true, true, 512> : k_skm_count<1, 3072, false, 31, false, true, 512>;
                if (kv_knob("KV_SKM_DEDUP_RS") && atoi(kv_knob("KV_SKM_DEDUP_RS")) >= 1024)
                    kernel = sg.compact ? k_skm_count<1, 4096, false, 31, true, true, 1024> : k_skm_count<1, 4096, false, 31, false, true, 1024>;
            }
        }
        if (!sg.oriented) sg.dd_maxn = 0;
        const size_t lds = ((pl && SKM_PL ? 1024 : 256) + ((ns + 3u) & ~3u) + (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(sg.sbw)) * 4;
        hipLaunchKernelGGL(kernel, dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, (const SketchDev *)s->d_desc, d_mask, filter, plan.g);
    }
    {
        KvProfScope prof("k_skm_loose_count");
        if (sg.kw == 1) hipLaunchKernelGGL(k_skm_loose_count<1>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, (const SketchDev *)s->d_desc, d_mask, filter, plan.g);
        else hipLaunchKernelGGL(k_skm_loose_count<2>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, (const SketchDev *)s->d_desc, d_mask, filter, plan.g);
    }
    KV_HIP(hipGetLastError());
    // a lost record (loose list overflow) must stop the apply stage, which looks at the partition's own flag
    hipLaunchKernelGGL(k_skm_forward_flag, dim3(1), dim3(1), 0, st, sg.ctr, plan.g.ctr);
    KV_HIP(hipGetLastError());
    sg.abl_keys = nullptr; sg.abl_cnts = nullptr; sg.abl_bstart = nullptr; sg.abl_bcount = nullptr;
    sg.dl_keys = nullptr; sg.dl_hash = nullptr; sg.dl_bstart = nullptr; sg.dl_bcount = nullptr;
    const int rc = kv_bin_finish(s, plan, true, 0, n_added);     // synchronises the stream
    if (abl_new) s->abl.valid = rc == KV_OK;
    {
        // what the batch looked like: if most k-mers are distinct (low coverage per batch) cutting and bucketing the
        // reads buys nothing, and if many occurrences missed the LDS tables the buckets were too full; either way
        // the next batches into this sketch take the one-item-per-k-mer partition (until the sketch is cleared)
        unsigned long long sc[13] = {0};
        const bool got = hipMemcpy(sc, sg.ctr, sizeof(sc), hipMemcpyDeviceToHost) == hipSuccess;
        idx.dl_valid = dl_new && got && rc == KV_OK && sc[9] == 0;
        if (dl_new && kv_knob("KV_SKM_VERBOSE")) fprintf(stderr, "[kv_skm] distinct list: %s (%llu workgroups ran out of %u entries)\n", idx.dl_valid ? "kept" : "dropped", sc[9], idx.dl_cap_wg);
        if (got && n_kmers) {
            const double alone = (double)(sc[0] > sc[6] ? sc[0] - sc[6] : 0) / (double)n_kmers;
            const double distinct = (double)sc[7] / (double)n_kmers + alone;
            s->skm_off = rc != KV_OK || alone > 0.03 || distinct > 0.45;
            // (what the batch itself showed -- or buffers sized for sequencing coverage running over, which is how a batch without repeats ends:
            // its figures are then those of an aborted count -- not an error of another kind)
            s->skm_off_kmers = (distinct > 0.45 || rc == KV_ERR_CAPACITY) ? n_kmers : 0;
            s->skm_distinct = distinct;
            { std::lock_guard<std::mutex> glk(g_skm_mu); g_skm_last_distinct = distinct; }
            if (kv_knob("KV_SKM_VERBOSE"))
                fprintf(stderr, "[kv_skm] batch of %llu k-mers: %.1f%% distinct, %.2f%% outside the LDS tables, %llu of %llu records (%d bytes each) outside their segments, %llu of %u buckets walked record by record%s\n",
                        (unsigned long long)n_kmers, 100 * distinct, 100 * alone, sc[6], sc[5], 8 * sg.recw, sg.dd_maxn ? sc[12] : (unsigned long long)sg.n_buckets, sg.n_buckets,
                        s->skm_off ? " -> next batches take the plain partition" : "");
        }
    }
    if (rc != KV_OK) {
        idx.valid = false;
        idx.mex_scan_ready = false;
        if (kv_knob("KV_SKM_VERBOSE")) {
            unsigned long long sc[8] = {0}, bc[4] = {0};
            (void)hipMemcpy(sc, sg.ctr, sizeof(sc), hipMemcpyDeviceToHost);
            (void)hipMemcpy(bc, plan.g.ctr, sizeof(bc), hipMemcpyDeviceToHost);
            fprintf(stderr, "[kv_skm] count fell back: rc %d, loose %llu of %llu (flag %llu), records %llu, spill %llu of %llu (flag %llu); C1 %u F2 %u nwg1 %u nwg2 %u cap1 %u cap2 %u\n",
                    rc, sc[0], (unsigned long long)sg.loose_cap, sc[1], sc[5], bc[0], (unsigned long long)plan.g.spill_cap, bc[1],
                    sg.C1, sg.F2, sg.nwg1, sg.nwg2, sg.cap1, sg.cap2);
        }
    }
    if (s->skm_off && idx.arena.bytes + idx.dl.bytes >= ((size_t)1 << 30)) {
        // The batches of this sketch will not come this way again, and what was bucketed here is of no use to a scan of a batch with
        // nothing to combine: the gigabytes the attempt took (records twice over with their slack: 30 GB per stream for a 37.5 M-read
        // batch of config 4, which is what kept batches of twice the size from fitting beside the resident reads) go back now instead of
        // waiting for kv_scratch_trim.
        idx.valid = false; idx.mex_scan_ready = false; idx.dl_valid = false;
        idx.arena.release();
        idx.dl.release();
        idx.dl_keys = nullptr; idx.dl_hash = nullptr; idx.dl_bstart = nullptr; idx.dl_bcount = nullptr;
        if (kv_knob("KV_SKM_VERBOSE")) fprintf(stderr, "[kv_skm] the stream's bucket buffers are given back\n");
    }
    return rc;
}

// bits[w] bit j = (table[32 w + j] >= case_min) for a table of byte counters (k_case_bits): the first probe of a scan as a bit map
void kv_case_bits_launch(const uint8_t *d_table, uint64_t size, int case_min, uint32_t *d_bits, hipStream_t st)
{
    KvProfScope prof("k_case_bits");
    hipLaunchKernelGGL(k_case_bits, dim3(4096), dim3(256), 0, st, d_table, size, case_min, d_bits);
}

void kv_tile_hits_launch(const kv_reads *reads, const NovelParams &p, hipStream_t st)
{
    KvProfScope prof("k_tile_hits");
    hipLaunchKernelGGL(k_tile_hits, dim3(reads->n_tiles), dim3(64), 0, st, reads_dev(reads), p);
}

bool kv_skm_list_ready(const kv_reads *reads, int ksize)
{
    if (kv_knob("KV_SKM_NO_REUSE") || (kv_knob("KV_SKM_DL") && atoi(kv_knob("KV_SKM_DL")) == 0)) return false;
    std::lock_guard<std::mutex> lk(g_skm_mu);
    for (auto &kv : g_skm)
        if (kv.second.valid && kv.second.dl_valid && kv.second.reads_uid == reads->uid && kv.second.k == ksize) return true;
    return false;
}

int kv_skm_novel_mark(const kv_reads *reads, const NovelParams &p, uint64_t n_kmers)
{
    hipStream_t st = kv_stream();
    const int k = p.hp.k;
    // the bucketed batch may sit in the arena of whichever stream counted it
    SkmIndex *idx = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_skm_mu);
        for (auto &kv : g_skm)
            if (kv.second.valid && kv.second.reads_uid == reads->uid && kv.second.k == k) { idx = &kv.second; break; }
        if (!idx) idx = &g_skm[kv_stream_key(st)];
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    const bool reuse = idx->valid && idx->reads_uid == reads->uid && idx->k == k && !idx->g.compact && !kv_knob("KV_SKM_NO_REUSE");
    if (!reuse) {
        double hint;
        { std::lock_guard<std::mutex> glk(g_skm_mu); hint = g_skm_last_distinct; }
        const int rc = skm_build(*idx, reads, k, n_kmers, st, hint);
        if (rc != KV_OK) return rc;
    }
    SkmGeom &sg = idx->g;
    // the count pass of this very batch left key + hash of every distinct k-mer (kv_sketch_scan_hint): scan from that list
    const bool from_list = reuse && idx->dl_valid && !p.set_keys && !(kv_knob("KV_SKM_DL") && atoi(kv_knob("KV_SKM_DL")) == 0);
    if (reuse) {
        // records that did not fit their S1/S2 segment are the first ctr[6] entries of the loose list; the entries the
        // count pass added behind them (single occurrences that missed its LDS tables) are re-created by the scan's
        // own pass 0, so the list is cut back to what S1/S2 left -- unless the scan goes by the distinct list, which does
        // not hold those k-mers: then they stay, and k_skm_loose_novel evaluates them
        if (!from_list) KV_HIP(hipMemcpyAsync(&sg.ctr[0], &sg.ctr[6], 8, hipMemcpyDeviceToDevice, st));
        KV_HIP(hipMemsetAsync(&sg.ctr[4], 0, 8, st));
    }
    const uint32_t nwg3 = skm_nwg3(sg);
    const ReadsDev rd = reads_dev(reads);
    SkmAblSet abls;
    memset(&abls, 0, sizeof(abls));
    abls.ctrl_max = p.ctrl_max;
    if (p.host_ctrls && !p.set_keys && !(kv_knob("KV_SKM_ABL") && atoi(kv_knob("KV_SKM_ABL")) == 0)) {
        const kv_sketch *const *ctrls = (const kv_sketch *const *)p.host_ctrls;
        for (int c = 0; c < p.host_nctrl && abls.n < SKM_MAX_ABL; ++c) {
            const KvAbundList &al = ctrls[c]->abl;
            if (!al.valid || al.k != k || al.m != sg.m || al.C1 != sg.C1 || al.F2 != sg.F2 || al.n_buckets != sg.n_buckets) continue;
            const int a = abls.n++;
            abls.keys[a] = al.keys; abls.cnts[a] = al.cnts; abls.bstart[a] = al.bstart; abls.bcount[a] = al.bcount;
            abls.maxv[a] = ctrls[c]->h.storage == ST_BYTE ? 255u : (ctrls[c]->h.storage == ST_NIBBLE ? 15u : 1u);
        }
    }
    if (kv_knob("KV_SKM_VERBOSE")) fprintf(stderr, "[kv_skm] scan: %d of %d controls bring an abundance list in this bucket geometry\n", abls.n, p.host_nctrl);
    if (kv_knob("KV_SKM_VERBOSE")) fprintf(stderr, "[kv_skm] scan: %s\n", from_list ? "from the count pass's distinct list" : "by walking the buckets");
    if (const char *e = kv_knob("KV_SKM_SCAN_DEBUG")) sg.dbg = (uint32_t)atoi(e);       // scratch/scan_phases.py
    NovelParams pl = p;
    pl.case0_bits = nullptr;
    // (KV_NOVEL_BITS=0: probe the table.  Measured at config 2: the bit map costs 0.135 ms, the list scan goes from 2.57-2.77 to 2.34 ms)
    if (from_list && p.host_case0 && !(kv_knob("KV_NOVEL_BITS") && atoi(kv_knob("KV_NOVEL_BITS")) == 0)) {
        const kv_sketch *c0 = (const kv_sketch *)p.host_case0;
        if (c0->h.storage == ST_BYTE && !c0->lazy_zero && idx->bits.need(kv_round_up(((c0->h.size[0] + 31) >> 5) * 4, 256)) == hipSuccess) {
            KvProfScope prof("k_case_bits");
            hipLaunchKernelGGL(k_case_bits, dim3(4096), dim3(256), 0, st, (const uint8_t *)c0->h.tab[0], (uint64_t)c0->h.size[0], p.case_min, (uint32_t *)idx->bits.p);
            pl.case0_bits = (const uint32_t *)idx->bits.p;
        } else {
            (void)hipGetLastError();
        }
    }
    if (from_list) {
        KvProfScope prof("k_skm_novel_list");
        sg.dl_keys = idx->dl_keys; sg.dl_hash = idx->dl_hash; sg.dl_bstart = idx->dl_bstart; sg.dl_bcount = idx->dl_bcount; sg.dl_cap_wg = idx->dl_cap_wg;
        const size_t lds = (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(sg.sbw) * 4;
        void (*kernel)(SkmGeom, ReadsDev, NovelParams, SkmAblSet) =
            sg.oriented ? (sg.kw == 1 ? (sg.dbg ? k_skm_novel_list<1, 2048, true, true> : k_skm_novel_list<1, 2048, false, true>) : (sg.dbg ? k_skm_novel_list<2, 1024, true, true> : k_skm_novel_list<2, 1024, false, true>))
                        : (sg.kw == 1 ? (sg.dbg ? k_skm_novel_list<1, 2048, true> : k_skm_novel_list<1, 2048, false>) : (sg.dbg ? k_skm_novel_list<2, 1024, true> : k_skm_novel_list<2, 1024, false>));
        hipLaunchKernelGGL(kernel, dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, rd, pl, abls);
        if (p.ab_keys) hipLaunchKernelGGL(k_ab_fill, dim3(2048), dim3(256), 0, st, p);
        if (p.ab_list && kv_knob("KV_SKM_VERBOSE")) {
            unsigned int claimed = 0;
            if (hipMemcpyAsync(&claimed, p.ab_count, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess)
                fprintf(stderr, "[kv_skm] scan: %u interesting k-mers hold a slot for their abundances (the list takes %u)\n", claimed, p.ab_list_cap);
        }
        sg.dl_keys = nullptr; sg.dl_hash = nullptr; sg.dl_bstart = nullptr; sg.dl_bcount = nullptr;
    } else {
        KvProfScope prof("k_skm_novel");
        const size_t lds = (256 + (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(sg.sbw)) * 4;
        if (sg.oriented) {
            if (sg.kw == 1) hipLaunchKernelGGL((k_skm_novel<1, 4096, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, rd, p, abls);
            else hipLaunchKernelGGL((k_skm_novel<2, 2048, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, rd, p, abls);
        } else if (sg.kw == 1) hipLaunchKernelGGL((k_skm_novel<1, 4096>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, rd, p, abls);
        else hipLaunchKernelGGL((k_skm_novel<2, 2048>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, rd, p, abls);
    }
    {
        KvProfScope prof("k_skm_loose_novel");
        if (sg.kw == 1) hipLaunchKernelGGL(k_skm_loose_novel<1>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, rd, p);
        else hipLaunchKernelGGL(k_skm_loose_novel<2>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, rd, p);
    }
    {
        KvProfScope prof("k_tile_hits");
        hipLaunchKernelGGL(k_tile_hits, dim3(reads->n_tiles), dim3(64), 0, st, rd, p);
    }
    KV_HIP(hipGetLastError());
    unsigned long long sctr[2] = {0, 0};
    KV_HIP(hipMemcpyAsync(sctr, sg.ctr, sizeof(sctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    idx->valid = false;     // one scan per build: the loose list now holds this scan's entries
    idx->dl_valid = false;
    if (sctr[1] != 0) {
        kv_set_error("super-k-mer scan: loose record list overflow (%llu records)", sctr[0]);
        return KV_ERR_CAPACITY;
    }
    return KV_OK;
}

int kv_skm_route_distinct(const kv_reads *reads, int ksize, uint64_t n_kmers, int ndest,
                          int (*alloc)(void *ctx, uint32_t nwg, KvRouteSink *sink), void *ctx)
{
    KV_REQUIRE(ndest >= 1 && ndest <= SKM_ROUTE_MAX_DEST, KV_ERR_ARG, "kv_skm_route_distinct: 1..%d destinations", SKM_ROUTE_MAX_DEST);
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    { const int rc = skm_build(idx, reads, ksize, n_kmers, st, idx.distinct_hint); if (rc != KV_OK) return rc; }
    SkmGeom &sg = idx.g;
    const uint32_t nwg3 = skm_nwg3(sg);
    KvRouteSink rs;
    memset(&rs, 0, sizeof(rs));
    { const int rc = alloc(ctx, nwg3, &rs); if (rc != KV_OK) return rc; }
    const HashParams hp = make_hash_params(ksize, HF_MURMUR);
    {
        KvProfScope prof("k_skm_route");
        const size_t lds = (256 + (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(sg.sbw)) * 4;
        if (sg.oriented) {
            if (sg.kw == 1) hipLaunchKernelGGL((k_skm_route<1, 4096, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, hp, rs);
            else hipLaunchKernelGGL((k_skm_route<2, 2048, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, hp, rs);
        } else if (sg.kw == 1) hipLaunchKernelGGL((k_skm_route<1, 4096>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, hp, rs);
        else hipLaunchKernelGGL((k_skm_route<2, 2048>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, hp, rs);
    }
    {
        KvProfScope prof("k_skm_loose_route");
        if (sg.kw == 1) hipLaunchKernelGGL(k_skm_loose_route<1>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, hp, rs);
        else hipLaunchKernelGGL(k_skm_loose_route<2>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, hp, rs);
    }
    KV_HIP(hipGetLastError());
    unsigned long long sctr[8] = {0};
    KV_HIP(hipMemcpyAsync(sctr, sg.ctr, sizeof(sctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    // the next shard routed on this stream (same sample or a sibling: same coverage) gets buckets sized for what this one held
    const double alone = (double)(sctr[0] > sctr[6] ? sctr[0] - sctr[6] : 0) / (double)n_kmers;
    idx.distinct_hint = std::min(1.0, (double)sctr[7] / (double)n_kmers + alone);
    if (kv_knob("KV_SKM_VERBOSE"))
        fprintf(stderr, "[kv_skm] routed %llu k-mers: %.1f%% distinct, %.2f%% outside the LDS tables, %u buckets\n",
                (unsigned long long)n_kmers, 100 * idx.distinct_hint, 100 * alone, sg.n_buckets);
    if (sctr[1] != 0) {
        idx.valid = false;
        idx.mex_scan_ready = false;
        kv_set_error("super-k-mer route: loose record list overflow (%llu records)", sctr[0]);
        return KV_ERR_CAPACITY;
    }
    return KV_OK;
}


// ---- minimizer-sharded exchange (kevlar_amd/shardrun.py) ------------------------------------------------
// N ranks each hold 1/N of a sample's reads.  Deduplicating a shard on its own finds little to combine at 1/8 of the
// coverage, so the shards are cut into super-k-mers first (S1, here), the RECORDS travel to the rank that owns their
// minimizer bucket -- every occurrence of a k-mer, from whichever shard, meets there -- and that rank combines them at the
// sample's full coverage (S2 + the distinct route below) before (hash, occurrences) pairs go on to the band owners.
int kv_skm_mex_plan(int ksize, uint64_t n_reads_global, uint32_t read_len, int ndest, kv_mex_plan *plan)
{
    KV_REQUIRE(plan && ndest >= 1 && ndest <= SKM_ROUTE_MAX_DEST && ksize >= SKM_MIN_K && ksize <= SKM_MAX_K && read_len >= (uint32_t)ksize,
               KV_ERR_ARG, "kv_mex_plan: k in %d..%d, 1..%d destinations, reads of at least k bases", SKM_MIN_K, SKM_MAX_K, SKM_ROUTE_MAX_DEST);
    memset(plan, 0, sizeof(*plan));
    SkmGeom g;
    skm_geom_k(g, ksize);
    const uint64_t nk_read = read_len - (uint32_t)ksize + 1u;
    const uint64_t n_kmers = n_reads_global * nk_read;
    const uint32_t table_slots = g.kw == 1 ? 4096u : 2048u;
    const uint64_t target = g.kw == 1 ? 2ull * table_slots : table_slots + table_slots / 2;       // a whole sample at sequencing coverage (~0.2 distinct)
    const uint64_t nfine = std::max<uint64_t>(1, (n_kmers + target - 1) / target);
    uint32_t F2 = std::min<uint32_t>(512u, pow2_ceil((uint64_t)std::ceil(std::sqrt((double)nfine))));
    if (nfine > 255ull * F2) F2 = std::min<uint32_t>(SKM_MAX_F2, pow2_ceil((nfine + 254) / 255));
    uint32_t C1 = (uint32_t)std::min<uint64_t>(255, std::max<uint64_t>(1, (nfine + F2 - 1) / F2));
    C1 = (uint32_t)kv_round_up(C1, (uint64_t)ndest);
    if (C1 > 255u) C1 = 255u / (uint32_t)ndest * (uint32_t)ndest;
    uint32_t fbits = 0;
    while ((1u << fbits) < F2) ++fbits;
    const uint64_t shard_reads = (n_reads_global + ndest - 1) / ndest;
    const uint64_t tiles = (shard_reads + KV_TILE_MAX_READS - 1) / KV_TILE_MAX_READS;
    // Writers per shard; every one owns a segment of every coarse bucket.  One per CU, 1024 threads each: a shard is an N-th of a
    // sample, and with the 768 writers of a whole sample its ~190 000 segments held ~40 records each at N = 8 -- the cut, the packing
    // and the owner's split all pay per segment (per rank of config 2, N = 8 / N = 2: 8.58 / 25.07 ms with 768 writers of 512 threads,
    // 7.73 / 23.10 with 256 of 1024).  KV_MEX_NWG1 (the same on every rank) overrides.
    const uint64_t nwg1_max = kv_knob("KV_MEX_NWG1") ? (uint64_t)std::max(1, std::min(768, atoi(kv_knob("KV_MEX_NWG1")))) : 256;
    const uint32_t nwg1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((tiles + SKM_TILES_PER_TICKET - 1) / SKM_TILES_PER_TICKET, nwg1_max));
    const double rec_est = (double)(shard_reads * nk_read) * 2.2 / (double)(g.w + 1) + (double)shard_reads + 1024.0;
    const double m1 = rec_est / ((double)C1 * nwg1);
    const uint32_t cap1 = (uint32_t)kv_round_up((uint64_t)(m1 * 2.0 + 8.0 * std::sqrt(m1)) + 64, 16);
    plan->ksize = ksize; plan->ndest = ndest; plan->C1 = C1; plan->F2 = F2; plan->fbits = fbits; plan->nwg1 = nwg1; plan->cap1 = cap1;
    plan->recw = (uint32_t)g.recw; plan->m = (uint32_t)g.m;
    plan->seg_words = (uint64_t)C1 * nwg1 * cap1 * g.recw;
    plan->cnt_entries = (uint64_t)C1 * nwg1;
    for (int d = 0; d <= ndest; ++d) plan->c_lo[d] = (uint32_t)((uint64_t)C1 * d / ndest);
    plan->n_kmers_global = n_kmers; plan->n_reads_global = n_reads_global; plan->read_len = read_len;
    return KV_OK;
}

// the plan with 16-byte records (compact: kv_skm_device.h): the lane-per-read S1 writes them, the sorted S2 moves them
int kv_skm_mex_plan_short(kv_mex_plan *plan)
{
    SkmGeom g;
    skm_geom_k(g, plan->ksize);
    // (the owner's S2 takes 16-byte records in either form: the sorted split up to 1024 fine buckets, the plain one up to 4096 -- the 12
    // bits the record has for its fine bucket)
    KV_REQUIRE(g.kw == 1 && g.w == SKM_LANE_B && skm_lane_fits_len(g, plan->read_len) &&
               plan->F2 <= 4096u && !(kv_knob("KV_SKM_COMPACT") && atoi(kv_knob("KV_SKM_COMPACT")) == 0),
               KV_ERR_NOTIMPL, "kv_mex_plan_short: no 16-byte records for k = %d, reads of %u bases, %u fine buckets", plan->ksize, plan->read_len, plan->F2);
    if (plan->flags & 1u) return KV_OK;
    plan->flags |= 1u;
    plan->seg_words = plan->seg_words / plan->recw * 2u;
    plan->recw = 2u;
    return KV_OK;
}
// what the plan's flags mean for the geometry
static void skm_mex_apply_flags(SkmGeom &g, const kv_mex_plan *plan)
{
    if (plan->flags & 1u) { g.compact = 1u; g.recw = 2; g.ncap = std::min(g.ncap, SKM_C_BASES + 1 - g.k); g.sbw = ((64u * (uint32_t)g.ncap) >> 5) + 2u; }
}

// S1 of one shard into the caller's buffers ([C1][nwg1][cap1] records, [C1][nwg1] counts: what the plan says).
// d_out != NULL: the filled part of the segments is packed into it as well, destination after destination (kv_skm_mex_pack), if it
// holds out_cap_words; *packed says whether it did.  One stream synchronisation either way.
int kv_skm_mex_emit(const kv_reads *reads, const kv_mex_plan *plan, uint64_t read_base, uint64_t *d_seg, uint32_t *d_cnt,
                    uint64_t *d_out, uint64_t out_cap_words, uint64_t *records_per_dest, int *packed)
{
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    idx.valid = false;
    idx.mex_scan_ready = false;
    SkmGeom &g = idx.g;
    skm_geom_k(g, plan->ksize);
    skm_mex_apply_flags(g, plan);
    g.C1 = plan->C1; g.F2 = plan->F2; g.fbits = plan->fbits; g.n_buckets = g.C1 * g.F2;
    g.nwg1 = plan->nwg1; g.cap1 = plan->cap1;
    g.quota1 = 0xfffffff0u;                                   // tiles are dealt dynamically; the plan's capacity has the slack
    g.np_max = std::max<uint32_t>(reads->tile_max_bases, 64u);
    g.stride = plan->read_len - (uint32_t)plan->ksize + 1u;
    KV_REQUIRE(reads->max_len <= plan->read_len, KV_ERR_ARG, "kv_mex_emit: a read of %u bases in a plan for %u", reads->max_len, plan->read_len);
    // (a rank whose shard holds no read -- fewer reads than ranks -- still takes part: it sends empty segments)
    KV_REQUIRE(reads->n_tiles == 0 || (reads->tile_max_bases > 0 && reads->tile_max_bases <= 8192u), KV_ERR_ARG, "kv_mex_emit: reads too long for the super-k-mer front end");
    g.read_base = read_base;
    // (the exchange's records are oriented like a single GPU's: every rank cuts with the same rule, so the owner of a bucket finds a
    // k-mer under one key whichever shard it came from; KV_SKM_ORIENT=0 on every rank keeps the classic form)
    { const char *e = kv_knob("KV_SKM_ORIENT"); g.oriented = (e && atoi(e) == 0) ? 0u : 1u; }
    g.seg1 = d_seg; g.cnt1 = d_cnt;
    g.loose_cap = 1u << 16;
    const size_t b_loose = kv_round_up(g.loose_cap * (size_t)g.lrecw * 8, 256), b_ctr = 256;
    const uint64_t n_seg = plan->cnt_entries;
    // (a shard without reads cuts nothing and fits any plan)
    KV_REQUIRE(!g.compact || reads->n_tiles == 0 || skm_lane_fits(g, reads), KV_ERR_NOTIMPL, "kv_mex_emit: 16-byte records need a shard of equal-length reads (the lane-per-read cut)");
    const size_t b_off = d_out ? mex_scan_bytes(n_seg) : 0;
    KV_HIP(idx.arena.need(b_loose + b_ctr + b_off));
    g.loose = (uint64_t *)idx.arena.p;
    g.ctr = (unsigned long long *)((unsigned char *)idx.arena.p + b_loose);
    uint64_t *d_off = d_out ? (uint64_t *)((unsigned char *)idx.arena.p + b_loose + b_ctr) : nullptr;
    KV_HIP(hipMemsetAsync(g.ctr, 0, b_ctr, st));
    // a workgroup that takes no tile still writes its counts; workgroups beyond the grid never run: zero them
    KV_HIP(hipMemsetAsync(d_cnt, 0, plan->cnt_entries * 4, st));
    if (reads->n_tiles) {
        g.nwg1 = plan->nwg1;
        tl_s1_threads = plan->nwg1 <= 256u ? 1024u : 0u;
        skm_launch_emit(g, reads, st);
        tl_s1_threads = 0;
    }
    KV_HIP(hipGetLastError());
    if (d_out) {
        KvProfScope prof("k_mex_pack");
        mex_scan_launch(d_cnt, n_seg, plan->cap1, d_off, st);
        hipLaunchKernelGGL(k_mex_gather, dim3((unsigned)std::min<uint64_t>((n_seg + 3) / 4, 1u << 20)), dim3(256), 0, st, d_seg, d_cnt, d_off, n_seg, plan->cap1,
                           plan->recw, d_out, out_cap_words / plan->recw);
        KV_HIP(hipGetLastError());
    }
    KvReadback rb;
    hipError_t rb_err = hipSuccess;
    const unsigned long long *ctr = rb.add(g.ctr, 2, st, &rb_err);
    std::vector<const uint64_t *> bounds(plan->ndest + 1, nullptr);
    if (d_out)
        for (int d = 0; d <= plan->ndest; ++d) bounds[d] = rb.add(d_off + (uint64_t)plan->c_lo[d] * plan->nwg1, 1, st, &rb_err);
    KV_HIP(rb_err);
    KV_HIP(rb.wait(st));
    if (d_out) {
        for (int d = 0; d < plan->ndest; ++d) records_per_dest[d] = *bounds[d + 1] - *bounds[d];
        *packed = (*bounds[plan->ndest] - *bounds[0]) * (uint64_t)plan->recw <= out_cap_words ? 1 : 0;
    }
    // records that miss their segment have nowhere to travel in: the plan's capacity is twice the expected fill, so this
    // means a pathological input (one minimizer everywhere); no silent change of layout
    KV_REQUIRE(ctr[0] == 0 && ctr[1] == 0, KV_ERR_CAPACITY, "kv_mex_emit: %llu records did not fit their exchange segment", ctr[0]);
    // a later scan of this shard against the set of interesting k-mers buckets the shard on its own: 1 / ndest of the coverage
    // leaves more of its k-mers distinct than the sample's ~20 % (measured: 30 / 33 / 49 % at 1/2, 1/4, 1/8 of 30x)
    { std::lock_guard<std::mutex> glk(g_skm_mu); g_skm_last_distinct = std::min(0.9, 0.2 * std::sqrt((double)plan->ndest)); }
    return KV_OK;
}

// the records n_src ranks sent for this rank's Cl coarse buckets -> S2 -> every distinct k-mer once as a (hash, occurrences)
// pair for its band's owner (the callback allocates the sink, as for kv_skm_route_distinct)
int kv_skm_mex_route(const kv_mex_plan *plan, int my_dest, const uint64_t *d_recv_seg, const uint32_t *d_recv_cnt, int n_src, int compact, int keep_scan,
                     int (*alloc)(void *ctx, uint32_t nwg, KvRouteSink *sink), int (*after)(void *ctx), void *ctx, uint64_t *n_kmers_in)
{
    KV_REQUIRE(plan && my_dest >= 0 && my_dest < plan->ndest && n_src >= 1, KV_ERR_ARG, "kv_mex_route: bad argument");
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    idx.valid = false;
    idx.mex_scan_ready = false;
    SkmGeom &g = idx.g;
    skm_geom_k(g, plan->ksize);
    skm_mex_apply_flags(g, plan);
    KV_REQUIRE(!(g.compact && keep_scan), KV_ERR_ARG, "kv_mex_route: the sample the scan is answered from needs records with positions (not a short-record plan)");
    const uint32_t Cl = plan->c_lo[my_dest + 1] - plan->c_lo[my_dest];
    g.C1 = Cl; g.F2 = plan->F2; g.fbits = plan->fbits; g.n_buckets = Cl * g.F2;
    g.nwg1 = plan->nwg1; g.cap1 = plan->cap1; g.n_src = (uint32_t)n_src;
    { const char *e = kv_knob("KV_SKM_ORIENT"); g.oriented = (e && atoi(e) == 0) ? 0u : 1u; }       // as kv_skm_mex_emit cut them
    g.seg1 = const_cast<uint64_t *>(d_recv_seg); g.cnt1 = const_cast<uint32_t *>(d_recv_cnt);
    g.stride = plan->read_len - (uint32_t)plan->ksize + 1u;
    if (Cl == 0) { *n_kmers_in = 0; KvRouteSink rs; memset(&rs, 0, sizeof(rs)); return alloc(ctx, 1, &rs); }
    const uint32_t nseg = g.nwg1 * g.n_src;
    g.nwg2 = std::max<uint32_t>(std::max<uint32_t>(1u, std::min<uint32_t>(16u, 1024u / Cl)), (nseg + 767u) / 768u);
    g.nwg2 = std::min<uint32_t>(g.nwg2, nseg);
    KV_REQUIRE((nseg + g.nwg2 - 1) / g.nwg2 <= 768u, KV_ERR_ARG, "kv_mex_route: %u segments per bucket", nseg);
    // this rank's share of the sample: its buckets hold 1 / ndest of the k-mers, give or take the hash's evenness
    const uint64_t n_kmers_exp = plan->n_kmers_global / (uint64_t)plan->ndest + 1;
    const double rec_est = (double)n_kmers_exp * 2.2 / (double)(g.w + 1) + (double)plan->n_reads_global / plan->ndest + 1024.0;
    const double m2 = rec_est / ((double)g.n_buckets * g.nwg2);
    // (a geometry at its limit -- 4096 fine buckets per coarse one: a handful of distinct 12-base minimizers per bucket -- has buckets of
    // very different sizes: at 248 x 4096 buckets a tenth of the records missed segments of 2 x the even share; KV_MEX_CAP2_SLACK overrides)
    double slack2 = plan->F2 > SKM_S2_MAXF ? 3.0 : 1.4;
    if (const char *e = kv_knob("KV_MEX_CAP2_SLACK")) slack2 = std::max(1.1, atof(e));
    g.cap2 = (uint32_t)kv_round_up((uint64_t)(m2 * slack2 + 8.0 * std::sqrt(m2)) + 32, 16);
    g.loose_cap = (uint64_t)(rec_est / 8.0) + n_kmers_exp / 16 + (1u << 20);
    const size_t rb = (size_t)g.recw * 8;
    const size_t b_seg2 = kv_round_up((uint64_t)g.n_buckets * g.nwg2 * g.cap2 * rb, 256), b_cnt2 = kv_round_up((uint64_t)g.n_buckets * g.nwg2 * 4, 256);
    const size_t b_loose = kv_round_up(g.loose_cap * (size_t)g.lrecw * 8, 256), b_ctr = 256;
    const size_t b_off = compact ? mex_scan_bytes((uint64_t)Cl * nseg) : 0;
    KV_HIP(idx.arena.need(b_seg2 + b_cnt2 + b_loose + b_off + b_ctr));
    unsigned char *base = (unsigned char *)idx.arena.p;
    g.seg2 = (uint64_t *)base; base += b_seg2;
    g.cnt2 = (uint32_t *)base; base += b_cnt2;
    g.loose = (uint64_t *)base; base += b_loose;
    uint64_t *d_off = compact ? (uint64_t *)base : nullptr; base += b_off;
    g.ctr = (unsigned long long *)base;
    KV_HIP(hipMemsetAsync(g.ctr, 0, b_ctr, st));
    g.bucket_kmers = std::max<uint64_t>(1, n_kmers_exp / g.n_buckets);
    {
        // buckets the geometry could not make small enough (plan->F2 at its limit): k_skm_route combines them in passes, each taking
        // the k-mers of one hash class -- as many passes as bring a pass's distinct k-mers (a fifth of the occurrences at sequencing
        // coverage) under 0.65 of the LDS table.  KV_MEX_PASSES=n (tests): at least that many whatever the size.
        const double distinct = (double)g.bucket_kmers * 0.2, room = 0.5 * (g.kw == 1 ? 4096.0 : 2048.0);
        // (the least any bucket gets -- k_skm_route goes by a bucket's own record count: what the average bucket needs at 0.65 of the table)
        g.passes = distinct > room ? std::min<uint32_t>(16u, std::max<uint32_t>(2u, (uint32_t)std::ceil(distinct / (0.65 * 2.0 * room)))) : 1u;
        if (const char *e = kv_knob("KV_MEX_PASSES")) { const int v = atoi(e); if (v >= 1 && v <= 16) g.passes = (uint32_t)v; }
    }
    if (compact) {
        // the sources sent only the filled part of their segments, in segment order: a segment starts where the counts in
        // front of it end
        mex_scan_launch(g.cnt1, (uint64_t)Cl * nseg, g.cap1, d_off, st);
        g.seg1_off = d_off;
    }
    // how many k-mer occurrences arrived (the caller's buffer must hold a pair for each in the worst case)
    hipLaunchKernelGGL(k_mex_sum_kmers, dim3(1024), dim3(256), 0, st, g.seg1, g.cnt1, g.seg1_off, (uint64_t)Cl * nseg, g.cap1, (uint32_t)g.recw, &g.ctr[8]);
    skm_launch_split(g, st);
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(&g.ctr[6], &g.ctr[0], 8, hipMemcpyDeviceToDevice, st));
    skm_pick_bpt(g);
    const uint32_t nwg3 = skm_nwg3(g);
    {
        const uint64_t avg = (g.n_buckets + nwg3 - 1) / nwg3;
        g.quota3 = (uint32_t)kv_round_up(avg + avg / 2 + 1, g.bpt);
    }
    KvRouteSink rs;
    memset(&rs, 0, sizeof(rs));
    { const int rc = alloc(ctx, nwg3, &rs); if (rc != KV_OK) return rc; }
    const HashParams hp = make_hash_params(plan->ksize, HF_MURMUR);
    // keep_scan (the case sample): key + hash of every distinct k-mer stay, bucket by bucket, and the buckets themselves stay where
    // they are -- this rank will answer the scan for its buckets (kv_skm_mex_scan_set).  A stretch holds one and a half average shares
    // of 0.45 distinct k-mers per occurrence; one that runs out drops the list (ctr[9]) and the scan goes the other way.
    bool dl_new = false;
    const char *dl_why = "not asked for";
    // Room: 0.72 entries per occurrence in one stretch per workgroup covers a shard's 4-15 x with slack for uneven shares.  An owner of
    // config 4 holds 7.9 G occurrences at 30 x, a fifth of them distinct, and 16 bytes for each of 0.72 x 7.9 G entries is memory it does
    // not have: half of that next, and last a POOL with room for 0.22 -- chunks the workgroups draw one after the other (ctr[13]), so that
    // what one workgroup needs beyond its even share comes out of what another leaves (a quarter more than the even share per workgroup
    // was not enough there: 12-base minimizers make a million buckets far from even).  KV_MEX_DL_POOL=1 (tests): the pool at once.
    static const double dl_room[3] = {0.72, 0.36, 0.22};
    const bool pool_first = kv_knob("KV_MEX_DL_POOL") && atoi(kv_knob("KV_MEX_DL_POOL")) != 0;
    for (int attempt = pool_first ? 2 : 0; keep_scan && attempt < 3 && !dl_new; ++attempt) {
        const bool pool = attempt == 2;
        const double room = pool && pool_first ? dl_room[0] : dl_room[attempt];
        uint64_t cap_wg = std::min<uint64_t>((uint64_t)((double)n_kmers_exp * room / nwg3) + 4096, 0xfffffff0ull / nwg3);
        uint64_t entries = cap_wg * nwg3, chunk = 0, nchunks = 0;
        if (pool) {
            // a chunk holds any bucket's list (a table's worth per pass) four times over and is an eighth of a workgroup's even share
            // where that is more; every workgroup leaves its last chunk part empty: one chunk each on top
            chunk = std::min<uint64_t>(std::max<uint64_t>(entries / ((uint64_t)nwg3 * 8), 4ull * std::max<uint32_t>(g.passes, 2u) * 4096ull), 1ull << 20) & ~255ull;
            nchunks = std::min<uint64_t>(entries / chunk + nwg3, 0xfffffff0ull / chunk);
            entries = nchunks * chunk;
            cap_wg = 0;
        }
        const size_t b_keys = kv_round_up(entries * 8 * g.kw, 256), b_hash = kv_round_up(entries * 8, 256);
        const size_t b_idx = kv_round_up((uint64_t)g.n_buckets * 4, 256);
        const size_t b_all = b_keys + b_hash + 2 * b_idx;
        if (idx.dl.bytes == 0) idx.dl_refused = SIZE_MAX;
        if (b_all > idx.dl.bytes) {
            // not worth asking: refused before, or more than the device has left beside what the arena itself would give back
            size_t mem_free = 0, mem_total = 0;
            if (b_all >= idx.dl_refused || (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && b_all > mem_free + idx.dl.bytes)) {
                dl_why = "no memory for it";
                continue;
            }
        }
        if (idx.dl.need(b_all) == hipSuccess) {
            unsigned char *dbase = (unsigned char *)idx.dl.p;
            idx.dl_keys = (uint64_t *)dbase; dbase += b_keys;
            idx.dl_hash = (uint64_t *)dbase; dbase += b_hash;
            idx.dl_bstart = (uint32_t *)dbase; dbase += b_idx;
            idx.dl_bcount = (uint32_t *)dbase;
            idx.dl_cap_wg = (uint32_t)cap_wg;
            KV_HIP(hipMemsetAsync(idx.dl_bstart, 0, 2 * b_idx, st));
            g.dl_keys = idx.dl_keys; g.dl_hash = idx.dl_hash; g.dl_bstart = idx.dl_bstart; g.dl_bcount = idx.dl_bcount; g.dl_cap_wg = idx.dl_cap_wg;
            g.dl_chunk = (uint32_t)chunk; g.dl_nchunks = (uint32_t)nchunks;
            dl_new = true;
            dl_why = attempt == 0 ? "kept" : attempt == 1 ? "kept (room for 0.36 entries per occurrence)" : "kept (a pool of chunks the workgroups draw from)";
        } else {
            (void)hipGetLastError();                    // no room: no list, the scan goes the other way
            dl_why = "no memory for it";
            idx.dl_refused = std::min(idx.dl_refused, b_all);
        }
    }
    {
        KvProfScope prof("k_skm_route");
        const size_t lds = (256 + (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(g.sbw)) * 4;
        if (g.compact) {
            if (g.oriented) hipLaunchKernelGGL((k_skm_route<1, 4096, true, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
            else hipLaunchKernelGGL((k_skm_route<1, 4096, false, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
        } else if (g.oriented) {
            if (g.kw == 1) hipLaunchKernelGGL((k_skm_route<1, 4096, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
            else hipLaunchKernelGGL((k_skm_route<2, 2048, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
        } else if (g.kw == 1) hipLaunchKernelGGL((k_skm_route<1, 4096>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
        else hipLaunchKernelGGL((k_skm_route<2, 2048>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, g, hp, rs);
    }
    {
        KvProfScope prof("k_skm_loose_route");
        if (g.kw == 1) hipLaunchKernelGGL(k_skm_loose_route<1>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, g, hp, rs);
        else hipLaunchKernelGGL(k_skm_loose_route<2>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, g, hp, rs);
    }
    KV_HIP(hipGetLastError());
    // the packing of the pairs (the caller's `after`) is queued behind the route kernels and everything is waited for once: its
    // kernels bound what they write by the output's capacity, so what the counters say can be judged afterwards
    if (after) { const int rc = after(ctx); if (rc != KV_OK) return rc; }
    KvReadback back;
    hipError_t rb_err = hipSuccess;
    const unsigned long long *sctr = back.add(g.ctr, 14, st, &rb_err);
    KV_HIP(rb_err);
    KV_HIP(back.wait(st));
    const unsigned long long arrived = sctr[8];
    *n_kmers_in = arrived;
    idx.mex_scan_ready = dl_new && sctr[9] == 0 && sctr[1] == 0;
    idx.k = plan->ksize;
    if (kv_knob("KV_SKM_VERBOSE"))
        fprintf(stderr, "[kv_skm] exchange owner: %llu k-mers arrived in %u buckets (%u passes, segments of %u records), %.1f%% distinct, %.2f%% outside the LDS tables, "
                        "%llu records outside their bucket's segments\n",
                arrived, g.n_buckets, g.passes, g.cap2, arrived ? 100.0 * (double)sctr[7] / (double)arrived : 0.0,
                arrived ? 100.0 * (double)(sctr[0] > sctr[6] ? sctr[0] - sctr[6] : 0) / (double)arrived : 0.0, sctr[6]);
    if (keep_scan && kv_knob("KV_SKM_VERBOSE")) {
        fprintf(stderr, "[kv_skm] exchange owner: the distinct list its scan answers from: %s\n",
                !dl_new ? dl_why : sctr[9] != 0 ? (g.dl_chunk ? "dropped (the pool of chunks ran out)" : "dropped (a workgroup's stretch ran out)")
                                 : sctr[1] != 0 ? "dropped (loose list overflow)" : dl_why);
        if (dl_new && g.dl_chunk)
            fprintf(stderr, "[kv_skm] exchange owner: %llu of %u chunks of %u entries drawn for %llu distinct k-mers\n", sctr[13], g.dl_nchunks, g.dl_chunk, sctr[7]);
    }
    if (sctr[1] != 0) {
        kv_set_error("kv_mex_route: loose record list overflow (%llu records)", sctr[0]);
        return KV_ERR_CAPACITY;
    }
    return KV_OK;
}


// The hits of this rank's minimizer buckets against the gathered set of interesting hashes (p.set_*): see k_skm_set_hits.  Needs the
// state kv_skm_mex_route(keep_scan) left on this stream; KV_ERR_CAPACITY when it is not there (or a bucket holds more members than its
// table takes, or the hits do not fit): the caller then scans its shard against the set (kv_novel_scan_set) as before.
int kv_skm_mex_scan_set(const NovelParams &p, int ksize, uint64_t *d_tags, uint8_t *d_abund, uint64_t cap, uint64_t *n_hits)
{
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    if (!idx.mex_scan_ready || idx.k != ksize) {
        kv_set_error("kv_mex_scan_set: no combined buckets with a distinct list on this stream (kv_mex_route with keep_scan, nothing bucketed since)");
        return KV_ERR_CAPACITY;
    }
    SkmGeom sg = idx.g;
    sg.dl_keys = idx.dl_keys; sg.dl_hash = idx.dl_hash; sg.dl_bstart = idx.dl_bstart; sg.dl_bcount = idx.dl_bcount; sg.dl_cap_wg = idx.dl_cap_wg;
    KV_HIP(hipMemsetAsync(&sg.ctr[4], 0, 8, st));
    KV_HIP(hipMemsetAsync(&sg.ctr[10], 0, 8, st));
    SetHitSink out;
    out.tags = (unsigned long long *)d_tags; out.abund = d_abund; out.count = &sg.ctr[10]; out.cap = cap;
    const uint32_t nwg3 = skm_nwg3(sg);
    {
        KvProfScope prof("k_skm_set_hits");
        const size_t lds = (size_t)(SKM_THREADS3 / 64) * skm_wave_scratch_words(sg.sbw) * 4;
        if (sg.oriented) {
            if (sg.kw == 1) hipLaunchKernelGGL((k_skm_set_hits<1, 2048, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, p, out);
            else hipLaunchKernelGGL((k_skm_set_hits<2, 1024, true>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, p, out);
        } else if (sg.kw == 1) hipLaunchKernelGGL((k_skm_set_hits<1, 2048>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, p, out);
        else hipLaunchKernelGGL((k_skm_set_hits<2, 1024>), dim3(nwg3), dim3(SKM_THREADS3), lds, st, sg, p, out);
        if (sg.kw == 1) hipLaunchKernelGGL(k_skm_loose_set_hits<1>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, p, out);
        else hipLaunchKernelGGL(k_skm_loose_set_hits<2>, dim3(SKM_LOOSE_WGS), dim3(256), 0, st, sg, p, out);
    }
    KV_HIP(hipGetLastError());
    KvReadback back;
    hipError_t rb_err = hipSuccess;
    const unsigned long long *c = back.add(sg.ctr, 11, st, &rb_err);
    KV_HIP(rb_err);
    KV_HIP(back.wait(st));
    if (c[1] != 0) {
        idx.mex_scan_ready = false;
        kv_set_error("kv_mex_scan_set: a bucket holds more members of the set than its table takes");
        return KV_ERR_CAPACITY;
    }
    KV_REQUIRE(c[10] <= cap, KV_ERR_CAPACITY, "kv_mex_scan_set: %llu hits exceed the buffer of %llu", c[10], (unsigned long long)cap);
    *n_hits = c[10];
    return KV_OK;
}

// the filled part of a shard's exchange segments, destination after destination (what actually travels): d_out receives the
// records, records_per_dest[d] how many go to rank d.  The counts slab travels as it is.
int kv_skm_mex_pack(const kv_mex_plan *plan, const uint64_t *d_seg, const uint32_t *d_cnt, uint64_t *d_out, uint64_t *records_per_dest)
{
    hipStream_t st = kv_stream();
    SkmIndex &idx = skm_index_for(st);
    std::lock_guard<std::mutex> lk(idx.mu);
    const uint64_t n_seg = plan->cnt_entries;
    // the offsets go where the stream's bucketed batch lies: whatever that was -- a batch a scan could reuse, an owner's combined
    // buckets kept for kv_skm_mex_scan_set, a distinct list -- is gone after this call
    idx.valid = false; idx.mex_scan_ready = false; idx.dl_valid = false;
    KV_HIP(idx.arena.need(mex_scan_bytes(n_seg)));
    uint64_t *d_off = (uint64_t *)idx.arena.p;
    mex_scan_launch(d_cnt, n_seg, plan->cap1, d_off, st);
    hipLaunchKernelGGL(k_mex_gather, dim3((unsigned)std::min<uint64_t>((n_seg + 3) / 4, 1u << 20)), dim3(256), 0, st, d_seg, d_cnt, d_off, n_seg, plan->cap1,
                       plan->recw, d_out, ~0ull);
    KV_HIP(hipGetLastError());
    KvReadback rb;
    hipError_t rb_err = hipSuccess;
    std::vector<const uint64_t *> bounds(plan->ndest + 1, nullptr);
    for (int d = 0; d <= plan->ndest; ++d) bounds[d] = rb.add(d_off + (uint64_t)plan->c_lo[d] * plan->nwg1, 1, st, &rb_err);
    KV_HIP(rb_err);
    KV_HIP(rb.wait(st));
    for (int d = 0; d < plan->ndest; ++d) records_per_dest[d] = *bounds[d + 1] - *bounds[d];
    return KV_OK;
}
