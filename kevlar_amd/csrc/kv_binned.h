// kv_binned.h -- geometry and host entry points of the partitioned count (kv_binned.hip), shared with the
// super-k-mer front end (kv_skm.hip), which feeds the same coarse buckets with (bin, count) items.
#pragma once
#include <map>
#include <mutex>

#include "kv_internal.h"

#define BIN_C 64            // most coarse buckets per table
#define BIN_MAX_T 4         // tables handled by the partitioned path
#define BIN_MAX_F 512       // slices per coarse bucket (9 bits of a coarse item)

// A coarse item (stage A -> stage B) is one u32.  Unweighted (one item per k-mer):
//   bits  0..15  offset of the bin inside its 65536-bin slice        bits 16..24  slice inside the coarse bucket (< F <= 512)
// Weighted (one item per distinct k-mer of a batch; saturating adds commute, so a k-mer seen c times is one item of
// weight c instead of c items): the same layout plus bits 25..31 = weight - 1.  (A 32768-bin variant exists behind
// KV_BIN_SLICE15=1 -- offset 15 bits, slice 10 bits -- it measured no faster, see kv_bin_plan.)
// A fine item (stage B -> stage C) is the u16 offset, or for weighted items offset | weight << 16.
#define BIN_W_SHIFT 25
#define BIN_W_MAX 128u
#define BIN_SLICE_BITS 16      // unweighted path
#define BIN_SLICE_BITS_W 15    // weighted path

struct BinGeom {
    int T, F, C;                     // tables, slices per coarse bucket, coarse buckets in use (<= BIN_C)
    int sbits;                       // log2 of the bins per slice (BIN_SLICE_BITS or BIN_SLICE_BITS_W)
    uint32_t ringA, ringB;           // LDS ring entries per stream in stages A / B (powers of two)
    uint32_t recipF;                 // floor(2^32 / F) + 1: slice / F by multiply-high
    uint32_t nslices[BIN_MAX_T];
    uint32_t tile_lds;               // bytes of dynamic LDS in front of the stage-A rings
    uint32_t nwgA, nwgB;             // writers per coarse bucket (stage-A workgroups) / per slice (stage-B workgroups of the bucket)
    uint32_t quotaA;                 // work units (tiles / list chunks) one stage-A workgroup may take: bounds its segments' fill
    uint64_t cap1, cap2, spill_cap;  // items per PRIVATE segment: every writer owns its own region of every stream,
                                     // so appending needs no global atomic (and no round trip) at all
    uint32_t *gbuf1;                 // [T*C][nwgA][cap1] coarse items
    uint16_t *gbuf2;                 // [T*C*F][nwgB][cap2] fine items (u16, or u32 when weighted)
    uint32_t *gcnt1, *gcnt2;         // [T*C][nwgA] / [T*C*F][nwgB] items written per segment
    unsigned long long *spill;       // bin | table << 32 | (increment - 1) << 40
    unsigned long long *ctr;         // [0] spill count, [1] overflow flag, [2] k-mers added, [3] occupancy delta, [4] work ticket
    uint64_t tsize[BIN_MAX_T];       // table sizes and base pointers by value: stage C starts without a round trip to the descriptor
    uint8_t *ttab[BIN_MAX_T];
    uint32_t dbg;                    // KV_BIN_DEBUG: timing experiments that skip parts of stage C (results are then wrong)
    int zero_tables;                 // the tables are all zero by decree (kv_sketch::lazy_zero): stage C writes every slice without loading it
    // fast4: four tables of 2^16 <= size < 2^31 bins each and less than 2^32 coarse-item slots in all: the super-k-mer count's drain then
    // takes h % size with the FP64 quotient and 32-bit remainders (the remainder's sign is bit 31) and appends its four items with 32-bit
    // index arithmetic; tmagic[t] = kv_fastmod_magic(size[t]) by value (no trip to the sketch's descriptor per k-mer)
    int fast4;
    uint64_t tmagic[BIN_MAX_T];
};

// host-side description of one partitioned count in flight
struct BinPlan {
    BinGeom g;
    int cmax;                // bucket budget the geometry was chosen for (<= 32: 512-thread stage A, else 1024)
    uint32_t threadsA;
    uint32_t maxsl;          // most slices of any table
    bool weighted;
};

#ifdef __HIPCC__
__device__ __forceinline__ void spill_item(const BinGeom &g, int t, uint64_t bin, uint32_t weight = 1u)
{
    const unsigned long long pos = atomicAdd(&g.ctr[0], 1ull);
    if (pos < g.spill_cap) g.spill[pos] = ((unsigned long long)(weight - 1u) << 40) | ((unsigned long long)t << 32) | bin;
    else g.ctr[1] = 1;
}
// the same for a whole wave at once: every lane calls, `live` lanes append their T items (one per table); ONE atomic on the
// shared counter per call.  (The counter is one address for the whole device: it takes ~90 updates per microsecond however
// they are issued -- an atomic per item, or per table, made the 9 M loose items of a k = 51 sample cost 2 ms.)
__device__ __forceinline__ void spill_items_wave(const BinGeom &g, const uint64_t *bins, uint32_t weight, bool live)
{
    const unsigned long long vote = __ballot(live);
    if (!vote) return;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)vote) - 1;
    const unsigned long long T = (unsigned long long)g.T;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&g.ctr[0], (unsigned long long)__popcll(vote) * T);
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)base, leader), hi = (uint32_t)__shfl((int)(uint32_t)(base >> 32), leader);
    base = (unsigned long long)lo | ((unsigned long long)hi << 32);
    if (live) {
        const unsigned long long first = base + (unsigned long long)__popcll(vote & ((1ull << lane) - 1ull)) * T;
#pragma unroll
        for (int t = 0; t < BIN_MAX_T; ++t) {
            if (t >= g.T) break;
            if (first + t < g.spill_cap) g.spill[first + t] = ((unsigned long long)(weight - 1u) << 40) | ((unsigned long long)t << 32) | bins[t];
            else g.ctr[1] = 1;
        }
    }
}
#endif

// grow-only device scratch; one arena per stream so host threads counting different samples do not share buffers
struct KvArena {
    void *p = nullptr;
    size_t bytes = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t need(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        kv_thread_device();
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        // an eighth of headroom: the geometry of the next batch (bucket sizes follow the previous batch's statistics) may
        // ask for a little more, and releasing and allocating gigabytes costs a hundred milliseconds
        const size_t roomy = n + n / 8;
        hipError_t e = hipMalloc(&p, roomy);
        if (e == hipSuccess) { bytes = roomy; return e; }
        (void)hipGetLastError();
        e = hipMalloc(&p, n);
        if (e != hipSuccess) {                          // the table buffers kept for future sketches are worth less than this
            (void)hipGetLastError();
            kv_table_cache_release();
            kv_unique_scratch_release();                // (skips the arena of a kv_unique_new that is itself the caller)
            e = hipMalloc(&p, n);
        }
        if (e == hipSuccess) bytes = n;
        return e;
    }
};

// device buffers of the gzip inflater (kv_gunzip.hip); gigabytes for a big file, so the FASTQ reader pools them with its own
struct KvGunzipArenas {
    KvArena comp, syms, tails, meta, window, small, crc;
    void release()
    {
        for (KvArena *a : {&comp, &syms, &tails, &meta, &window, &small, &crc})
            if (a->p) { (void)hipFree(a->p); a->p = nullptr; a->bytes = 0; }
    }
};

bool kv_bin_two_bit(const kv_sketch *s, const kv_reads *reads);      // stage A can hash this batch from its 2-bit form (k_bin_hash_2bit)
int kv_device_cus();
static inline uint64_t kv_round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// Geometry + scratch for a count of at most `n_items_max` k-mers into `s` on the calling thread's stream.
// work_units / threads: the stage-A front end's units of work and workgroup size (0 threads: pick by geometry);
// nwgA_fixed != 0 pins the number of stage-A writers (the super-k-mer front end brings its own grid).
int kv_bin_plan(kv_sketch *s, uint64_t n_items_max, int nbands, bool use_mask, uint64_t work_units, uint32_t lds_front,
                uint32_t nwgA_fixed, bool weighted, BinPlan *plan);
// stages B, C, spill + bookkeeping (n_occupied, n_unique estimate); synchronises the stream.
// n_added_fixed: the caller knows the number of k-mers added (hash lists), else it is read from ctr[2].
int kv_bin_finish(kv_sketch *s, BinPlan &plan, bool added_from_ctr, uint64_t n_added_fixed, uint64_t *n_added);
