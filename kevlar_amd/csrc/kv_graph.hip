// kv_graph.hip -- (1) read-graph connected components for `kevlar partition`
// (kevlar/readgraph.py:43-84,104-137) as a hash-grouped GPU union-find, and (2) the exact
// single-thread value of khmer's n_unique_kmers() (kevlar/count.py:82-84 log line).
//
// Read graph: nodes = reads (by name), a canonical interesting k-mer links every node that
// contains it when minabund <= #nodes <= maxabund.  Instead of materialising the O(n^2)
// edges per k-mer the kernels (a) build each annotation's canonical 2-bit key, (b) group
// equal keys through an open-addressing table whose slots store the index of the first
// annotation that claimed them (keys are immutable, so no multi-word publish race),
// (c) count distinct (k-mer, node) pairs, (d) union every node of a retained k-mer with that
// k-mer's representative node (lock-free hooking, larger root under smaller), (e) flatten.
#include <algorithm>
#include <map>
#include <mutex>

#include "kv_binned.h"
#include "kv_device.h"

namespace {

#define KEY_WORDS 4  // canonical 2-bit k-mer, k <= 128, most significant word first
#define EMPTY32 0xFFFFFFFFu
#define EMPTY64 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint32_t packed_base(const uint32_t *words, uint64_t w0, uint32_t j)
{
    return (words[w0 + (j >> 4)] >> (2 * (j & 15))) & 3u;
}

__global__ void k_ann_keys(const uint32_t *words, const uint64_t *woff, const uint32_t *ann_read,
                           const uint32_t *ann_off, uint64_t n, int k, uint64_t *keys)
{
    for (uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; a < n; a += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w0 = woff[ann_read[a]];
        const uint32_t off = ann_off[a];
        uint64_t f[KEY_WORDS] = {0, 0, 0, 0}, r[KEY_WORDS] = {0, 0, 0, 0};
        for (int j = 0; j < k; ++j) {
            const uint64_t c = packed_base(words, w0, off + (uint32_t)j);
            // forward: shift the 256-bit value left by 2, append c
#pragma unroll
            for (int w = 0; w < KEY_WORDS - 1; ++w) f[w] = (f[w] << 2) | (f[w + 1] >> 62);
            f[KEY_WORDS - 1] = (f[KEY_WORDS - 1] << 2) | c;
            // reverse complement: complement lands at base position j counted from the END
            const int bit = 2 * j, word = KEY_WORDS - 1 - (bit >> 6);
#pragma unroll
            for (int w = 0; w < KEY_WORDS; ++w) r[w] |= (w == word) ? ((3ull - c) << (bit & 63)) : 0ull;
        }
        bool f_less = false, decided = false;
#pragma unroll
        for (int w = 0; w < KEY_WORDS; ++w)
            if (!decided && f[w] != r[w]) { f_less = f[w] < r[w]; decided = true; }
        const bool use_f = f_less || !decided;
#pragma unroll
        for (int w = 0; w < KEY_WORDS; ++w) keys[a * KEY_WORDS + w] = use_f ? f[w] : r[w];
    }
}

__device__ __forceinline__ uint64_t key_hash(const uint64_t *key)
{
    uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
    for (int w = 0; w < KEY_WORDS; ++w) h = fmix64(h ^ key[w]) + 0x632be59bd9b4e019ull * (uint64_t)(w + 1);
    return h;
}

__global__ void k_group(const uint64_t *keys, uint64_t n, uint32_t *owner, uint64_t capmask, uint32_t *grp)
{
    for (uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; a < n; a += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t *ka = keys + a * KEY_WORDS;
        uint64_t s = key_hash(ka) & capmask;
        for (;;) {
            const uint32_t prev = atomicCAS(&owner[s], EMPTY32, (uint32_t)a);
            if (prev == EMPTY32) { grp[a] = (uint32_t)a; break; }
            const uint64_t *kb = keys + (uint64_t)prev * KEY_WORDS;
            bool same = true;
#pragma unroll
            for (int w = 0; w < KEY_WORDS; ++w) same &= ka[w] == kb[w];
            if (same) { grp[a] = prev; break; }
            s = (s + 1) & capmask;
        }
    }
}

// distinct (k-mer group, node) pairs; per group: node count and a linked list of its pairs
__global__ void k_pairs(const uint32_t *grp, const uint32_t *ann_read, const uint32_t *node_of_read, uint64_t n,
                        unsigned long long *pairset, uint64_t capmask, uint32_t *cnt, uint32_t *head, uint32_t *next,
                        unsigned long long *pairs, unsigned long long *npairs)
{
    for (uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; a < n; a += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = grp[a];
        const uint32_t node = node_of_read[ann_read[a]];
        const unsigned long long pair = ((unsigned long long)g << 32) | node;
        uint64_t s = fmix64(pair) & capmask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&pairset[s], EMPTY64, pair);
            if (prev == EMPTY64) {
                atomicAdd(&cnt[g], 1u);
                const unsigned long long idx = atomicAdd(npairs, 1ull);
                pairs[idx] = pair;
                next[idx] = atomicExch(&head[g], (uint32_t)idx);
                break;
            }
            if (prev == pair) break;
            s = (s + 1) & capmask;
        }
    }
}

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
    uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {
        const uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) atomicCAS(&parent[x], p, gp);  // path halving; failure is harmless
        x = p;
        p = gp;
    }
    return x;
}

__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (;;) {
        uint32_t ra = uf_find(parent, a), rb = uf_find(parent, b);
        if (ra == rb) return;
        if (ra > rb) { const uint32_t t = ra; ra = rb; rb = t; }
        if (atomicCAS(&parent[rb], rb, ra) == rb) return;  // hook the larger root under the smaller
    }
}

__device__ __forceinline__ bool group_retained(uint32_t c, uint32_t minabund, uint32_t maxabund)
{
    return !(minabund && c < minabund) && !(maxabund && c > maxabund);
}

__global__ void k_union(const unsigned long long *pairs, uint64_t npairs, const uint32_t *cnt, const uint32_t *ann_read,
                        const uint32_t *node_of_read, uint32_t minabund, uint32_t maxabund, uint32_t *parent)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < npairs; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t g = (uint32_t)(pairs[i] >> 32), node = (uint32_t)pairs[i];
        if (!group_retained(cnt[g], minabund, maxabund)) continue;
        uf_union(parent, node, node_of_read[ann_read[g]]);
    }
}

__global__ void k_iota(uint32_t *p, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}

__global__ void k_flatten(uint32_t *parent, uint32_t n, uint32_t *labels)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) labels[i] = uf_find(parent, i);
}

// distinct node pairs sharing a retained k-mer (networkx number_of_edges in relaxed mode)
__global__ void k_edges(const uint32_t *grp, uint64_t n, const uint32_t *cnt, const uint32_t *head, const uint32_t *next,
                        const unsigned long long *pairs, uint32_t minabund, uint32_t maxabund,
                        unsigned long long *edgeset, uint64_t capmask, unsigned long long *nedges)
{
    for (uint64_t a = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; a < n; a += (uint64_t)gridDim.x * blockDim.x) {
        if (grp[a] != (uint32_t)a || !group_retained(cnt[a], minabund, maxabund)) continue;
        for (uint32_t i = head[a]; i != EMPTY32; i = next[i]) {
            const uint32_t u = (uint32_t)pairs[i];
            for (uint32_t j = next[i]; j != EMPTY32; j = next[j]) {
                const uint32_t v = (uint32_t)pairs[j];
                if (u == v) continue;
                const unsigned long long e = u < v ? (((unsigned long long)u << 32) | v) : (((unsigned long long)v << 32) | u);
                uint64_t s = fmix64(e) & capmask;
                for (;;) {
                    const unsigned long long prev = atomicCAS(&edgeset[s], EMPTY64, e);
                    if (prev == EMPTY64) { atomicAdd(nedges, 1ull); break; }
                    if (prev == e) break;
                    s = (s + 1) & capmask;
                }
            }
        }
    }
}

// ---- exact n_unique_kmers -------------------------------------------------------------
struct FirstTouchParams {
    ConsumeFilter f;
    uint64_t ordinal_base;
    const uint64_t *kprefix;  // per read: k-mers before it in this batch
    uint32_t *first[KV_MAX_TABLES];
};

__global__ __launch_bounds__(KV_TILE_THREADS) void k_first_touch(ReadsDev rd, const SketchDev *__restrict__ sk,
                                                                const SketchDev *__restrict__ mask, FirstTouchParams p)
{
    __shared__ TileShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)tile_smem;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.f.hp.k, 0, 0, read0);
    const uint32_t total = sh.kpre[nr];
    for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
        uint32_t r, i;
        locate_kmer(sh, nr, q, r, i);
        const uint32_t fwd = sh.foff[r] + i;
        const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.f.hp.k - i);
        const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, p.f.hp);
        if (!consume_filter_pass(p.f, mask, h)) continue;
        const uint32_t ordinal = (uint32_t)(p.ordinal_base + p.kprefix[read0 + r] + sh.seg_start + i);
        for (int t = 0; t < sk->ntables; ++t)
            atomicMin(&p.first[t][fastmod(h, sk->size[t], sk->magic[t])], ordinal);
    }
}

__global__ void k_mark_first(const uint32_t *first, uint64_t nbins, uint32_t *bitmap)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nbins; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = first[i];
        if (o != EMPTY32) atomicOr(&bitmap[o >> 5], 1u << (o & 31));
    }
}

__global__ void k_popcount(const uint32_t *w, uint64_t n, unsigned long long *out)
{
    uint64_t c = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) c += __popc(w[i]);
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// ---- kevlar dist: abundance distribution over first occurrences ---------------------------------
__device__ __forceinline__ uint32_t bin_value(const SketchDev *s, int t, uint64_t bin)
{
    const uint8_t *tab = s->tab[t];
    if (s->storage == ST_BYTE) return tab[bin];
    if (s->storage == ST_NIBBLE) return (tab[bin >> 1] >> ((bin & 1) ? 0 : 4)) & 15u;
    return (tab[bin >> 3] >> (bin & 7)) & 1u;
}

// a bin's first toucher is a NEW k-mer only if the tracking table had not recorded the bin before this call
__global__ void k_mark_first_untracked(const uint32_t *first, const SketchDev *__restrict__ tracking, int t, uint32_t *bitmap)
{
    const uint64_t nbins = tracking->size[t];
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nbins; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = first[i];
        if (o != EMPTY32 && bin_value(tracking, t, i) == 0) atomicOr(&bitmap[o >> 5], 1u << (o & 31));
    }
}

struct AbundParams {
    HashParams hp;
    uint64_t ordinal_base;
    const uint64_t *kprefix;
    const uint32_t *bitmap;          // bit per k-mer ordinal: first occurrence of a k-mer new to tracking
    unsigned long long *hist;        // 256 entries (counts saturate at 255 / 15 / 1)
};

__global__ __launch_bounds__(KV_TILE_THREADS) void k_abund_hist(ReadsDev rd, const SketchDev *__restrict__ counts, AbundParams p)
{
    __shared__ TileShared sh;
    __shared__ uint32_t lhist[256];
    extern __shared__ __attribute__((aligned(16))) unsigned char tile_smem[];
    if (threadIdx.x == 0) sh.ascii = (uint32_t *)tile_smem;
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) lhist[i] = 0;
    uint32_t read0;
    const uint32_t nr = stage_tile(sh, rd, blockIdx.x, p.hp.k, 0, 0, read0);
    const uint32_t total = sh.kpre[nr];
    for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
        uint32_t r, i;
        locate_kmer(sh, nr, q, r, i);
        const uint64_t ordinal = p.ordinal_base + p.kprefix[read0 + r] + sh.seg_start + i;
        if (!((p.bitmap[ordinal >> 5] >> (ordinal & 31)) & 1u)) continue;
        const uint32_t fwd = sh.foff[r] + i;
        const uint32_t rc = sh.roff[r] + (sh.len[r] - (uint32_t)p.hp.k - i);
        const uint64_t h = kmer_hash_lds(sh.ascii, fwd, rc, p.hp);
        atomicAdd(&lhist[sketch_get(counts, h) & 255u], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x)
        if (lhist[i]) atomicAdd(&p.hist[i], (unsigned long long)lhist[i]);
}

struct DevBuf {  // frees on scope exit
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return kv_hip_malloc(&p, n ? n : 4); }
    template <typename T> T *as() { return (T *)p; }
};

inline unsigned grid_for(uint64_t n) { return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 255) / 256, 4096)); }
inline uint64_t pow2_at_least(uint64_t n) { uint64_t c = 1024; while (c < n) c <<= 1; return c; }

}  // namespace

extern "C" int kv_readgraph_components(const kv_reads *reads, int ksize, const uint32_t *ann_read,
                                       const uint32_t *ann_offset, uint64_t n_ann, const uint32_t *node_of_read,
                                       uint32_t n_nodes, uint32_t minabund, uint32_t maxabund, uint32_t *labels_out,
                                       uint64_t *n_edges_out)
{
    KV_REQUIRE(reads && labels_out && (n_ann == 0 || (ann_read && ann_offset)) && node_of_read, KV_ERR_ARG,
               "kv_readgraph_components: null argument");
    KV_REQUIRE(ksize >= 1 && ksize <= 32 * KEY_WORDS, KV_ERR_ARG, "partition supports k <= %d (got %d)", 32 * KEY_WORDS, ksize);
    KV_REQUIRE(n_ann < 0xFFFFFFF0ull, KV_ERR_ARG, "too many annotations");
    for (uint64_t a = 0; a < n_ann; ++a) {
        KV_REQUIRE(ann_read[a] < reads->n_reads, KV_ERR_ARG, "annotation %llu names read %u of %llu",
                   (unsigned long long)a, ann_read[a], (unsigned long long)reads->n_reads);
        KV_REQUIRE((uint64_t)ann_offset[a] + (uint64_t)ksize <= reads->h_len[ann_read[a]], KV_ERR_ARG,
                   "annotation %llu runs past the end of its read", (unsigned long long)a);
    }
    for (uint64_t r = 0; r < reads->n_reads; ++r)
        KV_REQUIRE(node_of_read[r] < n_nodes, KV_ERR_ARG, "node id %u out of range", node_of_read[r]);
    hipStream_t st = kv_stream();
    DevBuf d_ar, d_ao, d_nr, d_keys, d_owner, d_grp, d_pairset, d_cnt, d_head, d_next, d_pairs, d_np, d_parent, d_labels;
    const uint64_t cap = pow2_at_least(2 * n_ann + 16);
    hipError_t e = d_ar.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_ao.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_nr.alloc(reads->n_reads * 4);
    if (e == hipSuccess) e = d_keys.alloc(n_ann * KEY_WORDS * 8);
    if (e == hipSuccess) e = d_owner.alloc(cap * 4);
    if (e == hipSuccess) e = d_grp.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_pairset.alloc(cap * 8);
    if (e == hipSuccess) e = d_cnt.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_head.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_next.alloc(n_ann * 4);
    if (e == hipSuccess) e = d_pairs.alloc(n_ann * 8);
    if (e == hipSuccess) e = d_np.alloc(16);
    if (e == hipSuccess) e = d_parent.alloc((uint64_t)n_nodes * 4);
    if (e == hipSuccess) e = d_labels.alloc((uint64_t)n_nodes * 4);
    KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "read graph allocation failed: %s", hipGetErrorString(e));
    if (n_ann) {
        KV_HIP(hipMemcpyAsync(d_ar.p, ann_read, n_ann * 4, hipMemcpyHostToDevice, st));
        KV_HIP(hipMemcpyAsync(d_ao.p, ann_offset, n_ann * 4, hipMemcpyHostToDevice, st));
    }
    if (reads->n_reads) KV_HIP(hipMemcpyAsync(d_nr.p, node_of_read, reads->n_reads * 4, hipMemcpyHostToDevice, st));
    KV_HIP(hipMemsetAsync(d_owner.p, 0xFF, cap * 4, st));
    KV_HIP(hipMemsetAsync(d_pairset.p, 0xFF, cap * 8, st));
    KV_HIP(hipMemsetAsync(d_cnt.p, 0, n_ann ? n_ann * 4 : 4, st));
    KV_HIP(hipMemsetAsync(d_head.p, 0xFF, n_ann ? n_ann * 4 : 4, st));
    KV_HIP(hipMemsetAsync(d_np.p, 0, 16, st));
    uint64_t npairs = 0;
    {
        KvProfScope prof("k_readgraph");
        hipLaunchKernelGGL(k_iota, dim3(grid_for(n_nodes)), dim3(256), 0, st, d_parent.as<uint32_t>(), n_nodes);
        if (n_ann) {
            hipLaunchKernelGGL(k_ann_keys, dim3(grid_for(n_ann)), dim3(256), 0, st, reads->d_words, reads->d_woff,
                               d_ar.as<uint32_t>(), d_ao.as<uint32_t>(), n_ann, ksize, d_keys.as<uint64_t>());
            hipLaunchKernelGGL(k_group, dim3(grid_for(n_ann)), dim3(256), 0, st, d_keys.as<uint64_t>(), n_ann,
                               d_owner.as<uint32_t>(), cap - 1, d_grp.as<uint32_t>());
            hipLaunchKernelGGL(k_pairs, dim3(grid_for(n_ann)), dim3(256), 0, st, d_grp.as<uint32_t>(), d_ar.as<uint32_t>(),
                               d_nr.as<uint32_t>(), n_ann, d_pairset.as<unsigned long long>(), cap - 1, d_cnt.as<uint32_t>(),
                               d_head.as<uint32_t>(), d_next.as<uint32_t>(), d_pairs.as<unsigned long long>(),
                               d_np.as<unsigned long long>());
            KV_HIP(hipMemcpyAsync(&npairs, d_np.p, 8, hipMemcpyDeviceToHost, st));
            KV_HIP(hipStreamSynchronize(st));
            hipLaunchKernelGGL(k_union, dim3(grid_for(npairs)), dim3(256), 0, st, d_pairs.as<unsigned long long>(), npairs,
                               d_cnt.as<uint32_t>(), d_ar.as<uint32_t>(), d_nr.as<uint32_t>(), minabund, maxabund,
                               d_parent.as<uint32_t>());
        }
        hipLaunchKernelGGL(k_flatten, dim3(grid_for(n_nodes)), dim3(256), 0, st, d_parent.as<uint32_t>(), n_nodes,
                           d_labels.as<uint32_t>());
    }
    KV_HIP(hipGetLastError());
    if (n_nodes) KV_HIP(hipMemcpyAsync(labels_out, d_labels.p, (uint64_t)n_nodes * 4, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    if (n_edges_out) {
        *n_edges_out = 0;
        if (n_ann) {
            std::vector<uint32_t> cnt(n_ann), grp(n_ann);
            KV_HIP(hipMemcpy(cnt.data(), d_cnt.p, n_ann * 4, hipMemcpyDeviceToHost));
            KV_HIP(hipMemcpy(grp.data(), d_grp.p, n_ann * 4, hipMemcpyDeviceToHost));
            uint64_t bound = 0;
            for (uint64_t a = 0; a < n_ann; ++a)
                if (grp[a] == a && !(minabund && cnt[a] < minabund) && !(maxabund && cnt[a] > maxabund))
                    bound += (uint64_t)cnt[a] * (cnt[a] - 1) / 2;
            KV_REQUIRE(bound < (1ull << 31), KV_ERR_CAPACITY, "edge count bound %llu too large to enumerate",
                       (unsigned long long)bound);
            const uint64_t ecap = pow2_at_least(2 * bound + 16);
            DevBuf d_edges;
            KV_HIP(d_edges.alloc(ecap * 8));
            KV_HIP(hipMemsetAsync(d_edges.p, 0xFF, ecap * 8, st));
            KV_HIP(hipMemsetAsync(d_np.p, 0, 16, st));
            hipLaunchKernelGGL(k_edges, dim3(grid_for(n_ann)), dim3(256), 0, st, d_grp.as<uint32_t>(), n_ann, d_cnt.as<uint32_t>(),
                               d_head.as<uint32_t>(), d_next.as<uint32_t>(), d_pairs.as<unsigned long long>(), minabund,
                               maxabund, d_edges.as<unsigned long long>(), ecap - 1, d_np.as<unsigned long long>());
            KV_HIP(hipGetLastError());
            KV_HIP(hipMemcpyAsync(n_edges_out, d_np.p, 8, hipMemcpyDeviceToHost, st));
            KV_HIP(hipStreamSynchronize(st));
        }
    }
    return KV_OK;
}

extern "C" int kv_unique_exact(kv_sketch *s, const kv_reads *const *batches, int n_batches, int nbands, int band,
                               const kv_sketch *mask, int threshold, int consume_masked, uint64_t *n_unique_out)
{
    KV_REQUIRE(s && n_unique_out && (batches || n_batches == 0) && n_batches >= 0, KV_ERR_ARG, "kv_unique_exact: bad argument");
    KV_REQUIRE(nbands >= 0 && (nbands == 0 || (band >= 0 && band < nbands)), KV_ERR_ARG,
               "band %d out of range for %d bands", band, nbands);
    { const int rc = kv_sketch_ready(mask); if (rc != KV_OK) return rc; }
    std::lock_guard<std::mutex> lk(s->mu);
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    hipStream_t st = kv_stream();
    const int k = s->h.ksize;
    uint64_t total = 0;
    for (int b = 0; b < n_batches; ++b) {
        uint64_t nk = 0;
        KV_REQUIRE(batches[b], KV_ERR_ARG, "kv_unique_exact: null batch");
        kv_reads_num_kmers(batches[b], k, &nk);
        total += nk;
    }
    KV_REQUIRE(total < 0xFFFFFFF0ull, KV_ERR_CAPACITY,
               "exact distinct k-mer counting handles up to 4.29e9 k-mers per sketch (got %llu)", (unsigned long long)total);
    *n_unique_out = 0;
    if (total == 0) return KV_OK;
    FirstTouchParams p;
    memset(&p, 0, sizeof(p));
    p.f = make_consume_filter(k, s->h.hashfam, nbands, band, mask != nullptr, threshold, consume_masked);
    std::vector<DevBuf> first(s->h.ntables);
    for (int t = 0; t < s->h.ntables; ++t) {
        hipError_t e = first[t].alloc(s->h.size[t] * 4);
        KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "first-touch scratch (%llu bytes) allocation failed: %s",
                   (unsigned long long)(s->h.size[t] * 4), hipGetErrorString(e));
        KV_HIP(hipMemsetAsync(first[t].p, 0xFF, s->h.size[t] * 4, st));
        p.first[t] = first[t].as<uint32_t>();
    }
    const uint64_t bm_words = (total + 31) / 32;
    DevBuf d_bm, d_out;
    KV_HIP(d_bm.alloc(bm_words * 4));
    KV_HIP(d_out.alloc(8));
    KV_HIP(hipMemsetAsync(d_bm.p, 0, bm_words * 4, st));
    KV_HIP(hipMemsetAsync(d_out.p, 0, 8, st));
    uint64_t base = 0;
    for (int b = 0; b < n_batches; ++b) {
        const kv_reads *r = batches[b];
        std::vector<uint64_t> kpre(r->n_reads + 1, 0);
        for (uint64_t i = 0; i < r->n_reads; ++i)
            kpre[i + 1] = kpre[i] + (r->h_len[i] >= (uint32_t)k ? r->h_len[i] - (uint32_t)k + 1 : 0);
        DevBuf d_kpre;
        KV_HIP(d_kpre.alloc(kpre.size() * 8));
        KV_HIP(hipMemcpyAsync(d_kpre.p, kpre.data(), kpre.size() * 8, hipMemcpyHostToDevice, st));
        p.ordinal_base = base;
        p.kprefix = d_kpre.as<uint64_t>();
        if (r->n_tiles) {
            KvProfScope prof("k_first_touch");
            kv_ensure_dynamic_lds((const void *)k_first_touch, r->tile_lds_bytes);
            hipLaunchKernelGGL(k_first_touch, dim3(r->n_tiles), dim3(KV_TILE_THREADS), r->tile_lds_bytes, st, reads_dev(r),
                               (const SketchDev *)s->d_desc, (const SketchDev *)(mask ? mask->d_desc : nullptr), p);
        }
        KV_HIP(hipGetLastError());
        KV_HIP(hipStreamSynchronize(st));
        base += kpre[r->n_reads];
    }
    for (int t = 0; t < s->h.ntables; ++t)
        hipLaunchKernelGGL(k_mark_first, dim3(grid_for(s->h.size[t])), dim3(256), 0, st, p.first[t], s->h.size[t], d_bm.as<uint32_t>());
    hipLaunchKernelGGL(k_popcount, dim3(grid_for(bm_words)), dim3(256), 0, st, d_bm.as<uint32_t>(), bm_words,
                       d_out.as<unsigned long long>());
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(n_unique_out, d_out.p, 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

// kv_unique_exact one batch at a time, against the sketch AS IT STANDS: how many k-mers of `batch` khmer's single thread would call
// new if the batch were consumed now -- some bin of the k-mer is clear in the tables (nothing counted so far touched it) and no earlier
// k-mer of this very batch touches it first.  Called before every kv_consume of a sketch that tracks the exact figure, the sum over the
// batches is kv_unique_exact's number without keeping any batch resident (round 5: the 4 GiB retention limit is gone; a sample of any
// size reports the reference's "distinct k-mers stored").  The first-toucher arrays (4 bytes per bin and table, ~4.5 x the sketch) come
// from a per-stream arena that is kept between the calls of one sample and given back by kv_unique_release (the wrapper calls it when
// tracking ends), kv_scratch_trim, and any allocation of the library that runs out of memory (kv_unique_scratch_release skips an
// arena whose call is still running).
namespace {
struct UniqueArena {
    KvArena a;
    std::mutex busy;            // held for the whole of a kv_unique_new call on this stream
};
std::map<hipStream_t, UniqueArena> g_unique_arena;
std::mutex g_unique_arena_mu;
thread_local const UniqueArena *tl_unique_held = nullptr;      // the arena the calling thread's own kv_unique_new holds (never try_lock a mutex one owns)
}
void kv_unique_scratch_release()
{
    std::lock_guard<std::mutex> alk(g_unique_arena_mu);
    for (auto &kv : g_unique_arena) {
        if (&kv.second == tl_unique_held) continue;     // the caller's own call is using it
        if (!kv.second.busy.try_lock()) continue;       // in use by another thread's running call: not ours to free
        kv.second.a.release();
        kv.second.busy.unlock();
    }
}
extern "C" int kv_unique_release(void)
{
    kv_unique_scratch_release();
    return KV_OK;
}
extern "C" int kv_unique_new(kv_sketch *s, const kv_reads *batch, int nbands, int band, const kv_sketch *mask, int threshold,
                             int consume_masked, uint64_t *n_new_out)
{
    KV_REQUIRE(s && batch && n_new_out, KV_ERR_ARG, "kv_unique_new: bad argument");
    KV_REQUIRE(nbands >= 0 && (nbands == 0 || (band >= 0 && band < nbands)), KV_ERR_ARG, "band %d out of range for %d bands", band, nbands);
    { const int rc = kv_sketch_ready(mask); if (rc != KV_OK) return rc; }
    std::lock_guard<std::mutex> lk(s->mu);
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    hipStream_t st = kv_stream();
    const int k = s->h.ksize;
    uint64_t total = 0;
    kv_reads_num_kmers(batch, k, &total);
    KV_REQUIRE(total < 0xFFFFFFF0ull, KV_ERR_CAPACITY, "exact distinct k-mer counting handles up to 4.29e9 k-mers per batch (got %llu)",
               (unsigned long long)total);
    *n_new_out = 0;
    if (total == 0 || batch->n_tiles == 0) return KV_OK;
    FirstTouchParams p;
    memset(&p, 0, sizeof(p));
    p.f = make_consume_filter(k, s->h.hashfam, nbands, band, mask != nullptr, threshold, consume_masked);
    UniqueArena *ua;
    {
        std::lock_guard<std::mutex> alk(g_unique_arena_mu);
        ua = &g_unique_arena[kv_stream_key(st)];
    }
    std::lock_guard<std::mutex> busy(ua->busy);
    struct Held { Held(const UniqueArena *u) { tl_unique_held = u; } ~Held() { tl_unique_held = nullptr; } } held(ua);
    KvArena *arena = &ua->a;
    const uint64_t bm_words = (total + 31) / 32;
    size_t need = kv_round_up(bm_words * 4, 256) + 256;
    for (int t = 0; t < s->h.ntables; ++t) need += kv_round_up(s->h.size[t] * 4, 256);
    {
        const hipError_t e = arena->need(need);
        KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "first-touch scratch (%llu bytes) allocation failed: %s", (unsigned long long)need, hipGetErrorString(e));
    }
    unsigned char *base = (unsigned char *)arena->p;
    for (int t = 0; t < s->h.ntables; ++t) {
        p.first[t] = (uint32_t *)base;
        KV_HIP(hipMemsetAsync(base, 0xFF, s->h.size[t] * 4, st));
        base += kv_round_up(s->h.size[t] * 4, 256);
    }
    uint32_t *d_bm = (uint32_t *)base; base += kv_round_up(bm_words * 4, 256);
    unsigned long long *d_out = (unsigned long long *)base;
    KV_HIP(hipMemsetAsync(d_bm, 0, bm_words * 4, st));
    KV_HIP(hipMemsetAsync(d_out, 0, 8, st));
    std::vector<uint64_t> kpre(batch->n_reads + 1, 0);
    for (uint64_t i = 0; i < batch->n_reads; ++i)
        kpre[i + 1] = kpre[i] + (batch->h_len[i] >= (uint32_t)k ? batch->h_len[i] - (uint32_t)k + 1 : 0);
    DevBuf d_kpre;
    KV_HIP(d_kpre.alloc(kpre.size() * 8));
    KV_HIP(hipMemcpyAsync(d_kpre.p, kpre.data(), kpre.size() * 8, hipMemcpyHostToDevice, st));
    p.ordinal_base = 0;
    p.kprefix = d_kpre.as<uint64_t>();
    {
        KvProfScope prof("k_first_touch");
        kv_ensure_dynamic_lds((const void *)k_first_touch, batch->tile_lds_bytes);
        hipLaunchKernelGGL(k_first_touch, dim3(batch->n_tiles), dim3(KV_TILE_THREADS), batch->tile_lds_bytes, st, reads_dev(batch),
                           (const SketchDev *)s->d_desc, (const SketchDev *)(mask ? mask->d_desc : nullptr), p);
    }
    // a bin's first toucher is new only if the tables had not recorded the bin before this batch
    for (int t = 0; t < s->h.ntables; ++t)
        hipLaunchKernelGGL(k_mark_first_untracked, dim3(grid_for(s->h.size[t])), dim3(256), 0, st, p.first[t], (const SketchDev *)s->d_desc, t, d_bm);
    hipLaunchKernelGGL(k_popcount, dim3(grid_for(bm_words)), dim3(256), 0, st, d_bm, bm_words, d_out);
    KV_HIP(hipGetLastError());
    KV_HIP(hipMemcpyAsync(n_new_out, d_out, 8, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    return KV_OK;
}

// khmer's Hashtable::abundance_distribution(parser, tracking), the second pass of `kevlar dist`
// (kevlar/dist.py:47-77): walking the k-mers in order, a k-mer that `tracking` has not seen
// (get == 0) is recorded there and bumps hist[counts.get(kmer)].  "Not seen" at occurrence o means
// some bin of the k-mer was clear before the call and no earlier occurrence of the call touched it --
// the same first-toucher rule as kv_unique_exact, so the result is the single-thread one.  hist_out
// has 65536 entries like khmer's (only the first 256 can be non-zero); tracking is updated.
extern "C" int kv_abundance_distribution(kv_sketch *counts, kv_sketch *tracking, const kv_reads *const *batches,
                                         int n_batches, uint64_t *hist_out)
{
    KV_REQUIRE(counts && tracking && hist_out && (batches || n_batches == 0) && n_batches >= 0, KV_ERR_ARG,
               "kv_abundance_distribution: bad argument");
    KV_REQUIRE(counts->h.ksize == tracking->h.ksize && counts->h.hashfam == tracking->h.hashfam, KV_ERR_ARG,
               "counts and tracking sketches must share k and hash function");
    memset(hist_out, 0, 65536 * sizeof(uint64_t));
    { const int rc = kv_sketch_ready(counts); if (rc != KV_OK) return rc; }
    { const int rc = kv_sketch_ready(tracking); if (rc != KV_OK) return rc; }
    hipStream_t st = kv_stream();
    const int k = counts->h.ksize;
    uint64_t total = 0;
    for (int b = 0; b < n_batches; ++b) {
        uint64_t nk = 0;
        KV_REQUIRE(batches[b], KV_ERR_ARG, "kv_abundance_distribution: null batch");
        kv_reads_num_kmers(batches[b], k, &nk);
        total += nk;
    }
    KV_REQUIRE(total < 0xFFFFFFF0ull, KV_ERR_CAPACITY,
               "abundance_distribution handles up to 4.29e9 k-mers per call (got %llu)", (unsigned long long)total);
    if (total == 0) return KV_OK;
    {
        std::lock_guard<std::mutex> lk(tracking->mu);
        FirstTouchParams p;
        memset(&p, 0, sizeof(p));
        p.f = make_consume_filter(k, tracking->h.hashfam, 0, 0, false, 0, 0);
        std::vector<DevBuf> first(tracking->h.ntables);
        for (int t = 0; t < tracking->h.ntables; ++t) {
            hipError_t e = first[t].alloc(tracking->h.size[t] * 4);
            KV_REQUIRE(e == hipSuccess, KV_ERR_HIP, "first-touch scratch (%llu bytes) allocation failed: %s",
                       (unsigned long long)(tracking->h.size[t] * 4), hipGetErrorString(e));
            KV_HIP(hipMemsetAsync(first[t].p, 0xFF, tracking->h.size[t] * 4, st));
            p.first[t] = first[t].as<uint32_t>();
        }
        const uint64_t bm_words = (total + 31) / 32;
        DevBuf d_bm, d_hist;
        KV_HIP(d_bm.alloc(bm_words * 4));
        KV_HIP(d_hist.alloc(256 * 8));
        KV_HIP(hipMemsetAsync(d_bm.p, 0, bm_words * 4, st));
        KV_HIP(hipMemsetAsync(d_hist.p, 0, 256 * 8, st));
        std::vector<DevBuf> d_kpre(n_batches);
        std::vector<uint64_t> bases(n_batches, 0);
        uint64_t base = 0;
        for (int b = 0; b < n_batches; ++b) {
            const kv_reads *r = batches[b];
            std::vector<uint64_t> kpre(r->n_reads + 1, 0);
            for (uint64_t i = 0; i < r->n_reads; ++i)
                kpre[i + 1] = kpre[i] + (r->h_len[i] >= (uint32_t)k ? r->h_len[i] - (uint32_t)k + 1 : 0);
            KV_HIP(d_kpre[b].alloc(kpre.size() * 8));
            KV_HIP(hipMemcpyAsync(d_kpre[b].p, kpre.data(), kpre.size() * 8, hipMemcpyHostToDevice, st));
            p.ordinal_base = base;
            p.kprefix = d_kpre[b].as<uint64_t>();
            bases[b] = base;
            if (r->n_tiles) {
                KvProfScope prof("k_first_touch");
                kv_ensure_dynamic_lds((const void *)k_first_touch, r->tile_lds_bytes);
                hipLaunchKernelGGL(k_first_touch, dim3(r->n_tiles), dim3(KV_TILE_THREADS), r->tile_lds_bytes, st, reads_dev(r),
                                   (const SketchDev *)tracking->d_desc, (const SketchDev *)nullptr, p);
            }
            KV_HIP(hipGetLastError());
            KV_HIP(hipStreamSynchronize(st));     // kpre (host vector) must outlive its copy
            base += kpre[r->n_reads];
        }
        for (int t = 0; t < tracking->h.ntables; ++t)
            hipLaunchKernelGGL(k_mark_first_untracked, dim3(grid_for(tracking->h.size[t])), dim3(256), 0, st, p.first[t],
                               (const SketchDev *)tracking->d_desc, t, d_bm.as<uint32_t>());
        AbundParams a;
        memset(&a, 0, sizeof(a));
        a.hp = make_hash_params(k, counts->h.hashfam);
        a.bitmap = d_bm.as<uint32_t>();
        a.hist = d_hist.as<unsigned long long>();
        for (int b = 0; b < n_batches; ++b) {
            const kv_reads *r = batches[b];
            if (!r->n_tiles) continue;
            a.ordinal_base = bases[b];
            a.kprefix = d_kpre[b].as<uint64_t>();
            KvProfScope prof("k_abund_hist");
            kv_ensure_dynamic_lds((const void *)k_abund_hist, r->tile_lds_bytes);
            hipLaunchKernelGGL(k_abund_hist, dim3(r->n_tiles), dim3(KV_TILE_THREADS), r->tile_lds_bytes, st, reads_dev(r),
                               (const SketchDev *)counts->d_desc, a);
        }
        KV_HIP(hipGetLastError());
        KV_HIP(hipMemcpyAsync(hist_out, d_hist.p, 256 * 8, hipMemcpyDeviceToHost, st));
        KV_HIP(hipStreamSynchronize(st));
    }
    // record every k-mer of the call in tracking (bins of k-mers that were not new are set already)
    for (int b = 0; b < n_batches; ++b) {
        uint64_t n = 0;
        const int rc = kv_consume(tracking, batches[b], 0, 0, nullptr, 0, 0, &n);
        if (rc != KV_OK) return rc;
    }
    return KV_OK;
}
