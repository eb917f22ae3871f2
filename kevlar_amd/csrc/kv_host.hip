// kv_host.hip -- host side of libkvsketch_hip: handles, OXLI v4 file I/O, table sizing,
// read packing and the live profiler.  All table memory is HBM (hipMalloc).
#include <algorithm>
#include <functional>
#include <atomic>
#include <cstdarg>
#include <map>

#include "kv_internal.h"

// ---------------------------------------------------------------------------------------
// errors / device / stream
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
thread_local int kv_last_hip_code = 0;

void kv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

uint64_t kv_next_uid()
{
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1);
}

static thread_local hipStream_t g_stream = nullptr;   // per host thread: concurrent samples use concurrent streams
static std::atomic<int> g_device{-1};                  // kv_set_device: the GPU of this process (-1: never said -- HIP's default, device 0)
static thread_local int tl_device = -1;                // the device this thread was last switched to by the library
void kv_thread_device()
{
    const int d = g_device.load(std::memory_order_relaxed);
    if (d >= 0 && tl_device != d) {
        if (hipSetDevice(d) == hipSuccess) tl_device = d;
        else (void)hipGetLastError();
    }
}
hipStream_t kv_stream() { kv_thread_device(); return g_stream; }

extern "C" const char *kv_last_error(void) { return g_err; }
extern "C" const char *kv_version(void) { return "kvsketch-hip 0.1 (gfx950)"; }

extern "C" int kv_device_count(int *n)
{
    KV_REQUIRE(n, KV_ERR_ARG, "kv_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { (void)hipGetLastError(); c = 0; }
    *n = c;
    return KV_OK;
}

extern "C" int kv_set_device(int device)
{
    if (hipSetDevice(device) != hipSuccess) {
        kv_set_error("kv_set_device: device %d: %s", device, hipGetErrorString(hipGetLastError()));
        return KV_ERR_HIP;
    }
    g_device.store(device, std::memory_order_relaxed);     // every host thread that enters the library from now on is switched to it (kv_thread_device)
    tl_device = device;
    return KV_OK;
}

extern "C" int kv_thread_device_get(int *configured, int *this_thread, int *hip_current)
{
    // (tests: what kv_set_device said, what the library last switched the calling thread to -- -1: it has not been here --, and what HIP says;
    // the call itself switches nothing)
    if (configured) *configured = g_device.load(std::memory_order_relaxed);
    if (this_thread) *this_thread = tl_device;
    if (hip_current) { int d = -1; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = -1; } *hip_current = d; }
    return KV_OK;
}

extern "C" int kv_set_stream(void *s)
{
    kv_thread_device();
    g_stream = (hipStream_t)s;
    return KV_OK;
}

void kv_ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<const void *, size_t> granted;
    std::lock_guard<std::mutex> lk(mu);
    size_t &have = granted[kernel];
    if (bytes > have) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        have = bytes;
    }
}

extern "C" int kv_synchronize(void)
{
    KV_HIP(hipStreamSynchronize(g_stream));
    return KV_OK;
}

// The library keeps gigabytes of scratch per stream (bucketed batches, bin stages, scan arenas): whoever works on a stream finds its
// buffers from the last time.  Keyed by the raw handle that only worked when the runtime happened to hand a new stream the handle of a
// destroyed one -- `kevlar novel` makes three streams per run: otherwise every run allocated its gigabytes again (0.19 s, and the old
// ones were never freed).  A stream made here therefore gets the lowest free SLOT, gives it back when it is destroyed, and the
// per-stream tables are keyed by kv_stream_key(): the slot, as a pointer-sized tag, or the handle itself for streams made elsewhere.
namespace {
std::mutex g_slot_mu;
std::map<hipStream_t, uintptr_t> g_stream_slot;
std::vector<bool> g_slot_used(1, true);            // slot 0: the null stream
}

hipStream_t kv_stream_key(hipStream_t st)
{
    if (!st) return st;
    std::lock_guard<std::mutex> lk(g_slot_mu);
    auto it = g_stream_slot.find(st);
    return it == g_stream_slot.end() ? st : (hipStream_t)it->second;
}

extern "C" int kv_stream_create(void **out)
{
    KV_REQUIRE(out, KV_ERR_ARG, "kv_stream_create: null output");
    hipStream_t s = nullptr;
    KV_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    {
        std::lock_guard<std::mutex> lk(g_slot_mu);
        size_t slot = 1;
        while (slot < g_slot_used.size() && g_slot_used[slot]) ++slot;
        if (slot == g_slot_used.size()) g_slot_used.push_back(true); else g_slot_used[slot] = true;
        g_stream_slot[s] = (uintptr_t)slot;           // (small integers are not addresses the runtime hands out)
    }
    *out = (void *)s;
    return KV_OK;
}

extern "C" int kv_stream_destroy(void *s)
{
    if (s) {
        {
            std::lock_guard<std::mutex> lk(g_slot_mu);
            auto it = g_stream_slot.find((hipStream_t)s);
            if (it != g_stream_slot.end()) { g_slot_used[(size_t)it->second] = false; g_stream_slot.erase(it); }
        }
        KV_HIP(hipStreamDestroy((hipStream_t)s));
    }
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// live profiler: HIP events on the library's stream around each kernel launch
// ---------------------------------------------------------------------------------------
struct ProfRec { double ms = 0; uint64_t n = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
static bool g_prof_on = false;
static std::map<std::string, ProfRec> g_prof;
static std::mutex g_prof_mu;

static std::vector<hipEvent_t> g_event_pool;   // events are recycled: creating one costs far more than recording it

static hipEvent_t prof_event()
{
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

KvProfScope::KvProfScope(const char *n) : name(n), a(nullptr), b(nullptr), on(g_prof_on)
{
    if (!on) return;
    a = prof_event();
    b = prof_event();
    if (!a || !b) { on = false; return; }
    (void)hipEventRecord(a, kv_stream());
}

KvProfScope::~KvProfScope()
{
    if (!on) return;
    (void)hipEventRecord(b, kv_stream());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[name].pending.emplace_back(a, b);
}

static void prof_drain()
{
    for (auto &kv : g_prof) {
        for (auto &p : kv.second.pending) {
            float ms = 0;
            if (hipEventSynchronize(p.second) == hipSuccess &&
                hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.n += 1;
            }
            g_event_pool.push_back(p.first);
            g_event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

extern "C" int kv_prof_enable(int on) { g_prof_on = on != 0; return KV_OK; }

extern "C" int kv_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    g_prof.clear();
    return KV_OK;
}

extern "C" int kv_prof_get(const char *kernel, double *ms, uint64_t *launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    auto it = g_prof.find(kernel ? kernel : "");
    if (ms) *ms = it == g_prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == g_prof.end() ? 0 : it->second.n;
    return KV_OK;
}

extern "C" int kv_prof_names(char *buf, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::string s;
    for (auto &kv : g_prof) { if (!s.empty()) s += ","; s += kv.first; }
    KV_REQUIRE(buf && s.size() + 1 <= cap, KV_ERR_CAPACITY, "kv_prof_names: buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// host hashing helpers (single k-mers: khmer .hash(); the batch paths hash on the device)
// ---------------------------------------------------------------------------------------
static inline uint64_t rotl64h(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64h(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

uint64_t kv_host_murmur_lo(const void *data, int len, uint32_t seed)
{
    const uint8_t *p = (const uint8_t *)data;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    int nblocks = len / 16;
    for (int b = 0; b < nblocks; ++b) {
        uint64_t k1, k2;
        memcpy(&k1, p + 16 * b, 8);
        memcpy(&k2, p + 16 * b + 8, 8);
        k1 *= c1; k1 = rotl64h(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64h(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64h(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64h(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = p + 16 * nblocks;
    int rem = len & 15;
    uint64_t k1 = 0, k2 = 0;
    if (rem > 8) {
        memcpy(&k2, tail + 8, (size_t)(rem - 8));
        k2 *= c2; k2 = rotl64h(k2, 33); k2 *= c1; h2 ^= k2;
    }
    if (rem > 0) {
        memcpy(&k1, tail, (size_t)(rem > 8 ? 8 : rem));
        k1 *= c1; k1 = rotl64h(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64h(h1); h2 = fmix64h(h2);
    return h1 + h2;
}

static inline char comp_base(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}

uint64_t kv_host_hash(int hashfam, const char *kmer, int k, bool *ok)
{
    *ok = true;
    if (hashfam == HF_TWOBIT) {
        uint64_t f = 0, r = 0;
        for (int i = 0; i < k; ++i) {
            int c, cc;
            switch (kmer[i]) { case 'A': c = 0; break; case 'T': c = 1; break; case 'C': c = 2; break; case 'G': c = 3; break; default: *ok = false; return 0; }
            switch (kmer[k - 1 - i]) { case 'A': cc = 1; break; case 'T': cc = 0; break; case 'C': cc = 3; break; case 'G': cc = 2; break; default: *ok = false; return 0; }
            f = (f << 2) | (uint64_t)c;
            r = (r << 2) | (uint64_t)cc;
        }
        return f < r ? f : r;
    }
    char rc[KV_MAX_K + 1];
    for (int i = 0; i < k; ++i) rc[i] = comp_base(kmer[k - 1 - i]);
    return kv_host_murmur_lo(kmer, k, 0) ^ kv_host_murmur_lo(rc, k, 0);
}

extern "C" int kv_hash_kmer(int kind, const char *kmer, int k, uint64_t *out)
{
    KV_REQUIRE(kmer && out, KV_ERR_ARG, "kv_hash_kmer: null argument");
    KV_REQUIRE(k >= 1 && k <= KV_MAX_K, KV_ERR_ARG, "kv_hash_kmer: k=%d out of range", k);
    int fam = kv_hashfam_of(kind);
    KV_REQUIRE(fam != HF_TWOBIT || k <= 32, KV_ERR_ARG, "graph sketches need k <= 32 (got %d)", k);
    bool ok;
    *out = kv_host_hash(fam, kmer, k, &ok);
    KV_REQUIRE(ok, KV_ERR_ARG, "invalid DNA character in k-mer");
    return KV_OK;
}

extern "C" int kv_reverse_hash(int kind, uint64_t h, int k, char *out)
{
    KV_REQUIRE(out, KV_ERR_ARG, "kv_reverse_hash: null output");
    KV_REQUIRE(kv_hashfam_of(kind) == HF_TWOBIT, KV_ERR_NOTIMPL,
               "reverse_hash not implemented for this hash function");
    static const char alphabet[4] = {'A', 'T', 'C', 'G'};
    for (int i = k - 1; i >= 0; --i) { out[i] = alphabet[h & 3]; h >>= 2; }
    out[k] = '\0';
    return KV_OK;
}

extern "C" int kv_band_bounds(int nbands, int band, uint64_t *lo, uint64_t *hi)
{
    KV_REQUIRE(nbands > 0 && band >= 0 && band < nbands, KV_ERR_ARG,
               "band %d out of range for %d bands", band, nbands);
    uint64_t bs = UINT64_MAX / (uint64_t)nbands;
    *lo = bs * (uint64_t)band;
    *hi = (band == nbands - 1) ? UINT64_MAX : bs * (uint64_t)(band + 1);
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// table sizing
// ---------------------------------------------------------------------------------------
static bool is_prime_u64(uint64_t n)
{
    if (n < 2) return false;
    if (n == 2) return true;
    if ((n & 1) == 0) return false;
    for (uint64_t d = 3; d * d <= n; d += 2)
        if (n % d == 0) return false;
    return true;
}

extern "C" int kv_primes_below(double target, int n, uint64_t *out, int *found)
{
    KV_REQUIRE(out && found && n >= 0, KV_ERR_ARG, "kv_primes_below: bad argument");
    *found = 0;
    if (!(target >= 3.0)) return KV_OK;
    uint64_t i = (uint64_t)target - 1;  // fractional targets truncate
    if ((i & 1) == 0) i -= 1;
    while (*found < n && i >= 2) {
        if (is_prime_u64(i)) out[(*found)++] = i;
        if (i < 3) break;
        i -= 2;
    }
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// sketch handles
// ---------------------------------------------------------------------------------------
// Table buffers of destroyed sketches are kept for the next sketch of the same geometry (up to KV_TABLE_CACHE_GB, default 32): on
// some boxes -- not on others, same image -- hipMalloc of a 2 GB sketch took 0.18 s inside a process that holds tens of gigabytes
// (`kevlar novel` after the bench's main loop: every run 0.26 s instead of 0.075), and it stalls every other thread's HIP call meanwhile.
namespace {
std::mutex g_tabcache_mu;
std::multimap<std::pair<int, uint64_t>, uint8_t *> g_tabcache;       // (device, bytes) -> buffer
uint64_t g_tabcache_bytes = 0;
int current_device() { kv_thread_device(); int d = 0; (void)hipGetDevice(&d); return d; }
uint64_t tabcache_cap()
{
    const char *e = kv_knob("KV_TABLE_CACHE_GB");
    const double gb = e ? atof(e) : 32.0;
    return gb > 0 ? (uint64_t)(gb * (double)(1ull << 30)) : 0;       // (fractions of a gigabyte count)
}
hipError_t table_alloc(uint8_t **p, uint64_t bytes)
{
    {
        std::lock_guard<std::mutex> lk(g_tabcache_mu);
        auto it = g_tabcache.find(std::make_pair(current_device(), bytes));
        if (it != g_tabcache.end()) {
            *p = it->second;
            g_tabcache.erase(it);
            g_tabcache_bytes -= bytes;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipSuccess) {                           // out of memory: give the cache (and idle first-toucher arrays) back, try once more
        (void)hipGetLastError();
        kv_table_cache_release();
        kv_unique_scratch_release();
        e = hipMalloc((void **)p, bytes);
    }
    return e;
}
void table_free(uint8_t *p, uint64_t bytes, int device)
{
    {
        std::lock_guard<std::mutex> lk(g_tabcache_mu);
        if (bytes >= (16u << 20) && g_tabcache_bytes + bytes <= tabcache_cap()) {
            g_tabcache.emplace(std::make_pair(device, bytes), p);      // the device the buffer lives on, not the caller's current one
            g_tabcache_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}
}  // namespace

hipError_t kv_hip_malloc(void **p, size_t bytes)
{
    kv_thread_device();
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        kv_table_cache_release();
        kv_unique_scratch_release();
        e = hipMalloc(p, bytes);
    }
    return e;
}

extern "C" int kv_table_cache_trim(void)
{
    kv_table_cache_release();
    return KV_OK;
}

extern "C" int kv_scratch_trim(void)
{
    (void)hipDeviceSynchronize();
    kv_skm_scratch_release();
    kv_route_scratch_release();
    kv_bin_scratch_release();
    kv_novel_scratch_release();
    kv_unique_scratch_release();
    kv_table_cache_release();
    return KV_OK;
}

void kv_table_cache_release()
{
    std::lock_guard<std::mutex> lk(g_tabcache_mu);
    for (auto &kv : g_tabcache) (void)hipFree(kv.second);         // (hipFree takes a pointer of any device)
    g_tabcache.clear();
    g_tabcache_bytes = 0;
}

int kv_sketch_alloc(int kind, int ksize, int ntables, const uint64_t *sizes, kv_sketch **out)
{
    KV_REQUIRE(out && sizes, KV_ERR_ARG, "kv_sketch_create: null argument");
    KV_REQUIRE(kind >= KV_COUNTTABLE && kind <= KV_NODEGRAPH, KV_ERR_ARG, "unknown sketch kind %d", kind);
    KV_REQUIRE(ntables >= 1 && ntables <= KV_MAX_TABLES, KV_ERR_ARG,
               "number of tables must be in 1..%d (got %d)", KV_MAX_TABLES, ntables);
    KV_REQUIRE(ksize >= 1 && ksize <= KV_MAX_K, KV_ERR_ARG, "k=%d out of range 1..%d", ksize, KV_MAX_K);
    KV_REQUIRE(kv_hashfam_of(kind) != HF_TWOBIT || ksize <= 32, KV_ERR_ARG,
               "graph sketches need k <= 32 (got %d)", ksize);
    static std::atomic<uint64_t> next_uid{1};
    kv_sketch *s = new kv_sketch();
    s->uid = next_uid.fetch_add(1);
    s->device = current_device();
    s->kind = kind;
    memset(&s->h, 0, sizeof(s->h));
    s->h.ntables = ntables;
    s->h.storage = kv_storage_of(kind);
    s->h.hashfam = kv_hashfam_of(kind);
    s->h.ksize = ksize;
    s->d_desc = nullptr;
    s->d_counters = nullptr;
    s->n_occupied = 0;
    s->occ_dirty = false;
    s->n_unique = 0;
    for (int i = 0; i < ntables; ++i) {
        if (sizes[i] == 0) { kv_set_error("table size must be positive"); delete s; return KV_ERR_ARG; }
        s->h.size[i] = sizes[i];
        s->h.magic[i] = kv_fastmod_magic(sizes[i]);
        uint64_t nb = kv_table_nbytes(s->h.storage, sizes[i]);
        s->alloc_bytes[i] = (nb + 15) & ~15ull;  // word-granular atomics need padding
    }
    hipError_t e = hipSuccess;
    for (int i = 0; i < ntables && e == hipSuccess; ++i) {
        e = table_alloc(&s->h.tab[i], s->alloc_bytes[i]);
        if (e == hipSuccess) e = hipMemsetAsync(s->h.tab[i], 0, s->alloc_bytes[i], kv_stream());
    }
    if (e == hipSuccess) e = kv_hip_malloc((void **)&s->d_desc, sizeof(SketchDev));
    if (e == hipSuccess) e = kv_hip_malloc((void **)&s->d_counters, 4 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemcpy(s->d_desc, &s->h, sizeof(SketchDev), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->d_counters, 0, 4 * sizeof(uint64_t));
    if (e != hipSuccess) {
        kv_set_error("sketch allocation failed: %s", hipGetErrorString(e));
        kv_sketch_destroy(s);
        return KV_ERR_HIP;
    }
    *out = s;
    return KV_OK;
}

extern "C" int kv_sketch_create(int kind, int ksize, int ntables, const uint64_t *sizes, kv_sketch **out)
{
    return kv_sketch_alloc(kind, ksize, ntables, sizes, out);
}

extern "C" int kv_sketch_destroy(kv_sketch *s)
{
    if (!s) return KV_OK;
    (void)hipDeviceSynchronize();                   // (what hipFree did implicitly: nobody is still working on these tables)
    for (int i = 0; i < KV_MAX_TABLES; ++i)
        if (s->h.tab[i]) table_free(s->h.tab[i], s->alloc_bytes[i], s->device);
    if (s->d_desc) (void)hipFree(s->d_desc);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->abl.mem) (void)hipFree(s->abl.mem);
    delete s;
    return KV_OK;
}

extern "C" int kv_sketch_info_get(kv_sketch *s, kv_sketch_info *out)
{
    KV_REQUIRE(s && out, KV_ERR_ARG, "kv_sketch_info_get: null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->occ_dirty) {
        { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
        int rc = kv_sketch_refresh_occupancy(s);
        if (rc != KV_OK) return rc;
    }
    memset(out, 0, sizeof(*out));
    out->kind = s->kind;
    out->ksize = s->h.ksize;
    out->ntables = s->h.ntables;
    for (int i = 0; i < s->h.ntables; ++i) {
        out->sizes[i] = s->h.size[i];
        out->bytes_device += s->alloc_bytes[i];
    }
    out->n_occupied = s->n_occupied;
    out->n_unique = s->n_unique;
    return KV_OK;
}

extern "C" int kv_sketch_table_read(kv_sketch *s, int table, uint8_t *host_out, uint64_t nbytes)
{
    KV_REQUIRE(s && host_out && table >= 0 && table < s->h.ntables, KV_ERR_ARG, "kv_sketch_table_read: bad argument");
    uint64_t nb = kv_table_nbytes(s->h.storage, s->h.size[table]);
    KV_REQUIRE(nbytes >= nb, KV_ERR_CAPACITY, "table %d needs %llu bytes", table, (unsigned long long)nb);
    std::lock_guard<std::mutex> lk(s->mu);
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemcpyAsync(host_out, s->h.tab[table], nb, hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    return KV_OK;
}

extern "C" int kv_sketch_table_write(kv_sketch *s, int table, const uint8_t *host_in, uint64_t nbytes)
{
    KV_REQUIRE(s && host_in && table >= 0 && table < s->h.ntables, KV_ERR_ARG, "kv_sketch_table_write: bad argument");
    uint64_t nb = kv_table_nbytes(s->h.storage, s->h.size[table]);
    KV_REQUIRE(nbytes == nb, KV_ERR_ARG, "table %d holds %llu bytes", table, (unsigned long long)nb);
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    s->abl.valid = false;
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    KV_HIP(hipMemcpyAsync(s->h.tab[table], host_in, nb, hipMemcpyHostToDevice, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    s->occ_dirty = true;
    return KV_OK;
}

extern "C" int kv_sketch_scan_hint(kv_sketch *s, int on)
{
    KV_REQUIRE(s, KV_ERR_ARG, "kv_sketch_scan_hint: null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    s->scan_hint = on != 0;
    s->scan_steady = on == 2;
    return KV_OK;
}

extern "C" int kv_sketch_clear(kv_sketch *s)
{
    KV_REQUIRE(s, KV_ERR_ARG, "kv_sketch_clear: null handle");
    std::lock_guard<std::mutex> lk(s->mu);
    s->version++;
    const char *lazy = kv_knob("KV_LAZY_CLEAR");           // "0": zero the tables here and now
    if (lazy && atoi(lazy) == 0) {
        KvProfScope prof("memset_tables");
        for (int i = 0; i < s->h.ntables; ++i) KV_HIP(hipMemsetAsync(s->h.tab[i], 0, s->alloc_bytes[i], kv_stream()));
        s->lazy_zero = false;
    } else {
        s->lazy_zero = true;
    }
    s->n_occupied = 0;
    s->n_unique = 0;
    s->occ_dirty = false;
    s->skm_off = false;
    s->skm_scan_off = false;
    s->abl.valid = false;
    return KV_OK;
}

int kv_sketch_ready_locked(kv_sketch *s)
{
    if (!s->lazy_zero) return KV_OK;
    KvProfScope prof("memset_tables");
    for (int i = 0; i < s->h.ntables; ++i) KV_HIP(hipMemsetAsync(s->h.tab[i], 0, s->alloc_bytes[i], kv_stream()));
    s->lazy_zero = false;
    return KV_OK;
}

int kv_sketch_ready(const kv_sketch *cs)
{
    kv_sketch *s = const_cast<kv_sketch *>(cs);
    if (!s) return KV_OK;
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->lazy_zero) return KV_OK;
    const int rc = kv_sketch_ready_locked(s);
    if (rc != KV_OK) return rc;
    KV_HIP(hipStreamSynchronize(kv_stream()));      // the reader may be on another stream
    return KV_OK;
}

extern "C" int kv_sketch_table_devptr(kv_sketch *s, int table, void **devptr, uint64_t *nbytes)
{
    KV_REQUIRE(s && devptr && table >= 0 && table < s->h.ntables, KV_ERR_ARG, "kv_sketch_table_devptr: bad argument");
    { const int rc = kv_sketch_ready(s); if (rc != KV_OK) return rc; }
    *devptr = s->h.tab[table];
    if (nbytes) *nbytes = kv_table_nbytes(s->h.storage, s->h.size[table]);
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// OXLI v4 files.  Layouts (decoded from the reference's fixtures):
//   Byte   : "OXLI" 04 01 use_bigcount(u8)=0 k(u32) ntables(u8) n_occupied(u64)
//            { size(u64) bytes[size] }*  n_bigcounts(u64)=0
//   Bit    : "OXLI" 04 02 k(u32) ntables(u8) n_occupied(u64) { size(u64) bytes[size/8+1] }*
//   Nibble : "OXLI" 04 07 k(u32) ntables(u8) n_occupied(u64) { size(u64) bytes[size/2+1] }*
// ---------------------------------------------------------------------------------------
static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }

extern "C" int kv_sketch_load(const char *path, int kind, kv_sketch **out)
{
    KV_REQUIRE(path && out, KV_ERR_ARG, "kv_sketch_load: null argument");
    KV_REQUIRE(kind >= KV_COUNTTABLE && kind <= KV_NODEGRAPH, KV_ERR_ARG, "unknown sketch kind %d", kind);
    FILE *f = fopen(path, "rb");
    KV_REQUIRE(f, KV_ERR_IO, "cannot open sketch file %s", path);
    uint8_t hdr[6];
    uint8_t bigcount = 0, nt = 0;
    uint32_t k = 0;
    uint64_t occ = 0;
    int storage = -1;
    bool ok = rd(f, hdr, 6) && memcmp(hdr, "OXLI", 4) == 0 && hdr[4] == 4;
    if (ok) storage = hdr[5] == 1 ? ST_BYTE : (hdr[5] == 2 ? ST_BIT : (hdr[5] == 7 ? ST_NIBBLE : -1));
    if (!ok || storage < 0) { fclose(f); kv_set_error("%s is not an OXLI v4 sketch file", path); return KV_ERR_IO; }
    if (storage != kv_storage_of(kind)) {
        fclose(f);
        kv_set_error("sketch file %s holds storage type %d, which does not match the requested sketch class", path, (int)hdr[5]);
        return KV_ERR_TYPE;
    }
    if (storage == ST_BYTE) ok = rd(f, &bigcount, 1);
    ok = ok && rd(f, &k, 4) && rd(f, &nt, 1) && rd(f, &occ, 8);
    if (!ok || nt < 1 || nt > KV_MAX_TABLES) { fclose(f); kv_set_error("truncated or unsupported sketch header in %s", path); return KV_ERR_IO; }
    std::vector<uint64_t> sizes(nt);
    std::vector<std::vector<uint8_t>> data(nt);
    for (int i = 0; i < nt && ok; ++i) {
        ok = rd(f, &sizes[i], 8);
        if (!ok) break;
        data[i].resize(kv_table_nbytes(storage, sizes[i]));
        ok = rd(f, data[i].data(), data[i].size());
    }
    fclose(f);
    KV_REQUIRE(ok, KV_ERR_IO, "truncated sketch file %s", path);
    kv_sketch *s = nullptr;
    int rc = kv_sketch_alloc(kind, (int)k, nt, sizes.data(), &s);
    if (rc != KV_OK) return rc;
    for (int i = 0; i < nt; ++i) {
        hipError_t e = hipMemcpy(s->h.tab[i], data[i].data(), data[i].size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { kv_set_error("upload of %s failed: %s", path, hipGetErrorString(e)); kv_sketch_destroy(s); return KV_ERR_HIP; }
    }
    s->n_occupied = occ;  // khmer keeps the header value
    s->occ_dirty = false;
    *out = s;
    return KV_OK;
}

extern "C" int kv_sketch_save(kv_sketch *s, const char *path)
{
    KV_REQUIRE(s && path, KV_ERR_ARG, "kv_sketch_save: null argument");
    std::lock_guard<std::mutex> lk(s->mu);
    { const int rc = kv_sketch_ready_locked(s); if (rc != KV_OK) return rc; }
    if (s->occ_dirty) {
        int rc = kv_sketch_refresh_occupancy(s);
        if (rc != KV_OK) return rc;
    }
    FILE *f = fopen(path, "wb");
    KV_REQUIRE(f, KV_ERR_IO, "cannot open %s for writing", path);
    const int st = s->h.storage;
    uint8_t hdr[6] = {'O', 'X', 'L', 'I', 4, (uint8_t)(st == ST_BYTE ? 1 : (st == ST_BIT ? 2 : 7))};
    fwrite(hdr, 1, 6, f);
    if (st == ST_BYTE) fputc(0, f);
    uint32_t k = (uint32_t)s->h.ksize;
    uint8_t nt = (uint8_t)s->h.ntables;
    fwrite(&k, 4, 1, f);
    fwrite(&nt, 1, 1, f);
    fwrite(&s->n_occupied, 8, 1, f);
    std::vector<uint8_t> buf;
    for (int i = 0; i < s->h.ntables; ++i) {
        uint64_t nb = kv_table_nbytes(st, s->h.size[i]);
        buf.resize(nb);
        hipError_t e = hipMemcpyAsync(buf.data(), s->h.tab[i], nb, hipMemcpyDeviceToHost, kv_stream());
        if (e == hipSuccess) e = hipStreamSynchronize(kv_stream());
        if (e != hipSuccess) { fclose(f); kv_set_error("download of table %d failed: %s", i, hipGetErrorString(e)); return KV_ERR_HIP; }
        fwrite(&s->h.size[i], 8, 1, f);
        fwrite(buf.data(), 1, nb, f);
    }
    if (st == ST_BYTE) { uint64_t z = 0; fwrite(&z, 8, 1, f); }
    bool bad = ferror(f) != 0;
    bad = (fclose(f) != 0) || bad;
    KV_REQUIRE(!bad, KV_ERR_IO, "write to %s failed", path);
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// pinned host blocks, recycled (first fit within 2x of the request; at most 512 MB parked)
// ---------------------------------------------------------------------------------------
namespace {
struct PinnedBlock { void *p; size_t cap; };
std::vector<PinnedBlock> g_pinned_free;
size_t g_pinned_parked = 0;
std::mutex g_pinned_mu;
}  // namespace

void *kv_pinned_get(size_t bytes, size_t *capacity)
{
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        for (size_t i = 0; i < g_pinned_free.size(); ++i) {
            if (g_pinned_free[i].cap >= bytes && g_pinned_free[i].cap <= 2 * bytes + 4096) {
                PinnedBlock b = g_pinned_free[i];
                g_pinned_free.erase(g_pinned_free.begin() + (long)i);
                g_pinned_parked -= b.cap;
                *capacity = b.cap;
                return b.p;
            }
        }
    }
    void *p = nullptr;
    const size_t cap = (bytes + 4095) & ~(size_t)4095;
    kv_thread_device();
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *capacity = cap;
    return p;
}

void kv_pinned_put(void *p, size_t capacity)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        if (g_pinned_parked + capacity <= ((size_t)512 << 20) && g_pinned_free.size() < 64) {
            g_pinned_free.push_back({p, capacity});
            g_pinned_parked += capacity;
            return;
        }
    }
    (void)hipHostFree(p);
}

// ---------------------------------------------------------------------------------------
// reads: ASCII upload, 2-bit packing on the device, tile table
// ---------------------------------------------------------------------------------------
// One thread per packed word (16 bases): locate the read by binary search over the word offsets,
// translate, and flag the read if any base is outside upper-case ACGT (stand-in code: A; the novel
// scan skips flagged reads, the count keeps them -- same rule as the oracle's clean_base()).
__global__ void k_pack_reads(const char *__restrict__ ascii, const uint64_t *__restrict__ offs,
                             const uint64_t *__restrict__ woff, uint64_t n_reads, uint64_t n_words,
                             uint32_t *__restrict__ words, uint32_t *__restrict__ flags32)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_reads;            // largest r with woff[r] <= w (empty reads share their successor's offset)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (woff[mid] <= w) lo = mid; else hi = mid;
        }
        const uint64_t r = lo;
        const uint64_t j0 = (w - woff[r]) * 16;
        const uint64_t len = offs[r + 1] - offs[r];
        const char *src = ascii + (offs[r] - offs[0]) + j0;
        const uint32_t n = (uint32_t)(len - j0 < 16 ? len - j0 : 16);
        uint32_t out = 0, bad = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const char c = src[j];
            uint32_t code = 0;
            if (c == 'A') code = 0;
            else if (c == 'C') code = 1;
            else if (c == 'G') code = 2;
            else if (c == 'T') code = 3;
            else { bad = 1; code = (c == 'c') ? 1u : (c == 'g') ? 2u : (c == 't') ? 3u : 0u; }
            out |= code << (2 * j);
        }
        words[w] = out;
        if (bad) atomicOr(&flags32[r >> 2], 1u << ((r & 3) * 8));
    }
}

extern "C" int kv_reads_create(const char *bases, const uint64_t *offs, uint64_t n_reads, kv_reads **out)
{
    KV_REQUIRE(out && offs && (bases || n_reads == 0), KV_ERR_ARG, "kv_reads_create: null argument");
    KV_REQUIRE(n_reads < 0xFFFFFFF0ull, KV_ERR_ARG, "too many reads in one batch");
    kv_reads *r = new kv_reads();
    r->n_reads = n_reads;
    r->n_bases = n_reads ? offs[n_reads] - offs[0] : 0;
    r->d_words = nullptr; r->d_woff = nullptr; r->d_len = nullptr; r->d_flags = nullptr; r->d_tile = nullptr;
    r->h_len.resize(n_reads);
    std::vector<uint64_t> woff(n_reads + 1);
    std::vector<TileDesc> tiles;
    uint64_t nw = 0;
    uint32_t max_len = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint64_t len = offs[i + 1] - offs[i];
        if (len > KV_MAX_READ_LEN) {
            kv_set_error("read %llu has %llu bases; this build handles reads up to %d bases",
                         (unsigned long long)i, (unsigned long long)len, (int)KV_MAX_READ_LEN);
            delete r;
            return KV_ERR_ARG;
        }
        r->h_len[i] = (uint32_t)len;
        if (len > max_len) max_len = (uint32_t)len;
        woff[i] = nw;
        nw += (len + 15) / 16;
    }
    woff[n_reads] = nw;
    r->n_words = nw;
    r->max_len = max_len;
    // tiles: consecutive reads whose staged ASCII (both strands, padded) fits the LDS budget; a sequence
    // too long for one tile (contigs, the chromosomes of a reference genome counted into a mask) becomes a
    // series of segment tiles of KV_SEG_BASES k-mer starts each
    {
        const uint32_t budget = KV_TILE_LDS_BYTES - 64;
        uint32_t used = 0, count = 0, first = 0;
        uint32_t run_bases = 0, most_bases = 0;
        auto close_run = [&](uint32_t next_first) {
            if (count) tiles.push_back(TileDesc{first, count, 0u, 0u});
            most_bases = std::max(most_bases, run_bases);
            used = 0; count = 0; first = next_first; run_bases = 0;
        };
        for (uint64_t i = 0; i < n_reads; ++i) {
            const uint32_t need = 2 * ((r->h_len[i] + KV_READ_PAD + 3) & ~3u);
            if (need > budget) {
                close_run((uint32_t)i + 1);
                for (uint32_t start = 0; start < r->h_len[i]; start += KV_SEG_BASES)
                    tiles.push_back(TileDesc{(uint32_t)i, 1u, start, 1u});
                most_bases = std::max<uint32_t>(most_bases, std::min<uint32_t>(r->h_len[i], KV_SEG_BASES + KV_MAX_K));
                continue;
            }
            if (count > 0 && (count == KV_TILE_MAX_READS || used + need > budget)) close_run((uint32_t)i);
            if (count == 0) first = (uint32_t)i;
            used += need; count += 1; run_bases += r->h_len[i];
        }
        close_run((uint32_t)n_reads);
        r->tile_max_bases = most_bases;
        r->n_tiles = (uint32_t)tiles.size();
        // reads of one length (packed on the host: uniform_reads above takes only text on the device) lie exactly as uniform_reads
        // lays them out -- read i at word i * wpr, the same reads per tile: say so, the lane-per-read cut asks for it
        {
            bool same = n_reads > 0 && r->h_len[0] > 0 && 2 * ((r->h_len[0] + KV_READ_PAD + 3) & ~3u) <= budget;
            for (uint64_t i = 1; same && i < n_reads; ++i) same = r->h_len[i] == r->h_len[0];
            if (same) {
                r->uni_len = r->h_len[0];
                r->uni_per_tile = std::min<uint32_t>(KV_TILE_MAX_READS, budget / (2 * ((r->h_len[0] + KV_READ_PAD + 3) & ~3u)));
            }
        }
        r->tile_lds_bytes = KV_TILE_LDS_BYTES + 256;   // + rolling-window over-read
        if (tiles.empty()) tiles.push_back(TileDesc{0u, 0u, 0u, 0u});
    }
    const uint64_t flag_bytes = ((n_reads + 3) & ~3ull) + 4;
    char *d_ascii = nullptr;
    uint64_t *d_offs = nullptr;
    hipStream_t st = kv_stream();
    hipError_t e = kv_hip_malloc((void **)&r->d_words, (nw + 4) * 4);   // + slack: k-mer extraction reads up to two words ahead
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_woff, woff.size() * 8);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_len, (n_reads ? n_reads : 1) * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_flags, flag_bytes);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_tile, tiles.size() * sizeof(TileDesc));
    if (e == hipSuccess) e = kv_hip_malloc((void **)&d_ascii, r->n_bases ? r->n_bases : 1);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&d_offs, woff.size() * 8);
    if (e == hipSuccess && r->n_bases) e = hipMemcpyAsync(d_ascii, bases + offs[0], r->n_bases, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_offs, offs, (n_reads + 1) * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_woff, woff.data(), woff.size() * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->d_len, r->h_len.data(), n_reads * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_flags, 0, flag_bytes, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_tile, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nw) {
        KvProfScope prof("k_pack_reads");
        const unsigned grid = (unsigned)std::min<uint64_t>((nw + 255) / 256, 65536);
        hipLaunchKernelGGL(k_pack_reads, dim3(grid), dim3(256), 0, st, (const char *)d_ascii, (const uint64_t *)d_offs,
                           (const uint64_t *)r->d_woff, n_reads, nw, r->d_words, (uint32_t *)r->d_flags);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);      // the host vectors and the caller's blob are free after this
    if (d_ascii) (void)hipFree(d_ascii);
    if (d_offs) (void)hipFree(d_offs);
    if (e != hipSuccess) {
        kv_last_hip_code = (int)e;
        kv_set_error("read batch upload failed: %s", hipGetErrorString(e));
        kv_reads_destroy(r);
        return KV_ERR_HIP;
    }
    *out = r;
    return KV_OK;
}

// reads that are already 2-bit packed on the host, any lengths: word offsets, tile table and uploads as kv_reads_create
// builds them, minus the ASCII upload and the packing kernel (the packed-read cache of kv_fastx.hip comes through here)
namespace {
struct TextSource { const uint8_t *d_text; const uint64_t *d_seq_start; const uint32_t *d_seq_len; };
int reads_from_packed(const uint32_t *words, const TextSource *text, const uint32_t *lens, const uint8_t *flags, uint64_t n_reads, kv_reads **out);
}

int kv_reads_from_packed_var(const uint32_t *words, const uint32_t *lens, const uint8_t *flags, uint64_t n_reads, kv_reads **out)
{
    KV_REQUIRE(out && ((words && lens) || n_reads == 0), KV_ERR_ARG, "kv_reads_from_packed_var: null argument");
    return reads_from_packed(words, nullptr, lens, flags, n_reads, out);
}

int kv_reads_from_device_text(const uint8_t *d_text, const uint64_t *d_seq_start, const uint32_t *d_seq_len, const uint32_t *lens,
                              uint64_t n_reads, kv_reads **out)
{
    KV_REQUIRE(out && ((d_text && d_seq_start && d_seq_len && lens) || n_reads == 0), KV_ERR_ARG, "kv_reads_from_device_text: null argument");
    const TextSource src = {d_text, d_seq_start, d_seq_len};
    return reads_from_packed(nullptr, &src, lens, nullptr, n_reads, out);
}

namespace {
// word offsets and tile table of a batch whose reads all have the same length: nothing per read crosses PCIe
__global__ void k_uniform_layout(uint64_t *woff, TileDesc *tiles, uint64_t n_reads, uint64_t wpr, uint32_t per_tile, uint32_t n_tiles)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= n_reads; i += stride) woff[i] = i * wpr;
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < n_tiles; t += stride) {
        const uint64_t first = t * per_tile;
        tiles[t] = TileDesc{(uint32_t)first, (uint32_t)min((uint64_t)per_tile, n_reads - first), 0u, 0u};
    }
}

// the same layout reads_from_packed builds read by read, in closed form; false if the batch is not uniform (or its reads
// need segment tiles)
bool uniform_reads(kv_reads *r, const TextSource *text, const uint32_t *lens, uint64_t n_reads, int *rc)
{
    if (!text || n_reads == 0) return false;
    const uint32_t L = lens[0];
    if (L == 0) return false;
    for (uint64_t i = 1; i < n_reads; ++i)
        if (lens[i] != L) return false;
    const uint32_t budget = KV_TILE_LDS_BYTES - 64, need = 2 * ((L + KV_READ_PAD + 3) & ~3u);
    if (need > budget) return false;
    const uint32_t per_tile = std::min<uint32_t>(KV_TILE_MAX_READS, budget / need);
    const uint64_t wpr = ((uint64_t)L + 15) / 16, nw = n_reads * wpr;
    r->n_words = nw; r->n_bases = n_reads * (uint64_t)L; r->max_len = L;
    r->tile_max_bases = (uint32_t)std::min<uint64_t>(per_tile, n_reads) * L;
    r->n_tiles = (uint32_t)((n_reads + per_tile - 1) / per_tile);
    r->uni_len = L; r->uni_per_tile = per_tile;
    r->tile_lds_bytes = KV_TILE_LDS_BYTES + 256;
    const uint64_t flag_bytes = ((n_reads + 3) & ~3ull) + 4;
    hipStream_t st = kv_stream();
    hipError_t e = kv_hip_malloc((void **)&r->d_words, (nw + 4) * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_woff, (n_reads + 1) * 8);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_len, n_reads * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_flags, flag_bytes);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_tile, (size_t)r->n_tiles * sizeof(TileDesc));
    if (e == hipSuccess) e = hipMemsetAsync(r->d_words + nw, 0, 16, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_len, text->d_seq_len, n_reads * 4, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_flags, 0, flag_bytes, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_uniform_layout, dim3(1024), dim3(256), 0, st, r->d_woff, r->d_tile, n_reads, wpr, per_tile, r->n_tiles);
        kv_fastq_pack_launch(text->d_text, text->d_seq_start, text->d_seq_len, r->d_woff, n_reads, nw, r->d_words, (uint32_t *)r->d_flags, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        kv_last_hip_code = (int)e;
        kv_set_error("read batch upload failed: %s", hipGetErrorString(e));
        *rc = KV_ERR_HIP;
    }
    return true;
}

int reads_from_packed(const uint32_t *words, const TextSource *text, const uint32_t *lens, const uint8_t *flags, uint64_t n_reads, kv_reads **out)
{
    KV_REQUIRE(n_reads < 0xFFFFFFF0ull, KV_ERR_ARG, "too many reads in one batch");
    kv_reads *r = new kv_reads();
    r->n_reads = n_reads;
    r->d_words = nullptr; r->d_woff = nullptr; r->d_len = nullptr; r->d_flags = nullptr; r->d_tile = nullptr;
    r->h_len.assign(lens, lens + n_reads);
    {
        int rc = KV_OK;
        if (uniform_reads(r, text, lens, n_reads, &rc)) {
            if (rc != KV_OK) { kv_reads_destroy(r); return rc; }
            *out = r;
            return KV_OK;
        }
    }
    std::vector<uint64_t> woff(n_reads + 1);
    std::vector<TileDesc> tiles;
    uint64_t nw = 0, nb = 0;
    uint32_t max_len = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        woff[i] = nw;
        nw += ((uint64_t)lens[i] + 15) / 16;
        nb += lens[i];
        if (lens[i] > max_len) max_len = lens[i];
    }
    woff[n_reads] = nw;
    r->n_words = nw; r->n_bases = nb; r->max_len = max_len;
    {
        const uint32_t budget = KV_TILE_LDS_BYTES - 64;
        uint32_t used = 0, count = 0, first = 0, run_bases = 0, most_bases = 0;
        auto close_run = [&](uint32_t next_first) {
            if (count) tiles.push_back(TileDesc{first, count, 0u, 0u});
            most_bases = std::max(most_bases, run_bases);
            used = 0; count = 0; first = next_first; run_bases = 0;
        };
        for (uint64_t i = 0; i < n_reads; ++i) {
            const uint32_t need = 2 * ((r->h_len[i] + KV_READ_PAD + 3) & ~3u);
            if (need > budget) {
                close_run((uint32_t)i + 1);
                for (uint32_t start = 0; start < r->h_len[i]; start += KV_SEG_BASES)
                    tiles.push_back(TileDesc{(uint32_t)i, 1u, start, 1u});
                most_bases = std::max<uint32_t>(most_bases, std::min<uint32_t>(r->h_len[i], KV_SEG_BASES + KV_MAX_K));
                continue;
            }
            if (count > 0 && (count == KV_TILE_MAX_READS || used + need > budget)) close_run((uint32_t)i);
            if (count == 0) first = (uint32_t)i;
            used += need; count += 1; run_bases += r->h_len[i];
        }
        close_run((uint32_t)n_reads);
        r->tile_max_bases = most_bases;
        r->n_tiles = (uint32_t)tiles.size();
        r->tile_lds_bytes = KV_TILE_LDS_BYTES + 256;
        if (tiles.empty()) tiles.push_back(TileDesc{0u, 0u, 0u, 0u});
    }
    const uint64_t flag_bytes = ((n_reads + 3) & ~3ull) + 4;
    hipStream_t st = kv_stream();
    hipError_t e = kv_hip_malloc((void **)&r->d_words, (nw + 4) * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_woff, woff.size() * 8);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_len, (n_reads ? n_reads : 1) * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_flags, flag_bytes);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_tile, tiles.size() * sizeof(TileDesc));
    if (e == hipSuccess && nw && words) e = hipMemcpyAsync(r->d_words, words, nw * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_words + nw, 0, 16, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_woff, woff.data(), woff.size() * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->d_len, r->h_len.data(), n_reads * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_flags, 0, flag_bytes, st);
    if (e == hipSuccess && n_reads && flags) e = hipMemcpyAsync(r->d_flags, flags, n_reads, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_tile, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nw && text) {
        kv_fastq_pack_launch(text->d_text, text->d_seq_start, text->d_seq_len, r->d_woff, n_reads, nw, r->d_words, (uint32_t *)r->d_flags, st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        kv_last_hip_code = (int)e;
        kv_set_error("read batch upload failed: %s", hipGetErrorString(e));
        kv_reads_destroy(r);
        return KV_ERR_HIP;
    }
    *out = r;
    return KV_OK;
}
}  // namespace

// equal-length reads from packed words on the host, or (words == nullptr) written on the device by `fill(d_words, stream)`
static int reads_packed_uniform(const uint32_t *words, const std::function<void(uint32_t *, hipStream_t)> *fill, uint64_t n_reads,
                                uint32_t read_len, kv_reads **out)
{
    KV_REQUIRE(out && (words || fill || n_reads == 0), KV_ERR_ARG, "kv_reads_create_packed: null argument");
    KV_REQUIRE(n_reads < 0xFFFFFFF0ull, KV_ERR_ARG, "too many reads in one batch");
    KV_REQUIRE(read_len >= 1 && read_len <= KV_MAX_READ_LEN, KV_ERR_ARG, "read length %u out of range", read_len);
    kv_reads *r = new kv_reads();
    const uint64_t wpr = (read_len + 15) / 16;
    r->n_reads = n_reads; r->n_bases = n_reads * read_len; r->n_words = n_reads * wpr; r->max_len = read_len;
    r->d_words = nullptr; r->d_woff = nullptr; r->d_len = nullptr; r->d_flags = nullptr; r->d_tile = nullptr;
    r->h_len.assign(n_reads, read_len);
    const uint32_t need = 2 * ((read_len + KV_READ_PAD + 3) & ~3u);
    uint32_t per_tile = (KV_TILE_LDS_BYTES - 64) / need;
    if (per_tile > KV_TILE_MAX_READS) per_tile = KV_TILE_MAX_READS;
    if (per_tile < 1) per_tile = 1;
    KV_REQUIRE(need <= KV_TILE_LDS_BYTES - 64, KV_ERR_ARG, "kv_reads_create_packed: read length %u needs kv_reads_create", read_len);
    r->n_tiles = (uint32_t)((n_reads + per_tile - 1) / per_tile);
    r->tile_max_bases = (uint32_t)std::min<uint64_t>(per_tile, n_reads) * read_len;
    r->uni_len = read_len; r->uni_per_tile = per_tile;
    r->tile_lds_bytes = KV_TILE_LDS_BYTES + 256;
    // only the packed words cross PCIe: word offsets, lengths and the tile table of equal-length reads are written on
    // the device in closed form
    hipStream_t st = kv_stream();
    const uint32_t tiles_alloc = std::max<uint32_t>(r->n_tiles, 1u);
    hipError_t e = kv_hip_malloc((void **)&r->d_words, (r->n_words + 4) * 4);   // + slack: k-mer extraction reads up to two words ahead
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_woff, (n_reads + 1) * 8);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_len, (n_reads ? n_reads : 1) * 4);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_flags, n_reads ? n_reads : 1);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&r->d_tile, (size_t)tiles_alloc * sizeof(TileDesc));
    if (e == hipSuccess && r->n_words && words) e = hipMemcpyAsync(r->d_words, words, r->n_words * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && r->n_words && !words) { (*fill)(r->d_words, st); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemsetAsync(r->d_words + r->n_words, 0, 16, st);
    if (e == hipSuccess && n_reads) e = hipMemsetD32Async((hipDeviceptr_t)r->d_len, (int)read_len, n_reads, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_flags, 0, n_reads ? n_reads : 1, st);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_tile, 0, (size_t)tiles_alloc * sizeof(TileDesc), st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_uniform_layout, dim3(1024), dim3(256), 0, st, r->d_woff, r->d_tile, n_reads, wpr, per_tile, r->n_tiles);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        kv_last_hip_code = (int)e;
        kv_set_error("read batch upload failed: %s", hipGetErrorString(e));
        kv_reads_destroy(r);
        return KV_ERR_HIP;
    }
    *out = r;
    return KV_OK;
}

extern "C" int kv_reads_create_packed(const uint32_t *words, uint64_t n_reads, uint32_t read_len, kv_reads **out)
{
    KV_REQUIRE(out && (words || n_reads == 0), KV_ERR_ARG, "kv_reads_create_packed: null argument");
    return reads_packed_uniform(words, nullptr, n_reads, read_len, out);
}

extern "C" int kv_reads_words_read(const kv_reads *r, uint64_t first_word, uint64_t n_words, uint32_t *host_out)
{
    KV_REQUIRE(r && (host_out || n_words == 0) && first_word + n_words <= r->n_words, KV_ERR_ARG, "kv_reads_words_read: bad range");
    if (n_words == 0) return KV_OK;
    KV_HIP(hipMemcpyAsync(host_out, r->d_words + first_word, n_words * 4, hipMemcpyDeviceToHost, kv_stream()));
    KV_HIP(hipStreamSynchronize(kv_stream()));
    return KV_OK;
}

// kv_synth.hip
void kv_synth_fill(uint32_t *d_words, uint64_t genome_len, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                   uint32_t read_len, double error_rate, hipStream_t st);

extern "C" int kv_reads_generate(uint64_t genome_len, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                                 uint32_t read_len, double error_rate, kv_reads **out)
{
    KV_REQUIRE(out && sample >= 0 && sample <= 2 && read_len >= 1 && genome_len >= (uint64_t)read_len + 1 && error_rate >= 0.0 && error_rate < 1.0,
               KV_ERR_ARG, "kv_reads_generate: bad argument");
    const std::function<void(uint32_t *, hipStream_t)> fill = [&](uint32_t *d_words, hipStream_t st) {
        kv_synth_fill(d_words, genome_len, seed, sample, first_read, n_reads, read_len, error_rate, st);
    };
    return reads_packed_uniform(nullptr, &fill, n_reads, read_len, out);
}

extern "C" int kv_reads_destroy(kv_reads *r)
{
    if (!r) return KV_OK;
    if (r->d_words) (void)hipFree(r->d_words);
    if (r->d_woff) (void)hipFree(r->d_woff);
    if (r->d_len) (void)hipFree(r->d_len);
    if (r->d_flags) (void)hipFree(r->d_flags);
    if (r->d_tile) (void)hipFree(r->d_tile);
    delete r;
    return KV_OK;
}

extern "C" int kv_reads_count(const kv_reads *r, uint64_t *n_reads, uint64_t *n_bases)
{
    KV_REQUIRE(r, KV_ERR_ARG, "kv_reads_count: null handle");
    if (n_reads) *n_reads = r->n_reads;
    if (n_bases) *n_bases = r->n_bases;
    return KV_OK;
}

extern "C" int kv_reads_num_kmers(const kv_reads *r, int ksize, uint64_t *n_kmers)
{
    KV_REQUIRE(r && n_kmers && ksize >= 1, KV_ERR_ARG, "kv_reads_num_kmers: bad argument");
    if (r->nk_cached_k != ksize) {
        uint64_t n = 0;
        for (uint32_t len : r->h_len)
            if (len >= (uint32_t)ksize) n += len - (uint32_t)ksize + 1;
        kv_reads *w = const_cast<kv_reads *>(r);
        w->nk_cached = n;
        w->nk_cached_k = ksize;
    }
    *n_kmers = r->nk_cached;
    return KV_OK;
}

// ---- the knob registry (kv_knobs.h) --------------------------------------------------------------------------------------------
namespace {
#define S_ KV_KNOB_SETTING
#define T_ KV_KNOB_TUNING
#define X_ KV_KNOB_EXPERIMENT
const KvKnobDef g_knobs[] = {
    // settings: a user may set these
    {"KV_TABLE_CACHE_GB", S_, "table buffers of destroyed sketches kept for the next sketch of the same size, in GB (default 32; 0: none)"},
    {"KV_AUGFASTX_THREADS", S_, "host threads of the augmented FASTX reader / writer (default: by input size, up to the cores)"},
    {"KV_FORMAT_THREADS", S_, "host threads formatting annotated reads (default: min(16, cores))"},
    {"KV_STAGE_THREADS", S_, "host threads reading a file into the pinned staging buffers (1..16)"},
    {"KEVLAR_PACK_CACHE", S_, "packed-read cache files beside the inputs: unset / 0 never, 1 use and create"},
    {"KV_INGEST", S_, "host: parse every file on the host, never on the device"},
    {"KV_GUNZIP", S_, "host: ordinary gzip streams through zlib on the host instead of the device inflater (BGZF and plain FASTQ stay on the device)"},
    {"KV_GUNZIP_CRC", S_, "0: skip the CRC-32 check of inflated members"},
    {"KV_PARALLEL_SAMPLES", S_, "1 / 0: count the samples of `kevlar novel` side by side on their own streams / one after the other"},
    {"KV_INGEST_VERBOSE", S_, "wall time of every ingest step on stderr"},
    {"KV_GUNZIP_VERBOSE", S_, "statistics of the gzip inflater on stderr"},
    {"KV_SKM_VERBOSE", S_, "what the super-k-mer front end decided (geometry, lists kept or dropped, fallbacks) on stderr"},
    {"KV_MEX_VERBOSE", S_, "why a rank declined a step of the exchange layout, on stderr"},
    // tuning: same results on another path / in another geometry; honoured only with KV_TUNING=1
    {"KV_COUNT_PATH", T_, "atomic | binned | skm: pin the count path"},
    {"KV_NOVEL_PATH", T_, "tiles | skm: pin the scan path"},
    {"KV_ROUTE_PATH", T_, "plain: the exchange routes one item per k-mer (no combining)"},
    {"KV_SET_SCAN", T_, "skm: a shard's set scan keeps the bucketed kernel"},
    {"KV_LAZY_CLEAR", T_, "0: kv_sketch_clear zeroes the tables at once instead of leaving it to the apply stage"},
    {"KV_NO_ROLL", T_, "hash every k-mer from scratch (no rolling 2-bit window)"},
    {"KV_BIN_2BIT", T_, "0: stage A never hashes from the 2-bit form"},
    {"KV_BIN_DIRECT", T_, "0: stage A writes items through LDS rings instead of direct stores"},
    {"KV_BIN_FAST4", T_, "0: no FP64-quotient remainders for four tables"},
    {"KV_BIN_SPLIT", T_, "sorted: stage B sorts a chunk in LDS before it stores"},
    {"KV_BIN_SLICE15", T_, "1: 32 K-bin slices for weighted items"},
    {"KV_BIN_C", T_, "coarse buckets per table"},
    {"KV_SKM_S1", T_, "tile | wave | lane: pin the record cutter"},
    {"KV_SKM_S2", T_, "plain | sorted: pin the fine split"},
    {"KV_SKM_S1_THREADS", T_, "512 | 1024: workgroup size of the wave cutter"},
    {"KV_SKM_R", T_, "reads per wave pass of the wave cutter"},
    {"KV_SKM_CH", T_, "8 | 16: k-mers per lane chunk of the wave cutter"},
    {"KV_SKM_LANE_MAXWG", T_, "2 | 3: workgroups per CU the lane cutter asks for"},
    {"KV_SKM_LANE_FLUSH", T_, "blocks between two flushes of the lane cutter's run list"},
    {"KV_SKM_SEG1", T_, "bucket: S1 segments laid out bucket-major"},
    {"KV_SKM_ORIENT", T_, "0: records keep the read's strand (no oriented keys)"},
    {"KV_SKM_COMPACT", T_, "0: never the 16-byte records without positions"},
    {"KV_SKM_DEDUP", T_, "1: identical records merged before the k-mer table (k = 31)"},
    {"KV_SKM_DEDUP_MAXN", T_, "longest record the record table takes"},
    {"KV_SKM_DEDUP_RS", T_, "record-table slots (1024: two workgroups per CU)"},
    {"KV_SKM_ANY_K", T_, "use the kernels with k at run time, not the k = 31 / 51 instances"},
    {"KV_SKM_BUCKET_KMERS", T_, "k-mer occurrences aimed at per bucket (tests: many buckets on small inputs)"},
    {"KV_SKM_CAP_PCT", T_, "segment capacity in per cent of the estimate (tests: push records through the loose list)"},
    {"KV_SKM_LOOSE_CAP", T_, "entries of the loose list"},
    {"KV_SKM_FORCE_LOOSE", T_, "every record through the loose list"},
    {"KV_SKM_NWG1", T_, "S1 writers (upper bound)"},
    {"KV_SKM_NWG2", T_, "S2 writers per coarse stream"},
    {"KV_SKM_BPT", T_, "buckets per work ticket of stage S3"},
    {"KV_SKM_WG3_PER_CU", T_, "1..3: S3 workgroups per CU"},
    {"KV_SKM_ABL", T_, "0: no abundance list from a control's count"},
    {"KV_SKM_DL", T_, "0: no distinct list from a case sample's count; 1: from the first batch on"},
    {"KV_SKM_NO_REUSE", T_, "the scan re-buckets the batch instead of reusing the count's buckets"},
    {"KV_NOVEL_2BIT", T_, "0: the per-k-mer scan keeps the tile kernel"},
    {"KV_NOVEL_BITS", T_, "0: no bit map as the first probe of the scan"},
    {"KV_NOVEL_BITS_MIN", T_, "items from which the pairs scan builds its bit map"},
    {"KV_NOVEL_PAIRS", T_, "0: the set scan answers from hashes alone"},
    {"KV_NOVEL_ABCACHE", T_, "0: no cache of the interesting k-mers' abundances"},
    {"KV_NOVEL_VCACHE", T_, "0: no verdict cache in the tile scan"},
    {"KV_NOVEL_VCSETS", T_, "0: direct-mapped verdict cache"},
    {"KV_NOVEL_EMIT_TILES", T_, "hits listed by the tile kernel"},
    {"KV_NOVEL_EMIT_FUSED", T_, "hits listed by the fused kernel"},
    {"KV_NOVEL_REREAD", T_, "`kevlar novel` reads a case file again for the scan instead of keeping its batch"},
    {"KV_HOST_SORT", T_, "partition sorts its keys on the host"},
    {"KV_FORMAT_FD", T_, "0: annotated reads formatted into a buffer, not written to the descriptor"},
    {"KV_STAGE", T_, "0: uploads straight from the file mapping, no pinned staging"},
    {"KV_INGEST_TEXT_MB", T_, "text per ingest batch in MB (tests: small batches)"},
    {"KV_GUNZIP_TEXT_MIN_MB", T_, "gzip files from this size on are inflated on the device"},
    {"KV_GUNZIP_CHUNK_KB", T_, "1..64: compressed bytes per probe stretch"},
    {"KV_GUNZIP_RING_BITS", T_, "10..14: LDS window of the gzip decoder"},
    {"KV_GUNZIP_SPLIT_KB", T_, "cut DEFLATE blocks longer than this"},
    {"KV_INFLATE_WINDOW_BITS", T_, "10..15: LDS window of the BGZF inflater"},
    {"KV_MEX_NWG1", T_, "exchange: S1 writers of a shard (the same on every rank)"},
    {"KV_MEX_CAP2_SLACK", T_, "exchange: S2 segment slack"},
    {"KV_MEX_PASSES", T_, "exchange: combine passes per bucket, at least (1-16)"},
    {"KV_MEX_DL_POOL", T_, "exchange: 1 = the owner's distinct list as a pool of chunks at once (it is the last resort of an owner short of memory)"},
    {"KV_MEX_PAIRS", T_, "9: (hash, count) pairs travel in the 9-byte block form"},
    {"KV_MEX_TEST_DECLINE", T_, "point:rank -- that rank fails at that point of the exchange (tests of the agreed fallbacks)"},
    {"KV_ROUTE_OVF_CAP", T_, "entries of the route's overflow list (tests: force the capacity error)"},
    // experiments: parts of kernels skipped for timing, RESULTS ARE WRONG; a -DKV_EXPERIMENTS build only
    {"KV_SKM_DEBUG", X_, "bit mask: phases of the super-k-mer count kernels to skip"},
    {"KV_SKM_SCAN_DEBUG", X_, "bit mask: phases of k_skm_novel_list to skip"},
    {"KV_BIN_DEBUG", X_, "bit mask: phases of k_bin_apply to skip"},
};
#undef S_
#undef T_
#undef X_
}  // namespace

const KvKnobDef *kv_knob_table(size_t *n)
{
    if (n) *n = sizeof(g_knobs) / sizeof(g_knobs[0]);
    return g_knobs;
}

static bool knob_honoured(KvKnobClass cls)
{
    if (cls == KV_KNOB_SETTING) return true;
    const char *t = getenv("KV_TUNING");                 // (the registry's own switch: the one getenv of the library besides the lookup below)
    if (!(t && atoi(t) == 1)) return false;
#if defined(KV_EXPERIMENTS)
    return true;
#else
    return cls != KV_KNOB_EXPERIMENT;
#endif
}

const char *kv_knob(const char *name)
{
    for (const KvKnobDef &d : g_knobs)
        if (!strcmp(d.name, name)) return knob_honoured(d.cls) ? getenv(name) : nullptr;
    fprintf(stderr, "[kvsketch] kv_knob(\"%s\"): not in the registry of kv_host.hip\n", name);
    return nullptr;
}

extern "C" int kv_knob_get(const char *name, char *value_out, uint64_t cap)
{
    KV_REQUIRE(name, KV_ERR_ARG, "kv_knob_get: null name");
    bool known = false;
    for (const KvKnobDef &d : g_knobs) known = known || !strcmp(d.name, name);
    KV_REQUIRE(known, KV_ERR_ARG, "kv_knob_get: %s is not a registered knob", name);
    const char *v = kv_knob(name);
    if (!v) return 0;
    if (value_out && cap) { strncpy(value_out, v, cap - 1); value_out[cap - 1] = 0; }
    return 1;
}

extern "C" int kv_knobs_describe(int whole_table, char *out, uint64_t cap)
{
    KV_REQUIRE(out && cap > 0, KV_ERR_ARG, "kv_knobs_describe: no buffer");
    std::string s;
    static const char *cls_name[] = {"setting", "tuning", "experiment"};
    for (const KvKnobDef &d : g_knobs) {
        if (whole_table) {
            s += d.name; s += '\t'; s += cls_name[d.cls]; s += '\t'; s += d.doc; s += '\n';
        } else if (const char *v = getenv(d.name)) {
            if (!s.empty()) s += ' ';
            if (!knob_honoured(d.cls)) s += "ignored:";
            s += d.name; s += '='; s += v;
        }
    }
    KV_REQUIRE(s.size() < cap, KV_ERR_CAPACITY, "kv_knobs_describe: %llu bytes needed", (unsigned long long)s.size() + 1);
    memcpy(out, s.c_str(), s.size() + 1);
    return KV_OK;
}
