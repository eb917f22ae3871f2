// kv_fastq.hip -- FASTQ records split and 2-bit packed on the device, fed by kv_inflate.hip.
//
// The host path (kv_fastx.hip) inflates with zlib and splits lines with memchr on one core: ~2 M reads/s from a .gz
// file, ~20 M from an uncompressed one, whatever the GPU does next.  For blocked gzip (BGZF) input the text never exists
// on the host (an uncompressed FASTQ file takes the same route minus the inflate: its bytes are uploaded as they are):
//
//   file image (mmap) --H2D, compressed--> k_inflate (one wave per member) --> text in HBM
//   k_count_lines   newlines per 16-KB chunk            \
//   k_chunk_scan    exclusive scan of the chunk counts    > start of every line
//   k_line_starts   byte offset of each line start      /
//   k_records       four lines = one record: checks '@' / '+', start and length of the sequence line
//   k_pack_text     2-bit packs the sequence lines into the kv_reads layout (same rules as k_pack_reads)
//
// The read lengths come back to the host (the tile table is built there, as for every batch); names, sequences and
// qualities stay in HBM until the next batch and are gathered for the few records somebody asks for
// (kv_fastq_device_fetch: the reads with interesting k-mers).  A record cut by the end of a batch is carried to the
// front of the next one.  Text that is not strict four-line FASTQ (FASTA, blank lines, wrapped sequences) is
// reported as KV_ERR_TYPE and the caller reopens the file on the host path.
//
// Replaces khmer.ReadParser over a gzip stream (kevlar/count.py:40, kevlar/novel.py:123, kevlar/__init__.py:125-128).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kv_binned.h"
#include "kv_device.h"
#include "kv_internal.h"

namespace {

#define FQ_CHUNK 16384u           // bytes per line-counting workgroup
#define FQ_THREADS 256

// bit b of the result: byte b of the 16 is a newline
__device__ __forceinline__ uint32_t newline_mask16(const uint4 v)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = w[q] ^ 0x0a0a0a0au;                                    // newline bytes become zero
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // 0x80 exactly in the zero bytes
        mask |= (((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u)) << (4 * q);
    }
    return mask;
}

__device__ __forceinline__ uint32_t newline_mask_at(const uint8_t *__restrict__ text, uint64_t at, uint64_t n)
{
    if (at + 16 <= n) return newline_mask16(*(const uint4 *)(text + at));       // `at` is a multiple of 16 in an aligned buffer
    uint32_t mask = 0;
    for (uint32_t b = 0; b < 16 && at + b < n; ++b) mask |= (text[at + b] == '\n' ? 1u : 0u) << b;
    return mask;
}

// newlines in text[0, n) per chunk
__global__ __launch_bounds__(FQ_THREADS) void k_count_lines(const uint8_t *__restrict__ text, uint64_t n, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    const uint64_t c0 = (uint64_t)blockIdx.x * FQ_CHUNK;
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x * 16u; i < FQ_CHUNK; i += FQ_THREADS * 16u) mine += (uint32_t)__popc(newline_mask_at(text, c0 + i, n));
    mine = (uint32_t)wave_sum_u64(mine);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of n 32-bit counts into 64-bit bases, base[n] = total (single 1024-thread workgroup)
__global__ __launch_bounds__(1024) void k_chunk_scan(const uint32_t *counts, uint32_t n, uint64_t *base)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const uint64_t v = i < n ? counts[i] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t up = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (i < n) base[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[n] = carry;
}

// line_start[l + 1] = offset behind the l-th newline (line_start[0] = 0 is set by the host); only the first
// `max_lines` lines are recorded
__global__ __launch_bounds__(FQ_THREADS) void k_line_starts(const uint8_t *__restrict__ text, uint64_t n, const uint64_t *__restrict__ base,
                                                            uint64_t max_lines, uint64_t *__restrict__ line_start)
{
    __shared__ uint32_t wsum[FQ_THREADS / 64];
    const uint64_t c0 = (uint64_t)blockIdx.x * FQ_CHUNK;
    uint64_t line = base[blockIdx.x];
    if (line >= max_lines) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i0 = 0; i0 < FQ_CHUNK; i0 += FQ_THREADS * 16u) {
        const uint64_t at = c0 + i0 + threadIdx.x * 16u;
        uint32_t mask = newline_mask_at(text, at, n);
        const uint32_t c = (uint32_t)__popc(mask);
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < FQ_THREADS / 64; ++w) {
            if (w < wave) before += wsum[w];
            all += wsum[w];
        }
        uint64_t l = line + before + incl - c;
        while (mask) {
            const uint32_t b = (uint32_t)__ffs((int)mask) - 1u;
            mask &= mask - 1u;
            if (l + 1 <= max_lines) line_start[l + 1] = at + b + 1;
            ++l;
        }
        line += all;
        __syncthreads();
    }
}

// record r = lines 4r .. 4r+3: start and length of its sequence line (a trailing '\r' is not part of a line); any
// record that does not look like FASTQ raises *bad
__global__ void k_records(const uint8_t *__restrict__ text, const uint64_t *__restrict__ line_start, uint64_t n_records,
                          uint64_t *__restrict__ seq_start, uint32_t *__restrict__ seq_len, unsigned long long *bad)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_records; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t l0 = line_start[4 * r], l1 = line_start[4 * r + 1], l2 = line_start[4 * r + 2], l3 = line_start[4 * r + 3];
        uint64_t len = l2 - l1 - 1;                                  // without the newline
        if (len && text[l1 + len - 1] == '\r') --len;
        const bool ok = text[l0] == '@' && l1 - l0 >= 2 && text[l2] == '+' && l3 > l2 && len <= KV_MAX_READ_LEN;
        if (!ok) atomicMin(bad, (unsigned long long)r);
        seq_start[r] = l1;
        seq_len[r] = ok ? (uint32_t)len : 0u;
    }
}

// k_pack_reads (kv_host.hip) for sequences that sit at seq_start[r] in the text instead of back to back
__global__ void k_pack_text(const uint8_t *__restrict__ text, const uint64_t *__restrict__ seq_start, const uint32_t *__restrict__ seq_len,
                            const uint64_t *__restrict__ woff, uint64_t n_reads, uint64_t n_words, uint32_t *__restrict__ words,
                            uint32_t *__restrict__ flags32)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t lo = 0, hi = n_reads;            // largest r with woff[r] <= w (empty reads share their successor's offset)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (woff[mid] <= w) lo = mid; else hi = mid;
        }
        const uint64_t r = lo;
        const uint64_t j0 = (w - woff[r]) * 16;
        const uint64_t len = seq_len[r];
        const uint8_t *src = text + seq_start[r] + j0;
        const uint32_t n = (uint32_t)(len - j0 < 16 ? len - j0 : 16);
        uint32_t out = 0, bad = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const uint8_t c = src[j];
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

// the last line of a file may lack its newline: give it one (the buffer has room)
__global__ void k_terminate(uint8_t *text, uint64_t n, unsigned long long *n_out)
{
    if (threadIdx.x || blockIdx.x) return;
    if (n && text[n - 1] != '\n') { text[n] = '\n'; *n_out = n + 1; }
    else *n_out = n;
}

// extents [line_start[4 idx], line_start[4 idx + 4]) of the requested records
__global__ void k_record_extents(const uint64_t *__restrict__ line_start, const uint64_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ ext)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        ext[2 * i] = line_start[4 * idx[i]];
        ext[2 * i + 1] = line_start[4 * idx[i] + 4];
    }
}

// one workgroup per requested record: its text to out[dst[i] ..]
__global__ __launch_bounds__(64) void k_record_copy(const uint8_t *__restrict__ text, const uint64_t *__restrict__ ext, const uint64_t *__restrict__ dst,
                                                    uint64_t n, uint8_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint64_t a = ext[2 * i], len = ext[2 * i + 1] - a, to = dst[i];
        for (uint64_t j = threadIdx.x; j < len; j += 64) out[to + j] = text[a + j];
    }
}

}  // namespace

// device buffers of one reader; they outlive it in a small pool (a sample is usually read twice -- count, then novel --
// and hipMalloc / hipFree of gigabytes cost more than parsing a few million reads)
// Upload of a stretch of a file through pinned staging buffers.  hipMemcpyAsync from the (pageable) mapping of the file copies on
// the calling thread into the runtime's own staging buffers: ~8 GB/s per thread, 25 GB/s for three samples read side by side, half of
// what the link takes.  Here KV_STAGE_THREADS threads pread() a chunk of the file into one of three pinned buffers while the previous
// chunks are on their way (an event per buffer says when it may be filled again); the DMA then runs at the link's rate.
#define KV_STAGE_SLOTS 3
#define KV_STAGE_CHUNK (16u << 20)
struct KvStager {
    void *slot[KV_STAGE_SLOTS] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[KV_STAGE_SLOTS];
    bool busy[KV_STAGE_SLOTS] = {false, false, false};
    bool ready = false, broken = false;
    bool init()
    {
        if (ready || broken) return ready;
        for (int i = 0; i < KV_STAGE_SLOTS; ++i) {
            kv_thread_device();
            if (hipHostMalloc(&slot[i], KV_STAGE_CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                broken = true;
                return false;
            }
        }
        ready = true;
        return true;
    }
    void release()
    {
        if (!ready) return;
        for (int i = 0; i < KV_STAGE_SLOTS; ++i) { (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); (void)hipHostFree(slot[i]); slot[i] = nullptr; busy[i] = false; }
        ready = false;
    }
    // bytes [off, off + n) of fd to d_dst on stream st; false: nothing was issued (the caller copies from its mapping instead)
    bool upload(uint8_t *d_dst, int fd, uint64_t off, uint64_t n, hipStream_t st)
    {
        if (!init()) return false;
        int nthreads = 4;
        if (const char *e = kv_knob("KV_STAGE_THREADS")) nthreads = std::max(1, std::min(16, atoi(e)));
        int s = 0;
        for (uint64_t done = 0; done < n; done += KV_STAGE_CHUNK, s = (s + 1) % KV_STAGE_SLOTS) {
            const uint64_t len = std::min<uint64_t>(KV_STAGE_CHUNK, n - done);
            if (busy[s] && hipEventSynchronize(ev[s]) != hipSuccess) return false;
            char *dst = (char *)slot[s];
            std::vector<char> fine((size_t)nthreads, 1);
            auto fill = [&](int t) {
                uint64_t lo = len * (uint64_t)t / (uint64_t)nthreads, hi = len * (uint64_t)(t + 1) / (uint64_t)nthreads;
                while (lo < hi) {
                    const ssize_t got = pread(fd, dst + lo, (size_t)(hi - lo), (off_t)(off + done + lo));
                    if (got <= 0) { fine[(size_t)t] = 0; return; }
                    lo += (uint64_t)got;
                }
            };
            std::vector<std::thread> crew;
            for (int t = 1; t < nthreads; ++t) crew.emplace_back(fill, t);
            fill(0);
            for (std::thread &th : crew) th.join();
            for (char ok : fine) if (!ok) { kv_set_error("reading the file failed while staging it for upload"); return false; }
            if (hipMemcpyAsync(d_dst + done, dst, len, hipMemcpyHostToDevice, st) != hipSuccess) return false;
            if (hipEventRecord(ev[s], st) != hipSuccess) return false;
            busy[s] = true;
        }
        return true;
    }
};

struct FastqBuffers {
    KvArena text[2];                // the batch being served / the batch before it (the carried tail moves across)
    KvArena comp, lines, recs, scratch, fetch;
    KvGunzipArenas gz;
    KvStager stage;
    void release()
    {
        for (KvArena *a : {&text[0], &text[1], &comp, &lines, &recs, &scratch, &fetch})
            if (a->p) { (void)hipFree(a->p); a->p = nullptr; a->bytes = 0; }
        gz.release();
        stage.release();
    }
};
namespace {
std::vector<FastqBuffers *> g_fastq_pool;
std::mutex g_fastq_pool_mu;
}

struct KvFastqDevice {
    std::string path;
    int fd = -1;
    const uint8_t *image = nullptr;
    size_t image_size = 0;
    std::vector<KvBgzfMember> members;
    size_t next_member = 0;
    bool plain = false;             // uncompressed FASTQ: the file's bytes are the text, uploaded as they are
    KvGunzip *gz = nullptr;         // an ordinary gzip stream (not BGZF): inflated a segment at a time (kv_gunzip.hip)
    uint64_t next_byte = 0;
    FastqBuffers *buf = nullptr;
    KvArena *text = nullptr;        // = buf->text
    int cur = 0;
    uint64_t carry_at = 0, carry_len = 0;     // unconsumed tail of text[cur]
    uint64_t *d_line_start = nullptr;         // of the batch being served
    uint64_t n_batch = 0;
    double bytes_per_read = 0.0;
    bool done = false;
};

KvFastqDevice *kv_fastq_device_open(const char *path)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 28) { close(fd); return nullptr; }
    void *map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return nullptr; }
    KvFastqDevice *d = new KvFastqDevice();
    d->path = path; d->fd = fd; d->image = (const uint8_t *)map; d->image_size = (size_t)sb.st_size;
    int yes = 0;
    if (d->image[0] == '@') d->plain = true;
    else kv_bgzf_index(d->image, d->image_size, &d->members, &yes);
    if (!yes && !d->plain) {
        const char *off = kv_knob("KV_GUNZIP");
        if (off && !strcmp(off, "host")) { kv_fastq_device_close(d); return nullptr; }
    }
    {
        std::lock_guard<std::mutex> lk(g_fastq_pool_mu);
        if (!g_fastq_pool.empty()) { d->buf = g_fastq_pool.back(); g_fastq_pool.pop_back(); }
    }
    if (!d->buf) d->buf = new FastqBuffers();
    d->text = d->buf->text;
    if (!yes && !d->plain) {
        d->gz = kv_gunzip_open(d->image, d->image_size, &d->buf->gz);
        if (!d->gz) { kv_fastq_device_close(d); return nullptr; }
        {
            const char *stage_env = kv_knob("KV_STAGE");
            if (!(stage_env && atoi(stage_env) == 0)) {
                KvStager *stage = &d->buf->stage;
                const int fd_ = d->fd;
                kv_gunzip_set_uploader(d->gz, [stage, fd_](uint8_t *dst, uint64_t off, uint64_t n, hipStream_t st) { return stage->upload(dst, fd_, off, n, st); });
            }
        }
    }
    return d;
}

void kv_fastq_device_close(KvFastqDevice *d)
{
    if (!d) return;
    kv_gunzip_close(d->gz);
    if (d->image) munmap((void *)d->image, d->image_size);
    if (d->fd >= 0) close(d->fd);
    if (d->buf) {
        std::lock_guard<std::mutex> lk(g_fastq_pool_mu);
        if (g_fastq_pool.size() >= 3) d->buf->gz.release();            // (the gzip inflater's buffers are gigabytes: three sets are kept, a trio read side by side)
        if (g_fastq_pool.size() < 4) g_fastq_pool.push_back(d->buf);
        else { d->buf->release(); delete d->buf; }
    }
    delete d;
}

int kv_fastq_device_next(KvFastqDevice *d, uint64_t max_reads, kv_reads **reads_out, uint64_t *n_out)
{
    *n_out = 0;
    *reads_out = nullptr;
    if (d->done) return KV_OK;             // (the batch served last stays served: its records can still be fetched)
    hipStream_t st = kv_stream();
    const bool verbose = kv_knob("KV_INGEST_VERBOSE") != nullptr;   // wall time of the steps of a batch on stderr
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!verbose) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[kv_ingest] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_mark).count());
        t_mark = now;
    };
    const char *cap_env = kv_knob("KV_INGEST_TEXT_MB");            // tests shrink the batches
    const uint64_t text_cap = (cap_env ? strtoull(cap_env, nullptr, 10) : 4096ull) << 20;
    const double per_read = d->bytes_per_read > 0 ? d->bytes_per_read : 280.0;
    uint64_t want = std::min<uint64_t>((uint64_t)((double)max_reads * per_read * 1.02) + 65536, text_cap);
    for (;;) {
        // ---- members (or, for an uncompressed file, bytes) of this batch
        const size_t m0 = d->next_member;
        size_t m1 = m0;
        uint64_t fresh = 0;
        const uint64_t b0 = d->next_byte;
        bool gz_last = false;
        if (d->gz) {
            const uint64_t ask = want > d->carry_len + (1u << 20) ? want - d->carry_len : (1u << 20);
            const int rc = kv_gunzip_decode(d->gz, ask + ask / 20, &fresh, &gz_last);
            if (rc != KV_OK) return rc;
        } else if (d->plain) {
            fresh = std::min<uint64_t>(d->image_size - b0, want > d->carry_len + 65536 ? want - d->carry_len : 65536);
        } else {
            while (m1 < d->members.size() && (m1 == m0 || d->carry_len + fresh + d->members[m1].isize <= want)) fresh += d->members[m1++].isize;
        }
        lap(d->gz ? "gunzip: decode" : "select");
        const bool final = d->gz ? gz_last : d->plain ? b0 + fresh == d->image_size : m1 == d->members.size();
        const uint64_t total_in = d->carry_len + fresh;
        if (total_in == 0) { d->done = true; return KV_OK; }     // end of file: nothing of the previous batch has been touched
        // (the batch served last -- its text buffer, its line starts, n_batch -- stays fetchable until a NEW batch with records
        // in it has been produced: a caller that kept it asks once more only to learn that the file has ended, and a file that
        // ends in blank lines gets here with a carry of a byte or two)
        const int nxt = d->cur ^ 1;
        KV_HIP(d->text[nxt].need(kv_round_up(total_in + 64, 4096)));
        uint8_t *text = (uint8_t *)d->text[nxt].p;
        if (d->carry_len) KV_HIP(hipMemcpyAsync(text, (const uint8_t *)d->text[d->cur].p + d->carry_at, d->carry_len, hipMemcpyDeviceToDevice, st));
        if (d->plain && fresh) {
            // big stretches through the pinned staging buffers (KV_STAGE=0: straight from the mapping, as small ones go)
            const char *stage_env = kv_knob("KV_STAGE");
            const bool staged = fresh >= (64u << 20) && !(stage_env && atoi(stage_env) == 0) && d->buf->stage.upload(text + d->carry_len, d->fd, b0, fresh, st);
            if (!staged) KV_HIP(hipMemcpyAsync(text + d->carry_len, d->image + b0, fresh, hipMemcpyHostToDevice, st));
        }
        if (d->gz && fresh) { const int rc = kv_gunzip_emit(d->gz, text + d->carry_len); if (rc != KV_OK) return rc; }
        if (m1 > m0) {
            const uint64_t c0 = d->members[m0].in_off, c1 = d->members[m1 - 1].in_off + d->members[m1 - 1].in_len;
            // (a damaged member is read at most ~600 bytes past its end before k_inflate catches it: one dynamic block
            // header, or one row of a stored block; the slack is zeroed so that what it decodes there is an error, not noise)
            KV_HIP(d->buf->comp.need(kv_round_up(c1 - c0 + KV_INFLATE_SLACK, 4096)));
            {
                const char *stage_env = kv_knob("KV_STAGE");
                const bool staged = c1 - c0 >= (64u << 20) && !(stage_env && atoi(stage_env) == 0) && d->buf->stage.upload((uint8_t *)d->buf->comp.p, d->fd, c0, c1 - c0, st);
                if (!staged) KV_HIP(hipMemcpyAsync(d->buf->comp.p, d->image + c0, c1 - c0, hipMemcpyHostToDevice, st));
            }
            KV_HIP(hipMemsetAsync((uint8_t *)d->buf->comp.p + (c1 - c0), 0, KV_INFLATE_SLACK, st));
            std::vector<uint64_t> text_off(m1 - m0);
            uint64_t at = d->carry_len;
            for (size_t i = m0; i < m1; ++i) { text_off[i - m0] = at; at += d->members[i].isize; }
            const int rc = kv_bgzf_inflate((const uint8_t *)d->buf->comp.p, c0, d->members.data() + m0, m1 - m0, text_off.data(), text, d->buf->scratch);
            if (rc != KV_OK) return rc;
        }
        lap("text on the device");
        // ---- lines
        const uint32_t n_chunks = (uint32_t)((total_in + 1 + FQ_CHUNK - 1) / FQ_CHUNK);
        const size_t b_counts = kv_round_up((uint64_t)n_chunks * 4, 256), b_base = kv_round_up(((uint64_t)n_chunks + 1) * 8, 256);
        KV_HIP(d->buf->scratch.need(b_counts + b_base + 256));
        uint32_t *d_counts = (uint32_t *)d->buf->scratch.p;
        uint64_t *d_base = (uint64_t *)((unsigned char *)d->buf->scratch.p + b_counts);
        unsigned long long *d_ctr = (unsigned long long *)((unsigned char *)d->buf->scratch.p + b_counts + b_base);
        unsigned long long total = total_in;
        if (final) {
            hipLaunchKernelGGL(k_terminate, dim3(1), dim3(1), 0, st, text, total_in, d_ctr);
            KV_HIP(hipMemcpyAsync(&total, d_ctr, 8, hipMemcpyDeviceToHost, st));
            KV_HIP(hipStreamSynchronize(st));
        }
        unsigned long long n_lines = 0;
        {
            KvProfScope prof("k_count_lines");
            hipLaunchKernelGGL(k_count_lines, dim3(n_chunks), dim3(FQ_THREADS), 0, st, (const uint8_t *)text, (uint64_t)total, d_counts);
            hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *)d_counts, n_chunks, d_base);
        }
        KV_HIP(hipGetLastError());
        KV_HIP(hipMemcpyAsync(&n_lines, d_base + n_chunks, 8, hipMemcpyDeviceToHost, st));
        KV_HIP(hipStreamSynchronize(st));
        uint64_t n = std::min<uint64_t>(n_lines / 4, max_reads);
        if (d->gz && !final && n_lines / 4 < max_reads && (total_in < text_cap || n_lines < 4)) {
            // the segment held fewer records than asked for (its size is a guess from the compression ratio so far) and the text
            // budget is not spent: what has been inflated waits as the carry and another segment joins it
            const uint64_t have = n_lines / 4;
            const double each = have ? (double)total_in / (double)have : per_read;
            d->n_batch = 0;               // the other text buffer -- the previous batch's -- is written next
            d->cur = nxt;
            d->carry_at = 0;
            d->carry_len = total_in;
            want = std::min<uint64_t>(total_in + (uint64_t)((double)(max_reads - have) * each * 1.05) + (1u << 20), std::max<uint64_t>(text_cap, total_in + (1u << 20)));
            continue;
        }
        if (n == 0 && !final) {           // not one whole record yet (huge records or a tiny budget): take more members
            d->next_member = m0;
            d->next_byte = b0;
            want = want * 2 + 65536;
            continue;
        }
        if (n == 0 && final && n_lines % 4 != 0) {
            // blank lines behind the last record (a common way for a FASTQ file to end) are not a partial record: the few
            // bytes are looked at on the host; only a genuinely cut record sends the file to the host parser
            bool blank = total_in <= 4096;
            if (blank) {
                unsigned char tail[4096];
                KV_HIP(hipMemcpyAsync(tail, text, total_in, hipMemcpyDeviceToHost, st));
                KV_HIP(hipStreamSynchronize(st));
                for (uint64_t i = 0; i < total_in && blank; ++i) blank = tail[i] == '\n' || tail[i] == '\r' || tail[i] == ' ' || tail[i] == '\t';
            }
            if (!blank) {
                kv_set_error("%s ends inside a FASTQ record", d->path.c_str());
                return KV_ERR_TYPE;
            }
            d->done = true; d->carry_len = 0;
            return KV_OK;
        }
        d->next_member = m1;
        d->next_byte = b0 + (d->plain ? fresh : 0);
        if (n == 0) { d->done = true; d->carry_len = 0; return KV_OK; }      // blank tail: the previous batch is still the current one
        d->cur = nxt;
        d->n_batch = 0;
        // ---- line starts, records
        KV_HIP(d->buf->lines.need(kv_round_up((4 * n + 2) * 8, 256)));
        uint64_t *line_start = (uint64_t *)d->buf->lines.p;
        KV_HIP(hipMemsetAsync(line_start, 0, 8, st));
        KV_HIP(d->buf->recs.need(kv_round_up(n * 8, 256) + kv_round_up(n * 4, 256) + 256));
        uint64_t *d_seq_start = (uint64_t *)d->buf->recs.p;
        uint32_t *d_seq_len = (uint32_t *)((unsigned char *)d->buf->recs.p + kv_round_up(n * 8, 256));
        unsigned long long *d_bad = (unsigned long long *)((unsigned char *)d->buf->recs.p + kv_round_up(n * 8, 256) + kv_round_up(n * 4, 256));
        KV_HIP(hipMemsetAsync(d_bad, 0xFF, 8, st));
        {
            KvProfScope prof("k_line_starts");
            hipLaunchKernelGGL(k_line_starts, dim3(n_chunks), dim3(FQ_THREADS), 0, st, (const uint8_t *)text, (uint64_t)total, (const uint64_t *)d_base,
                               (uint64_t)(4 * n), line_start);
        }
        {
            KvProfScope prof("k_records");
            hipLaunchKernelGGL(k_records, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 8192)), dim3(256), 0, st, (const uint8_t *)text,
                               (const uint64_t *)line_start, n, d_seq_start, d_seq_len, d_bad);
        }
        KV_HIP(hipGetLastError());
        std::vector<uint32_t> lens(n);
        unsigned long long bad = 0, consumed = 0;
        KV_HIP(hipMemcpyAsync(lens.data(), d_seq_len, n * 4, hipMemcpyDeviceToHost, st));
        KV_HIP(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
        KV_HIP(hipMemcpyAsync(&consumed, line_start + 4 * n, 8, hipMemcpyDeviceToHost, st));
        KV_HIP(hipStreamSynchronize(st));
        if (bad != ~0ull) {
            kv_set_error("%s: record %llu of the batch is not four-line FASTQ", d->path.c_str(), bad);
            return KV_ERR_TYPE;
        }
        d->carry_at = consumed;
        d->carry_len = total - consumed;
        if (final && d->carry_len == 0) d->done = true;
        d->bytes_per_read = (double)consumed / (double)n;
        d->d_line_start = line_start;
        d->n_batch = n;
        lap("lines and records");
        const int rc = kv_reads_from_device_text(text, d_seq_start, d_seq_len, lens.data(), n, reads_out);
        if (rc != KV_OK) return rc;
        lap("packed batch");
        *n_out = n;
        return KV_OK;
    }
}

// launches k_pack_text for kv_reads_from_device_text (kv_host.hip owns the kv_reads layout)
void kv_fastq_pack_launch(const uint8_t *d_text, const uint64_t *d_seq_start, const uint32_t *d_seq_len, const uint64_t *d_woff, uint64_t n_reads,
                          uint64_t n_words, uint32_t *d_words, uint32_t *d_flags32, hipStream_t st)
{
    KvProfScope prof("k_pack_text");
    const unsigned grid = (unsigned)std::min<uint64_t>((n_words + 255) / 256, 65536);
    hipLaunchKernelGGL(k_pack_text, dim3(grid), dim3(256), 0, st, d_text, d_seq_start, d_seq_len, d_woff, n_reads, n_words, d_words, d_flags32);
}

int kv_fastq_device_fetch(KvFastqDevice *d, const uint64_t *idx, uint64_t n, std::string *blob, std::vector<uint64_t> *offs)
{
    blob->clear();
    offs->assign(1, 0);
    if (n == 0) return KV_OK;
    for (uint64_t i = 0; i < n; ++i)
        KV_REQUIRE(idx[i] < d->n_batch, KV_ERR_ARG, "record %llu is not in the current batch of %llu", (unsigned long long)idx[i], (unsigned long long)d->n_batch);
    hipStream_t st = kv_stream();
    const size_t b_idx = kv_round_up(n * 8, 256), b_ext = kv_round_up(n * 16, 256);
    KV_HIP(d->buf->fetch.need(2 * b_idx + b_ext));
    uint64_t *d_idx = (uint64_t *)d->buf->fetch.p;
    uint64_t *d_ext = (uint64_t *)((unsigned char *)d->buf->fetch.p + b_idx);
    uint64_t *d_dst = (uint64_t *)((unsigned char *)d->buf->fetch.p + b_idx + b_ext);
    KV_HIP(hipMemcpyAsync(d_idx, idx, n * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_record_extents, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, (const uint64_t *)d->d_line_start,
                       (const uint64_t *)d_idx, n, d_ext);
    KV_HIP(hipGetLastError());
    std::vector<uint64_t> ext(2 * n);
    KV_HIP(hipMemcpyAsync(ext.data(), d_ext, n * 16, hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    offs->resize(n + 1);
    for (uint64_t i = 0; i < n; ++i) (*offs)[i + 1] = (*offs)[i] + (ext[2 * i + 1] - ext[2 * i]);
    const uint64_t bytes = (*offs)[n];
    uint8_t *d_out = nullptr;
    KV_HIP(kv_hip_malloc((void **)&d_out, bytes + 16));
    hipError_t e = hipMemcpyAsync(d_dst, offs->data(), n * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_record_copy, dim3((unsigned)std::min<uint64_t>(n, 16384)), dim3(64), 0, st, (const uint8_t *)d->text[d->cur].p, (const uint64_t *)d_ext,
                           (const uint64_t *)d_dst, n, d_out);
        e = hipGetLastError();
    }
    blob->resize(bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(&(*blob)[0], d_out, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_out);
    if (e != hipSuccess) { kv_set_error("kv_fastq_device_fetch: %s", hipGetErrorString(e)); return KV_ERR_HIP; }
    return KV_OK;
}
