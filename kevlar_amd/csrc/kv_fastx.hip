// kv_fastx.hip -- native FASTA/FASTQ ingest (the khmer.ReadParser stand-in of kevlar/count.py:40).
//
// At the rates the count/novel kernels run, a Python record parser is the bottleneck by two orders
// of magnitude.  This reader inflates (zlib; plain files pass
// through), splits records, 2-bit packs the sequences straight into a kv_reads batch in HBM and
// keeps the batch's text (names / sequences / qualities as blobs + offsets) so that the host only
// materialises Python records for the few reads that end up annotated.
//
// Record rules (same as kevlar_amd.khmer._iter_fastx): '@name' / sequence / '+' / quality, or
// '>name' followed by sequence lines up to the next '>'; name = the header line after its first
// character; blank lines between records are ignored.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>

#include "kv_binned.h"          // KvArena, KvGunzipArenas
#include "kv_internal.h"

// ---------------------------------------------------------------------------------------
// Packed-read cache: `kevlar count` parses a sample, `kevlar novel` parses the case sample again
// (kevlar/count.py:40, kevlar/__init__.py:125-128) -- and a gzip stream inflates at ~250 MB/s on one core however fast
// the GPU is.  With KEVLAR_PACK_CACHE=1 the first complete pass over FILE leaves FILE.kvpack next to it: per block of
// reads the 2-bit packed words exactly as the device packed them, lengths, flags, names, qualities and the few
// characters outside ACGT, so that a later open of FILE (same size and mtime) streams blocks from the page cache
// straight into HBM -- no inflate, no record splitting, no packing kernel -- and can still reproduce every record's
// text byte for byte.
//   file  : "KVPK" u32 version  u64 src_size  u64 src_mtime_ns  u64 total_reads   then blocks, then a block with n = 0
//   block : "KVPB" u32 n  u64 n_words  u64 names_bytes  u64 quals_bytes  u64 n_exc
//           u32 len[n]  u32 name_len[n]  u32 qual_len[n]  u8 flags[n] (bit0 non-ACGT, bit1 FASTQ; padded to 8)
//           u32 words[n_words] (padded to 8)  names  quals (each padded to 8)  u64 exc_pos[n_exc]  u8 exc_char[n_exc] (padded)
// ---------------------------------------------------------------------------------------
#define KVPK_VERSION 1u
#define KVPK_BLOCK_READS 65536u

struct PackBlock {                 // pointers into the mapped cache file
    uint32_t n = 0;
    uint64_t n_words = 0, names_bytes = 0, quals_bytes = 0, n_exc = 0;
    const uint32_t *len = nullptr, *name_len = nullptr, *qual_len = nullptr, *words = nullptr;
    const uint8_t *flags = nullptr, *exc_char = nullptr;
    const char *names = nullptr, *quals = nullptr;
    const uint64_t *exc_pos = nullptr;
    std::vector<uint64_t> woff, noff, qoff, boff;   // per-read prefix sums (words, names, quals, bases), built on load
};

struct PackReader {
    int fd = -1;
    const uint8_t *base = nullptr;
    size_t size = 0, pos = 0;
    std::vector<PackBlock> batch;          // blocks of the batch last handed out
    std::vector<uint64_t> first;           // first read of each of them inside the batch
};

struct PackWriter {
    FILE *fh = nullptr;
    std::string tmp, final_path;
    uint64_t total = 0;
    bool ok = true;
};

// A gzip file whose text is not four-line FASTQ (FASTA: a reference genome, contigs; FASTQ with blank lines) is still
// inflated on the device -- kv_gunzip.hip, a segment of text at a time -- and the TEXT comes back for the host's record
// parser: zlib inflates at ~0.35 GB/s of text on one core, the device at ~10 GB/s, the copy back at PCIe speed
// (kevlar/__init__.py:125-128 reads every sequence file through one gzip stream).  What the device decoder declines
// (no dynamic-Huffman block for megabytes, damaged stream: KV_ERR_TYPE) goes on through zlib from the same text offset.
struct DevTextSource {
    int fd = -1;
    const uint8_t *image = nullptr;
    size_t image_size = 0;
    KvGunzip *gz = nullptr;        // one DEFLATE stream (gzip, pigz): parallel stretches between found block starts (kv_gunzip.hip)
    std::vector<KvBgzfMember> members;   // blocked gzip (bgzip, this package's writer): a wavefront per member, CRC-32 checked (kv_inflate.hip)
    size_t next_member = 0;
    KvGunzipArenas arenas;
    KvArena text, comp, scratch;
    uint64_t delivered = 0;        // bytes of text handed to the parser so far
    ~DevTextSource()
    {
        for (KvArena *a : {&comp, &scratch})
            if (a->p) (void)hipFree(a->p);
        if (gz) kv_gunzip_close(gz);
        arenas.release();
        if (text.p) (void)hipFree(text.p);
        if (image) munmap((void *)image, image_size);
        if (fd >= 0) close(fd);
    }
};

struct kv_fastx {
    gzFile fh = nullptr;
    DevTextSource *dsrc = nullptr; // non-null: fx_fill draws the text from the device inflater instead of gzread
    std::string path;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    std::string io_error;         // what zlib said when the stream turned out damaged or cut short
    std::string pending;          // a header line read ahead while scanning a FASTA record
    std::string pending_view;     // backing store of the view handed out for it
    bool have_pending = false;
    uint64_t num_reads = 0;
    // text of the last batch
    std::string names, seqs, quals;
    std::vector<uint64_t> name_offs, seq_offs, qual_offs;
    std::vector<uint8_t> is_fastq;   // per record: came with a quality line
    std::mutex mu;
    PackReader *cache = nullptr;     // serving from FILE.kvpack instead of parsing
    PackWriter *writer = nullptr;    // leaving FILE.kvpack behind
    // blocked gzip (BGZF) holding four-line FASTQ: inflated, split and packed on the device (kv_inflate.hip, kv_fastq.hip)
    bool dev_candidate = false;      // the file starts like BGZF and nothing has been parsed on the host yet
    KvFastqDevice *dev = nullptr;
};

static bool fx_fill(kv_fastx *f)
{
    if (f->eof) return false;
    if (f->pos > 0 && f->pos < f->end) memmove(f->buf.data(), f->buf.data() + f->pos, f->end - f->pos);
    f->end -= f->pos;
    f->pos = 0;
    if (f->end == f->buf.size()) f->buf.resize(f->buf.size() * 2);
    if (f->dsrc) {
        DevTextSource *d = f->dsrc;
        const char *seg_env = kv_knob("KV_INGEST_TEXT_MB");            // tests shrink the segments
        const uint64_t want = (seg_env ? strtoull(seg_env, nullptr, 10) : 256ull) << 20;
        uint64_t n = 0;
        bool last = false;
        int rc = KV_OK;
        if (!d->gz) {
            // blocked gzip: as many whole members as make up the segment
            const size_t m0 = d->next_member;
            size_t m1 = m0;
            while (m1 < d->members.size() && (m1 == m0 || n + d->members[m1].isize <= want)) n += d->members[m1++].isize;
            if (m1 > m0) {
                hipStream_t st = kv_stream();
                const uint64_t c0 = d->members[m0].in_off, c1 = d->members[m1 - 1].in_off + d->members[m1 - 1].in_len;
                std::vector<uint64_t> text_off(m1 - m0);
                uint64_t at = 0;
                for (size_t i = m0; i < m1; ++i) { text_off[i - m0] = at; at += d->members[i].isize; }
                if (d->comp.need(kv_round_up(c1 - c0 + KV_INFLATE_SLACK, 4096)) != hipSuccess || d->text.need(n + 256) != hipSuccess) { (void)hipGetLastError(); rc = KV_ERR_TYPE; }
                else if (hipMemcpyAsync(d->comp.p, d->image + c0, c1 - c0, hipMemcpyHostToDevice, st) != hipSuccess ||
                         hipMemsetAsync((uint8_t *)d->comp.p + (c1 - c0), 0, KV_INFLATE_SLACK, st) != hipSuccess) { (void)hipGetLastError(); rc = KV_ERR_TYPE; }
                else rc = kv_bgzf_inflate((const uint8_t *)d->comp.p, c0, d->members.data() + m0, m1 - m0, text_off.data(), (uint8_t *)d->text.p, d->scratch);
                if (rc == KV_OK) d->next_member = m1;
            }
        } else {
            while (rc == KV_OK && n == 0 && !kv_gunzip_done(d->gz)) rc = kv_gunzip_decode(d->gz, want, &n, &last);
            if (rc == KV_OK && n) {
                if (d->text.need(n + 256) != hipSuccess) { (void)hipGetLastError(); rc = KV_ERR_TYPE; }
                else rc = kv_gunzip_emit(d->gz, (uint8_t *)d->text.p);
            }
        }
        if (rc == KV_OK && n) {
            if (f->buf.size() - f->end < n) f->buf.resize(f->end + n);
            if (hipMemcpyAsync(f->buf.data() + f->end, d->text.p, n, hipMemcpyDeviceToHost, kv_stream()) != hipSuccess ||
                hipStreamSynchronize(kv_stream()) != hipSuccess) { (void)hipGetLastError(); rc = KV_ERR_TYPE; }
        }
        if (rc == KV_OK && n) {
            d->delivered += n;
            f->end += (size_t)n;
            return true;
        }
        if (rc == KV_OK) { f->eof = true; return false; }            // the stream has ended
        if (rc == KV_ERR_HIP && kv_last_hip_code != (int)hipErrorOutOfMemory) {
            // a device fault is not a reason to read the file another way: the parser reports it
            f->eof = true;
            f->io_error = kv_last_error();
            return false;
        }
        // the device path stops here: zlib takes over at the same text offset (it reports a damaged stream itself)
        const uint64_t skip = d->delivered;
        delete d;
        f->dsrc = nullptr;
        (void)hipGetLastError();
        if (gzseek(f->fh, (z_off_t)skip, SEEK_SET) < 0) { f->eof = true; f->io_error = "cannot resume the gzip stream on the host"; return false; }
    }
    const int got = gzread(f->fh, f->buf.data() + f->end, (unsigned)(f->buf.size() - f->end));
    if (got <= 0) {
        // the end of the file -- or of what can be read of it: a damaged or truncated gzip stream is an error, not a short file
        f->eof = true;
        int errnum = Z_OK;
        const char *msg = gzerror(f->fh, &errnum);
        if (got < 0 || (errnum != Z_OK && errnum != Z_STREAM_END)) f->io_error = msg && *msg ? msg : "read error";
        return false;
    }
    f->end += (size_t)got;
    return true;
}

// next line without its terminator (\n or \r\n) as a view into the inflate buffer, valid until the
// next call; false at end of file
static bool fx_view(kv_fastx *f, const char *&p, size_t &n)
{
    if (f->have_pending) {          // a header read ahead while scanning a FASTA record
        f->pending_view.swap(f->pending);
        f->have_pending = false;
        p = f->pending_view.data(); n = f->pending_view.size();
        return true;
    }
    for (;;) {
        const char *start = f->buf.data() + f->pos;
        const char *nl = (const char *)memchr(start, '\n', f->end - f->pos);
        if (nl) {
            size_t len = (size_t)(nl - start);
            f->pos += len + 1;
            if (len && start[len - 1] == '\r') --len;
            p = start; n = len;
            return true;
        }
        if (!fx_fill(f)) {
            if (f->pos < f->end) {          // last line without newline
                size_t len = f->end - f->pos;
                const char *s0 = f->buf.data() + f->pos;
                f->pos = f->end;
                if (len && s0[len - 1] == '\r') --len;
                p = s0; n = len;
                return true;
            }
            return false;
        }
    }
}

static bool fx_line(kv_fastx *f, std::string &out)
{
    const char *p;
    size_t n;
    if (!fx_view(f, p, n)) return false;
    out.assign(p, n);
    return true;
}

static inline bool fx_blank_view(const char *p, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (p[i] != ' ' && p[i] != '\t' && p[i] != '\r') return false;
    return true;
}

static inline bool fx_blank(const std::string &s)
{
    for (char c : s)
        if (c != ' ' && c != '\t' && c != '\r') return false;
    return true;
}

static inline void fx_strip(std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    if (a > 0 || b < s.size()) s = s.substr(a, b - a);
}

static inline size_t pad8(size_t v) { return (v + 7) & ~(size_t)7; }

static bool pack_source_signature(const char *path, uint64_t *size, uint64_t *mtime_ns)
{
    struct stat st;
    if (stat(path, &st) != 0) return false;
    *size = (uint64_t)st.st_size;
    *mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    return true;
}

static PackReader *pack_open(const char *path)
{
    uint64_t size = 0, mtime = 0;
    if (!pack_source_signature(path, &size, &mtime)) return nullptr;
    const std::string cpath = std::string(path) + ".kvpack";
    const int fd = open(cpath.c_str(), O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 40) { close(fd); return nullptr; }
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return nullptr; }
    const uint8_t *b = (const uint8_t *)m;
    uint32_t version;
    uint64_t hs, hm;
    memcpy(&version, b + 4, 4); memcpy(&hs, b + 8, 8); memcpy(&hm, b + 16, 8);
    if (memcmp(b, "KVPK", 4) != 0 || version != KVPK_VERSION || hs != size || hm != mtime) {   // stale or foreign: parse the source
        munmap(m, (size_t)st.st_size); close(fd);
        return nullptr;
    }
    PackReader *r = new PackReader();
    r->fd = fd; r->base = b; r->size = (size_t)st.st_size; r->pos = 32;
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    return r;
}

static void pack_close(PackReader *r)
{
    if (!r) return;
    if (r->base) munmap((void *)r->base, r->size);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

// next block of the cache, or false at the end marker / on a truncated file
static bool pack_next_block(PackReader *r, PackBlock &blk)
{
    if (r->pos + 48 > r->size || memcmp(r->base + r->pos, "KVPB", 4) != 0) return false;
    const uint8_t *p = r->base + r->pos;
    memcpy(&blk.n, p + 4, 4); memcpy(&blk.n_words, p + 8, 8); memcpy(&blk.names_bytes, p + 16, 8);
    memcpy(&blk.quals_bytes, p + 24, 8); memcpy(&blk.n_exc, p + 32, 8);
    if (blk.n == 0) return false;
    size_t at = r->pos + 40;
    auto take = [&](size_t bytes) { const uint8_t *q = r->base + at; at += pad8(bytes); return q; };
    blk.len = (const uint32_t *)take((size_t)blk.n * 4);
    blk.name_len = (const uint32_t *)take((size_t)blk.n * 4);
    blk.qual_len = (const uint32_t *)take((size_t)blk.n * 4);
    blk.flags = take(blk.n);
    blk.words = (const uint32_t *)take((size_t)blk.n_words * 4);
    blk.names = (const char *)take(blk.names_bytes);
    blk.quals = (const char *)take(blk.quals_bytes);
    blk.exc_pos = (const uint64_t *)take((size_t)blk.n_exc * 8);
    blk.exc_char = take(blk.n_exc);
    if (at > r->size) return false;
    r->pos = at;
    blk.woff.resize(blk.n + 1); blk.noff.resize(blk.n + 1); blk.qoff.resize(blk.n + 1); blk.boff.resize(blk.n + 1);
    uint64_t w = 0, nn = 0, q = 0, bb = 0;
    for (uint32_t i = 0; i < blk.n; ++i) {
        blk.woff[i] = w; blk.noff[i] = nn; blk.qoff[i] = q; blk.boff[i] = bb;
        w += ((uint64_t)blk.len[i] + 15) / 16; nn += blk.name_len[i]; q += blk.qual_len[i]; bb += blk.len[i];
    }
    blk.woff[blk.n] = w; blk.noff[blk.n] = nn; blk.qoff[blk.n] = q; blk.boff[blk.n] = bb;
    return w == blk.n_words && nn == blk.names_bytes && q == blk.quals_bytes;
}

static PackWriter *pack_writer_start(const char *path)
{
    uint64_t size = 0, mtime = 0;
    if (!pack_source_signature(path, &size, &mtime)) return nullptr;
    PackWriter *w = new PackWriter();
    w->final_path = std::string(path) + ".kvpack";
    w->tmp = w->final_path + ".tmp." + std::to_string((long)getpid());
    w->fh = fopen(w->tmp.c_str(), "wb");
    if (!w->fh) { delete w; return nullptr; }     // read-only directory: no cache, no error
    const uint32_t version = KVPK_VERSION;
    const uint64_t zero = 0;
    w->ok = fwrite("KVPK", 1, 4, w->fh) == 4 && fwrite(&version, 4, 1, w->fh) == 1 && fwrite(&size, 8, 1, w->fh) == 1 &&
            fwrite(&mtime, 8, 1, w->fh) == 1 && fwrite(&zero, 8, 1, w->fh) == 1;
    return w;
}

static void pack_write_padded(PackWriter *w, const void *data, size_t bytes)
{
    static const char zeros[8] = {0};
    if (bytes && fwrite(data, 1, bytes, w->fh) != bytes) w->ok = false;
    const size_t pad = pad8(bytes) - bytes;
    if (pad && fwrite(zeros, 1, pad, w->fh) != pad) w->ok = false;
}

// one parsed + uploaded batch -> cache blocks of at most KVPK_BLOCK_READS reads
static void pack_write_batch(PackWriter *w, const kv_fastx *f, const kv_reads *reads, uint64_t n)
{
    if (!w->ok) return;
    std::vector<uint32_t> words(reads->n_words ? reads->n_words : 1);
    std::vector<uint8_t> dflags(n ? n : 1);
    if (hipMemcpy(words.data(), reads->d_words, reads->n_words * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(dflags.data(), reads->d_flags, n, hipMemcpyDeviceToHost) != hipSuccess) { w->ok = false; return; }
    uint64_t wbase = 0;
    for (uint64_t lo = 0; lo < n; lo += KVPK_BLOCK_READS) {
        const uint32_t m = (uint32_t)std::min<uint64_t>(KVPK_BLOCK_READS, n - lo);
        std::vector<uint32_t> len(m), nlen(m), qlen(m);
        std::vector<uint8_t> flags(m);
        std::vector<uint64_t> epos;
        std::vector<uint8_t> echar;
        uint64_t nwords = 0, bases = 0;
        for (uint32_t i = 0; i < m; ++i) {
            const uint64_t r = lo + i;
            len[i] = (uint32_t)(f->seq_offs[r + 1] - f->seq_offs[r]);
            nlen[i] = (uint32_t)(f->name_offs[r + 1] - f->name_offs[r]);
            qlen[i] = (uint32_t)(f->qual_offs[r + 1] - f->qual_offs[r]);
            flags[i] = (uint8_t)((dflags[r] & 1u) | (f->is_fastq[r] ? 2u : 0u));
            if (dflags[r] & 1u) {            // keep what 2 bits cannot: every character that is not an upper-case A, C, G or T
                const char *sq = f->seqs.data() + f->seq_offs[r];
                for (uint32_t j = 0; j < len[i]; ++j)
                    if (sq[j] != 'A' && sq[j] != 'C' && sq[j] != 'G' && sq[j] != 'T') { epos.push_back(bases + j); echar.push_back((uint8_t)sq[j]); }
            }
            nwords += ((uint64_t)len[i] + 15) / 16;
            bases += len[i];
        }
        const uint64_t names_bytes = f->name_offs[lo + m] - f->name_offs[lo], quals_bytes = f->qual_offs[lo + m] - f->qual_offs[lo];
        const uint64_t n_exc = epos.size();
        if (fwrite("KVPB", 1, 4, w->fh) != 4 || fwrite(&m, 4, 1, w->fh) != 1 || fwrite(&nwords, 8, 1, w->fh) != 1 ||
            fwrite(&names_bytes, 8, 1, w->fh) != 1 || fwrite(&quals_bytes, 8, 1, w->fh) != 1 || fwrite(&n_exc, 8, 1, w->fh) != 1) w->ok = false;
        pack_write_padded(w, len.data(), (size_t)m * 4);
        pack_write_padded(w, nlen.data(), (size_t)m * 4);
        pack_write_padded(w, qlen.data(), (size_t)m * 4);
        pack_write_padded(w, flags.data(), m);
        pack_write_padded(w, words.data() + wbase, (size_t)nwords * 4);
        pack_write_padded(w, f->names.data() + f->name_offs[lo], names_bytes);
        pack_write_padded(w, f->quals.data() + f->qual_offs[lo], quals_bytes);
        pack_write_padded(w, epos.data(), (size_t)n_exc * 8);
        pack_write_padded(w, echar.data(), n_exc);
        wbase += nwords;
    }
    w->total += n;
}

static void pack_writer_finish(PackWriter *w, bool complete)
{
    if (!w) return;
    if (w->fh) {
        if (complete && w->ok) {
            const uint32_t zero32 = 0;
            const uint64_t zero = 0;
            w->ok = fwrite("KVPB", 1, 4, w->fh) == 4 && fwrite(&zero32, 4, 1, w->fh) == 1;
            for (int i = 0; i < 4 && w->ok; ++i) w->ok = fwrite(&zero, 8, 1, w->fh) == 1;      // end marker: an empty block header
            if (w->ok) w->ok = fseek(w->fh, 24, SEEK_SET) == 0 && fwrite(&w->total, 8, 1, w->fh) == 1;
        }
        if (fclose(w->fh) != 0) w->ok = false;
        if (complete && w->ok) (void)rename(w->tmp.c_str(), w->final_path.c_str());     // appears atomically, or not at all
        else (void)unlink(w->tmp.c_str());
    }
    delete w;
}

// map the file and start a device inflater over it; on any failure the handle simply keeps its zlib stream
static void fx_open_device_text(kv_fastx *f, size_t size)
{
    DevTextSource *d = new DevTextSource();
    d->fd = open(f->path.c_str(), O_RDONLY);
    if (d->fd < 0) { delete d; return; }
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, d->fd, 0);
    if (map == MAP_FAILED) { delete d; return; }
    d->image = (const uint8_t *)map; d->image_size = size;
    int is_bgzf = 0;
    if (kv_bgzf_index(d->image, size, &d->members, &is_bgzf) != KV_OK || !is_bgzf || d->members.empty()) {
        d->members.clear();
        d->gz = kv_gunzip_open(d->image, size, &d->arenas);
        if (!d->gz) { delete d; return; }
    }
    f->dsrc = d;
}

extern "C" int kv_fastx_open(const char *path, kv_fastx **out)
{
    KV_REQUIRE(path && out, KV_ERR_ARG, "kv_fastx_open: null argument");
    const char *mode = kv_knob("KEVLAR_PACK_CACHE");          // unset or "0": never look at caches; "1": use and create them
    const bool caching = mode && atoi(mode) != 0;
    kv_fastx *f = new kv_fastx();
    f->path = path;
    if (caching) f->cache = pack_open(path);
    if (!f->cache) {
        gzFile fh = gzopen(path, "rb");
        if (!fh) {
            delete f;
            kv_set_error("cannot open sequence file %s", path);
            return KV_ERR_IO;
        }
        gzbuffer(fh, 1 << 20);
        f->fh = fh;
        f->buf.resize(4 << 20);
        if (caching) f->writer = pack_writer_start(path);
        const char *ingest = kv_knob("KV_INGEST");               // "host": never parse on the device
        if (!caching && !(ingest && strcmp(ingest, "host") == 0)) {
            // the device takes four-line FASTQ, uncompressed ('@' first) or gzip of either kind: for a compressed file the first
            // byte of TEXT is looked at (a few KB of the file through zlib; nothing of the stream handle is touched)
            unsigned char head[8192];
            FILE *probe = fopen(path, "rb");
            if (probe) {
                const size_t got = fread(head, 1, sizeof(head), probe);
                fclose(probe);
                if (got >= 18 && head[0] == '@') f->dev_candidate = true;
                else if (got >= 18 && head[0] == 0x1f && head[1] == 0x8b && head[2] == 8) {
                    z_stream zs;
                    memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, 15 + 16) == Z_OK) {
                        unsigned char first = 0;
                        zs.next_in = head; zs.avail_in = (uInt)got;
                        zs.next_out = &first; zs.avail_out = 1;
                        const int rc = inflate(&zs, Z_SYNC_FLUSH);
                        f->dev_candidate = (rc == Z_OK || rc == Z_STREAM_END || rc == Z_BUF_ERROR) && zs.avail_out == 0 && first == '@';
                        inflateEnd(&zs);
                        // any other text in a gzip file of some size (FASTA): inflate on the device, parse here
                        const char *big = kv_knob("KV_GUNZIP_TEXT_MIN_MB");
                        const uint64_t min_bytes = (big ? strtoull(big, nullptr, 10) : 4ull) << 20;
                        struct stat sb;
                        int ndev = 0;
                        if (!f->dev_candidate && stat(path, &sb) == 0 && (uint64_t)sb.st_size >= min_bytes &&
                            hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0)
                            fx_open_device_text(f, (size_t)sb.st_size);
                        else (void)hipGetLastError();
                    }
                }
            }
        }
    }
    *out = f;
    return KV_OK;
}

extern "C" int kv_fastx_close(kv_fastx *f)
{
    if (!f) return KV_OK;
    if (f->fh) gzclose(f->fh);
    delete f->dsrc;
    kv_fastq_device_close(f->dev);
    pack_writer_finish(f->writer, false);       // a complete pass has already finished (and detached) its writer
    pack_close(f->cache);
    delete f;
    return KV_OK;
}

// 1 if this handle streams from a packed-read cache (the text of a batch is then fetched per record)
extern "C" int kv_fastx_from_cache(kv_fastx *f, int *yes)
{
    KV_REQUIRE(f && yes, KV_ERR_ARG, "kv_fastx_from_cache: null argument");
    *yes = f->cache ? 1 : 0;
    return KV_OK;
}

extern "C" int kv_fastx_num_reads(kv_fastx *f, uint64_t *n)
{
    KV_REQUIRE(f && n, KV_ERR_ARG, "kv_fastx_num_reads: null argument");
    *n = f->num_reads;
    return KV_OK;
}

// records parsed on the host into the handle's blobs (appended); *n_out = how many
static int host_parse(kv_fastx *f, uint64_t max_reads, uint64_t *n_out)
{
    uint64_t n = 0;
    const char *lp;
    size_t ln;
    while (n < max_reads && fx_view(f, lp, ln)) {
        if (fx_blank_view(lp, ln)) continue;
        const char first = lp[0];
        if (first == '@') {
            // four lines, each appended to its blob straight from the inflate buffer
            f->names.append(lp + 1, ln - 1);
            if (fx_view(f, lp, ln)) f->seqs.append(lp, ln);
            (void)fx_view(f, lp, ln);                               // '+' line
            if (fx_view(f, lp, ln)) f->quals.append(lp, ln);
            f->is_fastq.push_back(1);
        } else if (first == '>') {
            f->names.append(lp + 1, ln - 1);
            std::string piece;
            while (fx_line(f, piece)) {
                if (!piece.empty() && piece[0] == '>') { f->pending.swap(piece); f->have_pending = true; break; }
                fx_strip(piece);
                f->seqs += piece;
            }
            f->is_fastq.push_back(0);
        } else {
            kv_set_error("cannot parse sequence file %s", f->path.c_str());
            return KV_ERR_IO;
        }
        f->name_offs.push_back(f->names.size());
        f->seq_offs.push_back(f->seqs.size());
        f->qual_offs.push_back(f->quals.size());
        ++n;
    }
    if (!f->io_error.empty()) {
        kv_set_error("%s: %s", f->path.c_str(), f->io_error.c_str());
        return KV_ERR_IO;
    }
    *n_out = n;
    return KV_OK;
}

extern "C" int kv_fastx_next(kv_fastx *f, uint64_t max_reads, int upload, kv_reads **reads_out, uint64_t *n_reads_out)
{
    KV_REQUIRE(f && n_reads_out && max_reads > 0, KV_ERR_ARG, "kv_fastx_next: bad argument");
    std::lock_guard<std::mutex> lk(f->mu);
    f->names.clear(); f->seqs.clear(); f->quals.clear();
    f->name_offs.assign(1, 0); f->seq_offs.assign(1, 0); f->qual_offs.assign(1, 0);
    f->is_fastq.clear();
    if (f->cache) {
        // whole blocks until max_reads would be exceeded (at least one): lengths, flags and packed words are
        // concatenated and uploaded as they are; names are copied out (small), sequences and qualities stay in the
        // file mapping and are produced per record on request (kv_fastx_record_text)
        PackReader *c = f->cache;
        c->batch.clear(); c->first.clear();
        std::vector<uint32_t> lens, words;
        std::vector<uint8_t> flags;
        uint64_t n = 0;
        for (;;) {
            const size_t mark = c->pos;
            PackBlock blk;
            if (!pack_next_block(c, blk)) break;
            if (n > 0 && n + blk.n > max_reads) { c->pos = mark; break; }
            lens.insert(lens.end(), blk.len, blk.len + blk.n);
            flags.insert(flags.end(), blk.flags, blk.flags + blk.n);
            words.insert(words.end(), blk.words, blk.words + blk.n_words);
            f->names.append(blk.names, blk.names_bytes);
            for (uint32_t i = 0; i < blk.n; ++i) {
                f->name_offs.push_back(f->name_offs.back() + blk.name_len[i]);
                f->seq_offs.push_back(f->seq_offs.back() + blk.len[i]);
                f->qual_offs.push_back(f->qual_offs.back() + blk.qual_len[i]);
                f->is_fastq.push_back((blk.flags[i] >> 1) & 1u);
            }
            c->first.push_back(n);
            n += blk.n;
            c->batch.push_back(std::move(blk));
        }
        for (uint8_t &fl : flags) fl &= 1u;
        f->num_reads += n;
        *n_reads_out = n;
        if (reads_out) {
            *reads_out = nullptr;
            if (upload && n > 0) return kv_reads_from_packed_var(words.data(), lens.data(), flags.data(), n, reads_out);
        }
        return KV_OK;
    }
    // ---- device path: BGZF + FASTQ, asked for as packed batches from the first call on
    if (f->dev_candidate && !f->dev && f->num_reads == 0 && upload && reads_out) {
        f->dev = kv_fastq_device_open(f->path.c_str());
        if (!f->dev) f->dev_candidate = false;
    }
    if (f->dev) {
        int rc = upload && reads_out ? kv_fastq_device_next(f->dev, max_reads, reads_out, n_reads_out) : KV_ERR_TYPE;
        if (rc == KV_OK) {
            f->num_reads += *n_reads_out;
            return KV_OK;
        }
        // not four-line FASTQ after all (or the caller wants host text now): the host parser takes over where the device
        // path stopped
        kv_fastq_device_close(f->dev);
        f->dev = nullptr;
        f->dev_candidate = false;
        // (KV_ERR_HIP with hipErrorOutOfMemory: the device path's scratch did not fit next to big sketches, or a smaller GPU --
        // the host parser reads the same file with a few megabytes.  Any other HIP error -- an illegal address, a lost context --
        // is a fault of the device path, not a reason to read the file another way: it goes to the caller)
        if (rc == KV_ERR_HIP && kv_last_hip_code != (int)hipErrorOutOfMemory) return rc;
        if (rc != KV_ERR_TYPE && rc != KV_ERR_HIP) return rc;
        (void)hipGetLastError();
        uint64_t left = f->num_reads;
        while (left > 0) {
            uint64_t got = 0;
            rc = host_parse(f, std::min<uint64_t>(left, 65536), &got);
            if (rc != KV_OK) return rc;
            if (got == 0) break;
            left -= got;
            f->names.clear(); f->seqs.clear(); f->quals.clear();
            f->name_offs.assign(1, 0); f->seq_offs.assign(1, 0); f->qual_offs.assign(1, 0);
            f->is_fastq.clear();
        }
    }
    f->dev_candidate = false;
    uint64_t n = 0;
    { const int rc = host_parse(f, max_reads, &n); if (rc != KV_OK) return rc; }
    f->num_reads += n;
    *n_reads_out = n;
    if (n == 0 && f->writer) {                       // the source is exhausted and every batch went through: publish the cache
        pack_writer_finish(f->writer, true);
        f->writer = nullptr;
    }
    if (reads_out) {
        *reads_out = nullptr;
        if (upload && n > 0) {
            const int rc = kv_reads_create(f->seqs.data(), f->seq_offs.data(), n, reads_out);
            if (rc == KV_OK && f->writer) pack_write_batch(f->writer, f, *reads_out, n);
            return rc;
        }
    }
    if (n > 0 && f->writer) {                        // a batch that was not uploaded has no packed form: give the cache up
        pack_writer_finish(f->writer, false);
        f->writer = nullptr;
    }
    return KV_OK;
}

// sequence and quality of record i of the batch last returned, reconstructed from the packed-read cache (cache-backed
// handles only); buffers of seq_offs[i + 1] - seq_offs[i] and qual_offs[i + 1] - qual_offs[i] bytes
extern "C" int kv_fastx_record_text(kv_fastx *f, uint64_t i, char *seq_out, char *qual_out)
{
    KV_REQUIRE(f && f->cache, KV_ERR_ARG, "kv_fastx_record_text: the handle does not stream from a packed-read cache");
    std::lock_guard<std::mutex> lk(f->mu);
    const PackReader *c = f->cache;
    KV_REQUIRE(!c->batch.empty() && i < f->is_fastq.size(), KV_ERR_ARG, "kv_fastx_record_text: record %llu is not in the current batch",
               (unsigned long long)i);
    const size_t b = (size_t)(std::upper_bound(c->first.begin(), c->first.end(), i) - c->first.begin()) - 1;
    const PackBlock &blk = c->batch[b];
    const uint32_t r = (uint32_t)(i - c->first[b]);
    if (seq_out) {
        const uint32_t *w = blk.words + blk.woff[r];
        const uint32_t len = blk.len[r];
        for (uint32_t j = 0; j < len; ++j) seq_out[j] = "ACGT"[(w[j >> 4] >> (2 * (j & 15))) & 3u];
        if (blk.flags[r] & 1u) {
            const uint64_t lo = blk.boff[r], hi = lo + len;
            for (const uint64_t *e = std::lower_bound(blk.exc_pos, blk.exc_pos + blk.n_exc, lo); e < blk.exc_pos + blk.n_exc && *e < hi; ++e)
                seq_out[*e - lo] = (char)blk.exc_char[e - blk.exc_pos];
        }
    }
    if (qual_out && blk.qual_len[r]) memcpy(qual_out, blk.quals + blk.qoff[r], blk.qual_len[r]);
    return KV_OK;
}

extern "C" int kv_fastx_batch_text(kv_fastx *f, const char **names, const uint64_t **name_offs, const char **seqs,
                                   const uint64_t **seq_offs, const char **quals, const uint64_t **qual_offs,
                                   const uint8_t **is_fastq)
{
    KV_REQUIRE(f, KV_ERR_ARG, "kv_fastx_batch_text: null handle");
    if (names) *names = f->names.data();
    if (name_offs) *name_offs = f->name_offs.data();
    if (seqs) *seqs = f->seqs.data();
    if (seq_offs) *seq_offs = f->seq_offs.data();
    if (quals) *quals = f->quals.data();
    if (qual_offs) *qual_offs = f->qual_offs.data();
    if (is_fastq) *is_fastq = f->is_fastq.data();
    return KV_OK;
}

// 1 if the handle's batches are produced on the device (BGZF + FASTQ): a batch's text then stays in HBM and
// kv_fastx_batch_text describes only the records kv_fastx_fetch asked for
extern "C" int kv_fastx_on_device(kv_fastx *f, int *yes)
{
    KV_REQUIRE(f && yes, KV_ERR_ARG, "kv_fastx_on_device: null argument");
    *yes = f->dev ? 1 : 0;
    return KV_OK;
}

// text of records idx[0 .. n) of the batch last returned by a device-parsed handle, in that order: afterwards
// kv_fastx_batch_text shows exactly these n records
extern "C" int kv_fastx_fetch(kv_fastx *f, const uint64_t *idx, uint64_t n)
{
    KV_REQUIRE(f && (idx || n == 0), KV_ERR_ARG, "kv_fastx_fetch: null argument");
    KV_REQUIRE(f->dev, KV_ERR_ARG, "kv_fastx_fetch: the handle does not parse on the device");
    std::lock_guard<std::mutex> lk(f->mu);
    std::string blob;
    std::vector<uint64_t> offs;
    { const int rc = kv_fastq_device_fetch(f->dev, idx, n, &blob, &offs); if (rc != KV_OK) return rc; }
    f->names.clear(); f->seqs.clear(); f->quals.clear();
    f->name_offs.assign(1, 0); f->seq_offs.assign(1, 0); f->qual_offs.assign(1, 0);
    f->is_fastq.assign(n, 1);
    for (uint64_t i = 0; i < n; ++i) {
        const char *p = blob.data() + offs[i], *end = blob.data() + offs[i + 1];
        const char *line[5];
        line[0] = p;
        for (int l = 1; l <= 4; ++l) {
            const char *nl = (const char *)memchr(line[l - 1], '\n', (size_t)(end - line[l - 1]));
            line[l] = nl ? nl + 1 : end;
        }
        auto text_of = [&](int l, size_t skip) {
            size_t len = (size_t)(line[l + 1] - line[l]);
            if (len && line[l][len - 1] == '\n') --len;
            if (len && line[l][len - 1] == '\r') --len;
            return std::string(line[l] + std::min(skip, len), len - std::min(skip, len));
        };
        f->names += text_of(0, 1);
        f->seqs += text_of(1, 0);
        f->quals += text_of(3, 0);
        f->name_offs.push_back(f->names.size());
        f->seq_offs.push_back(f->seqs.size());
        f->qual_offs.push_back(f->quals.size());
    }
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// Augmented FASTA/FASTQ text (kevlar/sequence.py print_augmented_fastx): the record, then one line per interesting
// k-mer -- `offset` blanks, the k-mer, ten blanks, the abundances separated by blanks, '#'.  Formatting a few
// hundred thousand annotated reads record by record in Python costs more than the whole GPU pipeline, so the writer
// gets the text of a batch in one piece.  hits: (read, offset, abund[S]) sorted by (read, offset); the j-th distinct
// read of the hits is record rec_index[j] of the blobs.
// ---------------------------------------------------------------------------------------
extern "C" int kv_format_augmented(const uint32_t *hit_read, const uint32_t *hit_off, const uint8_t *abund, uint64_t n_hits, int nsamples,
                                   int ksize, const uint64_t *rec_index, const char *names, const uint64_t *name_offs, const char *seqs,
                                   const uint64_t *seq_offs, const char *quals, const uint64_t *qual_offs, const uint8_t *is_fastq,
                                   char **text_out, uint64_t *bytes_out, uint64_t *n_records_out)
{
    KV_REQUIRE(text_out && bytes_out && (n_hits == 0 || (hit_read && hit_off && abund && rec_index && names && name_offs && seqs && seq_offs)),
               KV_ERR_ARG, "kv_format_augmented: null argument");
    KV_REQUIRE(nsamples >= 1 && ksize >= 1, KV_ERR_ARG, "kv_format_augmented: bad argument");
    KvTextOut out;
    out.room((size_t)n_hits * (size_t)(ksize + 48) + 4096);
    uint64_t j = 0;         // distinct reads so far
    for (uint64_t i = 0; i < n_hits;) {
        const uint64_t r = rec_index[j++];
        const char *seq = seqs + seq_offs[r];
        const size_t seq_len = (size_t)(seq_offs[r + 1] - seq_offs[r]);
        const bool fq = is_fastq ? is_fastq[r] != 0 : false;
        out.put(fq ? '@' : '>');
        out.put(names + name_offs[r], (size_t)(name_offs[r + 1] - name_offs[r]));
        out.put('\n');
        out.put(seq, seq_len);
        if (fq) {
            out.put("\n+\n", 3);
            out.put(quals + qual_offs[r], (size_t)(qual_offs[r + 1] - qual_offs[r]));
        }
        out.put('\n');
        const uint32_t read = hit_read[i];
        for (; i < n_hits && hit_read[i] == read; ++i) {
            const uint32_t off = hit_off[i];
            KV_REQUIRE((size_t)off + (size_t)ksize <= seq_len, KV_ERR_ARG, "kv_format_augmented: a hit at offset %u does not fit its read of %llu bases",
                       off, (unsigned long long)seq_len);
            const uint8_t *row = abund + i * (uint64_t)nsamples;
            out.kmer_line(seq, off, ksize, nsamples, [&](int c) { return (int64_t)row[c]; });
        }
    }
    KV_REQUIRE(out.ok, KV_ERR_HIP, "kv_format_augmented: out of memory");
    *bytes_out = out.len;
    *text_out = out.release();
    KV_REQUIRE(*text_out, KV_ERR_HIP, "kv_format_augmented: out of memory");
    if (n_records_out) *n_records_out = j;
    return KV_OK;
}

extern "C" int kv_text_free(char *text)
{
    free(text);
    return KV_OK;
}
