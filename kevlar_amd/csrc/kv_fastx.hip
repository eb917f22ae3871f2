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
#include <zlib.h>

#include <algorithm>

#include "kv_internal.h"

struct kv_fastx {
    gzFile fh = nullptr;
    std::string path;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    std::string pending;          // a header line read ahead while scanning a FASTA record
    std::string pending_view;     // backing store of the view handed out for it
    bool have_pending = false;
    uint64_t num_reads = 0;
    // text of the last batch
    std::string names, seqs, quals;
    std::vector<uint64_t> name_offs, seq_offs, qual_offs;
    std::vector<uint8_t> is_fastq;   // per record: came with a quality line
    std::mutex mu;
};

static bool fx_fill(kv_fastx *f)
{
    if (f->eof) return false;
    if (f->pos > 0 && f->pos < f->end) memmove(f->buf.data(), f->buf.data() + f->pos, f->end - f->pos);
    f->end -= f->pos;
    f->pos = 0;
    if (f->end == f->buf.size()) f->buf.resize(f->buf.size() * 2);
    const int got = gzread(f->fh, f->buf.data() + f->end, (unsigned)(f->buf.size() - f->end));
    if (got <= 0) { f->eof = true; return false; }
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

extern "C" int kv_fastx_open(const char *path, kv_fastx **out)
{
    KV_REQUIRE(path && out, KV_ERR_ARG, "kv_fastx_open: null argument");
    gzFile fh = gzopen(path, "rb");
    KV_REQUIRE(fh, KV_ERR_IO, "cannot open sequence file %s", path);
    gzbuffer(fh, 1 << 20);
    kv_fastx *f = new kv_fastx();
    f->fh = fh;
    f->path = path;
    f->buf.resize(4 << 20);
    *out = f;
    return KV_OK;
}

extern "C" int kv_fastx_close(kv_fastx *f)
{
    if (!f) return KV_OK;
    if (f->fh) gzclose(f->fh);
    delete f;
    return KV_OK;
}

extern "C" int kv_fastx_num_reads(kv_fastx *f, uint64_t *n)
{
    KV_REQUIRE(f && n, KV_ERR_ARG, "kv_fastx_num_reads: null argument");
    *n = f->num_reads;
    return KV_OK;
}

extern "C" int kv_fastx_next(kv_fastx *f, uint64_t max_reads, int upload, kv_reads **reads_out, uint64_t *n_reads_out)
{
    KV_REQUIRE(f && n_reads_out && max_reads > 0, KV_ERR_ARG, "kv_fastx_next: bad argument");
    std::lock_guard<std::mutex> lk(f->mu);
    f->names.clear(); f->seqs.clear(); f->quals.clear();
    f->name_offs.assign(1, 0); f->seq_offs.assign(1, 0); f->qual_offs.assign(1, 0);
    f->is_fastq.clear();
    std::string seq;
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
    f->num_reads += n;
    *n_reads_out = n;
    if (reads_out) {
        *reads_out = nullptr;
        if (upload && n > 0) return kv_reads_create(f->seqs.data(), f->seq_offs.data(), n, reads_out);
    }
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
