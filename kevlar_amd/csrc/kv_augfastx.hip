// kv_augfastx.hip -- augmented FASTA/FASTQ files as flat arrays (host code; kevlar/sequence.pyx parse_augmented_fastx).
//
// `kevlar filter` and `kevlar partition` both start by reading what `kevlar novel` wrote: records followed by one
// line per interesting k-mer (`offset` blanks, the k-mer, blanks, the abundances, '#') and optional `#mateseq=SEQ#`
// lines.  At config 2 that is 120 k reads with 2.3 M annotation lines; the reference builds a Python object per record
// and per annotation (seconds, where the GPU work behind it takes milliseconds).  This parser reads the file once into
// the same blobs + offsets a kv_fastx batch exposes, plus per-annotation arrays (offset, abundances) and the
// annotations' extent per record, so the drivers can work on whole arrays and hand positions -- not k-mer strings --
// to the device (kv_hash_positions, kv_readgraph_components).
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kv_internal.h"

struct kv_augfastx {
    std::string names, seqs, quals, mates;
    std::vector<uint64_t> name_offs{0}, seq_offs{0}, qual_offs{0}, mate_offs{0};
    std::vector<uint8_t> is_fastq;
    std::vector<uint64_t> ann_first{0};        // annotations of record i: ann_first[i] .. ann_first[i + 1]
    std::vector<uint32_t> ann_offset;
    std::vector<int32_t> ann_abund;            // n_ann * nsamples
    std::vector<uint32_t> mate_record;         // record each mate sequence belongs to
    int ksize = 0, nsamples = 0;
};

namespace {

inline const char *trim(const char *p, const char *&end)
{
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    while (end > p && (end[-1] == ' ' || end[-1] == '\t' || end[-1] == '\r' || end[-1] == '\n')) --end;
    return p;
}

}  // namespace

extern "C" int kv_augfastx_load(const char *path, kv_augfastx **out)
{
    KV_REQUIRE(path && out, KV_ERR_ARG, "kv_augfastx_load: null argument");
    gzFile fh = gzopen(path, "rb");
    if (!fh) { kv_set_error("cannot open %s", path); return KV_ERR_IO; }
    gzbuffer(fh, 1 << 20);
    std::string text;
    {
        std::vector<char> buf(8 << 20);
        for (;;) {
            const int got = gzread(fh, buf.data(), (unsigned)buf.size());
            if (got < 0) { gzclose(fh); kv_set_error("cannot read %s", path); return KV_ERR_IO; }
            if (got == 0) break;
            text.append(buf.data(), (size_t)got);
        }
        gzclose(fh);
    }
    kv_augfastx *a = new kv_augfastx();
    const char *p = text.data(), *const end = text.data() + text.size();
    auto next_line = [&](const char *&lo, const char *&hi) -> bool {      // [lo, hi) without the newline; hi_nl = had one
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        lo = p;
        hi = nl ? nl : end;
        p = nl ? nl + 1 : end;
        return true;
    };
    bool have_record = false;
    int rc = KV_OK;
    const char *lo, *hi;
    while (rc == KV_OK && next_line(lo, hi)) {
        const char *t_hi = hi;
        const char *t_lo = trim(lo, t_hi);
        if (t_lo == t_hi) continue;                          // blank line
        const char first = lo[0];
        if (first == '@' || first == '>') {
            const char *n_hi = hi;
            const char *n_lo = trim(lo + 1, n_hi);
            a->names.append(n_lo, (size_t)(n_hi - n_lo));
            a->name_offs.push_back(a->names.size());
            const char *s_lo = nullptr, *s_hi = nullptr;
            if (next_line(s_lo, s_hi)) { s_lo = trim(s_lo, s_hi); a->seqs.append(s_lo, (size_t)(s_hi - s_lo)); }
            a->seq_offs.push_back(a->seqs.size());
            if (first == '@') {
                const char *q_lo, *q_hi;
                (void)next_line(q_lo, q_hi);                 // '+'
                if (next_line(q_lo, q_hi)) { q_lo = trim(q_lo, q_hi); a->quals.append(q_lo, (size_t)(q_hi - q_lo)); }
            }
            a->qual_offs.push_back(a->quals.size());
            a->is_fastq.push_back(first == '@' ? 1 : 0);
            a->ann_first.push_back(a->ann_offset.size());
            have_record = true;
            continue;
        }
        // an annotation or mate line ends in '#' + newline
        if (!(hi > lo && hi[-1] == '#' && hi < end) || !have_record) {
            kv_set_error("%s: unexpected line in an augmented FASTA/FASTQ stream: %.60s", path, std::string(lo, (size_t)(hi - lo)).c_str());
            rc = KV_ERR_IO;
            break;
        }
        const uint64_t rec = a->is_fastq.size() - 1;
        if ((size_t)(hi - lo) > 9 && memcmp(lo, "#mateseq=", 9) == 0) {
            a->mates.append(lo + 9, (size_t)(hi - 1 - (lo + 9)));
            a->mate_offs.push_back(a->mates.size());
            a->mate_record.push_back((uint32_t)rec);
            continue;
        }
        const char *q = lo;
        while (q < hi && (*q == ' ' || *q == '\t')) ++q;
        const uint32_t offset = (uint32_t)(q - lo);
        const char *kmer = q;
        while (q < hi - 1 && *q != ' ' && *q != '\t') ++q;
        const int k = (int)(q - kmer);
        if (a->ksize == 0) a->ksize = k;
        const uint64_t s0 = a->seq_offs[rec], slen = a->seq_offs[rec + 1] - s0;
        if (k <= 0 || (uint64_t)offset + (uint64_t)k > slen || memcmp(a->seqs.data() + s0 + offset, kmer, (size_t)k) != 0) {
            kv_set_error("%s: the k-mer of an annotation does not match its read at offset %u (record %llu)", path, offset, (unsigned long long)rec);
            rc = KV_ERR_IO;
            break;
        }
        if (k != a->ksize) { a->ksize = -1; }               // mixed k: the caller decides (filter / partition reject it)
        int count = 0;
        while (q < hi - 1) {
            while (q < hi - 1 && (*q == ' ' || *q == '\t')) ++q;
            if (q >= hi - 1) break;
            int32_t v = 0;
            bool digits = false;
            while (q < hi - 1 && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); ++q; digits = true; }
            if (!digits) { kv_set_error("%s: bad abundance in an annotation of record %llu", path, (unsigned long long)rec); rc = KV_ERR_IO; break; }
            a->ann_abund.push_back(v);
            ++count;
        }
        if (rc != KV_OK) break;
        if (a->nsamples == 0 && a->ann_offset.empty()) a->nsamples = count;
        if (count != a->nsamples) {
            kv_set_error("%s: annotations with %d and %d abundances in one stream", path, a->nsamples, count);
            rc = KV_ERR_IO;
            break;
        }
        a->ann_offset.push_back(offset);
        a->ann_first.back() = a->ann_offset.size();
    }
    if (rc != KV_OK) { delete a; return rc; }
    *out = a;
    return KV_OK;
}

extern "C" int kv_augfastx_info(const kv_augfastx *a, uint64_t *n_records, uint64_t *n_annotations, int *ksize, int *nsamples, uint64_t *n_mates)
{
    KV_REQUIRE(a, KV_ERR_ARG, "kv_augfastx_info: null handle");
    if (n_records) *n_records = a->is_fastq.size();
    if (n_annotations) *n_annotations = a->ann_offset.size();
    if (ksize) *ksize = a->ksize;
    if (nsamples) *nsamples = a->nsamples;
    if (n_mates) *n_mates = a->mate_record.size();
    return KV_OK;
}

extern "C" int kv_augfastx_view(const kv_augfastx *a, const char **names, const uint64_t **name_offs, const char **seqs, const uint64_t **seq_offs,
                                const char **quals, const uint64_t **qual_offs, const uint8_t **is_fastq, const uint64_t **ann_first,
                                const uint32_t **ann_offset, const int32_t **ann_abund, const uint32_t **mate_record, const char **mates,
                                const uint64_t **mate_offs)
{
    KV_REQUIRE(a, KV_ERR_ARG, "kv_augfastx_view: null handle");
    if (names) *names = a->names.data();
    if (name_offs) *name_offs = a->name_offs.data();
    if (seqs) *seqs = a->seqs.data();
    if (seq_offs) *seq_offs = a->seq_offs.data();
    if (quals) *quals = a->quals.data();
    if (qual_offs) *qual_offs = a->qual_offs.data();
    if (is_fastq) *is_fastq = a->is_fastq.data();
    if (ann_first) *ann_first = a->ann_first.data();
    if (ann_offset) *ann_offset = a->ann_offset.data();
    if (ann_abund) *ann_abund = a->ann_abund.data();
    if (mate_record) *mate_record = a->mate_record.data();
    if (mates) *mates = a->mates.data();
    if (mate_offs) *mate_offs = a->mate_offs.data();
    return KV_OK;
}

extern "C" int kv_augfastx_free(kv_augfastx *a)
{
    delete a;
    return KV_OK;
}

// ---------------------------------------------------------------------------------------
// The writer's side for streams held as arrays: output record j is record rec_index[j] of the blobs with the
// annotations ann_lo[j] <= i < ann_hi[j] of the annotation arrays for which keep[i] != 0 (keep NULL: all), in offset
// order (stable, as sorted(record.annotations, key=offset) in kevlar/sequence.pyx), then its mate lines.  case_abund
// (optional) replaces the first abundance of every annotation; suffix (optional blob + offsets per output record) is
// appended to the record's name (partition's " kvcc=N").
// ---------------------------------------------------------------------------------------
extern "C" int kv_format_records(uint64_t n_out, const uint64_t *rec_index, const uint64_t *ann_lo, const uint64_t *ann_hi,
                                 const uint32_t *ann_offset, const int32_t *ann_abund, const uint8_t *keep, const int32_t *case_abund,
                                 int nsamples, int ksize, const char *names, const uint64_t *name_offs, const char *seqs,
                                 const uint64_t *seq_offs, const char *quals, const uint64_t *qual_offs, const uint8_t *is_fastq,
                                 const char *suffix, const uint64_t *suffix_offs, const uint32_t *mate_record, uint64_t n_mates,
                                 const char *mates, const uint64_t *mate_offs, char **text_out, uint64_t *bytes_out)
{
    KV_REQUIRE(text_out && bytes_out && (n_out == 0 || (rec_index && ann_lo && ann_hi && names && name_offs && seqs && seq_offs)), KV_ERR_ARG,
               "kv_format_records: null argument");
    KvTextOut out;
    {
        uint64_t notes = 0;
        for (uint64_t j = 0; j < n_out; ++j) notes += ann_hi[j] - ann_lo[j];
        out.room((size_t)notes * (size_t)(ksize + 48) + (size_t)n_out * 320 + 4096);
    }
    std::vector<uint64_t> order;
    for (uint64_t j = 0; j < n_out; ++j) {
        const uint64_t r = rec_index[j];
        const char *seq = seqs + seq_offs[r];
        const size_t seq_len = (size_t)(seq_offs[r + 1] - seq_offs[r]);
        const bool fq = is_fastq ? is_fastq[r] != 0 : false;
        out.put(fq ? '@' : '>');
        out.put(names + name_offs[r], (size_t)(name_offs[r + 1] - name_offs[r]));
        if (suffix && suffix_offs) out.put(suffix + suffix_offs[j], (size_t)(suffix_offs[j + 1] - suffix_offs[j]));
        out.put('\n');
        out.put(seq, seq_len);
        if (fq) {
            out.put("\n+\n", 3);
            out.put(quals + qual_offs[r], (size_t)(qual_offs[r + 1] - qual_offs[r]));
        }
        out.put('\n');
        // the kept annotations in offset order (stable); nearly always they already are
        bool sorted = true;
        uint32_t last = 0;
        for (uint64_t i = ann_lo[j]; i < ann_hi[j] && sorted; ++i)
            if (!keep || keep[i]) { sorted = ann_offset[i] >= last; last = ann_offset[i]; }
        auto line = [&](uint64_t i) {
            const uint32_t off = ann_offset[i];
            if ((size_t)off + (size_t)ksize > seq_len) { out.ok = false; return; }
            const int32_t *row = ann_abund + i * (uint64_t)nsamples;
            out.kmer_line(seq, off, ksize, nsamples, [&](int c) { return (int64_t)((c == 0 && case_abund) ? case_abund[i] : row[c]); });
        };
        if (sorted) {
            for (uint64_t i = ann_lo[j]; i < ann_hi[j]; ++i)
                if (!keep || keep[i]) line(i);
        } else {
            order.clear();
            for (uint64_t i = ann_lo[j]; i < ann_hi[j]; ++i)
                if (!keep || keep[i]) order.push_back(i);
            std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return ann_offset[x] < ann_offset[y]; });
            for (const uint64_t i : order) line(i);
        }
        KV_REQUIRE(out.ok, KV_ERR_ARG, "kv_format_records: an annotation does not fit its read (record %llu), or out of memory", (unsigned long long)r);
        if (n_mates) {
            const uint32_t *m = std::lower_bound(mate_record, mate_record + n_mates, (uint32_t)r);
            for (; m < mate_record + n_mates && *m == (uint32_t)r; ++m) {
                const uint64_t mi = (uint64_t)(m - mate_record);
                out.put("#mateseq=", 9);
                out.put(mates + mate_offs[mi], (size_t)(mate_offs[mi + 1] - mate_offs[mi]));
                out.put("#\n", 2);
            }
        }
    }
    KV_REQUIRE(out.ok, KV_ERR_HIP, "kv_format_records: out of memory");
    *bytes_out = out.len;
    *text_out = out.release();
    KV_REQUIRE(*text_out, KV_ERR_HIP, "kv_format_records: out of memory");
    return KV_OK;
}
