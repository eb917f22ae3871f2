// kv_augfastx.hip -- augmented FASTA/FASTQ files as flat arrays (host code; kevlar/sequence.pyx parse_augmented_fastx).
//
// `kevlar filter` and `kevlar partition` both start by reading what `kevlar novel` wrote: records followed by one
// line per interesting k-mer (`offset` blanks, the k-mer, blanks, the abundances, '#') and optional `#mateseq=SEQ#`
// lines.  At config 2 that is 120 k reads with 2.3 M annotation lines; the reference builds a Python object per record
// and per annotation (seconds, where the GPU work behind it takes milliseconds).  This parser reads the file once into
// the same blobs + offsets a kv_fastx batch exposes, plus per-annotation arrays (offset, abundances) and the
// annotations' extent per record, so the drivers can work on whole arrays and hand positions -- not k-mer strings --
// to the device (kv_hash_positions, kv_readgraph_components).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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

// records and annotations of the lines in [begin, end) -> a; begin is the start of a record (or of the file)
int parse_range(const char *begin, const char *end, const char *path, kv_augfastx *a, std::string *err)
{
    const char *p = begin;
    auto fail = [&](const char *fmt, auto... args) {
        char buf[512];
        snprintf(buf, sizeof(buf), fmt, args...);
        *err = buf;
        return KV_ERR_IO;
    };
    bool had_nl = false;
    auto next_line = [&](const char *&lo, const char *&hi) -> bool {      // [lo, hi) without the newline
        if (p >= end) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        lo = p;
        hi = nl ? nl : end;
        p = nl ? nl + 1 : end;
        had_nl = nl != nullptr;
        return true;
    };
    bool have_record = false;
    const char *lo, *hi;
    while (next_line(lo, hi)) {
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
        if (!(hi > lo && hi[-1] == '#' && had_nl) || !have_record)
            return fail("%s: unexpected line in an augmented FASTA/FASTQ stream: %.60s", path, std::string(lo, (size_t)(hi - lo)).c_str());
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
        if (k <= 0 || (uint64_t)offset + (uint64_t)k > slen || memcmp(a->seqs.data() + s0 + offset, kmer, (size_t)k) != 0)
            return fail("%s: the k-mer of an annotation does not match its read at offset %u (record %llu)", path, offset, (unsigned long long)rec);
        if (k != a->ksize) { a->ksize = -1; }               // mixed k: the caller decides (filter / partition reject it)
        int count = 0;
        while (q < hi - 1) {
            while (q < hi - 1 && (*q == ' ' || *q == '\t')) ++q;
            if (q >= hi - 1) break;
            int32_t v = 0;
            bool digits = false;
            while (q < hi - 1 && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); ++q; digits = true; }
            if (!digits) return fail("%s: bad abundance in an annotation of record %llu", path, (unsigned long long)rec);
            a->ann_abund.push_back(v);
            ++count;
        }
        if (a->nsamples == 0 && a->ann_offset.empty()) a->nsamples = count;
        if (count != a->nsamples) return fail("%s: annotations with %d and %d abundances in one stream", path, a->nsamples, count);
        a->ann_offset.push_back(offset);
        a->ann_first.back() = a->ann_offset.size();
    }
    return KV_OK;
}

// `pieces` + 1 places to cut [begin, end) at, each the start of a record: the first, the end, and in between the first
// record start behind every i / pieces of the way.  A record starts with a line that begins with '>' in a FASTA stream; in a
// FASTQ stream with one that begins with '@' AND has a '+' line two lines on (a quality line may begin with '@', but what
// follows a quality line two lines on is an annotation, a header or a sequence).  Fewer places come back if no start is found
// within 4 MB of where one is wanted (the caller then parses in one piece).
std::vector<const char *> record_starts(const char *begin, const char *end, size_t pieces)
{
    std::vector<const char *> cuts(1, begin);
    const char *p = begin;
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    if (p >= end || (*p != '@' && *p != '>')) return cuts;
    const bool fastq = *p == '@';
    auto line_after = [&](const char *q) -> const char * {
        const char *nl = (const char *)memchr(q, '\n', (size_t)(end - q));
        return nl ? nl + 1 : end;
    };
    for (size_t i = 1; i < pieces; ++i) {
        const char *q = begin + (size_t)((double)(end - begin) * (double)i / (double)pieces);
        if (q <= cuts.back()) continue;
        q = line_after(q);                              // the start of a line
        const char *limit = std::min(end, q + (4u << 20));
        const char *found = nullptr;
        while (q < limit) {
            if (!fastq && *q == '>') { found = q; break; }
            if (fastq && *q == '@') {
                const char *two_on = line_after(line_after(q));
                if (two_on < end && *two_on == '+') { found = q; break; }
            }
            q = line_after(q);
        }
        if (!found) return std::vector<const char *>(1, begin);
        if (found > cuts.back()) cuts.push_back(found);
    }
    cuts.push_back(end);
    return cuts;
}

// the pieces one behind the other; false if they cannot be one stream (different numbers of abundance columns)
bool join_parts(std::vector<kv_augfastx> &part, kv_augfastx *a)
{
    size_t names = 0, seqs = 0, quals = 0, mates = 0, recs = 0, anns = 0, abund = 0, n_mates = 0;
    for (const kv_augfastx &b : part) {
        names += b.names.size(); seqs += b.seqs.size(); quals += b.quals.size(); mates += b.mates.size();
        recs += b.is_fastq.size(); anns += b.ann_offset.size(); abund += b.ann_abund.size(); n_mates += b.mate_record.size();
        if (b.ann_offset.empty()) continue;
        if (a->nsamples == 0) a->nsamples = b.nsamples;
        else if (b.nsamples != a->nsamples) return false;
        if (a->ksize == 0) a->ksize = b.ksize;
        else if (b.ksize != a->ksize) a->ksize = -1;
    }
    // Every piece knows where it goes, so the pieces are copied side by side (one thread doing it all -- 2 GB of a 3 GB file --
    // was a third of the load); the big blobs are sized (= zeroed, first touch) side by side too.
    const size_t np = part.size();
    std::vector<uint64_t> base_n(np), base_s(np), base_q(np), base_m(np), base_r(np), base_a(np), base_ab(np), base_nm(np);
    {
        uint64_t n_ = 0, s_ = 0, q_ = 0, m_ = 0, r_ = 0, a_ = 0, ab_ = 0, nm_ = 0;
        for (size_t i = 0; i < np; ++i) {
            base_n[i] = n_; base_s[i] = s_; base_q[i] = q_; base_m[i] = m_; base_r[i] = r_; base_a[i] = a_; base_ab[i] = ab_; base_nm[i] = nm_;
            n_ += part[i].names.size(); s_ += part[i].seqs.size(); q_ += part[i].quals.size(); m_ += part[i].mates.size();
            r_ += part[i].is_fastq.size(); a_ += part[i].ann_offset.size(); ab_ += part[i].ann_abund.size(); nm_ += part[i].mate_record.size();
        }
    }
    {
        std::thread sized_seqs([&] { a->seqs.resize(seqs); });
        std::thread sized_quals([&] { a->quals.resize(quals); });
        a->names.resize(names); a->mates.resize(mates);
        a->name_offs.assign(recs + 1, 0); a->seq_offs.assign(recs + 1, 0); a->qual_offs.assign(recs + 1, 0); a->ann_first.assign(recs + 1, 0);
        a->is_fastq.resize(recs); a->ann_offset.resize(anns); a->ann_abund.resize(abund);
        a->mate_record.resize(n_mates); a->mate_offs.assign(n_mates + 1, 0);
        sized_seqs.join(); sized_quals.join();
    }
    auto place = [&](size_t i) {
        kv_augfastx &b = part[i];
        if (!b.names.empty()) memcpy(&a->names[base_n[i]], b.names.data(), b.names.size());
        if (!b.seqs.empty()) memcpy(&a->seqs[base_s[i]], b.seqs.data(), b.seqs.size());
        if (!b.quals.empty()) memcpy(&a->quals[base_q[i]], b.quals.data(), b.quals.size());
        if (!b.mates.empty()) memcpy(&a->mates[base_m[i]], b.mates.data(), b.mates.size());
        for (size_t r = 1; r < b.name_offs.size(); ++r) {
            a->name_offs[base_r[i] + r] = base_n[i] + b.name_offs[r];
            a->seq_offs[base_r[i] + r] = base_s[i] + b.seq_offs[r];
            a->qual_offs[base_r[i] + r] = base_q[i] + b.qual_offs[r];
            a->ann_first[base_r[i] + r] = base_a[i] + b.ann_first[r];
        }
        if (!b.is_fastq.empty()) memcpy(&a->is_fastq[base_r[i]], b.is_fastq.data(), b.is_fastq.size() * sizeof(b.is_fastq[0]));
        if (!b.ann_offset.empty()) memcpy(&a->ann_offset[base_a[i]], b.ann_offset.data(), b.ann_offset.size() * sizeof(b.ann_offset[0]));
        if (!b.ann_abund.empty()) memcpy(&a->ann_abund[base_ab[i]], b.ann_abund.data(), b.ann_abund.size() * sizeof(b.ann_abund[0]));
        for (size_t m = 1; m < b.mate_offs.size(); ++m) a->mate_offs[base_nm[i] + m] = base_m[i] + b.mate_offs[m];
        for (size_t m = 0; m < b.mate_record.size(); ++m) a->mate_record[base_nm[i] + m] = (uint32_t)(base_r[i] + b.mate_record[m]);
        b = kv_augfastx();                               // let the piece go
    };
    std::vector<std::thread> crew;
    for (size_t i = 1; i < np; ++i) crew.emplace_back(place, i);
    if (np) place(0);
    for (std::thread &t : crew) t.join();
    return true;
}

}  // namespace

extern "C" int kv_augfastx_load(const char *path, kv_augfastx **out)
{
    KV_REQUIRE(path && out, KV_ERR_ARG, "kv_augfastx_load: null argument");
    // an uncompressed file is parsed where the page cache has it (going through zlib's transparent read copied it twice:
    // half of the load time); a compressed one is inflated into a buffer first
    std::string text;
    const char *image = nullptr;
    size_t image_size = 0;
    int fd = open(path, O_RDONLY);
    if (fd < 0) { kv_set_error("cannot open %s", path); return KV_ERR_IO; }
    {
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0;
        const bool gz = pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (regular && !gz) {
            void *map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map != MAP_FAILED) { image = (const char *)map; image_size = (size_t)sb.st_size; }
        }
    }
    struct Unmap {
        const char *p; size_t n; int fd;
        ~Unmap() { if (p) munmap((void *)p, n); if (fd >= 0) close(fd); }
    } unmap{image, image_size, fd};
    if (!image) {
        gzFile fh = gzopen(path, "rb");
        if (!fh) { kv_set_error("cannot open %s", path); return KV_ERR_IO; }
        gzbuffer(fh, 1 << 20);
        std::vector<char> buf(8 << 20);
        for (;;) {
            const int got = gzread(fh, buf.data(), (unsigned)buf.size());
            if (got < 0) { gzclose(fh); kv_set_error("cannot read %s", path); return KV_ERR_IO; }
            if (got == 0) break;
            text.append(buf.data(), (size_t)got);
        }
        gzclose(fh);
        image = text.data();
        image_size = text.size();
    }
    // ---- big files are cut at record starts and the pieces parsed side by side
    kv_augfastx *a = nullptr;
    const char *forced = kv_knob("KV_AUGFASTX_THREADS");             // 1: one pass, for comparison
    const unsigned hw = forced ? (unsigned)std::max(1, atoi(forced)) : std::max(1u, std::thread::hardware_concurrency());
    const size_t pieces = std::min<size_t>(std::min<size_t>(hw, 16), image_size / (4u << 20));
    if (pieces >= 2) {
        const std::vector<const char *> cuts = record_starts(image, image + image_size, pieces);
        if (cuts.size() >= 3) {
            const size_t n = cuts.size() - 1;
            std::vector<kv_augfastx> part(n);
            std::vector<int> rcs(n, KV_OK);
            std::vector<std::string> errs(n);
            std::vector<std::thread> crew;
            for (size_t i = 0; i < n; ++i)
                crew.emplace_back([&, i] { rcs[i] = parse_range(cuts[i], cuts[i + 1], path, &part[i], &errs[i]); });
            for (std::thread &t : crew) t.join();
            bool ok = true;
            for (size_t i = 0; i < n; ++i) ok = ok && rcs[i] == KV_OK;
            if (ok) {
                a = new kv_augfastx();
                if (!join_parts(part, a)) { delete a; a = nullptr; }        // (pieces that disagree on the columns: the plain pass below says so)
            }
        }
    }
    if (!a) {                                          // one pass over the whole file (small files; and whatever the pieces stumbled over,
        a = new kv_augfastx();                         // for the message with the record's true number)
        std::string err;
        const int rc = parse_range(image, image + image_size, path, a, &err);
        if (rc != KV_OK) { delete a; kv_set_error("%s", err.c_str()); return rc; }
    }
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
// flags[i] = 1 if read i holds a byte other than A, C, G, T (upper case): such reads cannot be 2-bit packed, their k-mers are
// hashed from the text (AnnotatedReads.hashes).  Host only, threads over the reads.
extern "C" int kv_reads_flag_other_bytes(const char *seqs, const uint64_t *seq_offs, uint64_t n, uint8_t *flags)
{
    KV_REQUIRE(n == 0 || (seqs && seq_offs && flags), KV_ERR_ARG, "kv_reads_flag_other_bytes: null argument");
    const char *forced = kv_knob("KV_AUGFASTX_THREADS");
    const unsigned hw = forced ? (unsigned)std::max(1, atoi(forced)) : std::max(1u, std::thread::hardware_concurrency());
    const uint64_t crew_n = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw, 32), n / 20000));
    auto work = [&](uint64_t lo, uint64_t hi) {
        bool other[256];
        for (int b = 0; b < 256; ++b) other[b] = !(b == 'A' || b == 'C' || b == 'G' || b == 'T');
        for (uint64_t r = lo; r < hi; ++r) {
            const unsigned char *p = (const unsigned char *)seqs + seq_offs[r], *end = (const unsigned char *)seqs + seq_offs[r + 1];
            unsigned any = 0;
            for (; p < end; ++p) any |= (unsigned)other[*p];
            flags[r] = (uint8_t)any;
        }
    };
    if (crew_n <= 1) { work(0, n); return KV_OK; }
    std::vector<std::thread> crew;
    for (uint64_t t = 0; t < crew_n; ++t) crew.emplace_back(work, n * t / crew_n, n * (t + 1) / crew_n);
    for (std::thread &t : crew) t.join();
    return KV_OK;
}

// Two independent 64-bit hashes of min(sequence, reverse complement) for the given reads: the key `kevlar partition` dedups a
// partition's reads by (kevlar/partition.py:37-47 via kevlar.revcommin).  complement: the 256-entry byte table of the caller's
// revcom() (IUPAC codes, both cases); order is that of the bytes, as Python compares the strings.  Host only, threads over the reads.
extern "C" int kv_canonical_read_hashes(const char *seqs, const uint64_t *seq_offs, const uint64_t *reads, uint64_t n,
                                        const uint8_t *complement, uint64_t *h1, uint64_t *h2)
{
    KV_REQUIRE(n == 0 || (seqs && seq_offs && reads && complement && h1 && h2), KV_ERR_ARG, "kv_canonical_read_hashes: null argument");
    const char *forced = kv_knob("KV_AUGFASTX_THREADS");
    const unsigned hw = forced ? (unsigned)std::max(1, atoi(forced)) : std::max(1u, std::thread::hardware_concurrency());
    const uint64_t crew_n = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(hw, 32), n / 20000));
    auto work = [&](uint64_t lo, uint64_t hi) {
        std::string rc;
        for (uint64_t j = lo; j < hi; ++j) {
            const uint64_t r = reads[j];
            const unsigned char *fw = (const unsigned char *)seqs + seq_offs[r];
            const size_t len = (size_t)(seq_offs[r + 1] - seq_offs[r]);
            rc.resize(len);
            for (size_t i = 0; i < len; ++i) rc[i] = (char)complement[fw[len - 1 - i]];
            const unsigned char *use = len && memcmp(rc.data(), fw, len) < 0 ? (const unsigned char *)rc.data() : fw;
            uint64_t a = 0xcbf29ce484222325ull ^ (uint64_t)len, c = 0x9e3779b97f4a7c15ull + (uint64_t)len;
            for (size_t i = 0; i < len; i += 8) {
                uint64_t w = 0;
                memcpy(&w, use + i, std::min<size_t>(8, len - i));
                a = (a ^ w) * 0x100000001b3ull; a ^= a >> 31;
                c = (c + w) * 0xff51afd7ed558ccdull; c ^= c >> 29;
            }
            h1[j] = a; h2[j] = c;
        }
    };
    if (crew_n <= 1) { work(0, n); return KV_OK; }
    std::vector<std::thread> crew;
    for (uint64_t t = 0; t < crew_n; ++t) crew.emplace_back(work, n * t / crew_n, n * (t + 1) / crew_n);
    for (std::thread &t : crew) t.join();
    return KV_OK;
}

// same[j] = 1 if reads a[j] and b[j] have the same canonical sequence -- min(sequence, reverse complement) by byte order, the string
// the reference compares (kevlar/partition.py:26-33 via kevlar.revcommin).  The canonical forms themselves are built and compared:
// revcom() is not an involution (it upper-cases what it complements), so "one is the other or the other's reverse complement"
// would be a different relation on mixed-case reads.  partition calls this for the pairs whose two hashes agree, so that a
// collision never drops a read.  Host only.
extern "C" int kv_canonical_reads_equal(const char *seqs, const uint64_t *seq_offs, const uint64_t *a, const uint64_t *b, uint64_t n,
                                        const uint8_t *complement, uint8_t *same)
{
    KV_REQUIRE(n == 0 || (seqs && seq_offs && a && b && complement && same), KV_ERR_ARG, "kv_canonical_reads_equal: null argument");
    std::string ra, rb;
    auto canonical = [&](uint64_t r, std::string &rc, size_t &len) -> const unsigned char * {
        const unsigned char *fw = (const unsigned char *)seqs + seq_offs[r];
        len = (size_t)(seq_offs[r + 1] - seq_offs[r]);
        rc.resize(len);
        for (size_t i = 0; i < len; ++i) rc[i] = (char)complement[fw[len - 1 - i]];
        return len && memcmp(rc.data(), fw, len) < 0 ? (const unsigned char *)rc.data() : fw;      // the rule of kv_canonical_read_hashes
    };
    for (uint64_t j = 0; j < n; ++j) {
        size_t la = 0, lb = 0;
        const unsigned char *ca = canonical(a[j], ra, la), *cb = canonical(b[j], rb, lb);
        same[j] = la == lb && (la == 0 || memcmp(ca, cb, la) == 0) ? 1 : 0;
    }
    return KV_OK;
}

// what kv_format_records and kv_format_records_fd render from
struct FormatArgs {
    const uint64_t *rec_index, *ann_lo, *ann_hi;
    const uint32_t *ann_offset; const int32_t *ann_abund; const uint8_t *keep; const int32_t *case_abund;
    int nsamples, ksize;
    const char *names; const uint64_t *name_offs; const char *seqs; const uint64_t *seq_offs; const char *quals; const uint64_t *qual_offs;
    const uint8_t *is_fastq; const char *suffix; const uint64_t *suffix_offs;
    const uint32_t *mate_record; uint64_t n_mates; const char *mates; const uint64_t *mate_offs;
};

// output record j as text; false if an annotation does not fit its read (out.ok stays true) or memory ran out (out.ok false)
static bool format_one(KvTextOut &out, const FormatArgs &a, uint64_t j, std::vector<uint64_t> &order)
{
    const uint64_t r = a.rec_index[j];
    const char *seq = a.seqs + a.seq_offs[r];
    const size_t seq_len = (size_t)(a.seq_offs[r + 1] - a.seq_offs[r]);
    const bool fq = a.is_fastq ? a.is_fastq[r] != 0 : false;
    out.put(fq ? '@' : '>');
    out.put(a.names + a.name_offs[r], (size_t)(a.name_offs[r + 1] - a.name_offs[r]));
    if (a.suffix && a.suffix_offs) out.put(a.suffix + a.suffix_offs[j], (size_t)(a.suffix_offs[j + 1] - a.suffix_offs[j]));
    out.put('\n');
    out.put(seq, seq_len);
    if (fq) {
        out.put("\n+\n", 3);
        out.put(a.quals + a.qual_offs[r], (size_t)(a.qual_offs[r + 1] - a.qual_offs[r]));
    }
    out.put('\n');
    // the kept annotations in offset order (stable); nearly always they already are
    bool sorted = true, fits = true;
    uint32_t last = 0;
    for (uint64_t i = a.ann_lo[j]; i < a.ann_hi[j] && sorted; ++i)
        if (!a.keep || a.keep[i]) { sorted = a.ann_offset[i] >= last; last = a.ann_offset[i]; }
    auto line = [&](uint64_t i) {
        const uint32_t off = a.ann_offset[i];
        if ((size_t)off + (size_t)a.ksize > seq_len) { fits = false; return; }
        const int32_t *row = a.ann_abund + i * (uint64_t)a.nsamples;
        out.kmer_line(seq, off, a.ksize, a.nsamples, [&](int c) { return (int64_t)((c == 0 && a.case_abund) ? a.case_abund[i] : row[c]); });
    };
    if (sorted) {
        for (uint64_t i = a.ann_lo[j]; i < a.ann_hi[j]; ++i)
            if (!a.keep || a.keep[i]) line(i);
    } else {
        order.clear();
        for (uint64_t i = a.ann_lo[j]; i < a.ann_hi[j]; ++i)
            if (!a.keep || a.keep[i]) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](uint64_t x, uint64_t y) { return a.ann_offset[x] < a.ann_offset[y]; });
        for (const uint64_t i : order) line(i);
    }
    if (!fits) return false;
    if (a.n_mates) {
        const uint32_t *m = std::lower_bound(a.mate_record, a.mate_record + a.n_mates, (uint32_t)r);
        for (; m < a.mate_record + a.n_mates && *m == (uint32_t)r; ++m) {
            const uint64_t mi = (uint64_t)(m - a.mate_record);
            out.put("#mateseq=", 9);
            out.put(a.mates + a.mate_offs[mi], (size_t)(a.mate_offs[mi + 1] - a.mate_offs[mi]));
            out.put("#\n", 2);
        }
    }
    return out.ok;
}

extern "C" int kv_format_records(uint64_t n_out, const uint64_t *rec_index, const uint64_t *ann_lo, const uint64_t *ann_hi,
                                 const uint32_t *ann_offset, const int32_t *ann_abund, const uint8_t *keep, const int32_t *case_abund,
                                 int nsamples, int ksize, const char *names, const uint64_t *name_offs, const char *seqs,
                                 const uint64_t *seq_offs, const char *quals, const uint64_t *qual_offs, const uint8_t *is_fastq,
                                 const char *suffix, const uint64_t *suffix_offs, const uint32_t *mate_record, uint64_t n_mates,
                                 const char *mates, const uint64_t *mate_offs, char **text_out, uint64_t *bytes_out)
{
    KV_REQUIRE(text_out && bytes_out && (n_out == 0 || (rec_index && ann_lo && ann_hi && names && name_offs && seqs && seq_offs)), KV_ERR_ARG,
               "kv_format_records: null argument");
    const FormatArgs a = {rec_index, ann_lo, ann_hi, ann_offset, ann_abund, keep, case_abund, nsamples, ksize, names, name_offs, seqs, seq_offs,
                          quals, qual_offs, is_fastq, suffix, suffix_offs, mate_record, n_mates, mates, mate_offs};
    KvTextOut out;
    {
        uint64_t notes = 0;
        for (uint64_t j = 0; j < n_out; ++j) notes += ann_hi[j] - ann_lo[j];
        out.room((size_t)notes * (size_t)(ksize + 48) + (size_t)n_out * 320 + 4096);
    }
    std::vector<uint64_t> order;
    for (uint64_t j = 0; j < n_out; ++j) {
        const bool fine = format_one(out, a, j, order);
        KV_REQUIRE(fine || !out.ok, KV_ERR_ARG, "kv_format_records: an annotation does not fit its read (record %llu)", (unsigned long long)rec_index[j]);
        KV_REQUIRE(out.ok, KV_ERR_HIP, "kv_format_records: out of memory");
    }
    *bytes_out = out.len;
    *text_out = out.release();
    KV_REQUIRE(*text_out, KV_ERR_HIP, "kv_format_records: out of memory");
    return KV_OK;
}

// The same text written straight to a file descriptor: the records are rendered a stretch at a time by `nthreads` threads, each
// into a buffer of its own that it keeps (warm pages: a fresh 3 GB buffer costs a page fault per 4 KB, which is what the
// single-buffer form spent most of its time on at config-4 scale), and the stretches are written in order.  Nothing of the size of
// the output is ever resident, and the Python side neither copies nor holds the text.
extern "C" int kv_format_records_fd(uint64_t n_out, const uint64_t *rec_index, const uint64_t *ann_lo, const uint64_t *ann_hi,
                                    const uint32_t *ann_offset, const int32_t *ann_abund, const uint8_t *keep, const int32_t *case_abund,
                                    int nsamples, int ksize, const char *names, const uint64_t *name_offs, const char *seqs,
                                    const uint64_t *seq_offs, const char *quals, const uint64_t *qual_offs, const uint8_t *is_fastq,
                                    const char *suffix, const uint64_t *suffix_offs, const uint32_t *mate_record, uint64_t n_mates,
                                    const char *mates, const uint64_t *mate_offs, int fd, int nthreads, uint64_t *bytes_out)
{
    KV_REQUIRE(bytes_out && fd >= 0 && (n_out == 0 || (rec_index && ann_lo && ann_hi && names && name_offs && seqs && seq_offs)), KV_ERR_ARG,
               "kv_format_records_fd: bad argument");
    const FormatArgs a = {rec_index, ann_lo, ann_hi, ann_offset, ann_abund, keep, case_abund, nsamples, ksize, names, name_offs, seqs, seq_offs,
                          quals, qual_offs, is_fastq, suffix, suffix_offs, mate_record, n_mates, mates, mate_offs};
    *bytes_out = 0;
    if (n_out == 0) return KV_OK;
    const uint64_t per = 32768;                                   // records per stretch (~15 MB of 100-bp reads)
    const uint64_t n_chunks = (n_out + per - 1) / per;
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > n_chunks) nthreads = (int)n_chunks;
    std::atomic<uint64_t> next_chunk{0};
    std::mutex mu;
    std::condition_variable cv;
    uint64_t turn = 0, total = 0;                                 // guarded by mu: the stretch that may be written, bytes written
    int failed = 0;                                               // 1 annotation out of range, 2 memory, 3 write error
    int write_errno = 0;                                          // errno of the write that failed, taken where it failed
    uint64_t bad_record = 0;
    const off_t start_at = lseek(fd, 0, SEEK_CUR);                // -1: not seekable (a pipe): nothing can be taken back
    auto work = [&]() {
        KvTextOut out;
        std::vector<uint64_t> order;
        for (;;) {
            const uint64_t c = next_chunk.fetch_add(1);
            if (c >= n_chunks) return;
            out.len = 0;
            int mine = 0;
            uint64_t bad = 0;
            for (uint64_t j = c * per; j < std::min(n_out, (c + 1) * per) && !mine; ++j)
                if (!format_one(out, a, j, order)) { mine = out.ok ? 1 : 2; bad = rec_index[j]; }
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return turn == c; });
            if (!failed && mine) { failed = mine; bad_record = bad; }
            if (!failed) {
                size_t done = 0;
                while (done < out.len) {
                    const ssize_t w = write(fd, out.buf + done, out.len - done);
                    if (w < 0) { if (errno == EINTR) continue; write_errno = errno; failed = 3; break; }
                    done += (size_t)w;
                }
                total += done;
            }
            turn = c + 1;
            lk.unlock();
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    *bytes_out = total;
    if (failed && start_at >= 0) {
        // earlier stretches are in the file already: a failed call leaves a regular file as it found it (what the buffered form,
        // which writes nothing before everything is rendered, leaves), not a truncated but plausible list of records
        if (ftruncate(fd, start_at) == 0 && lseek(fd, start_at, SEEK_SET) == start_at) *bytes_out = 0;
    }
    KV_REQUIRE(failed != 1, KV_ERR_ARG, "kv_format_records_fd: an annotation does not fit its read (record %llu)", (unsigned long long)bad_record);
    KV_REQUIRE(failed != 2, KV_ERR_HIP, "kv_format_records_fd: out of memory");
    KV_REQUIRE(failed != 3, KV_ERR_IO, "kv_format_records_fd: write failed: %s", strerror(write_errno));
    return KV_OK;
}
