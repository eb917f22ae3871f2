"""The khmer sketch surface that kevlar calls, backed by HIP kernels on an MI355X.

Same class and method names, argument meaning and error behaviour as the khmer objects used
by the reference:

    Counttable / SmallCounttable / Nodetable / Countgraph / SmallCountgraph / Nodegraph
        (k, tablesize, ntables[, primes])      kevlar/sketch.py:99-119, kevlar/filter.py:29
    .load(path) / .save(path)                  kevlar/sketch.py:14-27, kevlar/count.py:95
    .consume_seqfile[_banding][_with_mask]     kevlar/count.py:43-71
    .get .add .hash .get_kmers .get_kmer_hashes .reverse_hash
                                               kevlar/novel.py:38,48,143,145, filter.py:32-34,67
    .hashsizes .n_occupied .n_unique_kmers .ksize .n_tables
                                               kevlar/sketch.py:62-74, kevlar/count.py:84
    ReadParser, _buckets_per_byte, khmer_args.memory_setting
                                               kevlar/count.py:33,40, kevlar/cli/count.py:49

Every table lives in HBM; every method that touches a table is a kernel launch through the
C ABI in include/kvsketch.h.  There is no CPU implementation behind these classes.
"""
import ctypes
import gzip
import os
import threading

import numpy as np

from kevlar_amd import _lib
from kevlar_amd._lib import check

KIND = {'Counttable': 0, 'SmallCounttable': 1, 'Nodetable': 2,
        'Countgraph': 3, 'SmallCountgraph': 4, 'Nodegraph': 5}

_buckets_per_byte = {'countgraph': 1, 'smallcountgraph': 2, 'nodegraph': 8}

# reads handed to the device per kv_consume call when streaming a file
BATCH_READS = 1 << 23


def _u64p(arr):
    return arr.ctypes.data_as(_lib.u64p)


def _u32p(arr):
    return arr.ctypes.data_as(_lib.u32p)


def _u8p(arr):
    return arr.ctypes.data_as(_lib.u8p)


# ----------------------------------------------------------------------------------------
# streams: independent samples are counted by concurrent host threads on concurrent HIP streams
# ----------------------------------------------------------------------------------------
class Stream(object):
    def __init__(self):
        _lib.require_device()
        handle = ctypes.c_void_p()
        check(_lib.load().kv_stream_create(ctypes.byref(handle)))
        self._h = handle

    def bind(self):
        """Make this the stream of the calling host thread."""
        check(_lib.load().kv_set_stream(self._h))
        _bound.stream = self

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().kv_stream_destroy(h)
            except Exception:
                pass


_pool_lock = threading.Lock()
_pool_streams = []
_bound = threading.local()


def bound_stream():
    """The Stream the calling host thread was bound to (None: the library's default stream).  Worker threads a job
    starts bind it too, so that the whole job stays on one stream."""
    return getattr(_bound, 'stream', None)



def scratch_trim():
    """Give the library's grow-only working buffers back to the device (kv_scratch_trim): after a very large batch, or before another
    library allocates most of the HBM.  No other call may be running; buckets a later scan would have reused are cut again."""
    check(_lib.load().kv_scratch_trim())


def run_concurrently(jobs):
    """Run zero-argument callables on separate host threads, each bound to its own HIP stream;
    returns their results in order.  Kernels of different jobs overlap on the GPU.  May be nested (a job that runs
    jobs of its own): every job at every level gets a stream nobody else is using."""
    if len(jobs) <= 1:
        return [job() for job in jobs]
    with _pool_lock:
        streams = [_pool_streams.pop() if _pool_streams else Stream() for _ in jobs]
    results, errors = [None] * len(jobs), []

    def work(i):
        try:
            streams[i].bind()
            results[i] = jobs[i]()
        except BaseException as exc:
            errors.append(exc)
        finally:
            _lib.load().kv_set_stream(None)
            _bound.stream = None

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    with _pool_lock:
        _pool_streams.extend(streams)
    if errors:
        raise errors[0]
    return results


# ----------------------------------------------------------------------------------------
# reads
# ----------------------------------------------------------------------------------------
class Read(object):
    __slots__ = ('name', 'sequence', 'quality')

    def __init__(self, name, sequence, quality=None):
        self.name = name
        self.sequence = sequence
        self.quality = quality


def _open_maybe_gz(path):
    with open(path, 'rb') as fh:
        magic = fh.read(2)
    if magic == b'\x1f\x8b':
        return gzip.open(path, 'rt')
    return open(path, 'r')


def _iter_fastx(path):
    with _open_maybe_gz(path) as fh:
        line = fh.readline()
        while line:
            if line.strip() == '':
                line = fh.readline()
                continue
            first = line[0]
            if first == '@':
                name = line[1:].rstrip('\r\n')
                seq = fh.readline().rstrip('\r\n')
                fh.readline()
                qual = fh.readline().rstrip('\r\n')
                yield Read(name, seq, qual)
                line = fh.readline()
            elif first == '>':
                name = line[1:].rstrip('\r\n')
                chunks = []
                line = fh.readline()
                while line and line[0] != '>':
                    chunks.append(line.strip())
                    line = fh.readline()
                yield Read(name, ''.join(chunks), None)
            else:
                raise ValueError('cannot parse sequence file ' + path)


class TextBatch(object):
    """One parsed batch: the packed reads in HBM (optional) plus the records' text as blobs.
    Python Read objects are only built on demand (record(i))."""

    def __init__(self, n, names, name_offs, seqs, seq_offs, quals, qual_offs, is_fastq, batch, fetch=None):
        self.n = n
        self._names, self._no = names, name_offs
        self._seqs, self._so = seqs, seq_offs
        self._quals, self._qo = quals, qual_offs
        self._fq = is_fastq
        self.batch = batch
        self._fetch = fetch        # batches served from a packed-read cache: (sequence, quality) of record i on request

    def name(self, i):
        return self._names[self._no[i]:self._no[i + 1]].decode('latin-1')

    def sequence(self, i):
        if self._fetch is not None:
            return self._fetch(i, int(self._so[i + 1] - self._so[i]), 0)[0]
        return self._seqs[self._so[i]:self._so[i + 1]].decode('latin-1')

    def record(self, i):
        if self._fetch is not None:
            seq, qual = self._fetch(i, int(self._so[i + 1] - self._so[i]), int(self._qo[i + 1] - self._qo[i]) if self._fq[i] else 0)
            return Read(self.name(i), seq, qual if self._fq[i] else None)
        qual = self._quals[self._qo[i]:self._qo[i + 1]].decode('latin-1') if self._fq[i] else None
        return Read(self.name(i), self.sequence(i), qual)

    def prefetch(self, indices):
        """Hint that record(i) will be asked for these i (a no-op when the text is already on the host)."""

    def augmented_text(self, hits, ksize, rec_index=None):
        """Augmented FASTA/FASTQ text (bytes) of the reads that hold `hits` = (read, offset, abund[n, S]) sorted by
        (read, offset), formatted natively (kv_format_augmented); None if this batch keeps no text blobs."""
        if self._fetch is not None or self._seqs is None:
            return None
        reads, offsets, abunds = hits[:3]
        n = len(reads)
        if n == 0:
            return b''
        reads = np.ascontiguousarray(reads, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        abunds = np.ascontiguousarray(abunds, dtype=np.uint8)
        if rec_index is None:
            rec_index = np.unique(reads)
        rec_index = np.ascontiguousarray(rec_index, dtype=np.uint64)
        no, so, qo = (np.ascontiguousarray(a, dtype=np.uint64) for a in (self._no, self._so, self._qo))
        fq = np.ascontiguousarray(self._fq, dtype=np.uint8)
        text, size, nrec = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        lib = _lib.load()
        check(lib.kv_format_augmented(_u32p(reads), _u32p(offsets), _u8p(abunds), n, abunds.shape[1], int(ksize), _u64p(rec_index),
                                      ctypes.cast(ctypes.c_char_p(self._names), ctypes.c_void_p), _u64p(no),
                                      ctypes.cast(ctypes.c_char_p(self._seqs), ctypes.c_void_p), _u64p(so),
                                      ctypes.cast(ctypes.c_char_p(self._quals), ctypes.c_void_p), _u64p(qo), _u8p(fq),
                                      ctypes.byref(text), ctypes.byref(size), ctypes.byref(nrec)))
        try:
            return ctypes.string_at(text, size.value)
        finally:
            lib.kv_text_free(text)

    def find_name(self, name):
        """Index of the first record called `name`, or -1."""
        raw = name.encode('latin-1')
        start = 0
        while True:
            pos = self._names.find(raw, start)
            if pos < 0:
                return -1
            i = int(np.searchsorted(self._no, pos, side='right')) - 1
            if self._no[i] == pos and self._no[i + 1] - pos == len(raw):
                return i
            start = pos + 1


class DeviceTextBatch(object):
    """A batch whose text never left HBM (BGZF FASTQ parsed on the device, kv_fastq.hip): records come to the host
    when asked for -- prefetch(indices) brings a set in one gather -- and are cached."""

    def __init__(self, n, batch, fetch):
        self.n = n
        self.batch = batch
        self._fetch = fetch          # indices -> TextBatch of exactly those records
        self._have = {}

    def prefetch(self, indices):
        want = [int(i) for i in dict.fromkeys(int(i) for i in indices) if int(i) not in self._have]
        if not want:
            return
        sub = self._fetch(want)
        for pos, i in enumerate(want):
            self._have[i] = sub.record(pos)

    def record(self, i):
        i = int(i)
        if i not in self._have:
            self.prefetch([i])
        return self._have[i]

    def name(self, i):
        return self.record(i).name

    def sequence(self, i):
        return self.record(i).sequence

    def augmented_text(self, hits, ksize):
        """see TextBatch.augmented_text: the hit reads are gathered from HBM in one piece and formatted natively"""
        reads = hits[0]
        if len(reads) == 0:
            return b''
        owners = np.unique(reads)
        sub = self._fetch(owners.tolist())
        return sub.augmented_text(hits, ksize, rec_index=np.arange(len(owners), dtype=np.uint64))

    def find_name(self, name):
        step = 1 << 18
        for lo in range(0, self.n, step):
            sub = self._fetch(list(range(lo, min(self.n, lo + step))))
            at = sub.find_name(name)
            if at >= 0:
                return lo + at
        return -1


class ReadParser(object):
    """FASTA/FASTQ reader (gzip transparent) over the native parser (kv_fastx_*); name = the header
    line after '@' or '>'.  Iteration yields Read objects (khmer's interface); the drivers use
    take_batch()/text_batches(), which never build per-read Python objects.

    Iteration is thread-safe: kevlar/count.py:41-76 shares one parser between threads."""

    def __init__(self, filename):
        handle = ctypes.c_void_p()
        check(_lib.load().kv_fastx_open(filename.encode(), ctypes.byref(handle)))
        self._h = handle
        self._lock = threading.Lock()
        self._pending = None
        self._cursor = 0
        cached = ctypes.c_int()
        check(_lib.load().kv_fastx_from_cache(handle, ctypes.byref(cached)))
        self.from_cache = bool(cached.value)     # streaming FILE.kvpack (KEVLAR_PACK_CACHE=1) instead of parsing FILE

    def _record_text(self, i, seqlen, quallen):
        seq, qual = ctypes.create_string_buffer(seqlen + 1), ctypes.create_string_buffer(quallen + 1)
        check(_lib.load().kv_fastx_record_text(self._h, int(i), seq, qual))
        return seq.raw[:seqlen].decode('latin-1'), qual.raw[:quallen].decode('latin-1')

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().kv_fastx_close(h)
            except Exception:
                pass

    @property
    def num_reads(self):
        n = ctypes.c_uint64()
        check(_lib.load().kv_fastx_num_reads(self._h, ctypes.byref(n)))
        return n.value - (self._pending.n - self._cursor if self._pending is not None else 0)

    def _next_text(self, max_reads, upload):
        lib = _lib.load()
        n = ctypes.c_uint64()
        reads = ctypes.c_void_p()
        check(lib.kv_fastx_next(self._h, max_reads, 1 if upload else 0, ctypes.byref(reads), ctypes.byref(n)))
        if n.value == 0:
            return None
        batch = None
        if upload:
            batch = ReadBatch.__new__(ReadBatch)
            batch._h = reads
            batch.n_reads = n.value
        return n.value, batch

    def take_batch(self, max_reads):
        """Next max_reads records as a ReadBatch in HBM (None at end of file); no text is copied."""
        _lib.require_device()
        with self._lock:
            got = self._next_text(max_reads, True)
        return None if got is None else got[1]

    def text_batch(self, max_reads, upload=True):
        """Next records as a TextBatch (text copied out of the parser; packed reads in HBM if upload)."""
        if upload:
            _lib.require_device()
        lib = _lib.load()
        with self._lock:
            got = self._next_text(max_reads, upload)
            if got is None:
                return None
            n, batch = got
            on_device = ctypes.c_int()
            check(lib.kv_fastx_on_device(self._h, ctypes.byref(on_device)))
            if on_device.value:
                return DeviceTextBatch(n, batch, self._fetch_records)
            return self._host_text(n, batch)

    def _fetch_records(self, indices):
        """Text of the given records of the current device-parsed batch as a TextBatch of len(indices) records."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        with self._lock:
            check(_lib.load().kv_fastx_fetch(self._h, _u64p(idx), len(idx)))
            return self._host_text(len(idx), None)

    def _host_text(self, n, batch):
        lib = _lib.load()
        vp, u64p, u8p = ctypes.c_void_p, _lib.u64p, _lib.u8p
        names, seqs, quals = vp(), vp(), vp()
        no, so, qo, fq = u64p(), u64p(), u64p(), u8p()
        check(lib.kv_fastx_batch_text(self._h, ctypes.byref(names), ctypes.byref(no), ctypes.byref(seqs),
                                      ctypes.byref(so), ctypes.byref(quals), ctypes.byref(qo), ctypes.byref(fq)))
        name_offs = np.ctypeslib.as_array(no, shape=(n + 1,)).copy()
        seq_offs = np.ctypeslib.as_array(so, shape=(n + 1,)).copy()
        qual_offs = np.ctypeslib.as_array(qo, shape=(n + 1,)).copy()
        is_fastq = np.ctypeslib.as_array(fq, shape=(n,)).copy()
        if self.from_cache:        # sequences and qualities stay in the cache file until a record is asked for
            tb = TextBatch(n, ctypes.string_at(names, int(name_offs[n])), name_offs, None, seq_offs, None, qual_offs,
                           is_fastq, batch, fetch=self._record_text)
        else:
            tb = TextBatch(n, ctypes.string_at(names, int(name_offs[n])), name_offs,
                           ctypes.string_at(seqs, int(seq_offs[n])), seq_offs,
                           ctypes.string_at(quals, int(qual_offs[n])), qual_offs, is_fastq, batch)
        return tb

    def text_batches(self, max_reads, upload=True):
        while True:
            tb = self.text_batch(max_reads, upload)
            if tb is None:
                return
            yield tb

    def __iter__(self):
        return self

    def __next__(self):
        with self._lock:
            pass
        if self._pending is None or self._cursor >= self._pending.n:
            self._pending = self.text_batch(4096, upload=False)
            self._cursor = 0
            if self._pending is None:
                raise StopIteration
        rec = self._pending.record(self._cursor)
        self._cursor += 1
        return rec

    def take(self, n):
        """Up to n reads as a list of Read objects."""
        out = []
        for read in self:
            out.append(read)
            if len(out) >= n:
                break
        return out


class ReadBatch(object):
    """A batch of reads 2-bit packed in HBM (kv_reads)."""

    def __init__(self, sequences):
        _lib.require_device()
        lib = _lib.load()
        n = len(sequences)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            np.cumsum(np.fromiter((len(s) for s in sequences), dtype=np.uint64, count=n), out=offs[1:])
        blob = ''.join(sequences).encode('latin-1')
        handle = ctypes.c_void_p()
        check(lib.kv_reads_create(blob, _u64p(offs), n, ctypes.byref(handle)))
        self._h = handle
        self.n_reads = n

    @classmethod
    def from_arrays(cls, blob, offs):
        """blob: bytes of concatenated ASCII bases; offs: uint64 array with n+1 entries."""
        _lib.require_device()
        lib = _lib.load()
        self = cls.__new__(cls)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        handle = ctypes.c_void_p()
        check(lib.kv_reads_create(blob, _u64p(offs), len(offs) - 1, ctypes.byref(handle)))
        self._h = handle
        self.n_reads = len(offs) - 1
        return self

    @classmethod
    def from_blob(cls, blob, offs):
        """sequences given as one bytes blob with offsets (read i = blob[offs[i]:offs[i + 1]])"""
        _lib.require_device()
        self = cls.__new__(cls)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        handle = ctypes.c_void_p()
        check(_lib.load().kv_reads_create(blob, _u64p(offs), len(offs) - 1, ctypes.byref(handle)))
        self._h = handle
        self.n_reads = len(offs) - 1
        return self

    @classmethod
    def from_packed(cls, words, read_len):
        """words: uint32 array [n_reads, ceil(read_len/16)], 2 bits per base (A0 C1 G2 T3)."""
        _lib.require_device()
        self = cls.__new__(cls)
        words = np.ascontiguousarray(words, dtype=np.uint32)
        handle = ctypes.c_void_p()
        check(_lib.load().kv_reads_create_packed(_u32p(words), words.shape[0], int(read_len), ctypes.byref(handle)))
        self._h = handle
        self.n_reads = int(words.shape[0])
        return self

    @classmethod
    def generate(cls, genome_len, seed, sample, first_read, n_reads, read_len=100, error_rate=0.005):
        """reads [first_read, first_read + n_reads) of the device-generated family (kv_reads_generate; sample 0 proband,
        1 mother, 2 father): nothing crosses PCIe.  kevlar_amd.synth.device_family_reads is the numpy restatement."""
        _lib.require_device()
        self = cls.__new__(cls)
        handle = ctypes.c_void_p()
        check(_lib.load().kv_reads_generate(int(genome_len), int(seed), int(sample), int(first_read), int(n_reads), int(read_len),
                                            float(error_rate), ctypes.byref(handle)))
        self._h = handle
        self.n_reads = int(n_reads)
        return self

    def packed_words(self, first_word, n_words):
        """words [first_word, first_word + n_words) of the packed batch, from the device"""
        out = np.empty(int(n_words), dtype=np.uint32)
        check(_lib.load().kv_reads_words_read(self._h, int(first_word), int(n_words), _u32p(out)))
        return out

    def num_kmers(self, ksize):
        n = ctypes.c_uint64()
        check(_lib.load().kv_reads_num_kmers(self._h, ksize, ctypes.byref(n)))
        return n.value

    def flagged_reads(self):
        """indices of the reads with a byte outside ACGT -- counted with stand-in bases, skipped by the scan (kv_reads_flags);
        read once per batch and kept"""
        if getattr(self, '_flagged', None) is None:
            nreads, nbases = ctypes.c_uint64(), ctypes.c_uint64()
            check(_lib.load().kv_reads_count(self._h, ctypes.byref(nreads), ctypes.byref(nbases)))
            flags = np.zeros(nreads.value, dtype=np.uint8)
            check(_lib.load().kv_reads_flags(self._h, ctypes.c_void_p(flags.ctypes.data)))
            self._flagged = np.flatnonzero(flags & 1).astype(np.int64)
        return self._flagged

    def device_bytes(self):
        """HBM the packed batch occupies, to within rounding: 2 bits per base plus 17 bytes of index per read."""
        nreads, nbases = ctypes.c_uint64(), ctypes.c_uint64()
        check(_lib.load().kv_reads_count(self._h, ctypes.byref(nreads), ctypes.byref(nbases)))
        return nbases.value // 4 + 17 * nreads.value

    def close(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            _lib.load().kv_reads_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


# ----------------------------------------------------------------------------------------
# sketches
# ----------------------------------------------------------------------------------------
def primes_below(target, n):
    out = (ctypes.c_uint64 * max(1, n))()
    found = ctypes.c_int()
    check(_lib.load().kv_primes_below(float(target), n, out, ctypes.byref(found)))
    return [int(out[i]) for i in range(found.value)]


class _Sketch(object):
    _kind = None

    def __init__(self, k, starting_size, n_tables, primes=None, _handle=None):
        self._lock = threading.Lock()
        self._consume_lock = threading.Lock()   # unique-new + consume as one step while the exact figure is tracked
        self._exact = None      # when tracking the exact distinct-k-mer figure: the k-mers counted as new so far (kv_unique_new), else None
        if _handle is not None:
            self._h = _handle
            return
        _lib.require_device()
        lib = _lib.load()
        if not primes:
            primes = primes_below(starting_size, int(n_tables))
        if len(primes) == 0:
            raise ValueError('table size {} is too small to hold any table'.format(starting_size))
        arr = (ctypes.c_uint64 * len(primes))(*[int(p) for p in primes])
        handle = ctypes.c_void_p()
        check(lib.kv_sketch_create(self._kind, int(k), len(primes), arr, ctypes.byref(handle)))
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.load().kv_sketch_destroy(h)
            except Exception:
                pass

    @classmethod
    def load(cls, path):
        _lib.require_device()
        handle = ctypes.c_void_p()
        check(_lib.load().kv_sketch_load(path.encode(), cls._kind, ctypes.byref(handle)))
        return cls(0, 0, 0, _handle=handle)

    def save(self, path):
        check(_lib.load().kv_sketch_save(self._h, path.encode()))

    def expect_scan(self, on=True, steady=False):
        """This sketch holds a case sample: every batch counted into it is scanned next (kevlar/novel.py:92-121 loads the cases
        last and scans them).  The count then keeps the batch's distinct k-mers with their hashes for that scan.  steady: the
        process counts and scans sample after sample, so the list is worth its allocation from the first batch on (a one-shot
        `kevlar novel` skips it there).  A performance hint only (kv_sketch_scan_hint)."""
        check(_lib.load().kv_sketch_scan_hint(self._h, (2 if steady else 1) if on else 0))
        return self

    def clear(self):
        """Zero all tables (same geometry, fresh counts)."""
        check(_lib.load().kv_sketch_clear(self._h))
        if self._exact is not None:
            self._exact = 0

    # ---- info ---------------------------------------------------------------------------
    def _info(self):
        info = _lib.SketchInfo()
        check(_lib.load().kv_sketch_info_get(self._h, ctypes.byref(info)))
        return info

    def ksize(self):
        return int(self._info().ksize)

    def n_tables(self):
        return int(self._info().ntables)

    def hashsizes(self):
        info = self._info()
        return [int(info.sizes[i]) for i in range(info.ntables)]

    def n_occupied(self):
        return int(self._info().n_occupied)

    def n_unique_kmers(self):
        """Distinct k-mers seen.  With track_exact_unique(True) this is the value one khmer
        thread reports for the same files in the same order; otherwise it carries the
        semantics of khmer's multi-threaded consume (see include/kvsketch.h kv_consume)."""
        if self._exact is not None:
            return int(self._exact)
        return int(self._info().n_unique)

    def track_exact_unique(self, on=True, budget_bytes=None):
        """From now on every consume_batch() first asks the library how many of the batch's k-mers are new to the tables as they
        stand (kv_unique_new: khmer's single-thread rule, batch by batch), and n_unique_kmers() reports the sum: the reference's
        "distinct k-mers stored" for one thread (kevlar/count.py:82-84), for a sample of any size -- nothing is kept resident
        (budget_bytes: ignored, the retention limit of earlier rounds is gone).  The figure is the single-thread one, so
        consume_batch() calls on a tracking sketch run one at a time (they take the sketch's consume lock); switching the
        tracking off gives the library's first-toucher arrays (4.5 x the sketch) back."""
        with self._lock:
            was = self._exact is not None
            self._exact = 0 if on else None
        if was and not on:
            check(_lib.load().kv_unique_release())

    def table_bytes(self, i):
        """Raw on-disk form of table i (tests compare this against the oracle)."""
        info = self._info()
        size = int(info.sizes[i])
        storage = {0: 'byte', 3: 'byte', 1: 'nibble', 4: 'nibble'}.get(self._kind, 'bit')
        nbytes = size if storage == 'byte' else (size // 2 + 1 if storage == 'nibble' else size // 8 + 1)
        buf = np.empty(nbytes, dtype=np.uint8)
        check(_lib.load().kv_sketch_table_read(self._h, i, _u8p(buf), nbytes))
        return buf.tobytes()

    # ---- hashing ------------------------------------------------------------------------
    def hash(self, kmer):
        k = self.ksize()
        if len(kmer) != k:
            raise ValueError('k-mer length {} does not match the sketch k-size {}'.format(len(kmer), k))
        out = ctypes.c_uint64()
        check(_lib.load().kv_hash_kmer(self._kind, kmer.encode(), k, ctypes.byref(out)))
        return out.value

    def reverse_hash(self, h):
        k = self.ksize()
        buf = ctypes.create_string_buffer(k + 1)
        check(_lib.load().kv_reverse_hash(self._kind, int(h), k, buf))
        return buf.value.decode()

    def get_kmers(self, seq):
        k = self.ksize()
        return [seq[i:i + k] for i in range(len(seq) - k + 1)]

    def hash_kmers(self, kmers):
        """Device-side hashing of a list of k-mers -> numpy uint64 array."""
        k = self.ksize()
        n = len(kmers)
        out = np.empty(n, dtype=np.uint64)
        if n:
            blob = ''.join(kmers).encode('latin-1')
            if len(blob) != n * k:
                raise ValueError('every k-mer must have length {}'.format(k))
            check(_lib.load().kv_hash_kmers(self._kind, blob, k, n, _u64p(out)))
        return out

    def get_kmer_counts(self, seq):
        """khmer's Hashtable.get_kmer_counts: the count of every k-mer of `seq`, in order
        (kevlar/simlike.py:24-35,87)."""
        k = self.ksize()
        if len(seq) < k:
            return []
        return [int(c) for c in self.get_hashes(self.hash_kmers(self.get_kmers(seq)))]

    def hash_positions(self, batch, ann_read, ann_offset):
        """Hashes of the k-mers at (read, offset) positions of a packed batch (kv_hash_positions): no k-mer text."""
        ann_read = np.ascontiguousarray(ann_read, dtype=np.uint32)
        ann_offset = np.ascontiguousarray(ann_offset, dtype=np.uint32)
        out = np.empty(len(ann_read), dtype=np.uint64)
        if len(out):
            check(_lib.load().kv_hash_positions(batch._h, self._kind, self.ksize(), _u32p(ann_read), _u32p(ann_offset),
                                                len(out), _u64p(out)))
        return out

    def get_kmer_hashes(self, seq):
        return [int(h) for h in self.hash_kmers(self.get_kmers(seq))]

    def _tohash(self, kmer):
        return int(kmer) if isinstance(kmer, (int, np.integer)) else self.hash(kmer)

    # ---- add / get ------------------------------------------------------------------------
    def get_hashes(self, hashes):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.empty(len(hashes), dtype=np.uint8)
        if len(hashes):
            check(_lib.load().kv_get_hashes(self._h, _u64p(hashes), len(hashes), _u8p(out)))
        return out

    def add_hashes(self, hashes):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        out = np.empty(len(hashes), dtype=np.uint8)
        if len(hashes):
            check(_lib.load().kv_add_hashes(self._h, _u64p(hashes), len(hashes), _u8p(out)))
        return out

    def get(self, kmer):
        return int(self.get_hashes(np.array([self._tohash(kmer)], dtype=np.uint64))[0])

    def add(self, kmer):
        return bool(self.add_hashes(np.array([self._tohash(kmer)], dtype=np.uint64))[0])

    count = add

    # ---- consume --------------------------------------------------------------------------
    def consume_batch(self, batch, nbands=0, band=0, mask=None, threshold=0, consume_masked=False):
        lib = _lib.load()
        args = (nbands or 0, band or 0, mask._h if mask is not None else None, int(threshold), 1 if consume_masked else 0)
        n = ctypes.c_uint64()
        if self._exact is None:
            check(lib.kv_consume(self._h, batch._h, *args, ctypes.byref(n)))
            return n.value
        # "new" is judged against the tables the batch is about to change, so the question and the count are one step: another
        # thread's batch between the two would be counted twice where the batches share k-mers
        with self._consume_lock:
            fresh = None
            if self._exact is not None:
                fresh = ctypes.c_uint64()
                try:
                    check(lib.kv_unique_new(self._h, batch._h, *args, ctypes.byref(fresh)))
                except (_lib.KvError, _lib.KvCapacityError):
                    # no room for the first-toucher arrays (4 bytes per bin) or a batch beyond 4.29e9 k-mers: the count goes on and
                    # n_unique_kmers() reports the estimate of kv_consume
                    fresh = None
            with self._lock:
                if fresh is None:
                    self._exact = None
                elif self._exact is not None:
                    self._exact += fresh.value
            check(lib.kv_consume(self._h, batch._h, *args, ctypes.byref(n)))
        return n.value

    def retains(self, batch):
        """(no batch is kept for the exact distinct-k-mer figure any more: always False)"""
        return False

    def consume_hashes(self, hashes_ptr, n, stride_words=1):
        """Count n hashes resident in HBM (device address; element i at word i * stride_words)."""
        added = ctypes.c_uint64()
        check(_lib.load().kv_consume_hashes(self._h, ctypes.c_void_p(hashes_ptr), int(n), int(stride_words),
                                            ctypes.byref(added)))
        return added.value

    def consume_hashes_weighted(self, items_ptr, n):
        """Count n (hash, occurrences) pairs resident in HBM; returns the occurrences they stand for."""
        added = ctypes.c_uint64()
        check(_lib.load().kv_consume_hashes_weighted(self._h, ctypes.c_void_p(items_ptr), int(n), ctypes.byref(added)))
        return added.value

    def consume(self, seq):
        return self.consume_batch(ReadBatch([seq]))

    def _consume_file(self, parser, nbands, band, mask, threshold, consume_masked):
        if isinstance(parser, str):
            parser = ReadParser(parser)
        nreads = nkmers = 0
        while True:
            batch = parser.take_batch(BATCH_READS)     # parsed and packed natively, straight into HBM
            if batch is None:
                break
            nkmers += self.consume_batch(batch, nbands, band, mask, threshold, consume_masked)
            nreads += batch.n_reads
            if not self.retains(batch):
                batch.close()
        return nreads, nkmers

    def consume_seqfile(self, parser):
        return self._consume_file(parser, 0, 0, None, 0, False)

    def abundance_distribution(self, parser, tracking):
        """khmer's Hashtable.abundance_distribution (kevlar/dist.py:53-54): a 65536-entry list whose entry c
        is the number of k-mers, each counted at its first occurrence according to `tracking` (which is
        updated), whose count in this sketch is c.  `parser`: file name, ReadParser or ReadBatch."""
        lib = _lib.load()
        total = np.zeros(65536, dtype=np.uint64)

        def one(batch):
            hist = np.zeros(65536, dtype=np.uint64)
            arr = (ctypes.c_void_p * 1)(batch._h)
            check(lib.kv_abundance_distribution(self._h, tracking._h, arr, 1, _u64p(hist)))
            np.add(total, hist, out=total)
        if isinstance(parser, ReadBatch):
            one(parser)
        else:
            if isinstance(parser, str):
                parser = ReadParser(parser)
            while True:
                batch = parser.take_batch(BATCH_READS)
                if batch is None:
                    break
                one(batch)
        return [int(v) for v in total]

    def consume_seqfile_banding(self, parser, nbands, band):
        return self._consume_file(parser, nbands, band, None, 0, False)

    def consume_seqfile_with_mask(self, parser, mask, threshold=0, consume_masked=False):
        return self._consume_file(parser, 0, 0, mask, threshold, consume_masked)

    def consume_seqfile_banding_with_mask(self, parser, nbands, band, mask, threshold=0,
                                          consume_masked=False):
        return self._consume_file(parser, nbands, band, mask, threshold, consume_masked)


class Counttable(_Sketch):
    _kind = 0


class SmallCounttable(_Sketch):
    _kind = 1


class Nodetable(_Sketch):
    _kind = 2


class Countgraph(_Sketch):
    _kind = 3


class SmallCountgraph(_Sketch):
    _kind = 4


class Nodegraph(_Sketch):
    _kind = 5


# ----------------------------------------------------------------------------------------
# the fused novel scan (no khmer analogue: replaces the per-k-mer Python loop of
# kevlar/novel.py:123-169)
# ----------------------------------------------------------------------------------------
def novel_scan(cases, controls, batch, case_min, ctrl_max, screen=None, band_mode=0, nbands=0,
               band=0, first_read=0, mask_ptr=None, mask_stride=0, lazy=False):
    """Returns (read_idx, offset, abund[n, S], discarded_reads) as numpy arrays, hits sorted
    by (read, offset).  lazy: returns a LazyHits instead, as soon as the scan's kernels are done (kv_hits_lazy): the sketches may
    be cleared and counted into again while the hit arrays are still on their way to the host; .arrays() waits for them."""
    lib = _lib.load()
    S = len(cases) + len(controls)
    ca = (ctypes.c_void_p * len(cases))(*[c._h for c in cases])
    cb = (ctypes.c_void_p * max(1, len(controls)))(*[c._h for c in controls])
    hits = ctypes.c_void_p()
    if lazy:
        check(lib.kv_hits_lazy(1))
    try:
        check(lib.kv_novel_scan(ca, len(cases), cb, len(controls), batch._h, int(first_read), int(case_min),
                                int(ctrl_max), int(screen or 0), int(band_mode), int(nbands or 0),
                                int(band or 0), mask_ptr, int(mask_stride), ctypes.byref(hits)))
    finally:
        if lazy:
            lib.kv_hits_lazy(0)
    if lazy:
        return LazyHits(hits, S)
    return _hits_arrays(hits, S)


class LazyHits(object):
    """the hits of a novel_scan(lazy=True): how many there are is known at once, the arrays when they have arrived"""

    def __init__(self, handle, S):
        self._handle, self._S, self._arrays = handle, S, None
        n = ctypes.c_uint64()
        check(_lib.load().kv_hits_count(handle, ctypes.byref(n), None))
        self.n = int(n.value)

    def arrays(self):
        if self._arrays is None:
            self._arrays = _hits_arrays(self._handle, self._S)       # (kv_hits_view waits for the copy; the arrays keep the handle alive)
            self._handle = None
        return self._arrays

    def __len__(self):
        return self.n

    def __del__(self):
        if self._handle is not None and _lib is not None:
            try:
                _lib.load().kv_hits_destroy(self._handle)            # (waits for a copy that nobody looked at)
            except Exception:
                pass


def _hits_arrays(hits, S):
    """(read, offset, abund[n, S], discarded) views into a kv_hits handle's pinned arrays: no second
    copy; the handle lives as long as the arrays do."""
    lib = _lib.load()
    holder = _HitsHandle(hits)
    n, nd = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.kv_hits_count(hits, ctypes.byref(n), ctypes.byref(nd)))
    pr, po, pa, pd = _lib.u32p(), _lib.u32p(), _lib.u8p(), _lib.u32p()
    check(lib.kv_hits_view(hits, ctypes.byref(pr), ctypes.byref(po), ctypes.byref(pa), ctypes.byref(pd)))

    def view(ptr, shape, dtype):
        if int(np.prod(shape)) == 0:
            return np.empty(shape, dtype=dtype)
        # the array's memory belongs to the kv_hits handle (pinned, recycled when the handle goes): numpy keeps the
        # object that exposes the memory alive for every view derived from it, np.asarray() included
        return np.asarray(_HitsMemory(ctypes.addressof(ptr.contents), shape, dtype, holder))
    reads = view(pr, (n.value,), np.uint32)
    offs = view(po, (n.value,), np.uint32)
    abund = view(pa, (n.value, S), np.uint8)
    disc = np.array(np.ctypeslib.as_array(pd, shape=(nd.value,)), dtype=np.uint32) if nd.value else np.empty(0, dtype=np.uint32)
    disc = disc.view(_Discarded)
    disc.shadow = (np.empty(0, dtype=np.uint32), np.empty(0, dtype=np.uint32))
    if nd.value:
        ps, pso, ns = _lib.u32p(), _lib.u32p(), ctypes.c_uint64()
        check(lib.kv_hits_shadow(hits, ctypes.byref(ps), ctypes.byref(pso), ctypes.byref(ns)))
        if ns.value:
            disc.shadow = (np.array(np.ctypeslib.as_array(ps, shape=(ns.value,)), dtype=np.uint32),
                           np.array(np.ctypeslib.as_array(pso, shape=(ns.value,)), dtype=np.uint32))
    return reads, offs, abund, disc


class _Discarded(np.ndarray):
    """indices of the reads the abundance screen dropped; .shadow = (read, offset) of their interesting k-mers in
    front of the k-mer that tripped the screen (the reference tallies those as unique novel k-mers before it drops
    the read, kevlar/novel.py:152-164)"""
    shadow = None


# ----------------------------------------------------------------------------------------
# read-sharded multi-GPU primitives (kv_shard.hip; driven by kevlar_amd/shardrun.py).  The *_ptr
# arguments are device addresses of caller-owned buffers (torch tensors' data_ptr()).
# ----------------------------------------------------------------------------------------
def route_hashes(batch, sketch_cls, ksize, ndest, read_index_base, with_tags, out_ptr, cap_items):
    """Hash every k-mer of `batch` and write the hashes grouped by the band that owns them (destination 0's
    items, then destination 1's, ... back to back in the buffer at out_ptr, cap_items >= batch.num_kmers);
    returns the number of items per destination."""
    counts = (ctypes.c_uint64 * int(ndest))()
    check(_lib.load().kv_route_hashes(batch._h, sketch_cls._kind, int(ksize), int(ndest), int(read_index_base),
                                      1 if with_tags else 0, ctypes.c_void_p(out_ptr), int(cap_items), counts))
    return [int(c) for c in counts]


def route_distinct(batch, sketch_cls, ksize, ndest, out_ptr, cap_items):
    """route_hashes for a count, with the shard deduplicated first: the buffer receives (hash, occurrences) pairs,
    one per distinct k-mer of a super-k-mer bucket; returns the number of pairs per destination."""
    counts = (ctypes.c_uint64 * int(ndest))()
    check(_lib.load().kv_route_distinct(batch._h, sketch_cls._kind, int(ksize), int(ndest), ctypes.c_void_p(out_ptr),
                                        int(cap_items), counts))
    return [int(c) for c in counts]


def mex_plan(sketch_cls, ksize, n_reads_global, read_len, ndest, short=False):
    """The geometry of a minimizer-sharded exchange of one sample (kv_mex_plan_make): the same on every rank, because it
    depends on the sample's global size only.  short: 16-byte records without read positions where the plan's shape has them
    (kv_mex_plan_short; k = 31 and reads of up to 224 bases today) -- for a sample nobody scans from these records; a shape
    without them keeps the classic plan (plan.flags & 1 says which it is)."""
    plan = _lib.MexPlan()
    check(_lib.load().kv_mex_plan_make(sketch_cls._kind, int(ksize), int(n_reads_global), int(read_len), int(ndest), ctypes.byref(plan)))
    if short:
        rc = _lib.load().kv_mex_plan_short(ctypes.byref(plan))
        if rc != _lib.KV_ERR_NOTIMPL:
            check(rc)
    return plan


def mex_emit(batch, plan, read_base, seg_ptr, cnt_ptr):
    """Cut this rank's shard into super-k-mer records, grouped by minimizer bucket, in the exchange buffers (kv_mex_emit)."""
    check(_lib.load().kv_mex_emit(batch._h, ctypes.byref(plan), int(read_base), ctypes.c_void_p(seg_ptr), ctypes.c_void_p(cnt_ptr)))


def mex_pack(plan, seg_ptr, cnt_ptr, out_ptr):
    """The filled part of the exchange segments, destination after destination (kv_mex_pack); returns records per destination."""
    counts = (ctypes.c_uint64 * int(plan.ndest))()
    check(_lib.load().kv_mex_pack(ctypes.byref(plan), ctypes.c_void_p(seg_ptr), ctypes.c_void_p(cnt_ptr), ctypes.c_void_p(out_ptr), counts))
    return [int(c) for c in counts]


def mex_emit_pack(batch, plan, read_base, seg_ptr, cnt_ptr, out_ptr, out_cap_words):
    """mex_emit + mex_pack in one call and one synchronisation (kv_mex_emit_pack); returns (records per destination, packed):
    packed False = the filled part of the segments did not fit out_cap_words and mex_pack into a bigger buffer has to follow."""
    counts = (ctypes.c_uint64 * int(plan.ndest))()
    packed = ctypes.c_int(0)
    check(_lib.load().kv_mex_emit_pack(batch._h, ctypes.byref(plan), int(read_base), ctypes.c_void_p(seg_ptr), ctypes.c_void_p(cnt_ptr),
                                       ctypes.c_void_p(out_ptr), int(out_cap_words), counts, ctypes.byref(packed)))
    return [int(c) for c in counts], bool(packed.value)


def mex_route(plan, my_dest, recv_seg_ptr, recv_cnt_ptr, n_src, out_ptr, cap_items, compact=False, keep_scan=False):
    """Combine the records n_src ranks sent for this rank's buckets and write one (hash, occurrences) pair per distinct
    k-mer, grouped by band owner (kv_mex_route); returns (pairs per destination, k-mer occurrences that arrived).
    keep_scan: the combined buckets stay for mex_scan_set (the case sample)."""
    counts = (ctypes.c_uint64 * int(plan.ndest))()
    arrived = ctypes.c_uint64()
    check(_lib.load().kv_mex_route(ctypes.byref(plan), int(my_dest), ctypes.c_void_p(recv_seg_ptr), ctypes.c_void_p(recv_cnt_ptr), int(n_src),
                                   1 if compact else 0, 1 if keep_scan else 0, ctypes.c_void_p(out_ptr), int(cap_items), counts, ctypes.byref(arrived)))
    return [int(c) for c in counts], arrived.value


def pairs_pack(pairs_ptr, counts, out_ptr, out_cap_words):
    """(hash, occurrences) pairs, counts[d] of them for destination d, into their 9-byte travelling form (kv_pairs_pack); returns the
    64-bit words of every destination's block."""
    nd = len(counts)
    c = (ctypes.c_uint64 * nd)(*[int(v) for v in counts])
    w = (ctypes.c_uint64 * nd)()
    check(_lib.load().kv_pairs_pack(ctypes.c_void_p(pairs_ptr), c, nd, ctypes.c_void_p(out_ptr), int(out_cap_words), w))
    return [int(v) for v in w]


def pairs_unpack(in_ptr, words_per_src, pairs_ptr, cap_pairs):
    """Received blocks back into 16-byte (hash, count <= 255) pairs (kv_pairs_unpack); returns (pairs per source, occurrences they stand
    for -- exact, from the blocks' heads)."""
    ns = len(words_per_src)
    w = (ctypes.c_uint64 * ns)(*[int(v) for v in words_per_src])
    n = (ctypes.c_uint64 * ns)()
    occ = ctypes.c_uint64()
    check(_lib.load().kv_pairs_unpack(ctypes.c_void_p(in_ptr), w, ns, ctypes.c_void_p(pairs_ptr), int(cap_pairs), n, ctypes.byref(occ)))
    return [int(v) for v in n], int(occ.value)


def mex_scan_set(sketch_cls, ksize, nsamples, hashes_ptr, abund_ptr, n, hit_tags_ptr, hit_abund_ptr, hit_cap):
    """The hits of this rank's minimizer buckets against the gathered set of interesting hashes (kv_mex_scan_set): returns how
    many (tag, abundances) rows were written; raises KvCapacityError when the owner cannot answer (scan the shard instead)."""
    n_hits = ctypes.c_uint64()
    check(_lib.load().kv_mex_scan_set(sketch_cls._kind, int(ksize), int(nsamples), ctypes.c_void_p(hashes_ptr), ctypes.c_void_p(abund_ptr), int(n),
                                      ctypes.c_void_p(hit_tags_ptr), ctypes.c_void_p(hit_abund_ptr), int(hit_cap), ctypes.byref(n_hits)))
    return n_hits.value


def novel_scan_hashes(cases, controls, items_ptr, n_items, case_min, ctrl_max, hit_tags_ptr, hit_abund_ptr, hit_cap):
    """kmer_is_interesting() over (hash, tag) pairs in HBM; returns the number of hits written."""
    ca = (ctypes.c_void_p * len(cases))(*[c._h for c in cases])
    cb = (ctypes.c_void_p * max(1, len(controls)))(*[c._h for c in controls])
    n = ctypes.c_uint64()
    check(_lib.load().kv_novel_scan_hashes(ca, len(cases), cb, len(controls), ctypes.c_void_p(items_ptr), int(n_items),
                                           int(case_min), int(ctrl_max), ctypes.c_void_p(hit_tags_ptr),
                                           ctypes.c_void_p(hit_abund_ptr), int(hit_cap), ctypes.byref(n)))
    return n.value


def novel_scan_distinct(cases, controls, items_ptr, n_items, case_min, ctrl_max, hit_hashes_ptr, hit_abund_ptr, hit_cap):
    """kmer_is_interesting() over (hash, occurrences) pairs in HBM: the interesting ones leave as (hash, abundances);
    returns how many."""
    ca = (ctypes.c_void_p * len(cases))(*[c._h for c in cases])
    cb = (ctypes.c_void_p * max(1, len(controls)))(*[c._h for c in controls])
    n = ctypes.c_uint64()
    check(_lib.load().kv_novel_scan_distinct(ca, len(cases), cb, len(controls), ctypes.c_void_p(items_ptr), int(n_items),
                                             int(case_min), int(ctrl_max), ctypes.c_void_p(hit_hashes_ptr),
                                             ctypes.c_void_p(hit_abund_ptr), int(hit_cap), ctypes.byref(n)))
    return n.value


def novel_scan_set(batch, sketch_cls, ksize, nsamples, hashes_ptr, abund_ptr, n):
    """The hits of `batch` against a known set of interesting k-mers: n hashes in HBM (~0 = padding) with nsamples
    abundances each; returns (read, offset, abund[n, S]) in (read, offset) order."""
    hits = ctypes.c_void_p()
    check(_lib.load().kv_novel_scan_set(batch._h, sketch_cls._kind, int(ksize), int(nsamples), ctypes.c_void_p(hashes_ptr),
                                        ctypes.c_void_p(abund_ptr), int(n), ctypes.byref(hits)))
    r, o, a, _ = _hits_arrays(hits, nsamples)
    return r, o, a


def hits_from_tagged(tags_ptr, abund_ptr, n_total, n_valid, nsamples):
    """Sort gathered (tag, abundances) hits into (read, offset) order; returns (read, offset, abund)."""
    hits = ctypes.c_void_p()
    check(_lib.load().kv_hits_from_tagged(ctypes.c_void_p(tags_ptr), ctypes.c_void_p(abund_ptr), int(n_total),
                                          int(n_valid), int(nsamples), ctypes.byref(hits)))
    r, o, a, _ = _hits_arrays(hits, nsamples)
    return r, o, a


class _HitsHandle(object):
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = self._h, None
        if h:
            try:
                _lib.load().kv_hits_destroy(h)
            except Exception:
                pass


class _HitsMemory(object):
    """a block of a kv_hits handle's pinned memory, exposed through the array interface together with its owner"""

    def __init__(self, address, shape, dtype, owner):
        self._owner = owner
        self.__array_interface__ = {'data': (int(address), False), 'shape': tuple(int(d) for d in shape),
                                    'typestr': np.dtype(dtype).str, 'version': 3}


class _Owned(np.ndarray):
    """ndarray view that keeps the kv_hits handle (the owner of its memory) alive."""

    def __new__(cls, arr, owner):
        obj = np.asarray(arr).view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, '_owner', None)


def readgraph_components(batch, ksize, ann_read, ann_offset, node_of_read, n_nodes, minabund=0,
                         maxabund=0, want_edges=False):
    """labels[node] = smallest node id of its connected component (kv_readgraph_components)."""
    lib = _lib.load()
    ann_read = np.ascontiguousarray(ann_read, dtype=np.uint32)
    ann_offset = np.ascontiguousarray(ann_offset, dtype=np.uint32)
    node_of_read = np.ascontiguousarray(node_of_read, dtype=np.uint32)
    labels = np.empty(n_nodes, dtype=np.uint32)
    nedges = ctypes.c_uint64()
    check(lib.kv_readgraph_components(batch._h, int(ksize), _u32p(ann_read), _u32p(ann_offset), len(ann_read),
                                      _u32p(node_of_read), int(n_nodes), int(minabund or 0), int(maxabund or 0),
                                      _u32p(labels), ctypes.byref(nedges) if want_edges else None))
    return (labels, nedges.value) if want_edges else labels


# ----------------------------------------------------------------------------------------
# khmer.khmer_args.memory_setting (kevlar/cli/count.py:49): K/M/G/T are powers of 1000
# ----------------------------------------------------------------------------------------
class khmer_args(object):
    @staticmethod
    def memory_setting(label):
        suffixes = {'K': 1e3, 'M': 1e6, 'G': 1e9, 'T': 1e12}
        try:
            return float(label)
        except ValueError:
            prefix, suffix = label[:-1], label[-1:].upper()
            if suffix not in suffixes:
                raise ValueError('cannot parse memory setting "{}"'.format(label))
            try:
                return float(prefix) * suffixes[suffix]
            except ValueError:
                raise ValueError('cannot parse memory setting "{}"'.format(label))


def calc_expected_collisions(sketch, force=False, max_false_pos=.2):
    sizes = sketch.hashsizes()
    return (float(sketch.n_occupied()) / min(sizes)) ** len(sizes)
