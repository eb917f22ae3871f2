"""okhmer -- a khmer-shaped Python surface over the CPU oracle (libkvoracle.so).

TEST INFRASTRUCTURE ONLY (see oracle/kvoracle.h).  Two users:

* tests/ compare the HIP path against these classes on the same inputs;
* tests/golden/make_golden.py registers this module as ``khmer`` so that the reference's
  own Python drivers (kevlar/count.py, novel.py, filter.py, partition.py under
  /root/reference) can be imported *in the build container* to generate golden vectors.

The class/method names are khmer's, exactly as kevlar calls them:
kevlar/sketch.py:14-27,99-119, kevlar/count.py:40-71, kevlar/novel.py:38,48,143,145,
kevlar/filter.py:29-34,67.  Nothing in kevlar_amd/ imports this file.
"""
import ctypes
import gzip
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, 'libkvoracle.so')


def build(force=False):
    """Compile the oracle's C restatement (gcc) if the shared object is missing or stale."""
    src = os.path.join(_HERE, 'kvoracle.c')
    hdr = os.path.join(_HERE, 'kvoracle.h')
    stale = (not os.path.exists(_LIBPATH)
             or os.path.getmtime(_LIBPATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        import fcntl
        with open(_LIBPATH + '.lock', 'w') as lock:      # concurrent test workers / ranks: one builds
            fcntl.flock(lock, fcntl.LOCK_EX)
            if force or not os.path.exists(_LIBPATH) or \
                    os.path.getmtime(_LIBPATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
                subprocess.check_call(['make', '-C', _HERE, '-B', 'libkvoracle.so'], stdout=subprocess.DEVNULL)
    return _LIBPATH


def _load():
    build()
    lib = ctypes.CDLL(_LIBPATH)
    u64, i32, vp, cp = ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p
    pu64 = ctypes.POINTER(ctypes.c_uint64)
    sig = {
        'kvo_murmur3_x64_128_lo': (u64, [cp, i32, ctypes.c_uint32]),
        'kvo_hash_murmur': (u64, [cp, i32]),
        'kvo_hash_2bit': (i32, [cp, i32, pu64]),
        'kvo_hash': (u64, [i32, cp, i32]),
        'kvo_reverse_hash_2bit': (None, [u64, i32, ctypes.c_char_p]),
        'kvo_primes_below': (i32, [ctypes.c_double, i32, pu64]),
        'kvo_sketch_create': (vp, [i32, i32, i32, pu64]),
        'kvo_sketch_free': (None, [vp]),
        'kvo_sketch_load': (vp, [cp, i32]),
        'kvo_sketch_save': (i32, [vp, cp]),
        'kvo_kind': (i32, [vp]),
        'kvo_ksize': (i32, [vp]),
        'kvo_ntables': (i32, [vp]),
        'kvo_tablesize': (u64, [vp, i32]),
        'kvo_n_occupied': (u64, [vp]),
        'kvo_n_unique': (u64, [vp]),
        'kvo_table_bytes': (ctypes.POINTER(ctypes.c_uint8), [vp, i32, pu64]),
        'kvo_add_hash': (i32, [vp, u64]),
        'kvo_get_hash': (i32, [vp, u64]),
        'kvo_consume': (u64, [vp, cp, ctypes.c_size_t, i32, i32, vp, i32, i32]),
        'kvo_consume_reads': (u64, [vp, cp, pu64, u64, i32, i32, vp, i32, i32]),
        'kvo_band_bounds': (None, [i32, i32, pu64, pu64]),
        'kvo_consume_reads_mt': (u64, [vp, cp, pu64, u64, i32]),
        'kvo_consume_reads_mt_banded': (u64, [vp, cp, pu64, u64, i32, i32, i32]),
        'kvo_novel_scan_mt': (ctypes.c_int64, [
            ctypes.POINTER(vp), i32, ctypes.POINTER(vp), i32, cp, pu64, u64, i32, i32, i32, i32, i32, i32,
            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64, i32]),
        'kvo_consume_reads_mt_allbands': (u64, [ctypes.POINTER(vp), i32, cp, pu64, u64, i32]),
        'kvo_novel_scan_mt_allbands': (ctypes.c_int64, [
            ctypes.POINTER(vp), i32, ctypes.POINTER(vp), i32, i32, cp, pu64, u64, i32, i32, i32,
            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8),
            ctypes.c_int64, i32]),
        'kvo_novel_scan_count_mt': (ctypes.c_int64, [ctypes.POINTER(vp), i32, ctypes.POINTER(vp), i32, cp, pu64, u64,
                                                     i32, i32, i32, i32]),
        'kvo_abundance_distribution': (u64, [vp, vp, cp, ctypes.c_size_t, pu64]),
        'kvo_novel_scan': (ctypes.c_int64, [
            ctypes.POINTER(vp), i32, ctypes.POINTER(vp), i32, cp, pu64, u64, i32, i32, i32, i32,
            i32, i32, i32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint16),
            ctypes.POINTER(ctypes.c_uint8), ctypes.c_int64, ctypes.POINTER(ctypes.c_uint8)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

KIND = {'Counttable': 0, 'SmallCounttable': 1, 'Nodetable': 2,
        'Countgraph': 3, 'SmallCountgraph': 4, 'Nodegraph': 5}

# khmer/__init__.py constant used at kevlar/count.py:33
_buckets_per_byte = {'countgraph': 1, 'smallcountgraph': 2, 'nodegraph': 8}


def murmur_lo(data, seed=0):
    if isinstance(data, str):
        data = data.encode('ascii')
    return lib.kvo_murmur3_x64_128_lo(data, len(data), seed)


def primes_below(target, n):
    out = (ctypes.c_uint64 * n)()
    found = lib.kvo_primes_below(float(target), n, out)
    return [int(out[i]) for i in range(found)]


def band_bounds(nbands, band):
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    lib.kvo_band_bounds(nbands, band, ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


# ----------------------------------------------------------------------------------------
# FASTA/FASTQ reader standing in for khmer.ReadParser (kevlar/count.py:40,
# kevlar/__init__.py:125-128).  name = the whole header line after '@' / '>'.
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


def iter_fastx(path):
    with _open_maybe_gz(path) as fh:
        line = fh.readline()
        while line:
            if line.strip() == '':
                line = fh.readline()
                continue
            if line[0] == '@':
                name = line[1:].rstrip('\r\n')
                seq = fh.readline().rstrip('\r\n')
                fh.readline()
                qual = fh.readline().rstrip('\r\n')
                yield Read(name, seq, qual)
                line = fh.readline()
            elif line[0] == '>':
                name = line[1:].rstrip('\r\n')
                chunks = []
                line = fh.readline()
                while line and line[0] != '>':
                    chunks.append(line.strip())
                    line = fh.readline()
                yield Read(name, ''.join(chunks), None)
            else:
                raise ValueError('cannot parse sequence file ' + path)


class ReadParser(object):
    def __init__(self, filename):
        self._iter = iter_fastx(filename)
        self._lock = threading.Lock()
        self.num_reads = 0

    def __iter__(self):
        return self

    def __next__(self):
        with self._lock:
            read = next(self._iter)
            self.num_reads += 1
            return read


# ----------------------------------------------------------------------------------------
# sketches
# ----------------------------------------------------------------------------------------
class _Sketch(object):
    _kind = None

    def __init__(self, k, starting_size, n_tables, primes=None, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        if not primes:
            primes = primes_below(starting_size, n_tables)
        arr = (ctypes.c_uint64 * len(primes))(*primes)
        self._h = lib.kvo_sketch_create(self._kind, int(k), len(primes), arr)
        if not self._h:
            raise MemoryError('oracle sketch allocation failed')

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            lib.kvo_sketch_free(h)

    @classmethod
    def load(cls, path):
        h = lib.kvo_sketch_load(path.encode(), cls._kind)
        if not h:
            raise OSError('cannot load sketch of type {} from {}'.format(cls.__name__, path))
        return cls(0, 0, 0, _handle=h)

    def save(self, path):
        if lib.kvo_sketch_save(self._h, path.encode()) != 0:
            raise OSError('cannot write ' + path)

    # -- info
    def ksize(self):
        return lib.kvo_ksize(self._h)

    def n_tables(self):
        return lib.kvo_ntables(self._h)

    def hashsizes(self):
        return [int(lib.kvo_tablesize(self._h, i)) for i in range(self.n_tables())]

    def n_occupied(self):
        return int(lib.kvo_n_occupied(self._h))

    def n_unique_kmers(self):
        return int(lib.kvo_n_unique(self._h))

    def table_bytes(self, i):
        n = ctypes.c_uint64()
        p = lib.kvo_table_bytes(self._h, i, ctypes.byref(n))
        return ctypes.string_at(p, n.value)

    # -- hashing
    def hash(self, kmer):
        if len(kmer) != self.ksize():
            raise ValueError('k-mer length must equal the k-size')
        if self._kind >= 3:
            out = ctypes.c_uint64()
            if lib.kvo_hash_2bit(kmer.encode(), len(kmer), ctypes.byref(out)) != 0:
                raise ValueError('invalid DNA character in k-mer')
            return out.value
        return int(lib.kvo_hash_murmur(kmer.encode(), len(kmer)))

    def reverse_hash(self, h):
        if self._kind < 3:
            raise ValueError('not implemented for this hash function')
        buf = ctypes.create_string_buffer(self.ksize() + 1)
        lib.kvo_reverse_hash_2bit(h, self.ksize(), buf)
        return buf.value.decode()

    def get_kmers(self, seq):
        k = self.ksize()
        return [seq[i:i + k] for i in range(len(seq) - k + 1)]

    def get_kmer_counts(self, seq):
        return [self.get(kmer) for kmer in self.get_kmers(seq)]

    def get_kmer_hashes(self, seq):
        return [self.hash(km) for km in self.get_kmers(seq)]

    def _tohash(self, kmer):
        return kmer if isinstance(kmer, int) else self.hash(kmer)

    # -- add / get
    def get(self, kmer):
        return lib.kvo_get_hash(self._h, self._tohash(kmer))

    def add(self, kmer):
        return bool(lib.kvo_add_hash(self._h, self._tohash(kmer)))

    count = add

    def consume(self, seq):
        b = seq.encode()
        return int(lib.kvo_consume(self._h, b, len(b), 0, 0, None, 0, 0))

    # -- file consumers (kevlar/count.py:43-71)
    def _consume_file(self, parser, nbands, band, mask, threshold, consume_masked):
        if isinstance(parser, str):
            parser = ReadParser(parser)
        nreads = nkmers = 0
        mh = mask._h if mask is not None else None
        for read in parser:
            b = read.sequence.encode()
            nkmers += lib.kvo_consume(self._h, b, len(b), nbands, band, mh, threshold,
                                      1 if consume_masked else 0)
            nreads += 1
        return nreads, int(nkmers)

    def consume_seqfile(self, parser):
        return self._consume_file(parser, 0, 0, None, 0, False)

    def abundance_distribution(self, parser, tracking):
        """khmer's Hashtable.abundance_distribution (kevlar/dist.py:53-54): a 65536-entry list,
        entry c = number of distinct (per `tracking`) k-mers whose count in this sketch is c."""
        if isinstance(parser, str):
            parser = ReadParser(parser)
        hist = (ctypes.c_uint64 * 65536)()
        for read in parser:
            b = read.sequence.encode()
            lib.kvo_abundance_distribution(self._h, tracking._h, b, len(b), hist)
        return list(hist)

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
# khmer.khmer_args.memory_setting as used by kevlar/cli/count.py:49 etc.
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
            return float(prefix) * suffixes[suffix]


def calc_expected_collisions(sketch, force=False, max_false_pos=.2):
    sizes = sketch.hashsizes()
    fp_one = float(sketch.n_occupied()) / min(sizes)
    return fp_one ** len(sizes)


# ----------------------------------------------------------------------------------------
# batch helpers used by tests and bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------------------
def concat_reads(seqs):
    """list of str -> (bytes, c_uint64 array of n+1 offsets)"""
    offs = (ctypes.c_uint64 * (len(seqs) + 1))()
    total = 0
    for i, s in enumerate(seqs):
        offs[i] = total
        total += len(s)
    offs[len(seqs)] = total
    return ''.join(seqs).encode(), offs


def consume_reads(sketch, bases, offs, n_reads, nbands=0, band=0, mask=None, threshold=0,
                  consume_masked=False):
    mh = mask._h if mask is not None else None
    return int(lib.kvo_consume_reads(sketch._h, bases, offs, n_reads, nbands, band, mh,
                                     threshold, 1 if consume_masked else 0))


def consume_reads_mt(sketch, bases, offs, n_reads, nthreads):
    """khmer-style threaded consume (kevlar/count.py:41-76): nthreads threads, one sketch, atomic saturating adds"""
    return int(lib.kvo_consume_reads_mt(sketch._h, bases, offs, n_reads, int(nthreads)))


def consume_reads_mt_banded(sketch, bases, offs, n_reads, nthreads, nbands, band):
    """consume_reads_mt keeping only hash band `band` of `nbands` (consume_seqfile_banding, kevlar/count.py:62-66)"""
    return int(lib.kvo_consume_reads_mt_banded(sketch._h, bases, offs, n_reads, int(nthreads), int(nbands), int(band)))


def novel_scan_mt(cases, ctrls, bases, offs, n_reads, ksize, case_min, ctrl_max, nthreads, band_mode=0, nbands=0, band=0, cap=1 << 22):
    """novel_scan (no abundance screen) on nthreads threads -> (read u32[n], offset u16[n], abund u8[n, S]) numpy arrays in scan order"""
    import numpy as np
    S = len(cases) + len(ctrls)
    vp = ctypes.c_void_p
    ca = (vp * len(cases))(*[c._h for c in cases])
    cb = (vp * max(1, len(ctrls)))(*[c._h for c in ctrls])
    while True:
        hr, ho, ha = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.uint16), np.empty((cap, S), dtype=np.uint8)
        n = lib.kvo_novel_scan_mt(ca, len(cases), cb, len(ctrls), bases, offs, n_reads, ksize, case_min, ctrl_max, band_mode, nbands, band,
                                  hr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ho.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                  ha.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), cap, int(nthreads))
        if n < 0:
            raise MemoryError('kvo_novel_scan_mt')
        if n <= cap:                      # (a range that outgrew its share of the buffers comes back as a number above cap)
            return hr[:n].copy(), ho[:n].copy(), ha[:n].copy()
        cap = int(n) * 2


def consume_reads_mt_allbands(sketches, bases, offs, n_reads, nthreads):
    """ALL bands of a banded count in one pass: sketches[b] receives what consume_reads_mt_banded(sketches[b], .., len(sketches), b) adds"""
    arr = (ctypes.c_void_p * len(sketches))(*[s._h for s in sketches])
    return int(lib.kvo_consume_reads_mt_allbands(arr, len(sketches), bases, offs, n_reads, int(nthreads)))


def novel_scan_mt_allbands(cases_by_band, ctrls_by_band, bases, offs, n_reads, ksize, case_min, ctrl_max, nthreads, cap=1 << 22):
    """the scans of ALL bands in one pass: cases_by_band[b] / ctrls_by_band[b] are band b's sketches; returns (read u32[n], offset u16[n],
    abund u8[n, S], band u8[n]) in (read, offset) order -- the rows with band == b are novel_scan_mt(band_mode=1, band=b) over band b's sketches"""
    import numpy as np
    nbands, ncase, nctrl = len(cases_by_band), len(cases_by_band[0]), len(ctrls_by_band[0])
    S = ncase + nctrl
    vp = ctypes.c_void_p
    ca = (vp * (nbands * ncase))(*[c._h for band in cases_by_band for c in band])
    cb = (vp * max(1, nbands * nctrl))(*[c._h for band in ctrls_by_band for c in band])
    while True:
        hr, ho, ha = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.uint16), np.empty((cap, S), dtype=np.uint8)
        hb = np.empty(cap, dtype=np.uint8)
        n = lib.kvo_novel_scan_mt_allbands(ca, ncase, cb, nctrl, nbands, bases, offs, n_reads, ksize, case_min, ctrl_max,
                                           hr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), ho.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                           ha.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), hb.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                           cap, int(nthreads))
        if n < 0:
            raise MemoryError('kvo_novel_scan_mt_allbands')
        if n <= cap:
            return hr[:n].copy(), ho[:n].copy(), ha[:n].copy(), hb[:n].copy()
        cap = int(n) * 2


def novel_scan_count_mt(cases, ctrls, bases, offs, n_reads, ksize, case_min, ctrl_max, nthreads):
    """number of interesting k-mer instances, the reads split over nthreads threads (timing leg of bench.py)"""
    vp = ctypes.c_void_p
    ca = (vp * len(cases))(*[c._h for c in cases])
    cb = (vp * max(1, len(ctrls)))(*[c._h for c in ctrls])
    return int(lib.kvo_novel_scan_count_mt(ca, len(cases), cb, len(ctrls), bases, offs, n_reads, ksize, case_min,
                                           ctrl_max, int(nthreads)))


def novel_scan(cases, ctrls, bases, offs, n_reads, ksize, case_min, ctrl_max, screen=0,
               band_mode=0, nbands=0, band=0, cap=None):
    """Returns (hits, status): hits = list of (read, offset, abund tuple) in scan order."""
    S = len(cases) + len(ctrls)
    if cap is None:
        cap = 1 << 20
    vp = ctypes.c_void_p
    ca = (vp * len(cases))(*[c._h for c in cases])
    cb = (vp * max(1, len(ctrls)))(*[c._h for c in ctrls])
    hr = (ctypes.c_uint32 * cap)()
    ho = (ctypes.c_uint16 * cap)()
    ha = (ctypes.c_uint8 * (cap * S))()
    st = (ctypes.c_uint8 * max(1, n_reads))()
    n = lib.kvo_novel_scan(ca, len(cases), cb, len(ctrls), bases, offs, n_reads, ksize, case_min,
                           ctrl_max, screen or 0, band_mode, nbands, band, hr, ho, ha, cap, st)
    if n > cap:
        return novel_scan(cases, ctrls, bases, offs, n_reads, ksize, case_min, ctrl_max, screen,
                          band_mode, nbands, band, cap=int(n))
    hits = [(int(hr[i]), int(ho[i]), tuple(ha[i * S:(i + 1) * S])) for i in range(n)]
    return hits, bytes(st[:n_reads])
