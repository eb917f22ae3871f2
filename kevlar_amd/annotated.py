"""AnnotatedReads: the reads of an augmented FASTA/FASTQ stream with their interesting-k-mer annotations as flat
arrays -- record text as blobs with offsets; per annotation the read index, offset and abundances -- next to the
reads themselves 2-bit packed in HBM.

`filter` and `partition` both start from such a stream (kevlar/filter.py:15-82, kevlar/readgraph.py:43-84).  The
reference builds a Python object per record and per annotation, slices every annotated k-mer out of its read as a
string and calls the sketch (or a dictionary) once per k-mer.  Here a file is parsed natively straight into the
arrays (kv_augfastx_load), k-mers are addressed by position -- the device hashes them from the packed reads
(kv_hash_positions) -- everything per annotation is numpy arithmetic over whole arrays, and what is written back is
formatted natively from the same arrays (kv_format_records).  Record objects exist only for callers that ask for
them one by one."""
import ctypes
import os

import numpy as np

from kevlar_amd import _lib, khmer
from kevlar_amd.sequence import KmerOfInterest, Record


def _host_cores():
    """cores this process may use: the affinity mask, cut to the cgroup's CPU quota (a container on a 256-core box gets 16)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def _bytes_at(ptr, n):
    """ctypes.string_at for any size (its length argument is a C int: 2 GB of annotated reads overflow it)"""
    n = int(n)
    if n < (1 << 31) - 1:
        return ctypes.string_at(ptr, n)
    address = ptr.value if hasattr(ptr, 'value') else int(ptr)
    return bytes((ctypes.c_ubyte * n).from_address(address))


def _offsets(lengths):
    out = np.zeros(len(lengths) + 1, dtype=np.uint64)
    if len(lengths):
        np.cumsum(np.asarray(lengths, dtype=np.uint64), out=out[1:])
    return out


class AnnotatedReads(object):
    """names / seqs / quals: bytes blobs, record i at blob[offs[i]:offs[i + 1]]; annotations of record i are entries
    first[i] .. first[i + 1] of offset[], abund[, S]; read[] names the owner of every annotation."""

    def __init__(self, records=()):
        records = [r for r in records if r is not None]
        self.n = len(records)
        self.names = ''.join(r.name for r in records).encode('latin-1')
        self.name_offs = _offsets([len(r.name) for r in records])
        self.seqs = ''.join(r.sequence for r in records).encode('latin-1')
        self.seq_offs = _offsets([len(r.sequence) for r in records])
        self.quals = ''.join(r.quality or '' for r in records).encode('latin-1')
        self.qual_offs = _offsets([len(r.quality or '') for r in records])
        self.is_fastq = np.fromiter((r.quality is not None for r in records), dtype=np.uint8, count=self.n)
        per_read = np.fromiter((len(r.annotations) for r in records), dtype=np.int64, count=self.n)
        self.first = _offsets(per_read)
        total = int(self.first[-1])
        self.offset = np.fromiter((k.offset for r in records for k in r.annotations), dtype=np.uint32, count=total)
        ksizes = {k.ksize for r in records for k in r.annotations}
        if len(ksizes) > 1:
            raise ValueError('all interesting k-mers of one stream must share k (found {})'.format(sorted(ksizes)))
        self.ksize = ksizes.pop() if ksizes else None
        self.nsamples = next((len(k.abund) for r in records for k in r.annotations), 0)
        self.abund = np.fromiter((a for r in records for k in r.annotations for a in k.abund), dtype=np.int32,
                                 count=total * self.nsamples).reshape(total, self.nsamples) if total else np.zeros((0, 0), dtype=np.int32)
        mates = [(i, m) for i, r in enumerate(records) for m in r.mates]
        self.mate_record = np.array([i for i, _ in mates], dtype=np.uint32)
        self.mates = ''.join(m for _, m in mates).encode('latin-1')
        self.mate_offs = _offsets([len(m) for _, m in mates])
        self._finish()

    @classmethod
    def from_file(cls, filename):
        """Parse an augmented FASTA/FASTQ file (plain or gzip) natively; no Python object per record."""
        lib = _lib.load()
        handle = ctypes.c_void_p()
        _lib.check(lib.kv_augfastx_load(filename.encode(), ctypes.byref(handle)))
        try:
            n, na, nm = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
            k, S = ctypes.c_int(), ctypes.c_int()
            _lib.check(lib.kv_augfastx_info(handle, ctypes.byref(n), ctypes.byref(na), ctypes.byref(k), ctypes.byref(S), ctypes.byref(nm)))
            if k.value < 0:
                raise ValueError('all interesting k-mers of one stream must share k')
            ptr = [ctypes.c_void_p() for _ in range(13)]
            _lib.check(lib.kv_augfastx_view(handle, *[ctypes.byref(p) for p in ptr]))
            n, na, nm, S = n.value, na.value, nm.value, S.value

            def arr(p, count, dtype):
                if count == 0:
                    return np.zeros(0, dtype=dtype)
                # (one copy out of the handle's memory, not a bytes object first and an array from that)
                window = (ctypes.c_ubyte * (count * np.dtype(dtype).itemsize)).from_address(p.value)
                return np.frombuffer(window, dtype=dtype).copy()
            self = cls.__new__(cls)
            self.n = n
            self.name_offs, self.seq_offs, self.qual_offs = (arr(ptr[i], n + 1, np.uint64) for i in (1, 3, 5))
            self.names = _bytes_at(ptr[0], int(self.name_offs[n]))
            self.seqs = _bytes_at(ptr[2], int(self.seq_offs[n]))
            self.quals = _bytes_at(ptr[4], int(self.qual_offs[n]))
            self.is_fastq = arr(ptr[6], n, np.uint8)
            self.first = arr(ptr[7], n + 1, np.uint64)
            self.offset = arr(ptr[8], na, np.uint32)
            self.abund = arr(ptr[9], na * S, np.int32).reshape(na, S) if na else np.zeros((0, 0), dtype=np.int32)
            self.mate_record = arr(ptr[10], nm, np.uint32)
            self.mate_offs = arr(ptr[12], nm + 1, np.uint64)
            self.mates = _bytes_at(ptr[11], int(self.mate_offs[nm])) if nm else b''
            self.ksize = k.value if na else None
            self.nsamples = S if na else 0
        finally:
            lib.kv_augfastx_free(handle)
        self._finish()
        return self

    @classmethod
    def concat(cls, parts):
        """the records of several containers, one after the other"""
        parts = list(parts)
        ksizes = {p.ksize for p in parts if p.ksize is not None}
        if len(ksizes) > 1:
            raise ValueError('all interesting k-mers of one stream must share k (found {})'.format(sorted(ksizes)))
        self = cls.__new__(cls)
        self.n = sum(p.n for p in parts)

        def blobs(attr, offs):
            blob = b''.join(getattr(p, attr) for p in parts)
            shifted, base = [np.zeros(1, dtype=np.uint64)], 0
            for p in parts:
                o = getattr(p, offs)
                shifted.append(o[1:] + np.uint64(base))
                base += int(o[-1])
            return blob, np.concatenate(shifted)
        self.names, self.name_offs = blobs('names', 'name_offs')
        self.seqs, self.seq_offs = blobs('seqs', 'seq_offs')
        self.quals, self.qual_offs = blobs('quals', 'qual_offs')
        self.mates, self.mate_offs = blobs('mates', 'mate_offs')
        self.is_fastq = np.concatenate([p.is_fastq for p in parts]) if parts else np.zeros(0, dtype=np.uint8)
        first, base, rec_base, mate_record = [np.zeros(1, dtype=np.int64)], 0, 0, []
        for p in parts:
            first.append(p.first[1:] + base)
            base += int(p.first[-1])
            mate_record.append(p.mate_record + np.uint32(rec_base))
            rec_base += p.n
        self.first = np.concatenate(first)
        self.mate_record = np.concatenate(mate_record) if parts else np.zeros(0, dtype=np.uint32)
        self.offset = np.concatenate([p.offset for p in parts]) if parts else np.zeros(0, dtype=np.uint32)
        self.ksize = ksizes.pop() if ksizes else None
        self.nsamples = max([p.nsamples for p in parts] or [0])
        with_ann = [p.abund for p in parts if len(p)]
        self.abund = np.concatenate(with_ann) if with_ann else np.zeros((0, 0), dtype=np.int32)
        self._finish()
        return self

    def _finish(self):
        self.first = self.first.astype(np.int64)
        self.read = np.repeat(np.arange(self.n, dtype=np.uint32), np.diff(self.first))
        self._batch = None
        self._made = {}

    def __len__(self):
        return int(self.first[-1])

    # ---- record objects, on request ------------------------------------------------------------
    def _text(self, blob, offs, i):
        return blob[int(offs[i]):int(offs[i + 1])].decode('latin-1')

    def name(self, i):
        return self._text(self.names, self.name_offs, i)

    def sequence(self, i):
        return self._text(self.seqs, self.seq_offs, i)

    def record(self, i, keep=None, case_abund=None):
        """Record i with its annotations (only those where keep[...] is set; case_abund replaces the first abundance)"""
        lo, hi = int(self.first[i]), int(self.first[i + 1])
        notes = []
        for j in range(lo, hi):
            if keep is not None and not keep[j]:
                continue
            abund = tuple(int(a) for a in self.abund[j])
            if case_abund is not None:
                abund = (int(case_abund[j]),) + abund[1:]
            notes.append(KmerOfInterest(self.ksize, int(self.offset[j]), abund))
        mates = []
        if len(self.mate_record):
            for m in np.flatnonzero(self.mate_record == i).tolist():
                mates.append(self._text(self.mates, self.mate_offs, m))
        return Record(self.name(i), self.sequence(i), self._text(self.quals, self.qual_offs, i) if self.is_fastq[i] else None,
                      annotations=notes, mates=mates)

    @property
    def records(self):
        """every record as an object (built once)"""
        if 'all' not in self._made:
            self._made['all'] = [self.record(i) for i in range(self.n)]
        return self._made['all']

    # ---- device side -----------------------------------------------------------------------------
    @property
    def batch(self):
        """the read sequences packed in HBM (uploaded on first use)"""
        if self._batch is None:
            self._batch = khmer.ReadBatch.from_blob(self.seqs, self.seq_offs)
        return self._batch

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None

    def odd_reads(self):
        """indices of the reads with characters outside ACGT (the packed form cannot hold them)"""
        if not self.seqs:
            return np.zeros(0, dtype=np.int64)
        if 'odd' not in self._made:             # (a table look-up per byte in numpy took 3 ns a byte: seconds for a few million reads)
            flags = np.zeros(self.n, dtype=np.uint8)
            offs = np.ascontiguousarray(self.seq_offs, dtype=np.uint64)
            _lib.check(_lib.load().kv_reads_flag_other_bytes(
                ctypes.cast(ctypes.c_char_p(self.seqs), ctypes.c_void_p), offs.ctypes.data_as(ctypes.c_void_p), self.n, flags.ctypes.data_as(ctypes.c_void_p)))
            self._made['odd'] = np.flatnonzero(flags)
        return self._made['odd']

    def hashes(self, sketch):
        """`sketch`'s hash of every annotated k-mer, in stream order.  K-mers of reads with characters outside ACGT
        are hashed from their text instead."""
        if not len(self):
            return np.zeros(0, dtype=np.uint64)
        out = sketch.hash_positions(self.batch, self.read, self.offset)
        for i in self.odd_reads().tolist():
            lo, hi = int(self.first[i]), int(self.first[i + 1])
            if hi > lo:
                seq = self.sequence(i)
                out[lo:hi] = sketch.hash_kmers([seq[o:o + self.ksize] for o in self.offset[lo:hi].tolist()])
        return out

    # ---- selections ------------------------------------------------------------------------------
    def _kept_reads(self, keep):
        if not len(self):
            return np.zeros(0, dtype=np.int64)
        csum = np.concatenate(([0], np.cumsum(keep.astype(np.int64))))
        return np.flatnonzero(csum[self.first[1:]] - csum[self.first[:-1]])

    def select(self, keep, case_abund=None):
        """Records that still have an annotation where `keep` (boolean per annotation) is set, each with only those
        annotations; case_abund (per annotation) replaces the first abundance.  Generator, stream order."""
        for i in self._kept_reads(keep).tolist():
            yield self.record(i, keep, case_abund)

    def format_to(self, sink, reads, keep=None, case_abund=None, suffixes=None, regrouped=None, suffix_blob=None):
        """format() written to `sink` (kevlar_amd.open_sink): when the sink is a plain file the native formatter renders on several
        threads and writes the file itself (kv_format_records_fd: no text buffer of the output's size, no copy into a Python
        object); otherwise format() + write()."""
        fd = sink.raw_fd() if hasattr(sink, 'raw_fd') else None
        if fd is None or _lib.knob('KV_FORMAT_FD') == '0':
            sink.write(self.format(reads, keep, case_abund, suffixes, regrouped, suffix_blob))
            return
        self.format(reads, keep, case_abund, suffixes, regrouped, suffix_blob, _fd=fd)

    def format(self, reads, keep=None, case_abund=None, suffixes=None, regrouped=None, suffix_blob=None, _fd=None):
        """Augmented FASTA/FASTQ text (bytes) of the given reads (indices, in that order) with the annotations where
        `keep` is set (None: all); suffixes: one string per read appended to its name (suffix_blob: the same as one blob + offsets).  regrouped = (lo, hi, order):
        output read j carries annotations order[lo[j]:hi[j]] (indices into this container's annotations) instead of
        its own -- the union over several copies of a read (unband)."""
        reads = np.ascontiguousarray(reads, dtype=np.uint64)
        if not len(reads):
            return b''
        idx = reads.astype(np.int64)
        abund = np.ascontiguousarray(self.abund, dtype=np.int32)
        offset = np.ascontiguousarray(self.offset, dtype=np.uint32)
        if regrouped is None:
            lo = np.ascontiguousarray(self.first[idx], dtype=np.uint64)
            hi = np.ascontiguousarray(self.first[idx + 1], dtype=np.uint64)
        else:
            lo, hi, order = regrouped
            lo, hi = np.ascontiguousarray(lo, dtype=np.uint64), np.ascontiguousarray(hi, dtype=np.uint64)
            abund, offset = np.ascontiguousarray(abund[order]), np.ascontiguousarray(offset[order])
        keep8 = None if keep is None else np.ascontiguousarray(keep, dtype=np.uint8)
        case32 = None if case_abund is None else np.ascontiguousarray(case_abund, dtype=np.int32)
        sfx_blob = sfx_offs = None
        if suffixes is not None:
            sfx_blob = ''.join(suffixes).encode('latin-1')
            sfx_offs = _offsets([len(s) for s in suffixes])
        elif suffix_blob is not None:           # the same, already joined: (bytes, offsets[len(reads) + 1])
            sfx_blob, sfx_offs = suffix_blob[0], np.ascontiguousarray(suffix_blob[1], dtype=np.uint64)

        def ptr(a):
            return None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        text, size = ctypes.c_void_p(), ctypes.c_uint64()
        lib = _lib.load()
        common = (
            len(reads), ptr(reads), ptr(lo), ptr(hi), ptr(offset), ptr(abund), ptr(keep8), ptr(case32), int(self.nsamples), int(self.ksize or 1),
            ctypes.cast(ctypes.c_char_p(self.names), ctypes.c_void_p), ptr(self.name_offs), ctypes.cast(ctypes.c_char_p(self.seqs), ctypes.c_void_p),
            ptr(self.seq_offs), ctypes.cast(ctypes.c_char_p(self.quals), ctypes.c_void_p), ptr(self.qual_offs), ptr(self.is_fastq),
            None if sfx_blob is None else ctypes.cast(ctypes.c_char_p(sfx_blob), ctypes.c_void_p), ptr(sfx_offs),
            ptr(self.mate_record) if len(self.mate_record) else None, len(self.mate_record),
            ctypes.cast(ctypes.c_char_p(self.mates), ctypes.c_void_p) if len(self.mate_record) else None,
            ptr(self.mate_offs) if len(self.mate_record) else None)
        if _fd is not None:
            threads = int(_lib.knob('KV_FORMAT_THREADS', '0')) or min(16, _host_cores())
            _lib.check(lib.kv_format_records_fd(*(common + (int(_fd), threads, ctypes.byref(size)))))
            return size.value
        _lib.check(lib.kv_format_records(*(common + (ctypes.byref(text), ctypes.byref(size)))))
        try:
            return _bytes_at(text, size.value)
        finally:
            lib.kv_text_free(text)

    def select_text(self, keep, case_abund=None):
        """select() rendered to augmented FASTA/FASTQ text; returns (bytes, number of reads)"""
        reads = self._kept_reads(keep)
        return self.format(reads, keep, case_abund), len(reads)

    def select_to(self, sink, keep, case_abund=None):
        """select_text() written to `sink` (format_to); returns the number of reads"""
        reads = self._kept_reads(keep)
        self.format_to(sink, reads, keep, case_abund)
        return len(reads)
