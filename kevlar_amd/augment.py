"""`kevlar augment`: carry the interesting-k-mer annotations of one set of sequences over to another -- contigs
assembled from annotated reads, or reads that went through a tool that dropped the annotation lines
(kevlar/augment.py:13-45).

The annotated k-mers (both orientations) become a sorted table of 2-bit packed integers; every sequence to be
annotated is packed the same way, all of its windows at once with numpy, and matched against the table with one
binary search -- no per-offset string slicing.  Sequences with characters outside ACGT, and k > 32, fall back to a
regular-expression scan over the literal k-mers."""
import re

import numpy as np

import kevlar_amd
from kevlar_amd.sequence import Record, revcom

_CODES = np.full(256, 255, dtype=np.uint8)
_CODES[np.frombuffer(b'ACGT', dtype=np.uint8)] = np.arange(4, dtype=np.uint8)


def _pack_windows(text, ksize):
    """2 bits per base, first base in the low bits, for every window of `text`; None if that is not possible."""
    if ksize > 32 or len(text) < ksize:
        return None
    codes = _CODES[np.frombuffer(text.encode('ascii', 'replace'), dtype=np.uint8)]
    if codes.max(initial=0) > 3:
        return None
    nwin = len(codes) - ksize + 1
    packed = np.zeros(nwin, dtype=np.uint64)
    for j in range(ksize):
        packed |= codes[j:j + nwin].astype(np.uint64) << np.uint64(2 * j)
    return packed


class KmerBook(object):
    """The k-mers of interest of an annotated stream with their abundances, in both orientations."""

    def __init__(self):
        self.ksize = None
        self.abund = {}
        self._keys = self._texts = self._pattern = None

    def learn(self, record):
        for ikmer in record.annotations:
            text = record.ikmerseq(ikmer)
            self.abund[text] = self.abund[revcom(text)] = ikmer.abund
            self.ksize = ikmer.ksize
        self._keys = None

    def _freeze(self):
        texts = sorted(self.abund)
        packable = [t for t in texts if _pack_windows(t, self.ksize) is not None]
        keys = np.array([int(_pack_windows(t, self.ksize)[0]) for t in packable], dtype=np.uint64)
        order = np.argsort(keys, kind='stable')
        self._keys, self._texts = keys[order], [packable[i] for i in order]
        self._pattern = re.compile('(?=(' + '|'.join(map(re.escape, texts)) + '))') if texts else None

    def occurrences(self, sequence):
        """(offset, k-mer text) for every window of `sequence` that is a k-mer of interest, by offset."""
        if not self.abund:
            return []
        if self._keys is None:
            self._freeze()
        windows = _pack_windows(sequence, self.ksize)
        if windows is None:
            if len(sequence) < self.ksize:
                return []
            return [(m.start(), m.group(1)) for m in self._pattern.finditer(sequence)]
        if len(self._keys) == 0:
            return []
        slot = np.minimum(np.searchsorted(self._keys, windows), len(self._keys) - 1)
        return [(int(i), self._texts[slot[i]]) for i in np.flatnonzero(self._keys[slot] == windows)]


def augment(augseqstream, nakedseqstream, upint=10000):
    """Yield a copy of every record of `nakedseqstream` annotated with the k-mers of interest of `augseqstream`
    (same abundances) wherever they occur in it, on either strand."""
    book = KmerBook()
    for nread, record in enumerate(augseqstream):
        if nread and nread % upint == 0:
            kevlar_amd.plog('[kevlar::augment] processed', nread, 'input reads')
        book.learn(record)
    for bare in nakedseqstream:
        fresh = Record(name=bare.name, sequence=bare.sequence, quality=getattr(bare, 'quality', None))
        for offset, kmer in book.occurrences(bare.sequence):
            fresh.annotate(kmer, offset, book.abund[kmer])
        yield fresh


def main(args):
    annotated = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.augseqs, 'r'))
    bare = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.seqs, 'r'))
    sink = kevlar_amd.open(args.out, 'w')
    for record in augment(annotated, bare):
        kevlar_amd.print_augmented_fastx(record, sink)
    sink.flush()
