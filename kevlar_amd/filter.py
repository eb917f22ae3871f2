"""`kevlar filter`: second opinion on the interesting k-mers of `novel` (kevlar/filter.py:15-107).  Every annotated
k-mer occurrence that the mask (reference genome, contaminants) does not contain is counted again into a small fresh
Counttable -- one whose false positive rate is far below that of the big per-sample sketches -- and annotations whose
recount falls below the case threshold, or whose control abundances exceed theirs, are dropped; reads left without
annotations disappear.

The annotated stream is held as an AnnotatedReads (kevlar_amd/annotated.py): k-mers are hashed on the device from
their (read, offset) positions, masked / added / looked up as whole arrays, and the thresholds are numpy comparisons."""
import numpy as np

import kevlar_amd
from kevlar_amd import khmer
from kevlar_amd.annotated import AnnotatedReads
from kevlar_amd.sketch import KevlarUnsuitableFPRError, estimate_fpr

_TICK = '[kevlar::filter]     processed {counter} reads'


class Recount(object):
    """The two passes over one annotated stream (a file name, or an iterable of records).  After recount(): `table`
    holds the fresh counts (None if the stream carries no annotation), `nreads` the reads seen; survivors() yields what
    passes the thresholds, survivors_text() the same as augmented FASTQ text."""

    def __init__(self, readstream, mask=None, memory=1e6):
        self.mask, self.memory = mask, memory
        self.nreads = 0
        self._stream = readstream
        self.annotated = self.table = self._hashes = None

    def _load(self):
        ticker = kevlar_amd.ProgressIndicator(_TICK, interval=1e5, breaks=[1e6, 1e7])
        if isinstance(self._stream, str):
            self.annotated = AnnotatedReads.from_file(self._stream)      # parsed natively, no object per record
            self.nreads = self.annotated.n
            ticker.update(self.nreads)
            return

        def counted(stream):
            for read in stream:
                self.nreads += 1
                ticker.update()
                yield read
        self.annotated = AnnotatedReads(counted(self._stream))

    def recount(self):
        self._load()
        if not len(self.annotated):
            return
        self.table = khmer.Counttable(self.annotated.ksize, self.memory / 4, 4)
        self._hashes = self.annotated.hashes(self.table)
        fresh = self._hashes
        if self.mask:
            # a mask of another hash family (a *graph sketch) has to hash the k-mers itself
            seen_by_mask = self.mask.get_hashes(fresh if self.mask._kind < 3 else self.annotated.hashes(self.mask))
            fresh = fresh[seen_by_mask == 0]
        self.table.add_hashes(fresh)

    def _verdict(self, casemin, ctrlmax):
        again = self.table.get_hashes(self._hashes).astype(np.int64)
        verdict = again >= casemin
        if self.annotated.nsamples > 1:
            verdict &= (self.annotated.abund[:, 1:] <= ctrlmax).all(axis=1)
        return verdict, again

    def survivors(self, casemin, ctrlmax):
        """reads that keep an annotation with recount >= casemin and every control abundance <= ctrlmax; the kept
        annotations carry the recount as their case abundance"""
        verdict, again = self._verdict(casemin, ctrlmax)
        ticker = kevlar_amd.ProgressIndicator(_TICK, interval=1e5, breaks=[1e6, 1e7])
        for read in self.annotated.select(verdict, case_abund=again):
            ticker.update()
            yield read

    def survivors_text(self, casemin, ctrlmax, sink=None):
        """survivors() as (augmented FASTA/FASTQ bytes, number of reads), formatted natively from the arrays; with a `sink`
        (kevlar_amd.open_sink) the text is written there as it is rendered and (b'', number of reads) comes back"""
        verdict, again = self._verdict(casemin, ctrlmax)
        if sink is not None:
            text, n = b'', self.annotated.select_to(sink, verdict, case_abund=again)
        else:
            text, n = self.annotated.select_text(verdict, case_abund=again)
        kevlar_amd.ProgressIndicator(_TICK, interval=1e5, breaks=[1e6, 1e7]).update(n)
        return text, n


def _passes(readfile, mask, memory, maxfpr, casemin, ctrlmax, as_text, sink=None):
    timer = kevlar_amd.Timer()
    timer.start()
    if isinstance(readfile, str) and readfile != '-':
        stream = readfile                       # a file: parsed natively
    elif isinstance(readfile, str) or readfile is None:
        stream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(readfile, 'r'))      # standard input
    else:
        stream = readfile                       # records handed over by the caller
    work = Recount(stream, mask, memory)

    kevlar_amd.plog('[kevlar::filter] First pass: re-counting k-mers')
    timer.start('firstpass')
    work.recount()
    kevlar_amd.plog('[kevlar::filter]', 'First pass complete! Processed {:d} reads in {:.2f} seconds!'.format(
        work.nreads, timer.stop('firstpass')))

    if work.table is not None:
        fpr = estimate_fpr(work.table)
        verdict = 'FPR for re-computed k-mer counts: {:1.3f}'.format(fpr)
        kevlar_amd.plog('[kevlar::filter]', verdict)
        if fpr > maxfpr:
            raise KevlarUnsuitableFPRError(verdict + 'FPR too high, bailing out!!!')
        kevlar_amd.plog('[kevlar::filter] Second pass: discarding k-mers/reads')
        timer.start('secondpass')
        nkept = 0
        if as_text:
            text, nkept = work.survivors_text(casemin, ctrlmax, sink)
            yield text
        else:
            for nkept, read in enumerate(work.survivors(casemin, ctrlmax), 1):
                yield read
        kevlar_amd.plog('[kevlar::filter]', 'Second pass complete! Validated {:d} reads in {:.2f} seconds!'.format(
            nkept, timer.stop('secondpass')))
    work.annotated.close()
    kevlar_amd.plog('[kevlar::filter]', 'Total time: {:.2f} seconds'.format(timer.stop()))


def filter(readfile, mask=None, memory=1e6, maxfpr=0.01, casemin=6, ctrlmax=1):
    """Generator over the validated reads of the augmented FASTQ `readfile` (a file name as in the reference, '-', or an
    iterable of records)."""
    yield from _passes(readfile, mask, memory, maxfpr, casemin, ctrlmax, as_text=False)


def main(args):
    sink = kevlar_amd.open_sink(args.out)
    mask = kevlar_amd.sketch.load(args.mask) if args.mask else None
    for text in _passes(args.augfastq, mask, args.memory, args.max_fpr, args.case_min, args.ctrl_max, as_text=True, sink=sink):
        if text:
            sink.write(text)
    sink.close()
