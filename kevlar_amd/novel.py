"""`kevlar novel`: the reads of the case sample that carry k-mers abundant in the case and (nearly) absent from
every control, annotated with those k-mers (kevlar/novel.py:21-236).

The reference walks every k-mer of every case read in Python and asks each sketch for its abundance
(novel.py:123-169).  Here a batch of reads is parsed, 2-bit packed and uploaded by the native reader, the whole test
-- hash, band filter, Count-Min lookups in every case and control sketch, thresholds, abundance screen -- is one
device scan (kv_novel_scan) that returns the sparse hits sorted by (read, offset), and the host turns runs of hits
into annotated records, in input order, touching only the reads that have any."""
import os

import numpy as np

import kevlar_amd
from kevlar_amd import _lib, khmer
from kevlar_amd._lib import KV_BAND_NONE, KV_BAND_RANGE, KV_BAND_REFQUIRK

SCAN_BATCH_READS = 1 << 23


class KevlarCaseSampleMismatchError(ValueError):
    pass


def kmer_is_interesting(kmer, casecounts, controlcounts, case_min=5, ctrl_max=1, screen_thresh=None):
    """The abundance test for ONE k-mer, kept for API parity (kevlar/novel.py:21-53); the scan never calls it.
    (interesting, discard the read, case abundances, control abundances): a case abundance under case_min ends the
    test (and, under screen_thresh, condemns the read); so does a control abundance over ctrl_max."""
    observed = []
    for i, sketch in enumerate(list(casecounts) + list(controlcounts)):
        count = sketch.get(kmer)
        is_case = i < len(casecounts)
        if is_case and count < case_min:
            return False, bool(screen_thresh and count < screen_thresh), [], []
        if not is_case and count > ctrl_max:
            return False, False, [], []
        observed.append(count)
    return True, False, observed[:len(casecounts)], observed[len(casecounts):]


def save_counts(filelist, tablelist, log=None):
    log = log or kevlar_amd.plog
    if len(filelist) != len(tablelist):
        log('[kevlar::novel] WARNING:', 'number of filenames provided ({:d})does not match the number of samples '
                        'provided ({:d}); stubbornly refusing to save k-mer counts'.format(len(filelist), len(tablelist)))
        return
    for path, sketch in zip(filelist, tablelist):
        path += '' if path.endswith(('.ct', '.counttable')) else '.counttable'
        log('    saved to "{}"'.format(os.path.abspath(path)))
        sketch.save(path)


def _side_by_side(filelists=None):
    """Samples counted side by side (each on its own HIP stream and with its own scratch buffers) or one after the other?
    KV_PARALLEL_SAMPLES=1 / 0 says; otherwise side by side while the input is small enough (under 3 GB of files: reading
    one file then overlaps inflating and counting another), one after the other beyond (the first allocation of three
    sets of multi-gigabyte scratch buffers costs a cold one-shot run more than the overlap gains)."""
    env = _lib.knob('KV_PARALLEL_SAMPLES')
    if env is not None and env != '':
        return env not in ('0', 'no', 'false')
    if not filelists:
        return False
    try:
        total = sum(os.path.getsize(path) for files in filelists for path in files if isinstance(path, str))
    except OSError:
        return False
    return 0 < total < 3e9


def load_samples(counttables=None, filelists=None, ksize=31, memory=1e6, maxfpr=0.2, numbands=None, band=None, numthreads=1,
                 outfilelist=None, log=None, keep=None):
    """One sketch per sample: loaded from saved count tables if given, else counted from the sample's files."""
    assert counttables or filelists
    log = log or kevlar_amd.plog
    if counttables:
        log('[kevlar::novel]    INFO:', 'counttables for {:d} sample(s) provided, any corresponding FASTA/FASTQ input '
                        'will be ignored for computing k-mer abundances'.format(len(counttables)))
        return kevlar_amd.sketch.load_sketchfiles(counttables, maxfpr)
    # KV_PARALLEL_SAMPLES=1: the samples are counted side by side, each on its own HIP stream (reading one file overlaps with
    # counting another); what each has to say is held back and printed sample by sample, as a one-after-the-other run prints
    # it.  Not the default: every stream brings its own scratch buffers, and in a process that counts each sample once their
    # first allocation (gigabytes at config 2) costs more than the overlap gains (7.5 M reads per sample, cold: 1.09 s side by
    # side against 0.76 s one after the other; 2 M reads, buffers warm: 0.115 against 0.134 s).
    if not _side_by_side(filelists):
        sketches = [kevlar_amd.count.load_sample_seqfile(files, ksize, memory, maxfpr=maxfpr, numbands=numbands, band=band,
                                                         numthreads=numthreads, log=log, keep=keep) for files in filelists]
        if outfilelist:
            save_counts(outfilelist, sketches, log)
        return sketches
    said = [[] for _ in filelists]

    def one(i):
        return kevlar_amd.count.load_sample_seqfile(filelists[i], ksize, memory, maxfpr=maxfpr, numbands=numbands, band=band,
                                                    numthreads=numthreads, log=lambda *words: said[i].append(words), keep=keep)
    try:
        sketches = khmer.run_concurrently([lambda i=i: one(i) for i in range(len(filelists))])
    finally:
        for lines in said:
            for words in lines:
                log(*words)
    if outfilelist:
        save_counts(outfilelist, sketches, log)
    return sketches


class _Parsed(object):
    """records that arrived one by one (a Python iterable instead of the native reader), packed like a TextBatch"""

    def __init__(self, records, ksize):
        self.records, self.n = records, len(records)
        self.batch = khmer.ReadBatch([r.sequence if len(r.sequence) >= ksize else '' for r in records])

    def record(self, i):
        return self.records[i]

    def prefetch(self, indices):
        pass

    def find_name(self, name):
        return next((i for i, r in enumerate(self.records) if r.name == name), -1)


def _batches(stream, ksize, sketch_k, size):
    if hasattr(stream, 'text_batches') and ksize == sketch_k:        # the native reader hands over whole batches
        yield from stream.text_batches(size)
        return
    held = []
    for record in stream:
        held.append(record)
        if len(held) == size:
            yield _Parsed(held, ksize)
            held = []
    if held:
        yield _Parsed(held, ksize)


class _Tally(object):
    """what the closing log line reports"""

    def __init__(self, hasher=None):
        self.instances = self.reads = 0
        self.kmers = set()          # as read; reduced to one strand at the end
        self.hashes = []            # ... or, on the text path, the k-mers' hashes (one value for both strands)
        self.hasher = hasher        # the sketch whose hash function the text path used

    def line(self, seconds):
        """`unique novel kmers` counts canonical k-mers (kevlar/novel.py:161-162).  One run can tally through both paths
        (a batch of parsed records next to native batches): the k-mers then meet in ONE set -- the strings are hashed
        with the same strand-symmetric function -- so that a k-mer seen on both paths counts once."""
        canonical = {kevlar_amd.revcommin(kmer) for kmer in self.kmers}
        if not self.hashes:
            unique = len(canonical)
        else:
            parts = list(self.hashes)
            if canonical:
                parts.append(self.hasher.hash_kmers(sorted(canonical)))
            unique = len(np.unique(np.concatenate(parts)))
        return 'Found {:d} instances of {:d} unique novel kmers in {:d} reads in {:.2f} seconds'.format(
            self.instances, unique, self.reads, seconds)


def _annotate(text, hits, k, tally):
    """records for the runs of hits (read, offset, abundances; sorted by read) of one batch"""
    reads, offsets, abunds, dropped = hits
    text.prefetch(np.concatenate((np.unique(reads), np.unique(dropped.shadow[0]))) if len(dropped) else np.unique(reads))
    if len(dropped):
        # the reference tallies the interesting k-mers in front of the k-mer that tripped the screen, then drops the read
        for ridx, off in zip(*(a.tolist() for a in dropped.shadow)):
            tally.kmers.add(text.record(ridx).sequence[off:off + k])
    if not len(reads):
        return
    cuts = np.flatnonzero(np.diff(reads)) + 1
    firsts = np.concatenate(([0], cuts)).tolist()
    lasts = np.concatenate((cuts, [len(reads)])).tolist()
    owner = reads[firsts].tolist()
    offsets, abunds = offsets.tolist(), abunds.tolist()          # Python lists: numpy scalars one by one cost more than the scan
    for lo, hi, ridx in zip(firsts, lasts, owner):
        source = text.record(ridx)
        fresh = kevlar_amd.sequence.copy_record(source)
        for j in range(lo, hi):
            kmer = source.sequence[offsets[j]:offsets[j] + k]
            fresh.annotate(kmer, offsets[j], tuple(abunds[j]))
            tally.kmers.add(kmer)
        tally.reads += 1
        tally.instances += hi - lo
        yield fresh


def _scan(casestream, casecounts, controlcounts, ksize, abundscreen, casemin, ctrlmax, numbands, band, skipuntil, refbandquirk, batchsize):
    """(batch text, hits) of every batch of the case stream; the batch's packed reads are still open"""
    if (not numbands) != (not band and band != 0):
        raise ValueError('Must specify `numbands` and `band` together')
    if band is not None and band < 0:
        raise ValueError('`band` must be a value between 0 and {:d} (`numbands` - 1), inclusive'.format(numbands - 1))
    ticking = '[kevlar::novel]     processed {counter} reads'
    progress = kevlar_amd.ProgressIndicator(ticking + ('; skipping reads in search of {}'.format(skipuntil) if skipuntil else ''),
                                            interval=1e6, breaks=[1e7, 1e8, 1e9], usetimer=True)
    band_mode = (KV_BAND_REFQUIRK if refbandquirk else KV_BAND_RANGE) if numbands else KV_BAND_NONE
    k = casecounts[0].ksize() if casecounts else ksize
    passed = 0
    for text in _batches(casestream, ksize, k, batchsize):
        progress.update(text.n)
        start = 0
        if skipuntil:           # everything up to and including the named read is skipped (kevlar/novel.py:124-132)
            at = text.find_name(skipuntil)
            if at < 0:
                passed += text.n
                text.batch.close()
                continue
            kevlar_amd.plog('[kevlar::novel]', 'Found read {:s} (skipped {:d} reads)'.format(skipuntil, passed + at + 1))
            skipuntil, progress.message, start = None, ticking, at + 1
        passed += text.n
        hits = khmer.novel_scan(casecounts, controlcounts, text.batch, casemin, ctrlmax, screen=abundscreen, band_mode=band_mode,
                                nbands=numbands or 0, band=band or 0, first_read=start)
        yield text, hits, k


def novel(casestream, casecounts, controlcounts, ksize=31, abundscreen=None, casemin=5, ctrlmax=0, numbands=None, band=None,
          skipuntil=None, refbandquirk=False, batchsize=SCAN_BATCH_READS):
    """Yield the case reads that hold interesting k-mers, annotated, in input order.

    Banding: a band keeps the k-mers whose hash falls in its range -- the rule the banded *count* uses
    (kevlar/count.py:62-66) -- so the union over bands equals the unbanded result.  refbandquirk=True applies the
    reference's literal low-bits test instead (kevlar/novel.py:144-147), which is inconsistent with range-banded
    counts."""
    clock = kevlar_amd.Timer()
    clock.start()
    tally = _Tally()
    for text, hits, k in _scan(casestream, casecounts, controlcounts, ksize, abundscreen, casemin, ctrlmax, numbands, band, skipuntil,
                               refbandquirk, batchsize):
        text.batch.close()
        yield from _annotate(text, hits, k, tally)
    kevlar_amd.plog('[kevlar::novel]', tally.line(clock.stop()))


def novel_text(casestream, casecounts, controlcounts, ksize=31, abundscreen=None, casemin=5, ctrlmax=0, numbands=None, band=None,
               skipuntil=None, refbandquirk=False, batchsize=SCAN_BATCH_READS):
    """novel() for a writer: yields the augmented FASTA/FASTQ text (bytes) of each batch's annotated reads -- the same
    bytes print_augmented_fastx would produce record by record -- formatted natively where the batch has its text in
    one piece (every batch of the native reader), so no Python object is built per read or per k-mer."""
    clock = kevlar_amd.Timer()
    clock.start()
    tally = _Tally(casecounts[0] if casecounts else None)
    for text, hits, k in _scan(casestream, casecounts, controlcounts, ksize, abundscreen, casemin, ctrlmax, numbands, band, skipuntil,
                               refbandquirk, batchsize):
        reads, offsets, _, dropped = hits
        blob = text.augmented_text(hits, k) if hasattr(text, 'augmented_text') and hasattr(casecounts[0], 'hash_positions') else None
        if blob is None:
            text.batch.close()
            blob = ''.join(kevlar_amd.sequence.format_augmented_fastx(rec) for rec in _annotate(text, hits, k, tally)).encode('latin-1')
        else:
            if len(reads):
                tally.hashes.append(casecounts[0].hash_positions(text.batch, reads, offsets))
                tally.instances += len(reads)
                tally.reads += int(np.count_nonzero(np.diff(reads))) + 1
            if len(dropped) and len(dropped.shadow[0]):
                tally.hashes.append(casecounts[0].hash_positions(text.batch, dropped.shadow[0], dropped.shadow[1]))
            text.batch.close()
        if blob:
            yield blob
    kevlar_amd.plog('[kevlar::novel]', tally.line(clock.stop()))


def _load_side_by_side(args, band, keep=None):
    """KV_PARALLEL_SAMPLES=1: controls and cases loaded side by side, each sample on its own HIP stream; what they have to say
    is printed in the order of a one-after-the-other run"""
    said = {'ctrl': [], 'case': []}
    took = {}

    def load(which, counts, files, save):
        watch = kevlar_amd.Timer()
        watch.start()
        sketches = load_samples(counts, files, args.ksize, args.memory, args.max_fpr, args.num_bands, band, args.threads, save,
                                log=lambda *words: said[which].append(words), keep=keep if which == 'case' else None)
        took[which] = watch.stop()
        return sketches
    try:
        return khmer.run_concurrently([lambda: load('ctrl', args.control_counts, args.control, args.save_ctrl_counts),
                                       lambda: load('case', args.case_counts, args.case, args.save_case_counts)])
    finally:
        kevlar_amd.plog('[kevlar::novel] Loading control samples')
        for words in said['ctrl']:
            kevlar_amd.plog(*words)
        if 'ctrl' in took:
            kevlar_amd.plog('[kevlar::novel]', 'Control samples loaded in {:.2f} sec'.format(took['ctrl']))
        kevlar_amd.plog('[kevlar::novel] Loading case samples')
        for words in said['case']:
            kevlar_amd.plog(*words)
        if 'case' in took:
            kevlar_amd.plog('[kevlar::novel] Case samples loaded in {:.2f} sec'.format(took['case']))


def main(args):
    if (not args.num_bands) != (not args.band):
        raise ValueError('Must specify --num-bands and --band together')
    band = args.band - 1 if args.band else None
    clock = kevlar_amd.Timer()
    for key in (None, 'loadall', 'loadctrl'):
        clock.start(key)
    # a case sample that was counted as one batch is scanned from that batch (no second pass over its file)
    kept = {} if not _lib.knob('KV_NOVEL_REREAD') else None
    if not _side_by_side((args.control or []) + (args.case or [])) or args.control_counts or args.case_counts:
        kevlar_amd.plog('[kevlar::novel] Loading control samples')
        controls = load_samples(args.control_counts, args.control, args.ksize, args.memory, args.max_fpr, args.num_bands, band,
                                args.threads, args.save_ctrl_counts)
        kevlar_amd.plog('[kevlar::novel]', 'Control samples loaded in {:.2f} sec'.format(clock.stop('loadctrl')))
        kevlar_amd.plog('[kevlar::novel] Loading case samples')
        clock.start('loadcases')
        cases = load_samples(args.case_counts, args.case, args.ksize, args.memory, args.max_fpr, args.num_bands, band,
                             args.threads, args.save_case_counts, keep=kept)
        kevlar_amd.plog('[kevlar::novel] Case samples loaded in {:.2f} sec'.format(clock.stop('loadcases')))
    else:
        controls, cases = _load_side_by_side(args, band, keep=kept)
        clock.stop('loadctrl')
    kevlar_amd.plog('[kevlar::novel] All samples loaded in {:.2f} sec'.format(clock.stop('loadall')))

    clock.start('iter')
    kevlar_amd.plog('[kevlar::novel]', 'Iterating over reads from {:d} case sample(s)'.format(len(args.case)))
    sink = kevlar_amd.open_sink(args.out)
    case_reads = kevlar_amd.multi_file_iter_khmer([path for files in args.case for path in files], kept=kept)
    for blob in novel_text(case_reads, cases, controls, ksize=args.ksize, abundscreen=args.abund_screen, casemin=args.case_min,
                           ctrlmax=args.ctrl_max, numbands=args.num_bands, band=band, skipuntil=args.skip_until,
                           refbandquirk=getattr(args, 'ref_band_quirk', False)):
        sink.write(blob)
    sink.close()
    kevlar_amd.plog('[kevlar::novel]', 'Iterated over all case reads in {:.2f} seconds'.format(clock.stop('iter')))
    kevlar_amd.plog('[kevlar::novel]', 'Total time: {:.2f} seconds'.format(clock.stop()))
