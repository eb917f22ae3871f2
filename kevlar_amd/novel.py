"""`kevlar novel` driver (the reference's kevlar/novel.py:21-236).

The reference walks every k-mer of every case read in Python and asks each sketch for its
abundance (novel.py:123-169).  Here the whole test -- hash, band filter, Count-Min lookups in
every case and control sketch, case >= casemin / control <= ctrlmax, abundance screen -- is
one fused kernel over a batch of 2-bit packed reads (kv_novel_scan); the host only formats
the sparse hits back into annotated records, in input order.
"""
import os

import numpy as np

import kevlar_amd
from kevlar_amd import khmer
from kevlar_amd._lib import KV_BAND_NONE, KV_BAND_RANGE, KV_BAND_REFQUIRK

SCAN_BATCH_READS = 1 << 19


class KevlarCaseSampleMismatchError(ValueError):
    pass


def kmer_is_interesting(kmer, casecounts, controlcounts, case_min=5, ctrl_max=1, screen_thresh=None):
    """Single k-mer form of the abundance test (kevlar/novel.py:21-53), kept for API parity;
    the scan itself never calls it.  Returns (interesting, discard_read, case abunds, ctrl abunds)."""
    caseabunds = []
    for ct in casecounts:
        abund = ct.get(kmer)
        if abund < case_min:
            return False, bool(screen_thresh and abund < screen_thresh), [], []
        caseabunds.append(abund)
    ctrlabunds = []
    for ct in controlcounts:
        abund = ct.get(kmer)
        if abund > ctrl_max:
            return False, False, [], []
        ctrlabunds.append(abund)
    return True, False, caseabunds, ctrlabunds


def load_samples(counttables=None, filelists=None, ksize=31, memory=1e6, maxfpr=0.2,
                 numbands=None, band=None, numthreads=1, outfilelist=None):
    assert counttables or filelists
    if counttables:
        message = 'counttables for {:d} sample(s) provided'.format(len(counttables))
        message += ', any corresponding FASTA/FASTQ input will be ignored '
        message += 'for computing k-mer abundances'
        kevlar_amd.plog('[kevlar::novel]    INFO:', message)
        return kevlar_amd.sketch.load_sketchfiles(counttables, maxfpr)
    # (Counting the samples on concurrent streams -- khmer.run_concurrently -- was measured and buys
    # nothing: each count kernel already fills the GPU)
    samples = [
        kevlar_amd.count.load_sample_seqfile(filelist, ksize, memory, maxfpr=maxfpr, numbands=numbands,
                                             band=band, numthreads=numthreads)
        for filelist in filelists
    ]
    if outfilelist:
        save_counts(outfilelist, samples)
    return samples


def save_counts(filelist, tablelist):
    if len(filelist) != len(tablelist):
        msg = 'number of filenames provided ({:d})'.format(len(filelist))
        msg += 'does not match the number of samples provided ({:d})'.format(len(tablelist))
        msg += '; stubbornly refusing to save k-mer counts'
        kevlar_amd.plog('[kevlar::novel] WARNING:', msg)
        return
    for outfile, counttable in zip(filelist, tablelist):
        if not outfile.endswith(('.ct', '.counttable')):
            outfile += '.counttable'
        kevlar_amd.plog('    saved to "{}"'.format(os.path.abspath(outfile)))
        counttable.save(outfile)


class _RecordBatch(object):
    """Adapter giving a list of already-parsed records the interface of khmer.TextBatch."""

    def __init__(self, records, ksize):
        self.records = records
        self.n = len(records)
        self.batch = khmer.ReadBatch([r.sequence if len(r.sequence) >= ksize else '' for r in records])

    def record(self, i):
        return self.records[i]

    def find_name(self, name):
        for j, record in enumerate(self.records):
            if record.name == name:
                return j
        return -1


def _scan_batches(casestream, ksize, sketch_ksize, size):
    """Batches of case reads, packed in HBM.  Streams that can hand over whole parsed batches
    (kevlar_amd.multi_file_iter_khmer) skip per-read Python objects altogether."""
    if hasattr(casestream, 'text_batches') and ksize == sketch_ksize:
        for tb in casestream.text_batches(size):
            yield tb
        return
    chunk = []
    for record in casestream:
        chunk.append(record)
        if len(chunk) >= size:
            yield _RecordBatch(chunk, ksize)
            chunk = []
    if chunk:
        yield _RecordBatch(chunk, ksize)


def novel(casestream, casecounts, controlcounts, ksize=31, abundscreen=None, casemin=5, ctrlmax=0,
          numbands=None, band=None, skipuntil=None, refbandquirk=False, batchsize=SCAN_BATCH_READS):
    """Yield case reads annotated with their interesting k-mers, in input order.

    Banding: by default a band keeps the k-mers whose hash falls in its range -- the rule the
    banded *count* uses (kevlar/count.py:62-66) -- so the union over bands equals the unbanded
    result.  refbandquirk=True applies the reference's literal low-bits test instead
    (kevlar/novel.py:144-147), which is inconsistent with range-banded counts.
    """
    numbands_unset = not numbands
    band_unset = not band and band != 0
    if numbands_unset is not band_unset:
        raise ValueError('Must specify `numbands` and `band` together')
    if band is not None and band < 0:
        message = '`band` must be a value between 0 and {:d}'.format(numbands - 1)
        message += ' (`numbands` - 1), inclusive'
        raise ValueError(message)

    timer = kevlar_amd.Timer()
    timer.start()
    nkmers = nreads = 0
    update_message = '[kevlar::novel]     processed {counter} reads'
    first_message = update_message
    if skipuntil:
        first_message += '; skipping reads in search of {read}'.format(read=skipuntil)
    progress = kevlar_amd.ProgressIndicator(first_message, interval=1e6, breaks=[1e7, 1e8, 1e9], usetimer=True)
    seen_kmers = set()          # as read; canonicalised once at the end (distinct strings, not instances)
    band_mode = KV_BAND_NONE
    if numbands:
        band_mode = KV_BAND_REFQUIRK if refbandquirk else KV_BAND_RANGE
    nseen = 0
    k = casecounts[0].ksize() if casecounts else ksize

    for tb in _scan_batches(casestream, ksize, k, batchsize):
        progress.update(tb.n)
        first_read = 0
        if skipuntil:
            # reads up to and including the named one are skipped (kevlar/novel.py:124-132)
            found = tb.find_name(skipuntil)
            if found < 0:
                nseen += tb.n
                tb.batch.close()
                continue
            message = 'Found read {:s} (skipped {:d} reads)'.format(skipuntil, nseen + found + 1)
            kevlar_amd.plog('[kevlar::novel]', message)
            skipuntil = False
            progress.message = update_message
            first_read = found + 1
        nseen += tb.n
        hitread, hitoff, hitabund, _ = khmer.novel_scan(
            casecounts, controlcounts, tb.batch, casemin, ctrlmax, screen=abundscreen,
            band_mode=band_mode, nbands=numbands or 0, band=band or 0, first_read=first_read)
        tb.batch.close()
        if len(hitread) == 0:
            continue
        bounds = np.flatnonzero(np.diff(hitread)) + 1
        starts = np.concatenate(([0], bounds))
        ends = np.concatenate((bounds, [len(hitread)]))
        # plain Python lists: indexing numpy scalars one by one costs more than the scan itself
        offs, abunds, first = hitoff.tolist(), hitabund.tolist(), hitread[starts].tolist()
        for s, e, ridx in zip(starts.tolist(), ends.tolist(), first):
            record = tb.record(ridx)
            irecord = kevlar_amd.sequence.copy_record(record)
            sequence = record.sequence
            for j in range(s, e):
                offset = offs[j]
                kmer = sequence[offset:offset + k]
                irecord.annotate(kmer, offset, tuple(abunds[j]))
                seen_kmers.add(kmer)
            nreads += 1
            nkmers += e - s
            yield irecord

    elapsed = timer.stop()
    message = 'Found {:d} instances'.format(nkmers)
    unique_kmers = {kevlar_amd.revcommin(kmer) for kmer in seen_kmers}
    message += ' of {:d} unique novel kmers'.format(len(unique_kmers))
    message += ' in {:d} reads'.format(nreads)
    message += ' in {:.2f} seconds'.format(elapsed)
    kevlar_amd.plog('[kevlar::novel]', message)


def main(args):
    timer = kevlar_amd.Timer()
    timer.start()
    if (not args.num_bands) is not (not args.band):
        raise ValueError('Must specify --num-bands and --band together')
    myband = args.band - 1 if args.band else None

    timer.start('loadall')
    kevlar_amd.plog('[kevlar::novel] Loading control samples')
    timer.start('loadctrl')
    controls = load_samples(args.control_counts, args.control, args.ksize, args.memory, args.max_fpr,
                            args.num_bands, myband, args.threads, args.save_ctrl_counts)
    elapsed = timer.stop('loadctrl')
    kevlar_amd.plog('[kevlar::novel]', 'Control samples loaded in {:.2f} sec'.format(elapsed))

    kevlar_amd.plog('[kevlar::novel] Loading case samples')
    timer.start('loadcases')
    cases = load_samples(args.case_counts, args.case, args.ksize, args.memory, args.max_fpr,
                         args.num_bands, myband, args.threads, args.save_case_counts)
    elapsed = timer.stop('loadcases')
    kevlar_amd.plog('[kevlar::novel] Case samples loaded in {:.2f} sec'.format(elapsed))
    elapsed = timer.stop('loadall')
    kevlar_amd.plog('[kevlar::novel] All samples loaded in {:.2f} sec'.format(elapsed))

    timer.start('iter')
    message = 'Iterating over reads from {:d} case sample(s)'.format(len(args.case))
    kevlar_amd.plog('[kevlar::novel]', message)
    outstream = kevlar_amd.open(args.out, 'w')
    infiles = [f for filelist in args.case for f in filelist]
    caserecords = kevlar_amd.multi_file_iter_khmer(infiles)
    readstream = novel(caserecords, cases, controls, ksize=args.ksize, abundscreen=args.abund_screen,
                       casemin=args.case_min, ctrlmax=args.ctrl_max, numbands=args.num_bands, band=myband,
                       skipuntil=args.skip_until, refbandquirk=getattr(args, 'ref_band_quirk', False))
    for augmented_read in readstream:
        kevlar_amd.print_augmented_fastx(augmented_read, outstream)
    elapsed = timer.stop('iter')
    kevlar_amd.plog('[kevlar::novel]', 'Iterated over all case reads in {:.2f} seconds'.format(elapsed))
    total = timer.stop()
    kevlar_amd.plog('[kevlar::novel]', 'Total time: {:.2f} seconds'.format(total))
