"""`kevlar filter` driver (the reference's kevlar/filter.py:15-107).

Two passes over an augmented FASTQ: re-count every annotated k-mer that the mask does not
contain into a fresh Counttable, then re-threshold.  The k-mer instances (<= ~1e6) are hashed,
mask-tested, added and queried as device batches (kv_hash_kmers / kv_get_hashes /
kv_add_hashes) instead of one Python->sketch call per k-mer.
"""
import kevlar_amd
from kevlar_amd import khmer
from kevlar_amd.sequence import KmerOfInterest


def first_pass(reads, mask, memory, timer):
    """Returns (counts sketch or None, list of records, per-annotation recount array)."""
    kevlar_amd.plog('[kevlar::filter] First pass: re-counting k-mers')
    timer.start('firstpass')
    progress = kevlar_amd.ProgressIndicator('[kevlar::filter]     processed {counter} reads',
                                            interval=1e5, breaks=[1e6, 1e7])
    records, kmers = [], []
    n = 0
    for n, read in enumerate(reads, 1):
        progress.update()
        if read is None:
            continue
        records.append(read)
        for ikmer in read.annotations:
            kmers.append(read.ikmerseq(ikmer))
    counts, hashes = None, None
    if kmers:
        ksize = len(kmers[0])
        counts = khmer.Counttable(ksize, memory / 4, 4)
        hashes = counts.hash_kmers(kmers)
        keep = hashes
        if mask:
            mhashes = hashes if mask._kind < 3 else mask.hash_kmers(kmers)   # mask hashes with its own function
            keep = hashes[mask.get_hashes(mhashes) == 0]   # `if mask.get(ikseq) > 0: continue`
        counts.add_hashes(keep)
    elapsed = timer.stop('firstpass')
    message = 'First pass complete! Processed {:d} reads in {:.2f} seconds!'.format(n, elapsed)
    kevlar_amd.plog('[kevlar::filter]', message)
    return counts, records, hashes


def check_fpr(counts, maxfpr):
    fpr = kevlar_amd.sketch.estimate_fpr(counts)
    message = 'FPR for re-computed k-mer counts: {:1.3f}'.format(fpr)
    kevlar_amd.plog('[kevlar::filter]', message)
    if fpr > maxfpr:
        message += 'FPR too high, bailing out!!!'
        raise kevlar_amd.sketch.KevlarUnsuitableFPRError(message)


def second_pass(reads, counts, casemin, ctrlmax, timer, hashes=None):
    kevlar_amd.plog('[kevlar::filter] Second pass: discarding k-mers/reads')
    timer.start('secondpass')
    progress = kevlar_amd.ProgressIndicator('[kevlar::filter]     processed {counter} reads',
                                            interval=1e5, breaks=[1e6, 1e7])
    reads = list(reads)
    if hashes is None:
        kmers = [read.ikmerseq(ikmer) for read in reads for ikmer in read.annotations]
        hashes = counts.hash_kmers(kmers)
    recount = counts.get_hashes(hashes) if len(hashes) else []
    kept = 0
    cursor = 0
    for read in reads:
        progress.update()
        validated = []
        for ikmer in read.annotations:
            newcount = int(recount[cursor])
            cursor += 1
            if any(a > ctrlmax for a in ikmer.abund[1:]):
                continue
            if newcount < casemin:
                continue
            newabund = tuple([newcount] + list(ikmer.abund[1:]))
            validated.append(KmerOfInterest(ikmer.ksize, ikmer.offset, newabund))
        if not validated:
            continue
        read.annotations = validated
        yield read
        kept += 1
    elapsed = timer.stop('secondpass')
    message = 'Second pass complete! Validated {:d} reads in {:.2f} seconds!'.format(kept, elapsed)
    kevlar_amd.plog('[kevlar::filter]', message)


def filter(readfile, mask=None, memory=1e6, maxfpr=0.01, casemin=6, ctrlmax=1):
    timer = kevlar_amd.Timer()
    timer.start()
    reader = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(readfile, 'r'))
    counts, records, hashes = first_pass(reader, mask, memory, timer)
    if counts is not None:
        check_fpr(counts, maxfpr)
        for read in second_pass(records, counts, casemin, ctrlmax, timer, hashes=hashes):
            yield read
    total = timer.stop()
    kevlar_amd.plog('[kevlar::filter]', 'Total time: {:.2f} seconds'.format(total))


def main(args):
    mask = kevlar_amd.sketch.load(args.mask) if args.mask else None
    outstream = kevlar_amd.open(args.out, 'w')
    filterstream = filter(args.augfastq, mask=mask, memory=args.memory, maxfpr=args.max_fpr,
                          casemin=args.case_min, ctrlmax=args.ctrl_max)
    for record in filterstream:
        kevlar_amd.print_augmented_fastx(record, outstream)
