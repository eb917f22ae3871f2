"""`kevlar count` driver (the reference's kevlar/count.py:18-140) over the HIP sketch engine."""
import threading

import kevlar_amd
from kevlar_amd import khmer
from kevlar_amd.sketch import allocate, get_extension


def load_sample_seqfile(seqfiles, ksize, memory, maxfpr=0.2, count=True, smallcount=False,
                        mask=None, maskmaxabund=0, consume_masked=False, numbands=None,
                        band=None, outfile=None, numthreads=1, log=None):
    """Count the k-mers of one sample (one or more FASTA/FASTQ files) into a fresh sketch.

    tablesize = memory / 4 * buckets-per-byte, four tables (kevlar/count.py:29-35); the hot
    loop is kv_consume.  `numthreads` host threads pull read batches from one shared parser,
    as the reference's threads do; with one thread the "distinct k-mers stored" figure is the
    exact single-thread value.
    """
    log = log or kevlar_amd.plog
    numtables = 4
    sketchtype = 'nodegraph'
    if count:
        sketchtype = 'smallcountgraph' if smallcount else 'countgraph'
    tablesize = memory / numtables * khmer._buckets_per_byte[sketchtype]
    sketch = allocate(ksize, tablesize, num_tables=numtables, count=count, smallcount=smallcount)
    if numthreads == 1:
        sketch.track_exact_unique(True)
    numreads = 0
    for seqfile in seqfiles:
        log('[kevlar::count]', '- processing "{}"'.format(seqfile))
        parser = khmer.ReadParser(seqfile)
        if mask:
            kwargs = {'consume_masked': consume_masked,
                      'threshold': 1 if consume_masked else maskmaxabund}
            if numbands:
                target, args = sketch.consume_seqfile_banding_with_mask, (parser, numbands, band, mask)
            else:
                target, args = sketch.consume_seqfile_with_mask, (parser, mask)
        else:
            kwargs = {}
            if numbands:
                target, args = sketch.consume_seqfile_banding, (parser, numbands, band)
            else:
                target, args = sketch.consume_seqfile, (parser,)
        errors = []

        def work():
            try:
                target(*args, **kwargs)
            except BaseException as exc:  # surfaced after join
                errors.append(exc)

        threads = [threading.Thread(target=work) for _ in range(numthreads)]
        for thread in threads:
            thread.start()
        for thread in threads:
            thread.join()
        if errors:
            raise errors[0]
        numreads += parser.num_reads

    message = 'Done loading k-mers'
    if numbands:
        message += ' (band {:d}/{:d})'.format(band + 1, numbands)
    fpr = kevlar_amd.sketch.estimate_fpr(sketch)
    try:
        distinct = sketch.n_unique_kmers()
    except (kevlar_amd._lib.KvError, ValueError):
        # first-touch scratch did not fit: fall back to the multi-thread figure
        sketch.track_exact_unique(False)
        distinct = sketch.n_unique_kmers()
    message += ';\n    {:d} reads processed'.format(numreads)
    message += ', {:d} distinct k-mers stored'.format(distinct)
    message += ';\n    estimated false positive rate is {:1.3f}'.format(fpr)
    if fpr > maxfpr:
        message += ' (FPR too high, bailing out!!!)'
        raise kevlar_amd.sketch.KevlarUnsuitableFPRError('[kevlar::count] ' + message)
    sketch.track_exact_unique(False)  # release the packed batches

    if outfile:
        extensions = get_extension(count=count, smallcount=smallcount)
        if not outfile.endswith(extensions):
            outfile += extensions[1]
        sketch.save(outfile)
        message += ';\n    saved to "{:s}"'.format(outfile)
    log('[kevlar::count]', message)
    return sketch


def print_config(args):
    tabletype = {1: 'node', 4: 'small count', 8: 'count'}[args.counter_size]
    message = 'Storing k-mers in a {} table'.format(tabletype)
    if args.counter_size == 1:
        message += ' (Bloom filter) for k-mer presence/absence queries'
    else:
        maxcount = {4: 15, 8: 255}[args.counter_size]
        message += ', a CountMin sketch with a counter size of {} bits'.format(args.counter_size)
        message += ', for k-mer abundance queries (max abundance {})'.format(maxcount)
    kevlar_amd.plog('[kevlar::count]', message)


def main(args):
    if (args.num_bands is None) is not (args.band is None):
        raise ValueError('Must specify --num-bands and --band together')
    myband = args.band - 1 if args.band else None
    if args.mask:
        args.mask = kevlar_amd.sketch.load(args.mask)
    print_config(args)

    timer = kevlar_amd.Timer()
    timer.start()
    load_sample_seqfile(
        args.seqfile, args.ksize, args.memory, args.max_fpr, count=args.counter_size > 1,
        smallcount=args.counter_size == 4, mask=args.mask, consume_masked=args.count_masked,
        numbands=args.num_bands, band=myband, numthreads=args.threads, outfile=args.counttable,
    )
    total = timer.stop()
    kevlar_amd.plog('[kevlar::count] Total time: {:.2f} seconds'.format(total))
