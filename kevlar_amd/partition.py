"""`kevlar partition` driver (the reference's kevlar/partition.py:15-80)."""
import kevlar_amd


def partition(readstream, strict=False, minabund=None, maxabund=None, dedup=True, gmlfile=None):
    timer = kevlar_amd.Timer()
    timer.start()

    timer.start('loadreads')
    kevlar_amd.plog('[kevlar::partition] Loading reads')
    graph = kevlar_amd.ReadGraph()
    graph.load(readstream, minabund=minabund, maxabund=maxabund)
    elapsed = timer.stop('loadreads')
    kevlar_amd.plog('[kevlar::partition]', 'Reads loaded in {:.2f} sec'.format(elapsed))

    timer.start('buildgraph')
    mode = 'strict' if strict else 'relaxed'
    kevlar_amd.plog('[kevlar::partition]', 'Building read graph in {:s} mode'.format(mode))
    graph.populate_edges(strict=strict)
    elapsed = timer.stop('buildgraph')
    kevlar_amd.plog('[kevlar::partition]', 'Graph built in {:.2f} sec'.format(elapsed))

    if gmlfile:
        raise NotImplementedError('--gml needs the explicit edge list, which this build never materialises')

    timer.start('partition')
    kevlar_amd.plog('[kevlar::partition] Partition readgraph')
    part_iter = graph.partitions(dedup, minabund, maxabund, abundfilt=True)
    for n, part in enumerate(part_iter, 1):
        reads = [graph.get_record(readname) for readname in list(part)]
        for read in reads:
            read.name += ' kvcc={:d}'.format(n)
        yield n, reads
    elapsed = timer.stop('partition')
    kevlar_amd.plog('[kevlar::partition]', 'Partitioning done in {:.2f} sec'.format(elapsed))
    total = timer.stop()
    kevlar_amd.plog('[kevlar::partition]', 'Total time: {:.2f} seconds'.format(total))


def main(args):
    if args.split:
        kevlar_amd.mkdirp(args.split, trim=True)
    outstream = None if args.split else kevlar_amd.open(args.out, 'w')
    readstream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.infile, 'r'))
    partitioner = partition(readstream, strict=args.strict, minabund=args.min_abund,
                            maxabund=args.max_abund, dedup=args.dedup, gmlfile=args.gml)
    numreads = 0
    partnum = 0
    for partnum, part in partitioner:
        numreads += len(part)
        if args.split:
            ofname = '{:s}.cc{:d}.augfastq.gz'.format(args.split, partnum)
            with kevlar_amd.open(ofname, 'w') as outfile:
                for read in part:
                    kevlar_amd.print_augmented_fastx(read, outfile)
        else:
            for read in part:
                kevlar_amd.print_augmented_fastx(read, outstream)
    message = 'grouped {:d} reads into {:d} connected components'.format(numreads, partnum)
    kevlar_amd.plog('[kevlar::partition]', message)
