"""`kevlar split`: deal the partitions of a partitioned augmented FASTA/FASTQ file over N output files, the step
between `partition` and the per-partition assembly jobs of the workflow (kevlar/split.py:14-44,
workflows/mark-I/Snakefile:312-316).

A partition is rendered to text once and written with one call; which file a partition goes to is its position in
the input modulo N (an oversized partition that is discarded still uses up its turn, as in the reference)."""
import kevlar_amd
from kevlar_amd.sequence import format_augmented_fastx


def _emit(text, sink):
    try:
        sink.write(text)
    except TypeError:                      # binary sinks (gzip)
        sink.write(text.encode('ascii'))


def split(pstream, outstreams, maxreads=10000):
    """Round-robin the (partition id, reads) pairs of `pstream` over `outstreams`; partitions of more than
    `maxreads` reads are dropped with a warning."""
    progress = kevlar_amd.ProgressIndicator('[kevlar::split] processed {counter} partitions', interval=100,
                                            breaks=[1000, 10000, 100000], usetimer=True)
    nsinks = len(outstreams)
    for turn, (partid, reads) in enumerate(pstream):
        if len(reads) > maxreads:
            kevlar_amd.plog('[kevlar::split]', 'WARNING: discarding partition with {} reads'.format(len(reads)))
            continue
        _emit(''.join(map(format_augmented_fastx, reads)), outstreams[turn % nsinks])
        progress.update()


def main(args):
    suffix = '.augfastx.gz' if args.infile.endswith('.gz') else '.augfastx'
    sinks = [kevlar_amd.open('{:s}.{:d}{:s}'.format(args.base, i, suffix), 'w') for i in range(args.numfiles)]
    try:
        reads = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.infile, 'r'))
        split(kevlar_amd.parse_partitioned_reads(reads), sinks)
    finally:
        for sink in sinks:
            sink.close()
