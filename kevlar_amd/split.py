"""`kevlar split` driver (the reference's kevlar/split.py:14-44): deal the partitions of a partitioned
augmented FASTA/FASTQ file round-robin into N output files.  Host-side text plumbing, the step right
after `partition` in the mark-I workflow (Snakefile:312-316)."""
from itertools import cycle

import kevlar_amd


def split(pstream, outstreams, maxreads=10000):
    """Split the partitions across the N outstreams."""
    progress_indicator = kevlar_amd.ProgressIndicator(
        '[kevlar::split] processed {counter} partitions',
        interval=100, breaks=[1000, 10000, 100000], usetimer=True,
    )
    for partdata, outstream in zip(pstream, cycle(outstreams)):
        partid, partition = partdata
        if len(partition) > maxreads:
            kevlar_amd.plog('[kevlar::split]', 'WARNING: discarding partition with {} reads'.format(len(partition)))
            continue
        for read in partition:
            kevlar_amd.print_augmented_fastx(read, outstream)
        progress_indicator.update()


def main(args):
    partfile = kevlar_amd.open(args.infile, 'r')
    readstream = kevlar_amd.parse_augmented_fastx(partfile)
    partstream = kevlar_amd.parse_partitioned_reads(readstream)
    outstreams = list()
    for i in range(args.numfiles):
        outfile = '{:s}.{:d}.augfastx'.format(args.base, i)
        if args.infile.endswith('.gz'):
            outfile += '.gz'
        outstreams.append(kevlar_amd.open(outfile, 'w'))
    split(partstream, outstreams)
    for stream in outstreams:
        stream.close()
