"""`kevlar unband` driver (the reference's kevlar/unband.py:26-84): merge the per-band outputs
of a banded `novel` run -- union of the annotations of records that share a read name.
Host-side text plumbing; it is also the semantic model of the multi-GPU band merge."""
from tempfile import TemporaryDirectory
import zlib

import kevlar_amd


def create_batch_files(numbatches, tempdir):
    return [
        kevlar_amd.open('{:s}/kevlar-unband-batch{:d}.augfastq.gz'.format(tempdir, i), 'w')
        for i in range(numbatches)
    ]


def write_records_to_batches(recordstream, batchfiles):
    numbatches = len(batchfiles)
    kevlar_amd.plog('[kevlar::unband]', 'writing records to {:d} temp batch files'.format(numbatches))
    progress = kevlar_amd.ProgressIndicator('[kevlar::unband]     processed {counter} reads',
                                            interval=1e5, breaks=[1e6, 1e7])
    for record in recordstream:
        progress.update()
        # the reference buckets by Python's per-process hash(name); any stable function of the
        # name gives the same guarantee (all copies of a read meet in one batch)
        batch = zlib.crc32(record.name.encode()) % numbatches
        kevlar_amd.print_augmented_fastx(record, batchfiles[batch])


def resolve_batch(batchfile):
    filename = batchfile.name
    batchfile.close()
    reads = {}
    with kevlar_amd.open(filename, 'r') as fh:
        for read in kevlar_amd.parse_augmented_fastx(fh):
            if read is None:
                continue
            if read.name not in reads:
                reads[read.name] = read
            else:
                reads[read.name].annotations.extend(read.annotations)
    for readname in sorted(reads):
        read = reads[readname]
        read.annotations.sort(key=lambda k: k.offset)
        yield read


def resolve_batches(batchfiles):
    kevlar_amd.plog('[kevlar::unband]', 'resolving duplicate reads in {:d} batches'.format(len(batchfiles)))
    for n, batchfile in enumerate(batchfiles):
        for read in resolve_batch(batchfile):
            yield read
        kevlar_amd.plog('[kevlar::unband]     batch {:d} complete'.format(n))
    kevlar_amd.plog('[kevlar::unband] Done!')


def unband(recordstream, numbatches=16):
    with TemporaryDirectory() as tempdir:
        batchfiles = create_batch_files(numbatches, tempdir)
        write_records_to_batches(recordstream, batchfiles)
        for read in resolve_batches(batchfiles):
            yield read


def main(args):
    outstream = kevlar_amd.open(args.out, 'w')
    records = kevlar_amd.seqio.afxstream(args.infile)
    for read in unband(records, args.n_batches):
        kevlar_amd.print_augmented_fastx(read, outstream)
