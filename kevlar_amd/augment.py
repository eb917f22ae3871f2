"""`kevlar augment` driver (the reference's kevlar/augment.py:13-45): re-attach interesting-k-mer
annotations to sequences that lost them in a third-party tool (assembled contigs, re-processed reads),
using the annotations of an already augmented file.  Exact string matching on the host: the k-mer set is
the few thousand interesting k-mers of one partition, not a sketch."""
import kevlar_amd


def augment(augseqstream, nakedseqstream, upint=10000):
    """
    Augment an unannotated stream of sequences.

    - `augseqstream`: a stream of sequences annotated with k-mers of interest
    - `nakedseqstream`: a stream of unannotated sequences, to be augmented with
      k-mers of interest from `augseqstream`
    """
    ksize = None
    ikmers = dict()
    for n, record in enumerate(augseqstream):
        if n > 0 and n % upint == 0:
            kevlar_amd.plog('[kevlar::augment] processed', n, 'input reads')
        for ikmer in record.annotations:
            seq = record.ikmerseq(ikmer)
            ikmers[seq] = ikmer.abund
            ikmers[kevlar_amd.revcom(seq)] = ikmer.abund
            ksize = ikmer.ksize

    for record in nakedseqstream:
        qual = None
        if hasattr(record, 'quality') and record.quality is not None:
            qual = record.quality
        newrecord = kevlar_amd.sequence.Record(name=record.name, sequence=record.sequence, quality=qual)
        if ksize is not None:
            numkmers = len(record.sequence) - ksize + 1
            for offset in range(numkmers):
                kmer = record.sequence[offset:offset + ksize]
                if kmer in ikmers:
                    newrecord.annotate(kmer, offset, ikmers[kmer])
        yield newrecord


def main(args):
    augseqs = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.augseqs, 'r'))
    nakedseqs = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.seqs, 'r'))
    outstream = kevlar_amd.open(args.out, 'w')
    for record in augment(augseqs, nakedseqs):
        kevlar_amd.print_augmented_fastx(record, outstream)
