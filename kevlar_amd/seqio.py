"""Stream helpers around augmented FASTA/FASTQ files: plain-FASTA readers for small reference files, the
file-chaining record stream, and the grouping of a partitioned read stream by its `kvcc=N` labels.

Function names, arguments and errors are those of the reference's kevlar.seqio (call sites:
kevlar/split.py:31-34, kevlar/unband.py:67, kevlar/tests/test_seqio.py); the bodies are written around
itertools grouping rather than hand-rolled accumulators."""
from itertools import chain, groupby
import re

import kevlar_amd
from kevlar_amd.sequence import parse_augmented_fastx

_KVCC = re.compile(r'kvcc=(\d+)')


class KevlarPartitionLabelError(ValueError):
    pass


def parse_fasta(data):
    """(defline, sequence) pairs of FASTA text given line by line; sequence lines are joined, a defline directly
    followed by another one has an empty sequence, text in front of the first defline is ignored."""
    open_defline = None
    stripped = (raw.rstrip() for raw in data)
    for heads, block in groupby(stripped, key=lambda text: text[:1] == '>'):
        block = list(block)
        if not heads:
            if open_defline is not None:
                yield open_defline, ''.join(block)
                open_defline = None
            continue
        # a run of deflines: all but the last have no sequence at all
        yield from ((d, '') for d in ([open_defline] if open_defline is not None else []) + block[:-1])
        open_defline = block[-1]
    if open_defline is not None:
        yield open_defline, ''


def parse_seq_dict(data):
    """{sequence id: sequence}; the id is the defline up to the first blank or tab.  Duplicate ids are an error."""
    table = {}
    for defline, sequence in parse_fasta(data):
        seqid = re.match(r'>([^ \t]*)', defline).group(1)
        assert seqid not in table, seqid
        table[seqid] = sequence
    return table


def afxstream(filelist):
    """Records of several augmented FASTA/FASTQ files, one file after the other."""
    return chain.from_iterable(parse_augmented_fastx(kevlar_amd.open(path, 'r')) for path in filelist)


def partition_id(readname):
    """The N of a `kvcc=N` label (as text), or None for a read that carries none."""
    found = _KVCC.search(readname)
    return found.group(1) if found else None


def _label_of(read):
    return partition_id(read.name if hasattr(read, 'name') else read.defline)


def parse_partitioned_reads(readstream):
    """(partition id, reads) for each run of consecutive reads with the same `kvcc` label.

    An unlabelled stream is one partition with id None.  A labelled read after an unlabelled one is an error; the
    reverse (a labelled stream that ends in unlabelled reads) is tolerated the way the reference tolerates it: the
    tail joins the last partition, which is then reported with id None."""
    held_id, held_reads, unlabelled = None, [], False
    readstream = (read for read in readstream if read is not None)      # an empty augmented stream yields one None
    for label, run in groupby(readstream, key=_label_of):
        if label is None:
            held_reads.extend(run)
            unlabelled = True
            continue
        if unlabelled:
            raise KevlarPartitionLabelError('reads with and without partition labels (kvcc=#)')
        if held_id is not None:
            yield held_id, held_reads
            held_reads = []
        held_id = label
        held_reads.extend(run)
    yield (None if unlabelled else held_id), held_reads


def parse_single_partition(readstream, partid):
    """Only the partition(s) labelled `partid`."""
    return ((pid, reads) for pid, reads in parse_partitioned_reads(readstream) if pid == partid)
