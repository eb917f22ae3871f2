"""Record / augmented FASTA-FASTQ codec (the reference's kevlar/sequence.pyx:14-178).

Text format (must be byte-exact, kevlar/tests/test_seqio.py:135-182):
    @name\\nseq\\n+\\nqual\\n         or   >name\\nseq\\n
    then one line per interesting k-mer, sorted by offset:
        ' ' * offset + kmer + ' ' * 10 + 'a b c' + '#'
    then optional  #mateseq=SEQ#  lines.
"""
from collections import namedtuple
import re

KmerOfInterest = namedtuple('KmerOfInterest', 'ksize offset abund')

_COMPLEMENT = str.maketrans('ATUGCYRSWKMBDHVNatugcyrswkmbdhvn',
                            'TAACGRYSWMKVHDBNTAACGRYSWMKVHDBN')
_MATE_RE = re.compile(r'^#mateseq=(\S+)#\n$')


def revcom(sequence):
    return sequence.translate(_COMPLEMENT)[::-1]


def tostr(stringlike):
    try:
        return stringlike.decode('utf-8')
    except AttributeError:
        return stringlike


class Record(object):
    __slots__ = ('name', 'sequence', 'quality', 'annotations', 'mates', 'ikmers')

    def __init__(self, name, sequence, quality=None, annotations=None, mates=None, ikmers=None):
        self.name = name
        self.sequence = sequence
        self.quality = quality
        self.mates = [] if mates is None else mates
        self.ikmers = {}
        if annotations is None:
            self.annotations = []
        else:
            self.annotations = annotations
            if ikmers is not None:
                self.ikmers = ikmers
            else:
                for ikmer in annotations:
                    seq = self.ikmerseq(ikmer)
                    self.ikmers[seq] = ikmer
                    self.ikmers[revcom(seq)] = ikmer

    def __len__(self):
        return len(self.sequence)

    @property
    def id(self):
        return self.name.split()[0]

    def add_mate(self, mateseq):
        self.mates.append(mateseq)

    def annotate(self, sequence, offset, abundances):
        found = self.sequence[offset:offset + len(sequence)]
        assert found == sequence, (found, sequence)
        ikmer = KmerOfInterest(len(sequence), offset, abundances)
        self.annotations.append(ikmer)
        self.ikmers[sequence] = ikmer
        self.ikmers[revcom(sequence)] = ikmer

    def ikmerseq(self, ikmer):
        return self.sequence[ikmer.offset:ikmer.offset + ikmer.ksize]


def copy_record(record):
    quality = getattr(record, 'quality', None)
    return Record(record.name, record.sequence, quality)


def format_augmented_fastx(record):
    if record.quality is not None:
        parts = ['@', record.name, '\n', record.sequence, '\n+\n', record.quality, '\n']
    else:
        parts = ['>', record.name, '\n', record.sequence, '\n']
    for ikmer in sorted(record.annotations, key=lambda k: k.offset):
        parts.append(' ' * ikmer.offset)
        parts.append(record.sequence[ikmer.offset:ikmer.offset + ikmer.ksize])
        parts.append(' ' * 10)
        parts.append(' '.join(str(a) for a in ikmer.abund))
        parts.append('#\n')
    for mateseq in record.mates:
        parts.append('#mateseq={}#\n'.format(mateseq))
    return ''.join(parts)


def print_augmented_fastx(record, outstream):
    text = format_augmented_fastx(record)
    try:
        outstream.write(bytes(text, 'ascii'))
    except TypeError:
        outstream.write(text)


write_record = print_augmented_fastx


def parse_augmented_fastx(instream):
    """Generator over Records of an augmented FASTA/FASTQ stream."""
    record = None
    for line in instream:
        if line.strip() == '':
            continue
        first = line[0]
        if first in ('@', '>'):
            if record is not None:
                yield record
            name = line[1:].strip()
            seq = next(instream).strip()
            qual = None
            if first == '@':
                next(instream)
                qual = next(instream).strip()
            record = Record(name=name, sequence=seq, quality=qual)
        elif line.endswith('#\n'):
            if line.startswith('#mateseq='):
                record.add_mate(_MATE_RE.search(line).group(1))
                continue
            offset = len(line) - len(line.lstrip())
            fields = line.strip()[:-1].split()
            kmer = fields[0]
            record.annotate(kmer, offset, tuple(int(a) for a in fields[1:]))
        else:
            raise Exception(line)
    yield record
