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
    __slots__ = ('name', 'sequence', 'quality', 'annotations', 'mates', '_ikmers', '_ikmers_n')

    def __init__(self, name, sequence, quality=None, annotations=None, mates=None, ikmers=None):
        self.name = name
        self.sequence = sequence
        self.quality = quality
        self.mates = [] if mates is None else mates
        self.annotations = [] if annotations is None else annotations
        self._ikmers = ikmers          # k-mer sequence (both strands) -> KmerOfInterest; built on first use
        self._ikmers_n = len(self.annotations) if ikmers is not None else -1

    @property
    def ikmers(self):
        """{k-mer sequence or its reverse complement: KmerOfInterest} (sequence.pyx:38-49).  Only the strict
        read-pair logic looks k-mers up by sequence, so the dictionary is built on demand, not per annotation."""
        if self._ikmers is None or self._ikmers_n != len(self.annotations):
            table = {}
            for ikmer in self.annotations:
                seq = self.ikmerseq(ikmer)
                table[seq] = ikmer
                table[revcom(seq)] = ikmer
            self._ikmers = table
            self._ikmers_n = len(self.annotations)
        return self._ikmers

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
        self.annotations.append(KmerOfInterest(len(sequence), offset, abundances))

    def ikmerseq(self, ikmer):
        return self.sequence[ikmer.offset:ikmer.offset + ikmer.ksize]


def copy_record(record):
    quality = getattr(record, 'quality', None)
    return Record(record.name, record.sequence, quality)


def _by_offset(ikmer):
    return ikmer[1]


def format_augmented_fastx(record):
    if record.quality is not None:
        parts = ['@', record.name, '\n', record.sequence, '\n+\n', record.quality, '\n']
    else:
        parts = ['>', record.name, '\n', record.sequence, '\n']
    seq = record.sequence
    for ksize, offset, abund in sorted(record.annotations, key=_by_offset):
        parts.append('{}{}          {}#\n'.format(' ' * offset, seq[offset:offset + ksize], ' '.join(map(str, abund))))
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
            body = line.lstrip()
            fields = body[:-2].split()
            record.annotate(fields[0], len(line) - len(body), tuple(map(int, fields[1:])))
        else:
            raise Exception(line)
    yield record
