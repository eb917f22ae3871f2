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
    return Record(record.name, record.sequence, getattr(record, 'quality', None))


class _Renderer(object):
    """Record -> augmented text.  One instance, reused: the pieces of a record are collected in a list and joined
    once; the k-mer lines are produced in offset order whatever order the annotations were added in."""

    GAP = ' ' * 10

    def header(self, record):
        if record.quality is None:
            return '>{}\n{}\n'.format(record.name, record.sequence)
        return '@{}\n{}\n+\n{}\n'.format(record.name, record.sequence, record.quality)

    def kmer_line(self, sequence, note):
        ksize, offset, abund = note
        return ''.join((' ' * offset, sequence[offset:offset + ksize], self.GAP, ' '.join(str(a) for a in abund), '#\n'))

    def __call__(self, record):
        pieces = [self.header(record)]
        notes = record.annotations
        if any(notes[i][1] > notes[i + 1][1] for i in range(len(notes) - 1)):
            notes = sorted(notes, key=lambda note: note[1])        # stable: ties keep their order
        pieces.extend(self.kmer_line(record.sequence, note) for note in notes)
        pieces.extend('#mateseq={}#\n'.format(mate) for mate in record.mates)
        return ''.join(pieces)


format_augmented_fastx = _Renderer()


def print_augmented_fastx(record, outstream):
    text = format_augmented_fastx(record)
    try:
        outstream.write(text.encode('ascii'))
    except TypeError:
        outstream.write(text)


write_record = print_augmented_fastx


def _classify(line):
    """'blank', 'header', 'mate', 'kmer' or 'junk' for one line of an augmented stream"""
    if not line.strip():
        return 'blank'
    if line[0] in '@>':
        return 'header'
    if line[-2:] == '#\n':
        return 'mate' if line.startswith('#mateseq=') else 'kmer'
    return 'junk'


def parse_augmented_fastx(instream):
    """Generator over Records of an augmented FASTA/FASTQ stream (files go through the native parser instead:
    kevlar_amd.annotated.AnnotatedReads.from_file)."""
    lines = iter(instream)
    current = None
    for line in lines:
        kind = _classify(line)
        if kind == 'blank':
            continue
        if kind == 'header':
            if current is not None:
                yield current
            sequence = next(lines).strip()
            quality = None
            if line[0] == '@':
                next(lines)                                   # the '+' line
                quality = next(lines).strip()
            current = Record(name=line[1:].strip(), sequence=sequence, quality=quality)
        elif kind == 'mate':
            current.add_mate(_MATE_RE.search(line).group(1))
        elif kind == 'kmer':
            text = line.lstrip()
            kmer, *counts = text[:-2].split()
            current.annotate(kmer, len(line) - len(text), tuple(int(c) for c in counts))
        else:
            raise Exception(line)
    yield current
