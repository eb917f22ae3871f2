"""kevlar_amd -- MI355X-native drop-in for kevlar's novel-k-mer path.

Mirrors the reference package surface for count -> novel -> filter -> partition (+ unband):
kevlar/__init__.py:72-128.  The sketch engine underneath (kevlar_amd.khmer) is HIP-only.
"""
import builtins
from gzip import open as gzopen
from os import makedirs
from os.path import dirname
import sys

__version__ = '0.7+mi355x.r1'

logstream = None
teelog = False


def plog(*args, **kwargs):
    """Diagnostics go to the log stream (and to stderr when there is none, or with --tee)."""
    if logstream is not None:
        print(*args, **kwargs, file=logstream)
    if logstream is None or teelog:
        print(*args, **kwargs, file=sys.stderr)


def open(filename, mode):
    if mode not in ('r', 'w'):
        raise ValueError('invalid mode "{}"'.format(mode))
    if filename in ['-', None]:
        return sys.stdin if mode == 'r' else sys.stdout
    if filename.endswith('.gz'):
        if mode == 'w':         # blocked gzip: any gzip reader takes it, and the GPU can inflate its members in parallel
            from kevlar_amd.bgzf import BgzfWriter
            return BgzfWriter(filename)
        return gzopen(filename, mode + 't')
    return builtins.open(filename, mode)


class _Sink(object):
    """where a driver writes: takes bytes or str, whatever the stream underneath wants"""

    def __init__(self, stream, binary, close):
        self._stream, self._binary, self._close = stream, binary, close

    def write(self, data):
        if self._binary:
            self._stream.write(data.encode('latin-1') if isinstance(data, str) else data)
        else:
            self._stream.write(data.decode('latin-1') if isinstance(data, (bytes, bytearray, memoryview)) else data)

    def raw_fd(self):
        """file descriptor of a plain binary file underneath (flushed), for text a native formatter writes itself; None for
        standard output, gzip writers and anything else that is not a file of bytes"""
        import io
        if not self._binary or not isinstance(self._stream, io.BufferedWriter):
            return None
        try:
            self._stream.flush()
            return self._stream.fileno()
        except (OSError, ValueError):
            return None

    def close(self):
        if self._close:
            self._stream.close()
            self._close = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_sink(filename):
    """Output of a driver: '-' / None = standard output, *.gz = blocked gzip, else a plain file; written as bytes, so text
    rendered natively goes out without being decoded and encoded again."""
    if filename in ['-', None]:
        raw = getattr(sys.stdout, 'buffer', None)
        return _Sink(raw if raw is not None else sys.stdout, raw is not None, False)
    if filename.endswith('.gz'):
        from kevlar_amd.bgzf import BgzfWriter
        return _Sink(BgzfWriter(filename), True, True)
    return _Sink(builtins.open(filename, 'wb'), True, True)


def mkdirp(path, trim=False):
    outdir = dirname(path) if trim else path
    makedirs(outdir, exist_ok=True)
    return outdir


from kevlar_amd.sequence import (Record, KmerOfInterest, revcom, parse_augmented_fastx,  # noqa: E402
                                 print_augmented_fastx)


def revcommin(seq):
    rc = revcom(seq)
    return seq if seq <= rc else rc


def same_seq(seq1, seq2, seq2revcom=None):
    if seq2revcom is None:
        seq2revcom = revcom(seq2)
    return seq1 == seq2 or seq1 == seq2revcom


from kevlar_amd.timer import Timer  # noqa: E402
from kevlar_amd.progress import ProgressIndicator  # noqa: E402
from kevlar_amd import khmer  # noqa: E402
from kevlar_amd import sequence  # noqa: E402
from kevlar_amd import seqio  # noqa: E402
from kevlar_amd import sketch  # noqa: E402
from kevlar_amd.seqio import parse_partitioned_reads, parse_single_partition  # noqa: E402
from kevlar_amd import readgraph  # noqa: E402
from kevlar_amd.readgraph import ReadGraph  # noqa: E402
from kevlar_amd import count  # noqa: E402
from kevlar_amd import novel  # noqa: E402
from kevlar_amd import filter  # noqa: E402
from kevlar_amd import partition  # noqa: E402
from kevlar_amd import unband  # noqa: E402
from kevlar_amd import dist  # noqa: E402
from kevlar_amd import split  # noqa: E402
from kevlar_amd import augment  # noqa: E402
from kevlar_amd import gentrio  # noqa: E402
from kevlar_amd import cli  # noqa: E402


class multi_file_iter_khmer(object):
    """Records of several sequence files, one after the other (kevlar/__init__.py:125-128).
    Iterating yields records; text_batches() hands the novel scan whole parsed batches instead."""

    def __init__(self, filenames, kept=None):
        self.filenames = list(filenames)
        self.kept = kept or {}          # filename -> (parser, text batch) of files whose single batch is still open (kevlar_amd.count)

    def __iter__(self):
        for filename in self.filenames:
            for record in khmer.ReadParser(filename):
                yield record

    def text_batches(self, max_reads):
        for filename in self.filenames:
            held = self.kept.pop(filename, None)
            if held is not None and max_reads >= held[1].n:
                yield held[1]
                continue
            for tb in khmer.ReadParser(filename).text_batches(max_reads):
                yield tb
