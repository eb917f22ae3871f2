"""The native augmented FASTA/FASTQ codec (kv_augfastx.hip: host code, no GPU needed) against the Python one
(kevlar_amd/sequence.py, itself pinned to the reference's files in tests/test_host_logic.py): every golden augmented
file parses to the same records and annotations, and formats back to the same bytes; BGZF files index correctly."""
import glob
import gzip
import io
import os

import numpy as np
import pytest

import kevlar_amd
from kevlar_amd.annotated import AnnotatedReads
from kevlar_amd.sequence import format_augmented_fastx, parse_augmented_fastx

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'data')
FILES = sorted(glob.glob(os.path.join(DATA, '*.augfast*')) + glob.glob(os.path.join(DATA, '**', '*.augfast*'), recursive=True))


def python_records(path):
    with kevlar_amd.open(path, 'r') as stream:
        return [r for r in parse_augmented_fastx(stream)]


@pytest.mark.parametrize('path', FILES, ids=[os.path.basename(p) for p in FILES])
def test_native_parse_and_format_equal_python(path):
    want = python_records(path)
    ksizes = {k.ksize for r in want for k in r.annotations}
    if len(ksizes) > 1:
        with pytest.raises(ValueError):
            AnnotatedReads.from_file(path)
        return
    got = AnnotatedReads.from_file(path)
    assert got.n == len(want)
    assert len(got) == sum(len(r.annotations) for r in want)
    for i, rec in enumerate(want):
        mine = got.record(i)
        assert (mine.name, mine.sequence, mine.quality) == (rec.name, rec.sequence, rec.quality)
        assert [tuple(k) for k in mine.annotations] == [(k.ksize, k.offset, tuple(k.abund)) for k in rec.annotations]
        assert mine.mates == rec.mates
    text = ''.join(format_augmented_fastx(r) for r in want).encode('latin-1')
    assert got.format(np.arange(got.n)) == text
    # the container built from record objects renders the same bytes
    assert AnnotatedReads(want).format(np.arange(len(want))) == text
    if len(got):
        # a selection: every second annotation, recounted case abundance, a name suffix
        keep = np.zeros(len(got), dtype=bool)
        keep[::2] = True
        again = np.arange(len(got)) % 200
        chosen = [r for r in got.select(keep, case_abund=again)]
        body, n = got.select_text(keep, case_abund=again)
        assert n == len(chosen)
        assert body == ''.join(format_augmented_fastx(r) for r in chosen).encode('latin-1')
        first = np.arange(min(3, got.n))
        for rec, i in zip(want, first.tolist()):
            rec.name += ' kvcc=7'
        assert got.format(first, suffixes=[' kvcc=7'] * len(first)) == ''.join(format_augmented_fastx(r) for r in want[:len(first)]).encode('latin-1')


@pytest.mark.parametrize('threads', ['1', '5'])
def test_format_written_to_a_file_by_the_library_equals_format(tmp_path, threads, monkeypatch):
    """kv_format_records_fd (what filter / partition use when their output is a plain file: stretches of records rendered on
    several threads into buffers they keep, written in order) gives the bytes kv_format_records returns -- over many stretches
    (the records of a golden file repeated to 100 000 outputs), with a selection, recounted abundances and name suffixes; a
    sink that is not a plain file (gzip) takes the old route with the same result"""
    monkeypatch.setenv('KV_FORMAT_THREADS', threads)
    path = os.path.join(DATA, 'trio1', 'novel_3_1,2.txt')
    got = AnnotatedReads.from_file(path)
    reads = np.tile(np.arange(got.n), 100000 // got.n + 1)[:100000]
    rng = np.random.default_rng(3)
    rng.shuffle(reads)
    keep = rng.random(len(got)) < 0.7
    again = (np.arange(len(got)) % 250).astype(np.int32)
    blob = ''.join(' kvcc={}'.format(i % 977) for i in range(len(reads))).encode('latin-1')
    offs = np.cumsum([0] + [len(' kvcc={}'.format(i % 977)) for i in range(len(reads))]).astype(np.uint64)
    want = got.format(reads, keep, again, suffix_blob=(blob, offs))
    assert len(want) > 40_000_000
    out = str(tmp_path / 'out.augfastq')
    with kevlar_amd.open_sink(out) as sink:
        sink.write(b'# head\n')                       # what the sink already holds stays in front
        got.format_to(sink, reads, keep, again, suffix_blob=(blob, offs))
        sink.write(b'# tail\n')
    assert open(out, 'rb').read() == b'# head\n' + want + b'# tail\n'
    gz = str(tmp_path / 'out.augfastq.gz')
    with kevlar_amd.open_sink(gz) as sink:
        got.format_to(sink, reads[:5000], keep, again)
    assert gzip.open(gz, 'rb').read() == got.format(reads[:5000], keep, again)
    # an annotation that does not fit its read is an error on both routes
    bad = AnnotatedReads.from_file(path)
    bad.offset = bad.offset.copy()
    bad.offset[0] = 10 ** 6
    with pytest.raises(ValueError):
        bad.format(np.arange(bad.n))
    with kevlar_amd.open_sink(str(tmp_path / 'bad.augfastq')) as sink, pytest.raises(ValueError):
        bad.format_to(sink, np.arange(bad.n))
    # ... and when the bad record sits in a LATER stretch, the stretches already written are taken back: a failed call leaves the
    # file as it found it (the head), not a truncated but plausible list of records
    late = np.concatenate((reads[:70000][reads[:70000] != 0], [0]))       # record 0 (the damaged one) only at the very end: third stretch
    tail = str(tmp_path / 'late.augfastq')
    with kevlar_amd.open_sink(tail) as sink:
        sink.write(b'# head\n')
        with pytest.raises(ValueError):
            bad.format_to(sink, late)
        sink.write(b'# after\n')
    assert open(tail, 'rb').read() == b'# head\n# after\n'


def test_native_parser_rejects_what_the_python_parser_rejects(tmp_path):
    good = '@r1\nACGTACGTAC\n+\nIIIIIIIIII\n  GTACG          7 0 1#\n'
    bad_kmer = good.replace('  GTACG', '  GTACC')
    stray = good + 'something else\n'
    for name, text in (('kmer.augfastq', bad_kmer), ('stray.augfastq', stray)):
        path = str(tmp_path / name)
        with open(path, 'w') as fh:
            fh.write(text)
        with pytest.raises(Exception):
            AnnotatedReads.from_file(path)
        with pytest.raises(Exception):
            python_records(path)
    path = str(tmp_path / 'ok.augfastq')
    with open(path, 'w') as fh:
        fh.write('\n' + good + '\n#mateseq=TTTT#\n>r2 second\nAAAA\n')
    got = AnnotatedReads.from_file(path)
    assert got.n == 2 and len(got) == 1 and got.record(0).mates == ['TTTT'] and got.record(1).quality is None
    assert got.format([0, 1]) == ''.join(format_augmented_fastx(r) for r in python_records(path)).encode('latin-1')


def test_bgzf_writer_and_index(tmp_path):
    """kevlar_amd.open(name.gz, 'w') writes blocked gzip that every gzip reader takes and kv_bgzf_index walks"""
    import ctypes
    from kevlar_amd import _lib, bgzf
    path = str(tmp_path / 'out.augfastq.gz')
    text = ''.join('@read{}\n{}\n+\n{}\n'.format(i, 'ACGT' * 25, 'I' * 100) for i in range(3000))
    sink = kevlar_amd.open(path, 'w')
    sink.write(text[:1000])
    sink.write(text[1000:].encode('ascii'))
    sink.close()
    assert bgzf.is_bgzf(path)
    assert gzip.open(path, 'rt').read() == text
    with kevlar_amd.open(path, 'r') as stream:
        assert stream.read() == text
    image = open(path, 'rb').read()
    total, members = ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(_lib.load().kv_bgzf_text_size(image, len(image), ctypes.byref(total), ctypes.byref(members)))
    assert total.value == len(text) and members.value == (len(text) + bgzf.BLOCK_TEXT - 1) // bgzf.BLOCK_TEXT + 1
    plain = gzip.compress(text.encode('ascii'))
    with pytest.raises(Exception):
        _lib.check(_lib.load().kv_bgzf_text_size(plain, len(plain), ctypes.byref(total), ctypes.byref(members)))
    # the multi-threaded writer produces the same members
    other = str(tmp_path / 'mt.gz')
    bgzf.write_file(other, text.encode('ascii'), level=6, threads=4)
    assert open(other, 'rb').read() == image


def test_split_file_equals_split_of_records(tmp_path):
    """kevlar split on arrays (split_file) writes byte for byte what the record-stream split writes"""
    from io import BytesIO, StringIO
    infile = os.path.join(DATA, 'fiveparts.augfastq.gz')
    by_records = [StringIO() for _ in range(3)]
    stream = kevlar_amd.parse_partitioned_reads(parse_augmented_fastx(kevlar_amd.open(infile, 'r')))
    kevlar_amd.split.split(stream, by_records, maxreads=60)           # drops the 67-read partition
    by_arrays = [BytesIO() for _ in range(3)]
    kevlar_amd.split.split_file(infile, by_arrays, maxreads=60)
    assert [s.getvalue().encode('ascii') for s in by_records] == [s.getvalue() for s in by_arrays]
    assert sum(len(s.getvalue()) for s in by_arrays) > 1000
    # an unlabelled file is one partition
    plain = os.path.join(DATA, 'example1.augfastq')
    one, other = [StringIO(), StringIO()], [BytesIO(), BytesIO()]
    kevlar_amd.split.split(kevlar_amd.parse_partitioned_reads(parse_augmented_fastx(kevlar_amd.open(plain, 'r'))), one)
    kevlar_amd.split.split_file(plain, other)
    assert [s.getvalue().encode('ascii') for s in one] == [s.getvalue() for s in other]


def test_unband_files_equals_unband_of_records(tmp_path):
    """kevlar unband on arrays writes byte for byte what the record-stream unband writes: a banded novel output cut into
    band files (each copy of a read holding part of its annotations) folds back into the same text"""
    from kevlar_amd.sequence import Record
    src = python_records(os.path.join(DATA, 'fiveparts.augfastq.gz'))[:60] + python_records(os.path.join(DATA, 'seqs-mates.augfastq'))
    ksizes = [k.ksize for r in src for k in r.annotations]
    common = max(set(ksizes), key=ksizes.count)
    src = [r for r in src if all(k.ksize == common for k in r.annotations)]
    bands = [[], [], []]
    for i, rec in enumerate(src):
        for b in range(3):
            notes = [k for j, k in enumerate(rec.annotations) if (j + i) % 3 == b]
            if notes or b == i % 3:
                bands[b].append(Record(rec.name, rec.sequence, rec.quality, annotations=list(reversed(notes)), mates=list(rec.mates) if b == i % 3 else []))
    paths = []
    for b, records in enumerate(bands):
        paths.append(str(tmp_path / 'band{}.augfastq'.format(b)))
        with open(paths[-1], 'w') as fh:
            fh.write(''.join(format_augmented_fastx(r) for r in records))
    want = ''.join(format_augmented_fastx(r) for r in kevlar_amd.unband.unband(kevlar_amd.seqio.afxstream(paths), 4)).encode('latin-1')
    got = kevlar_amd.unband.unband_files(paths, 4)
    assert got == want and len(got) > 5000


def _big_augfastq(n, fastq, seed):
    """~100 bytes of record + annotations per read; quality lines that begin with '@' and '>', mates, blank lines"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        seq = ''.join('ACGT'[c] for c in rng.integers(0, 4, 60))
        if fastq:
            qual = ''.join(chr(c) for c in rng.integers(33, 74, 60))
            if i % 3 == 0:
                qual = '@' + qual[1:]
            if i % 5 == 0:
                qual = '>' + qual[1:]
            out.append('@read{} x\n{}\n+\n{}\n'.format(i, seq, qual))
        else:
            out.append('>read{} x\n{}\n'.format(i, seq))
        start = int(rng.integers(0, 20))
        for o in range(start, start + int(rng.integers(0, 4))):
            out.append(' ' * o + seq[o:o + 21] + ' ' * 10 + '{} 0 1#\n'.format(int(rng.integers(1, 99))))
        if i % 7 == 0:
            out.append('#mateseq={}#\n'.format(seq[::-1]))
        if i % 1000 == 0:
            out.append('\n')
    return ''.join(out)


@pytest.mark.parametrize('fastq', [True, False])
def test_native_parse_in_pieces_equals_one_pass(tmp_path, fastq):
    """a file big enough to be cut at record starts and parsed side by side gives the arrays of the one-pass parse; a damaged
    line in the middle is reported as the one-pass parse reports it"""
    text = _big_augfastq(60000, fastq, 5)
    assert len(text) > (9 << 20)
    path = str(tmp_path / ('big.augfastq' if fastq else 'big.augfasta'))
    with open(path, 'w') as fh:
        fh.write(text)
    loaded = {}
    for threads in ('1', '2', '7'):
        os.environ['KV_AUGFASTX_THREADS'] = threads
        try:
            loaded[threads] = AnnotatedReads.from_file(path)
        finally:
            os.environ.pop('KV_AUGFASTX_THREADS', None)
    one = loaded['1']
    assert one.n == 60000 and len(one.mate_record) == 60000 // 7 + 1
    # the writer (one pass; rendering stretches side by side was measured: page faults of the second buffer ate the gain) gives the
    # same bytes whatever the loader's thread count, for any selection and order of records
    picks = np.random.default_rng(3).permutation(one.n)[:50000]
    rendered = {}
    for threads in ('1', '2', '7'):
        os.environ['KV_AUGFASTX_THREADS'] = threads
        try:
            rendered[threads] = (one.format(np.arange(one.n)), one.format(picks, suffixes=[' kvcc={}'.format(i % 9) for i in range(len(picks))]))
        finally:
            os.environ.pop('KV_AUGFASTX_THREADS', None)
    assert rendered['1'][0].count(b'\n') > 3 * one.n
    assert rendered['2'] == rendered['1'] and rendered['7'] == rendered['1'] and len(rendered['1'][1]) > (5 << 20)
    for threads in ('2', '7'):
        got = loaded[threads]
        assert got.n == one.n and got.ksize == one.ksize and got.nsamples == one.nsamples
        for field in ('names', 'seqs', 'quals', 'mates'):
            assert getattr(got, field) == getattr(one, field), field
        for field in ('name_offs', 'seq_offs', 'qual_offs', 'is_fastq', 'first', 'offset', 'abund', 'mate_record', 'mate_offs'):
            assert np.array_equal(getattr(got, field), getattr(one, field)), field
    middle = text.index('\n', len(text) // 2) + 1
    with open(path, 'w') as fh:
        fh.write(text[:middle] + 'something else\n' + text[middle:])
    messages = []
    for threads in ('1', '7'):
        os.environ['KV_AUGFASTX_THREADS'] = threads
        try:
            with pytest.raises(Exception) as err:
                AnnotatedReads.from_file(path)
            messages.append(str(err.value))
        finally:
            os.environ.pop('KV_AUGFASTX_THREADS', None)
    assert messages[0] == messages[1] and 'unexpected line' in messages[0]
