"""Device ingest (kv_inflate.hip, kv_gunzip.hip, kv_fastq.hip): BGZF members and ordinary gzip streams inflated by the GPU
must give zlib's bytes, and a FASTQ file parsed on the device the same batches, record text and counts as the host parser."""
import ctypes
import gzip
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fastq_text(n, seed, read_len=100):
    from kevlar_amd import synth
    trio = synth.make_trio(100000, seed)
    reads = synth.unpack_reads(synth.sample_reads_packed(trio['proband'], n, read_len, 0.01, seed + 1), read_len)
    rng = np.random.default_rng(seed)
    quals = rng.integers(33, 74, size=(n, read_len), dtype=np.uint8)
    out = []
    for i, seq in enumerate(reads):
        out.append('@read{}/1 sample=proband\n{}\n+\n{}\n'.format(i, seq, quals[i].tobytes().decode('ascii')))
    return ''.join(out).encode('ascii')


def device_inflate(image):
    from kevlar_amd import _lib
    lib = _lib.load()
    total, members = ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(lib.kv_bgzf_text_size(image, len(image), ctypes.byref(total), ctypes.byref(members)))
    out = ctypes.create_string_buffer(total.value + 1)
    ms = ctypes.c_double()
    _lib.check(lib.kv_bgzf_inflate_host(image, len(image), out, total.value, ctypes.byref(ms)))
    return out.raw[:total.value], members.value, ms.value


@pytest.mark.parametrize('level', [1, 6, 9, 0])
def test_device_inflate_equals_zlib(hk, level, tmp_path):
    from kevlar_amd import bgzf
    text = fastq_text(40000, 3 + level)
    path = str(tmp_path / 'reads.fq.gz')
    with bgzf.BgzfWriter(path, level=level) as sink:
        sink.write(text)
    image = open(path, 'rb').read()
    assert gzip.decompress(image) == text
    got, members, ms = device_inflate(image)
    assert members == (len(text) + bgzf.BLOCK_TEXT - 1) // bgzf.BLOCK_TEXT + 1        # + the end-of-file marker
    assert got == text


def test_device_inflate_odd_members(hk):
    """fixed-Huffman members (tiny inputs), a one-byte member, long overlapping matches, incompressible bytes, empty members"""
    from kevlar_amd import bgzf
    rng = np.random.default_rng(5)
    parts = [b'A', b'ACGT' * 3, b'', b'x' * 65536, bytes(rng.integers(0, 256, 65280, dtype=np.uint8)), b'ab' * 30000 + b'c',
             bytes(rng.integers(0, 4, 60000, dtype=np.uint8)), b'\n' * 1000]
    image = b''.join(bgzf.member(p, level=lv) for p, lv in zip(parts, (6, 6, 6, 9, 6, 1, 6, 9))) + bgzf._EOF
    assert gzip.decompress(image) == b''.join(parts)
    got, members, _ = device_inflate(image)
    assert members == len(parts) + 1
    assert got == b''.join(parts)


def test_device_inflate_reports_corruption(hk):
    from kevlar_amd import _lib, bgzf
    text = fastq_text(3000, 9)
    good = bgzf.member(text[:60000]) + bgzf.member(text[60000:120000]) + bgzf._EOF
    bad = bytearray(good)
    for at in range(40, 60):
        bad[at] ^= 0x5a                     # inside the first member's deflate payload
    with pytest.raises(Exception) as err:
        device_inflate(bytes(bad))
    assert 'BGZF' in str(err.value)
    with pytest.raises(Exception):
        device_inflate(gzip.compress(text))          # plain gzip: not BGZF
    stored = bytearray(bgzf.member(text[:60000], level=0) + bgzf._EOF)
    stored[30000] ^= 0x01                   # inside a stored block: only the CRC-32 can tell
    with pytest.raises(Exception) as err:
        device_inflate(bytes(stored))
    assert 'CRC-32' in str(err.value)


def device_gunzip(image, segment=0, cap=None):
    """(text, [passes, stretches decoded, dropped, decoded again]) of a gzip image through kv_gunzip_host"""
    from kevlar_amd import _lib
    lib = _lib.load()
    cap = cap if cap is not None else len(gzip.decompress(image)) + 64
    out = ctypes.create_string_buffer(cap + 1)
    n, ms, stats = ctypes.c_uint64(), ctypes.c_double(), (ctypes.c_uint64 * 4)()
    _lib.check(lib.kv_gunzip_host(image, len(image), out, cap, segment, ctypes.byref(n), stats, ctypes.byref(ms)))
    return out.raw[:n.value], list(stats)


@pytest.mark.parametrize('level', [1, 6, 9])
@pytest.mark.parametrize('segment', [0, 1 << 20, 150000])
def test_device_gunzip_equals_zlib(hk, level, segment):
    """one DEFLATE stream (what gzip writes), whole and a segment at a time: window and bit position carried across"""
    text = fastq_text(40000, 30 + level)
    image = gzip.compress(text, compresslevel=level)
    got, stats = device_gunzip(image, segment)
    assert got == text
    assert stats[1] > 100                             # found and used block starts inside the stream
    if segment:
        assert stats[0] > 1


def test_device_gunzip_odd_streams(hk):
    """several members, every optional header field, stored and fixed-Huffman blocks, a 1000:1 stream (stretches run out
    of room and are decoded again), incompressible bytes, matches at the longest distance, tiny and empty inputs"""
    import io
    rng = np.random.default_rng(12)
    text = fastq_text(20000, 41)
    cut = len(text) // 3
    cases = {'members': gzip.compress(text[:cut], 6) + gzip.compress(text[cut:2 * cut], 1) + gzip.compress(b'', 6) + gzip.compress(text[2 * cut:], 9)}
    named = io.BytesIO()
    with gzip.GzipFile('reads of sample 1.fq', 'wb', 6, named, mtime=12345) as fh:
        fh.write(text[:700000])
    cases['name'] = named.getvalue()
    plain = gzip.compress(text[:500000], 6)
    # FEXTRA + FNAME + FCOMMENT + FHCRC written by hand around the same deflate data
    cases['all header fields'] = (plain[:3] + bytes([4 | 8 | 16 | 2]) + plain[4:10] + b'\x05\x00hello' + b'a name\x00' + b'a comment\x00' + b'\x12\x34' + plain[10:])
    cases['stored'] = gzip.compress(text[:300000], 0)
    cases['all A'] = gzip.compress(b'A' * 6000000, 6)
    cases['random'] = gzip.compress(bytes(rng.integers(0, 256, 1500000, dtype=np.uint8)), 6)
    far = bytes(rng.integers(65, 91, 32768, dtype=np.uint8))
    cases['far matches'] = gzip.compress(far * 40, 9)
    cases['tiny'] = gzip.compress(b'@r\nACGT\n+\nIIII\n', 6)
    cases['one byte'] = gzip.compress(b'x', 9)
    cases['empty'] = gzip.compress(b'', 6)
    for name, image in cases.items():
        want = gzip.decompress(image)
        for segment in (0, 1 << 19):
            got, stats = device_gunzip(image, segment)
            assert got == want, (name, segment, stats)
    assert device_gunzip(cases['all A'])[1][3] > 0             # ... decoded again with more room


def test_device_gunzip_refuses_what_it_cannot_read(hk):
    """trailing garbage, a truncated stream, damaged codes: an error (the reader then falls back to zlib), never wrong text"""
    from kevlar_amd import _lib
    text = fastq_text(20000, 43)
    image = gzip.compress(text, 6)
    bad = bytearray(image)
    for at in range(len(bad) // 2, len(bad) // 2 + 64):
        bad[at] ^= 0xa5
    for broken in (image + b'garbage behind the stream' * 3, image[:len(image) * 2 // 3], bytes(bad), b'\x1f\x8b\x08' + bytes(40)):
        try:
            got, _ = device_gunzip(broken, 0, cap=len(text) + 64)
        except (ValueError, OSError, _lib.KvError):
            continue
        # (a flipped bit inside a literal run decodes: that is what the CRC is for, which this decoder does not check)
        assert broken is not image and len(got) == len(text)


def test_damaged_gzip_is_an_error_on_both_paths(hk, tmp_path):
    """a truncated stream, and one with a byte changed where only the CRC-32 can tell (inside a stored block): the device
    path declines, zlib reports, and the reader raises instead of delivering a short or altered file"""
    text = fastq_text(20000, 44)
    whole = gzip.compress(text, 6)
    stored = bytearray(gzip.compress(text, 0))
    stored[len(stored) // 2] ^= 0x01
    with pytest.raises(ValueError) as err:
        device_gunzip(bytes(stored), 0, cap=len(text) + 64)
    assert 'CRC-32' in str(err.value)
    for name, image in (('cut.fq.gz', whole[:len(whole) * 2 // 3]), ('flipped.fq.gz', bytes(stored))):
        path = str(tmp_path / name)
        with open(path, 'wb') as fh:
            fh.write(image)
        for env in ({}, {'KV_INGEST': 'host'}):
            with pytest.raises((OSError, ValueError)):
                batches_of(hk, path, 100000, dict(env))
    good = str(tmp_path / 'good.fq.gz')
    with open(good, 'wb') as fh:
        fh.write(whole)
    assert batches_of(hk, good, 100000)[4] == 20000


def write_fastq(path, text, bgzf_level=6, kind='bgzf'):
    from kevlar_amd import bgzf
    if kind == 'gzip':
        with gzip.open(path, 'wb', compresslevel=bgzf_level) as sink:
            sink.write(text.encode('ascii') if isinstance(text, str) else text)
        return
    with bgzf.BgzfWriter(path, level=bgzf_level) as sink:
        sink.write(text)


def batches_of(hk, path, size, env=None):
    """(per batch: count tables of k=25, lengths) + every record's text, through ReadParser.text_batches"""
    env = env or {}
    os.environ.update(env)
    try:
        parser = hk.ReadParser(path)
        sketch = hk.Counttable(25, 4e6, 4)
        sizes, records, modes = [], [], []
        for tb in parser.text_batches(size):
            sizes.append(tb.n)
            modes.append(type(tb).__name__)
            sketch.consume_batch(tb.batch)
            tb.prefetch(range(0, tb.n, 7))
            for i in range(0, tb.n, 7):
                r = tb.record(i)
                records.append((r.name, r.sequence, r.quality))
            last = tb.record(tb.n - 1)
            records.append((last.name, last.sequence, last.quality))
            tb.batch.close()
        return sizes, records, [sketch.table_bytes(t) for t in range(4)], modes, parser.num_reads
    finally:
        for key in env:
            os.environ.pop(key, None)


@pytest.mark.parametrize('batch,text_mb,kind', [(100000, None, 'bgzf'), (7001, None, 'bgzf'), (100000, '1', 'bgzf'), (2500, '1', 'bgzf'),
                                                (100000, None, 'plain'), (7001, '1', 'plain'),
                                                (100000, None, 'gzip'), (7001, None, 'gzip'), (100000, '1', 'gzip'), (2500, '1', 'gzip')])
def test_device_parse_equals_host_parse(hk, tmp_path, batch, text_mb, kind):
    """BGZF FASTQ parsed on the device: same records, same packed reads (through the count tables they produce) as
    the host parser; reads with N / lower case, ragged lengths, CRLF, a last line without newline"""
    text = fastq_text(30000, 21).decode('ascii').split('\n')
    text[4 * 5 + 1] = text[4 * 5 + 1][:30] + 'N' + text[4 * 5 + 1][31:]
    text[4 * 9 + 1] = text[4 * 9 + 1].lower()
    text[4 * 11 + 1] = text[4 * 11 + 1][:20]; text[4 * 11 + 3] = text[4 * 11 + 3][:20]
    text[4 * 13 + 1] = ''; text[4 * 13 + 3] = ''
    text[4 * 17 + 1] += 'ACGTACGTAC' * 40; text[4 * 17 + 3] += 'I' * 400
    for line in range(4 * 20, 4 * 24):
        text[line] += '\r'
    blob = '\n'.join(text).rstrip('\n')              # no newline behind the last quality line
    if kind in ('bgzf', 'gzip'):                     # gzip: one DEFLATE stream, inflated a segment at a time (kv_gunzip.hip)
        path = str(tmp_path / 'reads.fq.gz')
        write_fastq(path, blob, kind=kind)
    else:                                            # an uncompressed file takes the same kernels, minus the inflate
        path = str(tmp_path / 'reads.fq')
        with open(path, 'w') as fh:
            fh.write(blob)
    host = batches_of(hk, path, batch, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, batch, {'KV_INGEST_TEXT_MB': text_mb} if text_mb else None)
    assert set(host[3]) == {'TextBatch'} and set(dev[3]) == {'DeviceTextBatch'}
    assert host[4] == dev[4] == 30000 and sum(host[0]) == sum(dev[0]) == 30000
    assert max(dev[0]) <= batch
    if text_mb is None:
        assert host[0] == dev[0]
    assert host[2] == dev[2]
    if host[0] == dev[0]:
        assert host[1] == dev[1]
    # take_batch (count's way in) and find_name
    parser = hk.ReadParser(path)
    tb = parser.text_batch(1000)
    assert tb.find_name('read777/1 sample=proband') == 777 and tb.find_name('nobody') == -1
    assert tb.record(20).sequence == text[4 * 20 + 1].rstrip('\r')


def test_device_parse_falls_back_to_host(hk, tmp_path):
    """BGZF that is not four-line FASTQ -- FASTA, or FASTQ with a blank line in the middle -- ends up on the host parser
    with nothing lost or repeated"""
    lines = fastq_text(5000, 4).decode('ascii').split('\n')
    broken = lines[:4 * 3000] + [''] + lines[4 * 3000:]
    path = str(tmp_path / 'gap.fq.gz')
    write_fastq(path, '\n'.join(broken))
    host = batches_of(hk, path, 1000, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, 1000)
    assert dev[3][0] == 'DeviceTextBatch' and dev[3][-1] == 'TextBatch'
    assert host[0] == dev[0] and host[1] == dev[1] and host[2] == dev[2] and dev[4] == 5000
    path = str(tmp_path / 'gap.plain.fq.gz')         # the same as one gzip stream
    write_fastq(path, '\n'.join(broken), kind='gzip')
    host = batches_of(hk, path, 1000, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, 1000)
    assert dev[3][0] == 'DeviceTextBatch' and dev[3][-1] == 'TextBatch'
    assert host[0] == dev[0] and host[1] == dev[1] and host[2] == dev[2] and dev[4] == 5000
    path = str(tmp_path / 'gap.fq')                  # the same, uncompressed
    with open(path, 'w') as fh:
        fh.write('\n'.join(broken))
    host = batches_of(hk, path, 1000, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, 1000)
    assert dev[3][0] == 'DeviceTextBatch' and dev[3][-1] == 'TextBatch'
    assert host[0] == dev[0] and host[1] == dev[1] and host[2] == dev[2] and dev[4] == 5000
    fasta = ''.join('>s{}\n{}\n'.format(i, lines[4 * i + 1]) for i in range(2000))
    path = str(tmp_path / 'seqs.fa.gz')
    write_fastq(path, fasta)
    host = batches_of(hk, path, 700, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, 700)
    assert set(dev[3]) == {'TextBatch'}
    assert host[:3] == dev[:3] and dev[4] == 2000


def test_novel_cli_same_output_from_bgzf_and_plain_gzip(hk, tmp_path):
    """kevlar novel end to end: reads as BGZF (one wavefront per member) and as plain gzip (one stream, inflated in
    parallel stretches) give the same augmented FASTQ as the host's zlib reader"""
    import gzip as gz
    import subprocess
    import sys
    from kevlar_amd import synth
    trio = synth.make_trio(60000, 8)
    files = {}
    for i, name in enumerate(('proband', 'mother', 'father')):
        reads = synth.unpack_reads(synth.sample_reads_packed(trio[name], 12000, 100, 0.005, 50 + i), 100)
        text = ''.join('@{}_{}\n{}\n+\n{}\n'.format(name, j, s, 'I' * len(s)) for j, s in enumerate(reads))
        files[name] = (str(tmp_path / (name + '.bgzf.fq.gz')), str(tmp_path / (name + '.plain.fq.gz')))
        write_fastq(files[name][0], text)
        with gz.open(files[name][1], 'wt') as fh:
            fh.write(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs, logs = [], []
    for which, env in ((0, {}), (1, {}), (1, {'KV_INGEST': 'host'}), (0, {'KV_PARALLEL_SAMPLES': '1'})):
        out = str(tmp_path / 'novel{}{}.augfastq'.format(which, ''.join(env)))
        cmd = [sys.executable, '-m', 'kevlar_amd', 'novel', '--case', files['proband'][which], '--control', files['mother'][which],
               '--control', files['father'][which], '--ksize', '25', '--memory', '2M', '--case-min', '5', '--ctrl-max', '1', '--out', out]
        done = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert done.returncode == 0, done.stderr[-2000:]
        outs.append(open(out).read())
        logs.append([line for line in done.stderr.split('\n') if line.startswith('[kevlar')])
    assert outs[0] == outs[1] == outs[2] == outs[3] and outs[0].count('\n') > 100
    # the samples counted side by side (KV_PARALLEL_SAMPLES=1) say what they say in the order of the one-after-the-other run

    def untimed(lines):
        import re
        return [re.sub(r'[0-9.]+ sec(onds)?', 'T sec', line) for line in lines]
    assert len(logs[0]) > 8 and untimed(logs[3]) == untimed(logs[0])


@pytest.mark.parametrize('kind', ['gzip', 'bgzf'])
def test_fasta_gz_is_inflated_on_the_device_and_parsed_on_the_host(hk, tmp_path, kind):
    """a gzip file that is not four-line FASTQ -- FASTA with wrapped sequence lines: a reference genome, contigs -- is inflated
    by the device (kv_gunzip.hip), a segment of text at a time, and its text parsed by the host's record parser
    (kevlar/__init__.py:125-128 reads it through one zlib stream): same records, same count tables as KV_INGEST=host, and the
    device inflater did run.  Segments of 1 MB of text cut records and lines anywhere."""
    import ctypes
    import gzip as gz
    from kevlar_amd import _lib
    rng = np.random.default_rng(31)
    letters = np.frombuffer(b'ACGT', dtype=np.uint8)
    recs = []
    for i in range(3000):
        n = int(rng.integers(40, 4000))
        seq = letters[rng.integers(0, 4, size=n)].tobytes().decode('ascii')
        if i % 97 == 0:
            seq = seq[:n // 2] + 'N' * 5 + seq[n // 2:]
        recs.append('>contig{} len={}\n'.format(i, len(seq)) + '\n'.join(seq[j:j + 70] for j in range(0, len(seq), 70)) + '\n')
    text = ''.join(recs)
    assert len(text) > (5 << 20)
    path = str(tmp_path / 'contigs.fa.gz')
    if kind == 'gzip':
        with gz.open(path, 'wt', compresslevel=6) as fh:
            fh.write(text)
    else:
        write_fastq(path, text)
    lib = _lib.load()
    host = batches_of(hk, path, 700, {'KV_INGEST': 'host'})
    lib.kv_prof_reset()
    lib.kv_prof_enable(1)
    try:
        dev = batches_of(hk, path, 700, {'KV_GUNZIP_TEXT_MIN_MB': '1', 'KV_INGEST_TEXT_MB': '1'})
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        # one DEFLATE stream: parallel stretches (kv_gunzip.hip); blocked gzip: a wavefront per member, CRC-32 checked (kv_inflate.hip)
        lib.kv_prof_get(b'k_gz_decode' if kind == 'gzip' else b'k_inflate', ctypes.byref(ms), ctypes.byref(n))
    finally:
        lib.kv_prof_enable(0)
    assert n.value >= 5, 'the device inflater must have decoded the segments'
    assert set(dev[3]) == {'TextBatch'} and dev[4] == host[4] == 3000
    assert dev[0] == host[0] and dev[1] == host[1] and dev[2] == host[2]
    # a damaged stream is an error on this path too (zlib takes over where the device stops and reports it)
    blob = bytearray(open(path, 'rb').read())
    mid = len(blob) // 2
    blob[mid:mid + 64] = bytes(64)
    bad = str(tmp_path / 'damaged.fa.gz')
    with open(bad, 'wb') as fh:
        fh.write(bytes(blob))
    for env in ({'KV_INGEST': 'host'}, {'KV_GUNZIP_TEXT_MIN_MB': '1'}):
        with pytest.raises(Exception):
            batches_of(hk, bad, 700, env)


@pytest.mark.parametrize('kind', ['plain', 'gzip'])
def test_novel_scans_a_kept_batch_of_a_case_file_that_ends_in_blank_lines(hk, tmp_path, kind):
    """`kevlar novel` keeps the batch a one-batch case sample was counted from and scans that (kevlar_amd/count.py keep=); it
    asks the reader once more only to learn that the file has ended.  A FASTQ file that ends in blank lines leaves a carry of a
    byte or two behind its last record: that second call must leave the kept batch fetchable (it reset n_batch
    and the hit records could not be fetched).  Reference run: KV_NOVEL_REREAD=1 reads the case file a second time."""
    import gzip as gz
    import subprocess
    import sys
    from kevlar_amd import synth
    trio = synth.make_trio(60000, 8)
    files = {}
    for i, name in enumerate(('proband', 'mother', 'father')):
        reads = synth.unpack_reads(synth.sample_reads_packed(trio[name], 12000, 100, 0.005, 50 + i), 100)
        text = ''.join('@{}_{}\n{}\n+\n{}\n'.format(name, j, s, 'I' * len(s)) for j, s in enumerate(reads))
        if name == 'proband':
            text += '\n'                               # the file ends "\n\n"
        files[name] = str(tmp_path / (name + ('.fq' if kind == 'plain' else '.fq.gz')))
        if kind == 'plain':
            with open(files[name], 'w', newline='') as fh:
                fh.write(text)
        else:
            with gz.open(files[name], 'wt', newline='') as fh:
                fh.write(text)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env in ({}, {'KV_NOVEL_REREAD': '1'}, {'KV_INGEST': 'host'}):
        out = str(tmp_path / 'novel{}.augfastq'.format(''.join(env)))
        cmd = [sys.executable, '-m', 'kevlar_amd', 'novel', '--case', files['proband'], '--control', files['mother'],
               '--control', files['father'], '--ksize', '25', '--memory', '2M', '--case-min', '5', '--ctrl-max', '1', '--out', out]
        done = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert done.returncode == 0, done.stderr[-2000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] == outs[2] and outs[0].count('\n') > 100


@pytest.mark.parametrize('tail', ['\n\n', '\n\r\n \n', '\n'])
def test_blank_lines_behind_the_last_record_stay_on_the_device(hk, tmp_path, tail):
    """a FASTQ file that ends in blank lines is still four-line FASTQ: the device reader finishes it (no hand-over to the host
    parser, which would read the whole file a second time) with the same records and counts"""
    text = fastq_text(9000, 77).decode('ascii').rstrip('\n') + tail
    path = str(tmp_path / 'tail.fq')
    with open(path, 'w', newline='') as fh:
        fh.write(text)
    host = batches_of(hk, path, 4000, {'KV_INGEST': 'host'})
    dev = batches_of(hk, path, 4000, None)
    assert set(dev[3]) == {'DeviceTextBatch'} and set(host[3]) == {'TextBatch'}
    assert host[4] == dev[4] == 9000 and dev[0] == host[0]
    assert dev[1] == host[1] and dev[2] == host[2]


@pytest.mark.parametrize('kind', ['plain', 'bgzf', 'gzip'])
def test_uploads_through_the_pinned_staging_buffers_give_the_same_reads(hk, tmp_path, kind):
    """stretches of 64 MB and more of a file reach the GPU through pinned staging buffers filled by pread threads (KvStager);
    KV_STAGE=0 copies from the file's mapping as small stretches always do: same records, same count tables -- with one, three
    and five reader threads, and with a last chunk that is not a whole one"""
    rng = np.random.default_rng(77)
    n, L = (330000, 100) if kind == 'plain' else (1300000, 100)          # 76 MB of text / ~70 MB compressed (four random bases a byte do not deflate)
    codes = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    rec = np.empty((n, 1 + 9 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord('@')
    digits = np.arange(n, dtype=np.int64)
    for d in range(9):
        rec[:, 9 - d] = 48 + digits % 10
        digits //= 10
    rec[:, 10] = 10
    rec[:, 11:11 + L] = np.frombuffer(b'ACGT', dtype=np.uint8)[codes]
    rec[:, 11 + L:14 + L] = np.frombuffer(b'\n+\n', dtype=np.uint8)
    rec[:, 14 + L:14 + 2 * L] = rng.integers(33, 74, size=(n, L), dtype=np.uint8) if kind != 'plain' else ord('I')
    rec[:, 14 + 2 * L] = 10
    text = rec.tobytes()
    if kind == 'plain':
        path = str(tmp_path / 'big.fq')
        with open(path, 'wb') as fh:
            fh.write(text)
        assert os.path.getsize(path) > (64 << 20)
    else:
        path = str(tmp_path / 'big.fq.gz')
        if kind == 'bgzf':
            from kevlar_amd import bgzf
            bgzf.write_file(path, text, level=1, threads=8)
        else:
            with gzip.open(path, 'wb', compresslevel=1) as sink:
                sink.write(text)
        assert os.path.getsize(path) > (64 << 20), os.path.getsize(path)
    del rec, text

    def tables(env):
        os.environ.update(env)
        try:
            parser = hk.ReadParser(path)
            sketch = hk.Counttable(25, 4e6, 4)
            names = []
            for tb in parser.text_batches(8000000):
                assert type(tb).__name__ == 'DeviceTextBatch'
                sketch.consume_batch(tb.batch)
                names.append(tb.record(tb.n - 1).name)
                tb.batch.close()
            return [sketch.table_bytes(t) for t in range(4)], names, parser.num_reads
        finally:
            for key in env:
                os.environ.pop(key, None)
    plain = tables({'KV_STAGE': '0'})
    assert plain[2] == n
    for threads in ('1', '3', '5'):
        assert tables({'KV_STAGE_THREADS': threads}) == plain
