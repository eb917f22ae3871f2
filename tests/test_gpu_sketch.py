"""Parity of the HIP sketch engine with the CPU oracle and the reference's golden files.
Everything here goes through the C ABI (kevlar_amd.khmer -> libkvsketch_hip.so)."""
import json
import os

import numpy as np
import pytest

from conftest import data_file, expected_file

pytestmark = pytest.mark.gpu

KINDS = ['Counttable', 'SmallCounttable', 'Nodetable', 'Countgraph', 'SmallCountgraph', 'Nodegraph']


def random_reads(seed, n, lo=20, hi=160, alphabet='ACGT'):
    rng = np.random.default_rng(seed)
    letters = np.array(list(alphabet))
    return [''.join(letters[rng.integers(0, len(letters), size=int(rng.integers(lo, hi + 1)))]) for _ in range(n)]


def test_library_loaded_and_device_present(hk):
    from kevlar_amd import _lib
    lib = _lib.load()
    assert os.path.basename(_lib.LIBPATH) == 'libkvsketch_hip.so'
    _lib.require_device()
    assert b'gfx950' in lib.kv_version()


def test_every_host_thread_is_switched_to_the_process_device(hk):
    """One process per GPU: kv_set_device(LOCAL_RANK) names the GPU, and HIP's current device is per host thread, 0 until said
    otherwise -- a worker thread of rank 3 (samples counted side by side, the readers of `kevlar count --threads`) must not allocate
    and launch on GPU 0.  The library switches every thread that enters it; on a one-GPU box the device is 0 either way, so what is
    held here is that a fresh thread HAS been switched (-1 before its first call, the process's device after) by each of the calls
    such a thread makes first."""
    import ctypes
    import threading
    from kevlar_amd import _lib
    lib = _lib.load()
    _lib.require_device()

    def probe():
        c, t, h = ctypes.c_int(-7), ctypes.c_int(-7), ctypes.c_int(-7)
        _lib.check(lib.kv_thread_device_get(ctypes.byref(c), ctypes.byref(t), ctypes.byref(h)))
        return c.value, t.value, h.value
    configured = probe()[0]
    assert configured == int(os.environ.get('LOCAL_RANK', '0'))
    seen = {}

    def first_call_is(name, call):
        def work():
            before = probe()
            call()
            seen[name] = (before, probe())
        th = threading.Thread(target=work)
        th.start(); th.join()
    first_call_is('sketch', lambda: hk.Counttable(21, 1e5, 4))
    first_call_is('stream', lambda: hk.Stream().bind())
    first_call_is('reads', lambda: hk.ReadBatch(['ACGTACGTACGTACGTACGTACGTACGT']))
    # the threads run_concurrently starts (the thread that calls it may touch nothing itself when the streams come from the pool)
    def job(name):
        def work():
            before = probe()
            hk.Counttable(21, 1e5, 4).n_occupied()
            seen[name] = (before[:2] + (None,), probe())
            return 0
        return work
    hk.run_concurrently([job('concurrent-a'), job('concurrent-b')])
    assert len(seen) == 5
    for name, (before, after) in seen.items():
        if not name.startswith('concurrent'):                             # (run_concurrently binds a stream before the job starts: switched already)
            assert before[:2] == (configured, -1), (name, before)         # a fresh thread: the library has not switched it yet
        assert after == (configured, configured, configured), (name, after)


@pytest.mark.parametrize('k', [1, 7, 15, 16, 17, 21, 25, 31, 32, 33, 47, 48, 51, 64, 65, 100])
def test_device_kmer_hashing_matches_oracle(hk, ok, k):
    reads = random_reads(k, 40, lo=k, hi=k + 60)
    kmers = [r[i:i + k] for r in reads for i in range(len(r) - k + 1)]
    dev = hk.Counttable(k, 1e4, 2)
    ref = ok.Counttable(k, 1e4, 2)
    got = dev.hash_kmers(kmers)
    want = np.array([ref.hash(km) for km in kmers], dtype=np.uint64)
    assert np.array_equal(got, want)
    assert dev.hash(kmers[0]) == int(want[0])           # host-side single k-mer hash
    if k <= 32:
        devg, refg = hk.Countgraph(k, 1e4, 2), ok.Countgraph(k, 1e4, 2)
        assert np.array_equal(devg.hash_kmers(kmers), np.array([refg.hash(km) for km in kmers], dtype=np.uint64))


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('k,tablesize', [(21, 5e3), (31, 4e4)])
def test_consume_tables_bit_exact(hk, ok, kind, k, tablesize):
    """The tile kernel (LDS-staged murmur + saturating atomics) == scalar oracle, byte for byte,
    for every storage width and both hash families; small tables force saturation."""
    reads = random_reads(11, 700, lo=10, hi=150) + ['A' * 150] * 30 + ['ACGT' * 30] * 5 + ['', 'ACG']
    dev = getattr(hk, kind)(k, tablesize, 4)
    ref = getattr(ok, kind)(k, tablesize, 4)
    assert dev.hashsizes() == ref.hashsizes()
    n_dev = dev.consume_batch(hk.ReadBatch(reads))
    bases, offs = ok.concat_reads(reads)
    n_ref = ok.consume_reads(ref, bases, offs, len(reads))
    assert n_dev == n_ref
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert dev.n_occupied() == ref.n_occupied()


def test_consume_ambiguous_and_lowercase_bases(hk, ok):
    reads = ['ACGTNNACGTTTGACCAGTACGATCAGTACGATCGATCGATCGACTAGCTAGCTAGC',
             'acgtacgtagctagctagcatcgatcgatcgatcgatcagctagctagctagctagct',
             'ACGTRYACGTAGCTAGCATCGATGCATGCATCGATCGATCGATCGATGCATGCATGCA']
    dev, ref = hk.Counttable(21, 1e4, 4), ok.Counttable(21, 1e4, 4)
    dev.consume_batch(hk.ReadBatch(reads))
    for r in reads:
        ref.consume(r)
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)


@pytest.mark.parametrize('nbands,band', [(2, 0), (2, 1), (16, 6), (9, 8)])
def test_consume_banding(hk, ok, nbands, band):
    reads = random_reads(5, 400, lo=40, hi=120)
    dev, ref = hk.Counttable(25, 2e4, 4), ok.Counttable(25, 2e4, 4)
    bases, offs = ok.concat_reads(reads)
    assert dev.consume_batch(hk.ReadBatch(reads), nbands, band) == \
        ok.consume_reads(ref, bases, offs, len(reads), nbands, band)
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)


@pytest.mark.parametrize('consume_masked', [False, True])
@pytest.mark.parametrize('maskkind', ['Nodetable', 'Counttable'])
def test_consume_with_mask(hk, ok, consume_masked, maskkind):
    reads = random_reads(6, 300, lo=50, hi=120)
    maskreads = reads[:60]
    dmask, rmask = getattr(hk, maskkind)(21, 3e4, 4), getattr(ok, maskkind)(21, 3e4, 4)
    dmask.consume_batch(hk.ReadBatch(maskreads))
    for r in maskreads:
        rmask.consume(r)
    threshold = 1 if consume_masked else 0
    dev, ref = hk.SmallCounttable(21, 2e4, 4), ok.SmallCounttable(21, 2e4, 4)
    bases, offs = ok.concat_reads(reads)
    n_dev = dev.consume_batch(hk.ReadBatch(reads), 0, 0, dmask, threshold, consume_masked)
    n_ref = ok.consume_reads(ref, bases, offs, len(reads), 0, 0, rmask, threshold, consume_masked)
    assert n_dev == n_ref and n_dev > 0
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)


def test_get_add_point_queries(hk, ok):
    reads = random_reads(8, 200, lo=60, hi=100)
    dev, ref = hk.Counttable(27, 1e6, 4), ok.Counttable(27, 1e6, 4)
    dev.consume_batch(hk.ReadBatch(reads))
    for r in reads:
        ref.consume(r)
    kmers = [r[i:i + 27] for r in reads[:20] for i in range(0, len(r) - 27 + 1, 5)] + ['GATTACA' * 3 + 'GATTAC']
    hashes = dev.hash_kmers(kmers)
    assert dev.get_hashes(hashes).tolist() == [ref.get(km) for km in kmers]
    assert dev.get(kmers[3]) == ref.get(kmers[3])
    new_dev = dev.add_hashes(hashes)
    for km in kmers:
        ref.add(km)
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert int(new_dev.sum()) >= 1          # the GATTACA k-mer is new
    assert dev.get_kmers('ACGTACGTACGTACGTACGTACGTACGTACG') == ref.get_kmers('ACGTACGTACGTACGTACGTACGTACGTACG')


@pytest.mark.parametrize('filename,cls,testkmer', [
    ('test.countgraph', 'Countgraph', 'TGGAACCGGCAACGACGAAAA'),
    ('test.smallcountgraph', 'SmallCountgraph', 'CTGTACTACAGCTACTACAGT'),
    ('test.counttable', 'Counttable', 'CCTGATATCCGGAATCTTAGC'),
    ('test.smallcounttable', 'SmallCounttable', 'GGGCCCCCATCTCTATCTTGC'),
    ('test.nodegraph', 'Nodegraph', 'GGGAACTTACCTGGGGGTGCG'),
    ('test.nodetable', 'Nodetable', 'CTGTTCGATATGAGGAATCTG'),
])
def test_sketch_load_save(hk, tmp_path, filename, cls, testkmer):
    """kevlar/tests/test_sketch.py:17-29 through kevlar_amd.sketch.load + byte-exact save."""
    import kevlar_amd
    infile = data_file(filename)
    if '.' + filename.split('.')[-1] in kevlar_amd.sketch.sketch_loader_by_filename_extension:
        sketch = kevlar_amd.sketch.load(infile)
        assert type(sketch).__name__ == cls
    sketch = getattr(hk, cls).load(infile)
    assert sketch.get(testkmer) > 0
    assert sketch.get('GATTACA' * 3) == 0
    out = str(tmp_path / filename)
    sketch.save(out)
    assert open(out, 'rb').read() == open(infile, 'rb').read()


def test_sketch_load_errors(hk):
    import kevlar_amd
    with pytest.raises(kevlar_amd.sketch.KevlarSketchTypeError, match='sketch type from filename'):
        kevlar_amd.sketch.load(data_file('test.notasketchtype'))
    with pytest.raises(ValueError):
        hk.Nodetable.load(data_file('test.counttable'))        # storage type mismatch
    with pytest.raises(OSError):
        hk.Counttable.load(data_file('does-not-exist.ct'))
    with pytest.raises(ValueError, match='not implemented'):
        hk.Counttable(35, 1e4, 4).reverse_hash(12345)
    g = hk.Countgraph(21, 1e4, 4)
    kmer = 'GCATAGTGTCTCTGCTGCGCA'
    assert kevlar_amd.same_seq(g.reverse_hash(g.hash(kmer)), kmer)
    with pytest.raises(ValueError):
        hk.Countgraph(35, 1e4, 4)


@pytest.mark.parametrize('infile,testout,numbands,band,kmers_stored', [
    ('case', 'case', 0, 0, 973),
    ('ctrl1', 'ctrl1', 0, 0, 973),
    ('ctrl2', 'ctrl2', 0, 0, 966),
    ('case', 'case-band-2-1', 2, 1, 501),
    ('case', 'case-band-16-7', 16, 7, 68),
])
def test_count_cli_golden_bytes(hk, tmp_path, kevlar_log, infile, testout, numbands, band, kmers_stored):
    """The reference's own golden test (kevlar/tests/test_count.py:45-68) against the HIP path:
    `kevlar count` output file bytes == committed .ct, and the asserted log lines."""
    import kevlar_amd
    out = str(tmp_path / 'out')
    arglist = ['count', '--ksize', '25', '--memory', '10K', '--num-bands', str(numbands), '--band', str(band),
               out, data_file('simple-genome-{}-reads.fa.gz'.format(infile))]
    args = kevlar_amd.cli.parser().parse_args(arglist)
    kevlar_amd.count.main(args)
    log = kevlar_log.getvalue()
    assert '600 reads processed' in log
    assert '{:d} distinct k-mers stored'.format(kmers_stored) in log
    with open(out + '.counttable', 'rb') as f1, open(data_file('simple-genome-{}.ct'.format(testout)), 'rb') as f2:
        assert f1.read() == f2.read()
    manifest = json.load(open(expected_file('manifest.json')))
    for line in manifest['cases']['count-' + testout]:
        assert line in log


def test_count_cli_with_mask_known_answer(hk, tmp_path, kevlar_log):
    """kevlar/tests/test_count.py:150-166: '36898 distinct k-mers stored'."""
    import kevlar_amd
    mask = hk.Nodetable(21, 1e4, 4)
    mask.consume('CACCAATCCGTACGGAGAGCCGTATATATAGACTGCTATACTATTGGATCGTACGGGGC')
    maskfile, countfile = str(tmp_path / 'mask.nt'), str(tmp_path / 'counts.sct')
    mask.save(maskfile)
    args = kevlar_amd.cli.parser().parse_args(['count', '--ksize', '21', '--mask', maskfile, '--memory', '1M',
                                               countfile, data_file('bogus-genome/refr.fa')])
    kevlar_amd.count.main(args)
    assert '36898 distinct k-mers stored' in kevlar_log.getvalue()


@pytest.mark.parametrize('count,smallcount,count_masked,kpresent,kabsent', [
    (True, True, True, 'CACCAATCCGTACGGAGAGCC', 'GAATCGGTGGCTGGTTGCCGT'),
    (True, False, True, 'CACCAATCCGTACGGAGAGCC', 'GAATCGGTGGCTGGTTGCCGT'),
    (False, False, True, 'CACCAATCCGTACGGAGAGCC', 'GAATCGGTGGCTGGTTGCCGT'),
    (True, True, False, 'GAATCGGTGGCTGGTTGCCGT', 'CACCAATCCGTACGGAGAGCC'),
    (True, False, False, 'GAATCGGTGGCTGGTTGCCGT', 'CACCAATCCGTACGGAGAGCC'),
    (False, False, False, 'GAATCGGTGGCTGGTTGCCGT', 'CACCAATCCGTACGGAGAGCC'),
])
def test_load_sample_seqfile_withmask(hk, kevlar_log, count, smallcount, count_masked, kpresent, kabsent):
    """kevlar/tests/test_count.py:130-147."""
    import kevlar_amd
    mask = hk.Nodetable(21, 1e4, 4)
    mask.consume('CACCAATCCGTACGGAGAGCCGTATATATAGACTGCTATACTATTGGATCGTACGGGGC')
    sketch = kevlar_amd.count.load_sample_seqfile(
        [data_file('bogus-genome/refr.fa')], 21, 1e6, mask=mask, consume_masked=count_masked, count=count,
        smallcount=smallcount)
    assert sketch.get(kpresent) > 0
    assert sketch.get(kabsent) == 0
    assert sketch.get('GATTACAGATTACAGATTACA') == 0


def test_count_problematic_and_threads(hk, tmp_path, kevlar_log):
    """kevlar/tests/test_count.py:71-105."""
    import kevlar_amd
    args = kevlar_amd.cli.parser().parse_args(['count', '--ksize', '21', '--memory', '200K', '--band', '2',
                                               'bogusoutput', data_file('trio1/ctrl1.fq.gz')])
    with pytest.raises(ValueError, match='Must specify --num-bands and --band together'):
        kevlar_amd.count.main(args)
    args = kevlar_amd.cli.parser().parse_args(['count', '--ksize', '21', '--memory', '97', 'bogusoutput',
                                               data_file('trio1/ctrl1.fq.gz')])
    with pytest.raises(kevlar_amd.sketch.KevlarUnsuitableFPRError):
        kevlar_amd.count.main(args)
    out = str(tmp_path / 'thr.counttable')
    args = kevlar_amd.cli.parser().parse_args(['count', '--ksize', '19', '--memory', '500K', '--threads', '2', out,
                                               data_file('trio1/case1.fq.gz')])
    kevlar_amd.count.main(args)
    one = kevlar_amd.count.load_sample_seqfile([data_file('trio1/case1.fq.gz')], 19, 5e5)
    two = hk.Counttable.load(out)
    for t in range(4):                      # threading never changes the tables
        assert one.table_bytes(t) == two.table_bytes(t)


def test_memory_to_table_size(hk, kevlar_log):
    """kevlar/tests/test_count.py:169-182."""
    import kevlar_amd
    for count, smallcount, sketchtype in [(False, False, 'nodegraph'), (True, False, 'countgraph'),
                                          (True, True, 'smallcountgraph')]:
        sketch = kevlar_amd.count.load_sample_seqfile([data_file('bogus-genome/refr.fa')], 21, 2e6, count=count,
                                                      smallcount=smallcount)
        actual = sum(sketch.hashsizes()) / hk._buckets_per_byte[sketchtype]
        assert actual / 2e6 == pytest.approx(1.0, rel=1e-4)


def test_exact_unique_matches_single_thread_oracle(hk, ok):
    """n_unique_kmers() with tracking == the value one khmer thread computes in file order."""
    reads = random_reads(21, 1500, lo=30, hi=140) * 2 + random_reads(22, 300, lo=30, hi=140)
    for nbands, band in [(0, 0), (4, 2)]:
        dev, ref = hk.Counttable(23, 6e3, 4), ok.Counttable(23, 6e3, 4)   # small tables: many collisions
        dev.track_exact_unique(True)
        half = len(reads) // 2
        dev.consume_batch(hk.ReadBatch(reads[:half]), nbands, band)
        dev.consume_batch(hk.ReadBatch(reads[half:]), nbands, band)
        bases, offs = ok.concat_reads(reads)
        ok.consume_reads(ref, bases, offs, len(reads), nbands, band)
        assert dev.n_unique_kmers() == ref.n_unique_kmers()
        assert dev.n_occupied() == ref.n_occupied()
    # batch by batch against the tables as they stand (kv_unique_new: nothing is kept resident, so no size limit): many small batches,
    # through every count path, on top of k-mers that were counted before tracking began, with a mask
    mask_dev, mask_ref = hk.Nodetable(23, 4e4, 2), ok.Nodetable(23, 4e4, 2)
    mseqs = random_reads(23, 200, lo=40, hi=90)
    mask_dev.consume_batch(hk.ReadBatch(mseqs))
    mb, mo = ok.concat_reads(mseqs)
    ok.consume_reads(mask_ref, mb, mo, len(mseqs))
    for path in ('atomic', 'binned', 'skm'):
        os.environ['KV_COUNT_PATH'] = path
        try:
            for use_mask in (False, True):
                dev, ref = hk.Counttable(23, 2e4, 4), ok.Counttable(23, 2e4, 4)
                before = random_reads(24, 400, lo=30, hi=140)
                dev.consume_batch(hk.ReadBatch(before))                         # (not tracked: the tables are not empty when tracking starts)
                bases, offs = ok.concat_reads(before)
                ok.consume_reads(ref, bases, offs, len(before))
                base_unique = ref.n_unique_kmers()
                dev.track_exact_unique(True)
                step = 211
                for lo in range(0, len(reads), step):
                    part = reads[lo:lo + step]
                    dev.consume_batch(hk.ReadBatch(part), 0, 0, mask_dev if use_mask else None, 0, False)
                    bases, offs = ok.concat_reads(part)
                    ok.consume_reads(ref, bases, offs, len(part), 0, 0, mask_ref if use_mask else None, 0, False)
                assert dev.n_unique_kmers() == ref.n_unique_kmers() - base_unique, (path, use_mask)
                for t in range(4):
                    assert dev.table_bytes(t) == ref.table_bytes(t)
        finally:
            os.environ.pop('KV_COUNT_PATH', None)


def test_long_sequences_are_cut_into_segment_tiles(hk, ok):
    """FASTA records of tens of kilobases (the reference counts bogus-genome/refr.fa, 3 x 12 kb) do not
    fit one tile: they become segment tiles (more cases in test_gpu_longreads.py)."""
    rng = np.random.default_rng(4)
    letters = np.array(list('ACGT'))
    reads = [''.join(letters[rng.integers(0, 4, size=n)]) for n in (12345, 100, 40000, 31, 30, 48999, 250)]
    for path in ('atomic', 'binned'):
        os.environ['KV_COUNT_PATH'] = path
        try:
            dev, ref = hk.Counttable(31, 3e5, 4), ok.Counttable(31, 3e5, 4)
            n_dev = dev.consume_batch(hk.ReadBatch(reads))
        finally:
            os.environ.pop('KV_COUNT_PATH', None)
        bases, offs = ok.concat_reads(reads)
        assert n_dev == ok.consume_reads(ref, bases, offs, len(reads))
        for t in range(4):
            assert dev.table_bytes(t) == ref.table_bytes(t)
    r, o, a, _ = hk.novel_scan([dev], [], hk.ReadBatch(reads), 1, 0)
    hits, _ = ok.novel_scan([ref], [], bases, offs, len(reads), 31, 1, 0)
    assert [(int(x), int(y)) for x, y in zip(r, o)] == [(h[0], h[1]) for h in hits]
    assert hk.ReadBatch(['A' * 60000]).num_kmers(31) == 60000 - 30        # no length limit any more


@pytest.mark.parametrize('kind,path', [('Counttable', 'skm'), ('Counttable', 'binned'), ('Counttable', 'atomic'),
                                       ('SmallCounttable', 'skm'), ('SmallCounttable', 'binned'), ('Nodetable', 'binned'),
                                       ('Nodetable', 'skm'), ('Countgraph', 'binned')])
def test_clear_is_lazy_but_invisible(hk, kind, path, tmp_path):
    """kv_sketch_clear defers the zeroing to the next partitioned count; whatever touches the tables first -- a big
    count, a small one, point queries, table reads, save, a scan, use as a mask -- must see zeroed tables"""
    from kevlar_amd import synth
    k = 25
    trio = synth.make_trio(120000, 77)
    reads = synth.unpack_reads(synth.sample_reads_packed(trio['proband'], 60000, 100, 0.01, 3), 100)
    junk = synth.unpack_reads(synth.sample_reads_packed(trio['mother'], 60000, 100, 0.01, 4), 100)
    big, small, dirty = hk.ReadBatch(reads), hk.ReadBatch(reads[:50]), hk.ReadBatch(junk)
    fresh = getattr(hk, kind)(k, 1.5e6, 4)
    os.environ['KV_COUNT_PATH'] = path
    try:
        fresh.consume_batch(big)
        want = [fresh.table_bytes(t) for t in range(4)]
        used = getattr(hk, kind)(k, 1.5e6, 4)
        for first in ('count', 'get', 'table', 'save', 'small', 'mask', 'scan'):
            used.consume_batch(dirty)
            used.clear()
            assert used.n_occupied() == 0 and used.n_unique_kmers() == 0
            if first == 'get':
                assert used.get(reads[0][:k]) == 0
                assert not used.get_hashes(used.hash_kmers([r[:k] for r in junk[:500]])).any()
            elif first == 'table':
                assert not any(used.table_bytes(t).strip(b'\x00') for t in range(4))
            elif first == 'save':
                out = str(tmp_path / 'empty.ct')
                used.save(out)
                back = getattr(hk, kind).load(out)
                assert not any(back.table_bytes(t).strip(b'\x00') for t in range(4))
            elif first == 'small':
                used.consume_batch(small)
                ref = getattr(hk, kind)(k, 1.5e6, 4)
                ref.consume_batch(small)
                assert [used.table_bytes(t) for t in range(4)] == [ref.table_bytes(t) for t in range(4)]
                used.clear()
            elif first == 'mask':
                other = getattr(hk, kind)(k, 1.5e6, 4)
                other.consume_batch(big, 0, 0, used)
                assert [other.table_bytes(t) for t in range(4)] == want          # an empty mask hides nothing
            elif first == 'scan':
                r, o, a, _ = hk.novel_scan([fresh], [used], hk.ReadBatch(reads[:2000]), 1, 0)
                assert len(r) > 0 and not a[:, 1].any()                        # the cleared control holds nothing
            used.consume_batch(big)
            assert [used.table_bytes(t) for t in range(4)] == want, first
            assert used.n_occupied() == fresh.n_occupied()
    finally:
        os.environ.pop('KV_COUNT_PATH', None)


def test_table_buffers_of_a_destroyed_sketch_serve_the_next_one_zeroed(hk):
    """kv_sketch_destroy keeps table buffers for the next sketch of the same geometry (a 2 GB hipMalloc can take 0.2 s and stalls
    every other thread's HIP call): the new sketch starts empty all the same, and KV_TABLE_CACHE_GB=0 gives the buffers back"""
    import ctypes
    import gc
    from kevlar_amd import _lib
    lib = _lib.load()

    def pointers(sketch):
        out = []
        for t in range(4):
            p, n = ctypes.c_void_p(), ctypes.c_uint64()
            _lib.check(lib.kv_sketch_table_devptr(sketch._h, t, ctypes.byref(p), ctypes.byref(n)))
            out.append(p.value)
        return out
    first = hk.Counttable(21, 3e7, 4)                 # 30 MB tables: above the 16 MB floor of the cache
    first.add('ACGTACGTACGTACGTACGTA')
    assert first.get('ACGTACGTACGTACGTACGTA') == 1
    held = pointers(first)
    del first
    gc.collect()
    second = hk.Counttable(21, 3e7, 4)
    assert sorted(pointers(second)) == sorted(held)
    assert second.get('ACGTACGTACGTACGTACGTA') == 0 and second.n_occupied() == 0
    os.environ['KV_TABLE_CACHE_GB'] = '0'
    try:
        del second
        gc.collect()
    finally:
        os.environ.pop('KV_TABLE_CACHE_GB')
