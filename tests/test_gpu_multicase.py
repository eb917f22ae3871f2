"""kmer_is_interesting() with more than one case sample and with no control at all (kevlar/novel.py:36-51; the reference's own
test is kevlar/tests/test_novel.py:108-144, `test_novel_two_cases`): the k-mer must reach case_min in EVERY case sample --
the loop leaves at the first that does not -- and stay at or below ctrl_max in every control, of which there may be none
(`novel()` called with controlcounts=[]; the CLI cannot ask for that, the API can).  Every scan kernel the product has --
the scan from the count pass's distinct list (k_skm_novel_list), the walk over re-built buckets (k_skm_novel), the per-k-mer
tile scans (k_novel_mark, k_novel_mark_2bit) -- against the oracle's literal restatement of that loop, hit for hit,
with the launch counts that prove which kernel answered."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KNOBS = ('KV_COUNT_PATH', 'KV_NOVEL_PATH', 'KV_SKM_BUCKET_KMERS', 'KV_SKM_DL', 'KV_NOVEL_2BIT', 'KV_NOVEL_BITS')


def launches(name):
    from kevlar_amd import _lib
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    _lib.load().kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return n.value


@pytest.fixture
def prof():
    from kevlar_amd import _lib
    lib = _lib.load()
    lib.kv_prof_reset()
    lib.kv_prof_enable(1)
    yield lib
    lib.kv_prof_enable(0)
    _ARRAYS[0] = False
    for name in KNOBS:
        os.environ.pop(name, None)


_ARRAYS = [False]          # run_scan returns numpy arrays instead of a tuple per hit (the no-control cases have millions of hits)


def family(genome_len, n_reads, read_len, seed):
    """two affected siblings and their parents: case 1 carries every de novo variant of the synthetic proband, case 2 shares only
    the ones on the proband's first haplotype (its other haplotype is the father's second), so a good part of what case 1 alone
    would report is rejected by the second pass of the case loop"""
    from kevlar_amd import synth
    trio = synth.make_trio(genome_len, seed, inherited_per_mb=400, denovo_per_mb=600)
    haps = {'case1': trio['proband'], 'case2': (trio['proband'][0], trio['father'][1]),
            'mother': trio['mother'], 'father': trio['father']}
    words = {s: synth.sample_reads_packed(haps[s], n_reads, read_len, 0.005, seed + 11 + i) for i, s in enumerate(haps)}
    reads = {s: synth.unpack_reads(words[s], read_len) for s in haps}
    return words, reads


def as_tuples(r, o, a):
    return [(int(r[i]), int(o[i]), tuple(int(x) for x in a[i])) for i in range(len(r))]


def run_scan(hk, cases, ctrls, batch, case_min, ctrl_max, path, **kw):
    """one scan with the path asked for by name; returns the hits and the launches of each scan kernel"""
    from kevlar_amd import _lib
    lib = _lib.load()
    if path in ('list', 'walk'):
        os.environ['KV_NOVEL_PATH'] = 'skm'
    else:
        # 'tiles' keeps the tile kernel (k_novel_mark); any other name than 'skm' leaves the choice among the per-k-mer kernels to
        # the library, which takes the 2-bit one for equal-length reads (kv_novel.hip, scan_reads)
        os.environ['KV_NOVEL_PATH'] = 'tiles' if path == 'tiles' else 'per-kmer'
    lib.kv_prof_reset()
    try:
        r, o, a, disc = hk.novel_scan(cases, ctrls, batch, case_min, ctrl_max, **kw)
    finally:
        os.environ.pop('KV_NOVEL_PATH', None)
    ran = {name: launches(name) for name in ('k_skm_novel_list', 'k_skm_novel', 'k_novel_mark', 'k_novel_mark_2bit')}
    if _ARRAYS[0]:
        return (np.asarray(r, dtype=np.uint32), np.asarray(o, dtype=np.uint32), np.asarray(a, dtype=np.uint8)), sorted(disc.tolist()), ran
    return as_tuples(r, o, a), sorted(disc.tolist()), ran


def same_hits(got, want):
    """hit arrays (read, offset, abundances) against the oracle's, without a Python tuple per hit (no control: millions of hits)"""
    return len(got[0]) == len(want[0]) and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1].astype(np.uint32)) and np.array_equal(got[2], want[2])


@pytest.mark.parametrize('k', [19, 31, 51])
@pytest.mark.parametrize('nctrl', [2, 0])
def test_two_cases_every_scan_kernel_equals_the_oracle(hk, ok, prof, k, nctrl):
    os.environ['KV_COUNT_PATH'] = 'skm'
    os.environ['KV_SKM_BUCKET_KMERS'] = '4096'
    os.environ['KV_SKM_DL'] = '1'
    read_len, n = 100, 24000
    words, reads = family(120000, n, read_len, 500 + k)
    names = ['mother', 'father'][:nctrl] + ['case1', 'case2']            # controls first, the cases last (kevlar/novel.py:197-216)
    case_min, ctrl_max = 5, 1
    oracle_hits = {}
    _ARRAYS[0] = True               # this test compares arrays (run_scan below; the other tests of the module keep their tuples)
    for hint in (True, False):
        dev = {s: hk.Counttable(k, 1.6e6, 4) for s in names}
        ref = {s: ok.Counttable(k, 1.6e6, 4) for s in names}
        batches = {s: hk.ReadBatch.from_packed(words[s], read_len) for s in names}
        if hint:
            dev['case1'].expect_scan()
            dev['case2'].expect_scan()
        for s in names:
            assert dev[s].consume_batch(batches[s]) == n * (read_len - k + 1)
            bases, offs = ok.concat_reads(reads[s])
            ok.consume_reads(ref[s], bases, offs, n)
            for t in range(4):
                assert dev[s].table_bytes(t) == ref[s].table_bytes(t)
        cases, ctrls = [dev['case1'], dev['case2']], [dev[s] for s in names[:nctrl]]
        rcases, rctrls = [ref['case1'], ref['case2']], [ref[s] for s in names[:nctrl]]
        # the reference scans the reads of ALL case samples, in file order, against all sketches (kevlar/novel.py:215-223)
        for which in ('case2', 'case1'):
            if which not in oracle_hits:
                bases, offs = ok.concat_reads(reads[which])
                oracle_hits[which] = ok.novel_scan_mt(rcases, rctrls, bases, offs, n, k, case_min, ctrl_max, 4)
            want = oracle_hits[which]
            assert len(want[0]) > (1000 if nctrl == 0 else 10)
            assert (want[2][:, :2] >= case_min).all() and (want[2][:, 2:] <= ctrl_max).all()
            # the batch counted last on the stream is the one whose buckets (and, with the hint, distinct list) are still there
            first = 'list' if (hint and which == 'case2') else 'walk'
            got, _, ran = run_scan(hk, cases, ctrls, batches[which], case_min, ctrl_max, first)
            if first == 'list':
                assert ran['k_skm_novel_list'] == 1 and ran['k_skm_novel'] == 0, ran
            else:
                assert ran['k_skm_novel'] == 1 and ran['k_skm_novel_list'] == 0, ran
            assert same_hits(got, want), '{} scan of {} (hint {}): {} hits, the oracle has {}'.format(first, which, hint, len(got[0]), len(want[0]))
            got, _, ran = run_scan(hk, cases, ctrls, batches[which], case_min, ctrl_max, 'walk')
            assert ran['k_skm_novel'] == 1 and ran['k_skm_novel_list'] == 0, ran
            assert same_hits(got, want), 'walk'
            got, _, ran = run_scan(hk, cases, ctrls, batches[which], case_min, ctrl_max, 'tiles')
            assert ran['k_novel_mark'] == 1 and ran['k_skm_novel'] + ran['k_skm_novel_list'] + ran['k_novel_mark_2bit'] == 0, ran
            assert same_hits(got, want), 'tiles'
            got, _, ran = run_scan(hk, cases, ctrls, batches[which], case_min, ctrl_max, 'tiles2bit')
            assert ran['k_novel_mark_2bit'] == 1 and ran['k_novel_mark'] == 0, ran
            assert same_hits(got, want), '2-bit tiles'
        if not hint:
            # the probe of the first case sample's table 0 through the bit map (k_case_bits) and through the table itself
            os.environ['KV_NOVEL_BITS'] = '0'
            dev['case2'].expect_scan()
            dev['case2'].clear()
            dev['case2'].consume_batch(batches['case2'])
            got, _, ran = run_scan(hk, cases, ctrls, batches['case2'], case_min, ctrl_max, 'list')
            assert ran['k_skm_novel_list'] == 1, ran
            assert same_hits(got, oracle_hits['case2'])
            os.environ.pop('KV_NOVEL_BITS')
    _ARRAYS[0] = False


def test_two_cases_screen_bands_and_order_of_cases(hk, ok, prof):
    """the abundance screen drops a read at the first case sample that falls below it -- which case that is depends on the ORDER of
    the cases (kevlar/novel.py:36-44) --, bands in both rules, and the abundances are reported cases first in the order given"""
    k, read_len, n = 25, 100, 9000
    words, reads = family(60000, n, read_len, 77)
    names = ['mother', 'father', 'case1', 'case2']
    dev = {s: hk.Counttable(k, 8e5, 4) for s in names}
    ref = {s: ok.Counttable(k, 8e5, 4) for s in names}
    batches = {s: hk.ReadBatch.from_packed(words[s], read_len) for s in names}
    for s in names:
        dev[s].consume_batch(batches[s])
        bases, offs = ok.concat_reads(reads[s])
        ok.consume_reads(ref[s], bases, offs, n)
    bases, offs = ok.concat_reads(reads['case1'])
    seen = set()
    for order in (('case1', 'case2'), ('case2', 'case1')):
        for nctrl in (2, 1, 0):
            for screen, band_mode, nbands, band in [(0, 0, 0, 0), (2, 0, 0, 0), (3, 0, 0, 0), (0, 1, 4, 2), (0, 2, 4, 1), (2, 1, 2, 1)]:
                cases, rcases = [dev[s] for s in order], [ref[s] for s in order]
                ctrls, rctrls = [dev[s] for s in names[:nctrl]], [ref[s] for s in names[:nctrl]]
                want, status = ok.novel_scan(rcases, rctrls, bases, offs, n, k, 5, 1, screen=screen, band_mode=band_mode, nbands=nbands,
                                             band=band, cap=1 << 21)
                dropped = [i for i, s in enumerate(status) if s == 2]
                for path in (('tiles',) if screen else ('walk', 'tiles', 'tiles2bit')):
                    got, disc, ran = run_scan(hk, cases, ctrls, batches['case1'], 5, 1, path, screen=screen, band_mode=band_mode,
                                              nbands=nbands, band=band)
                    assert got == want, (order, nctrl, screen, band_mode, path)
                    assert disc == dropped
                    seen.add((path, ran['k_skm_novel'], ran['k_novel_mark'], ran['k_novel_mark_2bit']))
                if screen == 0 and band_mode == 0:
                    assert len(want) > 0
    assert ('walk', 1, 0, 0) in seen and ('tiles', 0, 1, 0) in seen and ('tiles2bit', 0, 0, 1) in seen


def test_two_case_golden_from_the_reference(hk):
    """kevlar/tests/test_novel.py:108-144 through this build's CLI: two case samples (trio1/case6.fq, case6b.fq), two controls, counted
    by `kevlar count`, scanned from the saved tables; byte for byte what the reference's own drivers wrote over the oracle
    (tests/golden/make_golden.py), and the reference test's own assertion on every annotation line"""
    import re
    import tempfile
    from conftest import data_file, expected_file
    from test_gpu_pipeline import run_cli
    with tempfile.TemporaryDirectory() as tmp:
        inputs = [data_file('trio1/case6.fq.gz'), data_file('trio1/case6b.fq.gz'), data_file('trio1/ctrl5.fq.gz'), data_file('trio1/ctrl6.fq.gz')]
        tables = [os.path.join(tmp, n) for n in ('case1.ct', 'case2.ct', 'ctrl1.ct', 'ctrl2.ct')]
        for ct, fq in zip(tables, inputs):
            run_cli(['count', '--ksize', '19', '--memory', '1e7', ct, fq])
        out, log = run_cli(['novel', '--ksize', '19', '--memory', '1e7', '--ctrl-max', '1', '--case-min', '7', '--case', inputs[0],
                            '--case', inputs[1], '--case-counts', tables[0], tables[1], '--control-counts', tables[2], tables[3]])
        assert out == open(expected_file('novel-trio1-two-cases.augfastq')).read()
        assert out.strip() != ''
        nlines = 0
        for line in out.split('\n'):
            if not line.endswith('#') or line.startswith('#mateseq'):
                continue
            m = re.search(r'(\d+) (\d+) (\d+) (\d+)#$', line)
            assert m, line
            c1, c2, x1, x2 = (int(m.group(i)) for i in (1, 2, 3, 4))
            assert c1 >= 7 and c2 >= 7 and x1 <= 1 and x2 <= 1
            nlines += 1
        assert nlines > 0
        # the same from the reads, no saved tables: the cases are counted by `novel` itself
        out2, _ = run_cli(['novel', '--ksize', '19', '--memory', '1e7', '--ctrl-max', '1', '--case-min', '7', '--case', inputs[0],
                           '--case', inputs[1], '--control', inputs[2], '--control', inputs[3]])
        assert out2 == out


def test_no_control_golden_from_the_reference_api(hk):
    """novel(stream, [case, case], []) -- no control at all -- as the reference's own generator function returns it over the oracle"""
    import io
    import tempfile
    import kevlar_amd
    from conftest import data_file, expected_file
    from test_gpu_pipeline import run_cli
    with tempfile.TemporaryDirectory() as tmp:
        inputs = [data_file('trio1/case6.fq.gz'), data_file('trio1/case6b.fq.gz')]
        tables = [os.path.join(tmp, n) for n in ('case1.ct', 'case2.ct')]
        for ct, fq in zip(tables, inputs):
            run_cli(['count', '--ksize', '19', '--memory', '1e7', ct, fq])
        sketches = [kevlar_amd.sketch.load(t) for t in tables]
        stream = kevlar_amd.multi_file_iter_khmer(inputs)
        buf = io.StringIO()
        for rec in kevlar_amd.novel.novel(stream, sketches, [], ksize=19, casemin=12, ctrlmax=0):
            kevlar_amd.print_augmented_fastx(rec, buf)
        assert buf.getvalue() == open(expected_file('novel-trio1-two-cases-no-control.augfastq')).read()
        assert buf.getvalue().strip() != ''
