"""BASELINE.json config 2 at full size (25 Mb genome, 30x: 7.5 M reads = 525 M k-mers per sample,
2 GB sketch per sample) is far beyond what the scalar oracle can replay, so parity at this size is
checked through properties that do not depend on size:

* the three count implementations (global-atomic, one-item-per-k-mer partition, super-k-mer / per-distinct-k-mer)
  produce identical tables, and the two scans (per distinct k-mer over super-k-mer buckets, per k-mer over tiles)
  identical hits -- at config 2 (k=31) and config 5 (proband + 3 controls, k=51);
* saturating-add algebra: counting a batch twice gives min(255, 2 x) in every bin;
* banding is a partition of the k-mers: the band tables sum to the unbanded tables;
* the k-mer total matches n_reads x (L - k + 1); occupancy equals the non-zero count of table 0;
* every reported interesting k-mer satisfies the thresholds, the scan is idempotent, its bit
  mask and its hit list agree, and a seeded sample of hits re-derives bit-exactly from the oracle
  evaluated on the same table bytes.

Round 3 added the oracle itself at full size for the tables; round 4 for the scans too, whole samples: the oracle's scan loop
runs over ALL reads of the proband on the host cores (kvo_novel_scan_mt) and every hit of each of the three scan kernels --
k_skm_novel_list (what bench.py times), k_skm_novel, k_novel_mark -- is compared with it, configs 2 and 5.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

L, K, MEM = 100, 31, 2e9


@pytest.fixture(scope='module')
def trio(hk):
    from kevlar_amd import synth
    packed = synth.trio_reads_packed(25_000_000, 30, L)
    return packed, {n: hk.ReadBatch.from_packed(w, L) for n, w in packed.items()}


def tables(sketch):
    return [np.frombuffer(sketch.table_bytes(t), dtype=np.uint8) for t in range(4)]


def test_fullsize_count_paths_agree_and_saturating_algebra(hk, trio):
    packed, batches = trio
    n_reads = packed['proband'].shape[0]
    assert n_reads == 7_500_000
    os.environ['KV_COUNT_PATH'] = 'atomic'
    try:
        a = hk.Counttable(K, MEM / 4, 4)
        n_a = a.consume_batch(batches['proband'])
        os.environ['KV_COUNT_PATH'] = 'binned'
        c = hk.Counttable(K, MEM / 4, 4)
        n_c = c.consume_batch(batches['proband'])
    finally:
        os.environ.pop('KV_COUNT_PATH', None)
    b = hk.Counttable(K, MEM / 4, 4)                      # default at this size: the super-k-mer front end
    n_b = b.consume_batch(batches['proband'])
    assert n_a == n_b == n_c == n_reads * (L - K + 1)
    ta, tb = tables(a), tables(b)
    for x, y, z in zip(ta, tb, tables(c)):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    del c
    assert a.n_occupied() == b.n_occupied() == int(np.count_nonzero(tb[0]))
    for t in tb:                                    # every k-mer lands once in every table
        assert t.max() == 255 or int(t.sum(dtype=np.uint64)) == n_b
    b.consume_batch(batches['proband'])             # same batch again
    for once, twice in zip(ta, tables(b)):
        assert np.array_equal(np.minimum(255, 2 * once.astype(np.uint16)).astype(np.uint8), twice)


def test_fullsize_bands_partition_the_kmers(hk, trio):
    packed, batches = trio
    whole = hk.Counttable(K, MEM / 4, 4)
    total = whole.consume_batch(batches['mother'])
    acc = [np.zeros(s, dtype=np.uint32) for s in whole.hashsizes()]
    n_sum = 0
    for band in range(3):
        part = hk.Counttable(K, MEM / 4, 4)
        n_sum += part.consume_batch(batches['mother'], 3, band)
        for t, arr in enumerate(tables(part)):
            acc[t] += arr
        del part
    assert n_sum == total
    for t, arr in enumerate(tables(whole)):
        assert np.array_equal(np.minimum(acc[t], 255).astype(np.uint8), arr)


def test_fullsize_novel_scan_properties(hk, ok, trio):
    packed, batches = trio
    from kevlar_amd import synth
    names = ('proband', 'mother', 'father')
    sk = {n: hk.Counttable(K, MEM / 4, 4) for n in names}
    for n in names:
        sk[n].consume_batch(batches[n])
    nk = L - K + 1
    n_reads = packed['proband'].shape[0]
    import torch
    mask = torch.zeros((n_reads * nk + 31) // 32, dtype=torch.int32, device='cuda')
    r, o, a, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1,
                               mask_ptr=mask.data_ptr(), mask_stride=nk)
    torch.cuda.synchronize()
    assert len(r) > 100000
    assert (a[:, 0] >= 6).all() and (a[:, 1:] <= 1).all()
    key = r.astype(np.int64) * nk + o
    assert (np.diff(key) > 0).all()                  # (read, offset) order, no duplicates
    from kevlar_amd import bandmerge
    mr, mo = bandmerge.mask_to_hits(mask, nk)
    assert np.array_equal(mr, r) and np.array_equal(mo, o)
    r2, o2, a2, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1)
    assert np.array_equal(r, r2) and np.array_equal(o, o2) and np.array_equal(a, a2)
    # a seeded sample of reads, replayed by the oracle's Count-Min get on the SAME table bytes
    tabs = {n: tables(sk[n]) for n in names}
    sizes = sk['proband'].hashsizes()
    rng = np.random.default_rng(1)
    hit_reads = np.unique(r)
    sample = np.concatenate((rng.choice(hit_reads, size=150, replace=False), rng.integers(0, n_reads, size=150)))
    seqs = synth.unpack_reads(packed['proband'][sample], L)
    ref = ok.Counttable(K, MEM / 4, 4)
    want = []
    for ridx, seq in zip(sample.tolist(), seqs):
        for i in range(nk):
            h = ref.hash(seq[i:i + K])
            ab = [min(int(tabs[n][t][h % sizes[t]]) for t in range(4)) for n in names]
            if ab[0] >= 6 and ab[1] <= 1 and ab[2] <= 1:
                want.append((ridx, i, tuple(ab)))
    got = {}
    sel = np.isin(r, sample)
    for ridx, off, ab in zip(r[sel].tolist(), o[sel].tolist(), a[sel].tolist()):
        got[(ridx, off)] = tuple(ab)
    assert got == {(x, y): z for x, y, z in want}
    assert len(want) > 1000


@pytest.fixture(scope='module')
def quad(hk):
    """BASELINE.json config 5's family: proband + 3 controls (30x, 25 Mb), packed"""
    from kevlar_amd import synth
    return synth.trio_reads_packed(25_000_000, 30, L, extra_controls=1)


def test_fullsize_config5_k51_count_paths_agree(hk, quad):
    """k = 51 (two-word keys, three murmur blocks + tail): the one-item-per-k-mer partition and the super-k-mer count give the
    same tables and occupancy (the scans and the tables themselves meet the oracle below)"""
    k = 51
    batch = hk.ReadBatch.from_packed(quad['sibling1'], L)
    n_reads, nk = quad['sibling1'].shape[0], L - k + 1
    mine = hk.Counttable(k, MEM / 4, 4)
    assert mine.consume_batch(batch) == n_reads * nk
    os.environ['KV_COUNT_PATH'] = 'binned'
    try:
        other = hk.Counttable(k, MEM / 4, 4)
        other.consume_batch(batch)
    finally:
        os.environ.pop('KV_COUNT_PATH', None)
    for x, y in zip(tables(mine), tables(other)):
        assert np.array_equal(x, y)
    assert mine.n_occupied() == other.n_occupied() == int(np.count_nonzero(tables(other)[0]))


# ---- full size against the ORACLE itself (not against another device path) ------------------------------------------
# kvo_consume_reads_mt counts with atomic saturating adds, so its tables (and n_occupied: "was the bin zero" comes out of
# the same atomic) do not depend on the thread schedule: all host cores, one sample in well under a minute.

def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return cores


def ascii_block(words, read_len, chunk=500_000):
    """packed words -> (bytes of all reads back to back, c_uint64 offsets): the oracle's input, without a Python string per read"""
    import ctypes
    from kevlar_amd import synth
    n = words.shape[0]
    out = np.empty(n * read_len, dtype=np.uint8)
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    for lo in range(0, n, chunk):
        w = words[lo:lo + chunk]
        codes = ((w[:, :, None] >> shifts) & 3).reshape(w.shape[0], -1)[:, :read_len].astype(np.uint8)
        out[lo * read_len:(lo + w.shape[0]) * read_len] = synth.ALPHABET[codes].reshape(-1).view(np.uint8)
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len))
    return out.tobytes(), offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), offs


def assert_same_sketch(dev, ref):
    assert dev.hashsizes() == ref.hashsizes()
    for t in range(4):
        got = np.frombuffer(dev.table_bytes(t), dtype=np.uint8)
        want = np.frombuffer(ref.table_bytes(t), dtype=np.uint8)
        assert got.shape == want.shape and np.array_equal(got, want), 'table {} differs from the oracle'.format(t)
    assert dev.n_occupied() == ref.n_occupied()


def launches(hk, scope):
    """launches of a profile scope since kv_prof_reset"""
    import ctypes
    from kevlar_amd import _lib
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    _lib.load().kv_prof_get(scope.encode(), ctypes.byref(ms), ctypes.byref(n))
    return int(n.value)


class Profiled(object):
    """kernel launches by profile scope inside a with block"""
    def __init__(self, hk):
        from kevlar_amd import _lib
        self.hk, self.lib = hk, _lib.load()

    def __enter__(self):
        self.lib.kv_prof_reset()
        self.lib.kv_prof_enable(1)
        return self

    def __exit__(self, *exc):
        self.lib.kv_prof_enable(0)

    def count(self, scope):
        return launches(self.hk, scope)


def scan_all_ways(hk, cases, ctrls, batch, case_min, ctrl_max, first_from_list):
    """the hits of `batch` from every scan kernel the product has: the one that runs after a count with expect_scan()
    (k_skm_novel_list: what bench.py times and `kevlar novel` runs for a one-batch case sample), the walk over re-built buckets
    (k_skm_novel) and the per-k-mer tile scan (k_novel_mark); each with the launches that prove which one it was"""
    out = {}
    with Profiled(hk) as prof:
        out['list'] = hk.novel_scan(cases, ctrls, batch, case_min, ctrl_max)[:3]
        n_list, n_walk = prof.count('k_skm_novel_list'), prof.count('k_skm_novel')
    if first_from_list:
        assert n_list == 1 and n_walk == 0, 'the scan behind a hinted count must go by the distinct list ({} / {})'.format(n_list, n_walk)
    with Profiled(hk) as prof:          # one scan per bucketing: the next one cuts the reads again and walks
        out['walk'] = hk.novel_scan(cases, ctrls, batch, case_min, ctrl_max)[:3]
        assert prof.count('k_skm_novel') == 1 and prof.count('k_skm_novel_list') == 0
    os.environ['KV_NOVEL_PATH'] = 'tiles'
    try:
        with Profiled(hk) as prof:
            out['tiles'] = hk.novel_scan(cases, ctrls, batch, case_min, ctrl_max)[:3]
            assert prof.count('k_novel_mark') == 1 and prof.count('k_skm_novel') == 0 and prof.count('k_skm_novel_list') == 0
    finally:
        os.environ.pop('KV_NOVEL_PATH', None)
    return out


def assert_hits_equal(got, want, what):
    r, o, a = got
    wr, wo, wa = want
    assert len(r) == len(wr), '{}: {} hits, the oracle has {}'.format(what, len(r), len(wr))
    assert np.array_equal(np.asarray(r, dtype=np.uint32), wr), what
    assert np.array_equal(np.asarray(o, dtype=np.uint32), wo.astype(np.uint32)), what
    assert np.array_equal(np.asarray(a, dtype=np.uint8), wa), what


def test_fullsize_config2_tables_and_scan_equal_the_oracle(hk, ok, trio):
    """BASELINE.json config 2 against the oracle itself, whole samples: every table byte and n_occupied of all three 7.5 M-read
    samples (2 GB sketches) against the oracle's threaded count, and EVERY hit of the proband -- from each of the three scan
    kernels, first of all the one bench.py times (k_skm_novel_list behind a count with expect_scan) -- against the oracle's own
    scan loop (kevlar/novel.py:123-169 restated) over all 7.5 M reads, split over the host cores"""
    packed, batches = trio
    names = ('mother', 'father', 'proband')               # the case sample last: its bucketed batch is what the scan finds
    cores = host_cores()
    n_reads = packed['proband'].shape[0]
    dev, ref, keep = {}, {}, None
    had = os.environ.get('KV_SKM_DL')
    os.environ['KV_SKM_DL'] = '1'            # (a stream's first batch gets no list unless asked: bench.py asks the same way)
    try:
        for n in names:
            dev[n] = hk.Counttable(K, MEM / 4, 4)
            if n == 'proband':
                dev[n].expect_scan()
        with Profiled(hk) as prof:
            for n in names:
                assert dev[n].consume_batch(batches[n]) == n_reads * (L - K + 1)
            assert prof.count('k_skm_count') == 3 and prof.count('k_consume') == 0
        scans = scan_all_ways(hk, [dev['proband']], [dev['mother'], dev['father']], batches['proband'], 6, 1, first_from_list=True)
    finally:
        if had is None:
            os.environ.pop('KV_SKM_DL', None)
        else:
            os.environ['KV_SKM_DL'] = had
    for n in names:
        bases, offs_p, offs = ascii_block(packed[n], L)
        ref[n] = ok.Counttable(K, MEM / 4, 4)
        assert ok.consume_reads_mt(ref[n], bases, offs_p, n_reads, cores) == n_reads * (L - K + 1)
        assert_same_sketch(dev[n], ref[n])
        if n == 'proband':
            keep = (bases, offs_p, offs)
    want = ok.novel_scan_mt([ref['proband']], [ref['mother'], ref['father']], keep[0], keep[1], n_reads, K, 6, 1, cores)
    assert len(want[0]) > 1_000_000
    for way in ('list', 'walk', 'tiles'):
        assert_hits_equal(scans[way], want, 'config 2, scan by ' + way)


def test_fullsize_config5_tables_and_scan_equal_the_oracle(hk, ok, quad):
    """BASELINE.json config 5 -- proband + 3 controls, k = 51 (two-word keys, three murmur blocks + tail), 30x, 25 Mb -- whole samples
    against the oracle: every table byte of all four sketches, and every hit of each scan kernel against the oracle's scan loop
    over all reads of the proband with all four oracle-counted sketches"""
    k, names = 51, ('mother', 'father', 'sibling1', 'proband')
    packed = quad
    batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
    n_reads, nk = packed['proband'].shape[0], L - k + 1
    cores = host_cores()
    dev, ref, keep = {}, {}, None
    had = os.environ.get('KV_SKM_DL')
    os.environ['KV_SKM_DL'] = '1'
    try:
        for n in names:
            dev[n] = hk.Counttable(k, MEM / 4, 4)
            if n == 'proband':
                dev[n].expect_scan()
        with Profiled(hk) as prof:
            for n in names:
                assert dev[n].consume_batch(batches[n]) == n_reads * nk
            assert prof.count('k_skm_count') == 4
        scans = scan_all_ways(hk, [dev['proband']], [dev[n] for n in names[:3]], batches['proband'], 6, 1, first_from_list=True)
    finally:
        if had is None:
            os.environ.pop('KV_SKM_DL', None)
        else:
            os.environ['KV_SKM_DL'] = had
    for n in names:
        bases, offs_p, offs = ascii_block(packed[n], L)
        ref[n] = ok.Counttable(k, MEM / 4, 4)
        assert ok.consume_reads_mt(ref[n], bases, offs_p, n_reads, cores) == n_reads * nk
        assert_same_sketch(dev[n], ref[n])
        if n == 'proband':
            keep = (bases, offs_p, offs)
    want = ok.novel_scan_mt([ref['proband']], [ref[n] for n in names[:3]], keep[0], keep[1], n_reads, k, 6, 1, cores)
    assert len(want[0]) > 100_000 and want[2].shape[1] == 4
    for way in ('list', 'walk', 'tiles'):
        assert_hits_equal(scans[way], want, 'config 5, scan by ' + way)


def test_fullsize_config3_eight_bands_equal_the_oracle(hk, ok, trio):
    """BASELINE.json config 3 in its single-GPU form: the 25 Mb trio as EIGHT hash bands (what the eight GPUs of a node count and
    scan, one band each: docs/banding.rst, kevlar/count.py:62-66; sketch memory 2 GB / 8 per band as `bench.py --gpus 8` sizes
    it), replayed band by band on this GPU and held against the oracle, which hashes every k-mer once and sends it to its band's
    sketch (kvo_consume_reads_mt_allbands, kvo_novel_scan_mt_allbands: equal to the band-by-band scalar legs, tests/test_host_logic.py):
    every table byte and n_occupied of all 24 band sketches, every hit of every band, the merged hits (the all-gather +
    sort of kevlar_amd/bandmerge.py, kevlar/unband.py:41-77), and north_star's merge of per-band bit masks: the eight masks are
    disjoint, their sum (what the all-reduce computes) is their OR, and its set bits are the merged hits"""
    import torch
    from kevlar_amd import bandmerge
    packed, batches = trio
    NB = 8
    names = ('mother', 'father', 'proband')
    cores = host_cores()
    n_reads = packed['proband'].shape[0]
    nk = L - K + 1
    ref = {n: [ok.Counttable(K, MEM / NB / 4, 4) for _ in range(NB)] for n in names}
    keep = None
    for n in names:
        bases, offs_p, offs = ascii_block(packed[n], L)
        assert ok.consume_reads_mt_allbands(ref[n], bases, offs_p, n_reads, cores) == n_reads * nk
        if n == 'proband':
            keep = (bases, offs_p, offs)
    want = ok.novel_scan_mt_allbands([[ref['proband'][b]] for b in range(NB)], [[ref['mother'][b], ref['father'][b]] for b in range(NB)],
                                     keep[0], keep[1], n_reads, K, 6, 1, cores)
    del keep
    assert len(want[0]) > 1_000_000
    dev = {n: hk.Counttable(K, MEM / NB / 4, 4) for n in names}
    dev['proband'].expect_scan()
    merged = torch.zeros((n_reads * nk + 31) // 32, dtype=torch.int32, device='cuda')
    mask = torch.zeros_like(merged)
    parts, total = [], {n: 0 for n in names}
    for band in range(NB):
        for n in names:
            dev[n].clear()
            total[n] += dev[n].consume_batch(batches[n], NB, band)
            assert_same_sketch(dev[n], ref[n][band])
        mask.zero_()
        torch.cuda.synchronize()
        r, o, a, _ = hk.novel_scan([dev['proband']], [dev['mother'], dev['father']], batches['proband'], 6, 1, band_mode=1, nbands=NB,
                                   band=band, mask_ptr=mask.data_ptr(), mask_stride=nk)
        torch.cuda.synchronize()
        sel = want[3] == band
        assert_hits_equal((r, o, a), (want[0][sel], want[1][sel], want[2][sel]), 'config 3, band {} of {}'.format(band, NB))
        assert int(sel.sum()) > 50_000
        assert not bool((merged & mask).any()), 'the bands\' masks must be disjoint'
        merged += mask                                  # what all_reduce(SUM) does with the eight ranks' masks (bandmerge.allreduce_mask)
        parts.append((np.array(r, dtype=np.uint32), np.array(o, dtype=np.uint32), np.array(a, dtype=np.uint8)))
    for n in names:
        assert total[n] == n_reads * nk                 # every k-mer counted by exactly one band
    rr, oo, aa = (np.concatenate([p[i] for p in parts]) for i in range(3))
    order = np.lexsort((oo, rr))
    assert_hits_equal((rr[order], oo[order], aa[order]), want[:3], 'config 3, merged hits')
    mr, mo = bandmerge.mask_to_hits(merged, nk)
    assert np.array_equal(mr, want[0]) and np.array_equal(mo, want[1].astype(np.uint32)), 'config 3, merged bit masks'
