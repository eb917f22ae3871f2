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


def test_fullsize_config5_k51_four_samples(hk, ok):
    """BASELINE.json config 5: proband + 3 controls, k=51 (two-word keys, three murmur blocks + tail), 30x, 25 Mb"""
    from kevlar_amd import synth
    k, names = 51, ('proband', 'mother', 'father', 'sibling1')
    packed = synth.trio_reads_packed(25_000_000, 30, L, extra_controls=1)
    batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
    n_reads, nk = packed['proband'].shape[0], L - k + 1
    sk = {n: hk.Counttable(k, MEM / 4, 4) for n in names}
    for n in names[1:] + names[:1]:
        assert sk[n].consume_batch(batches[n]) == n_reads * nk
    os.environ['KV_COUNT_PATH'] = 'binned'
    try:
        other = hk.Counttable(k, MEM / 4, 4)
        other.consume_batch(batches['sibling1'])
    finally:
        os.environ.pop('KV_COUNT_PATH', None)
    for x, y in zip(tables(sk['sibling1']), tables(other)):
        assert np.array_equal(x, y)
    assert sk['sibling1'].n_occupied() == other.n_occupied() == int(np.count_nonzero(tables(other)[0]))
    del other
    cases, ctrls = [sk['proband']], [sk[n] for n in names[1:]]
    r, o, a, _ = hk.novel_scan(cases, ctrls, batches['proband'], 6, 1)
    os.environ['KV_NOVEL_PATH'] = 'tiles'
    try:
        r2, o2, a2, _ = hk.novel_scan(cases, ctrls, batches['proband'], 6, 1)
    finally:
        os.environ.pop('KV_NOVEL_PATH', None)
    assert np.array_equal(r, r2) and np.array_equal(o, o2) and np.array_equal(a, a2)
    assert len(r) > 100000 and a.shape[1] == 4
    assert (a[:, 0] >= 6).all() and (a[:, 1:] <= 1).all()
    assert (np.diff(r.astype(np.int64) * nk + o) > 0).all()
    # a seeded sample of reads replayed by the oracle's hash + Count-Min minimum over the SAME table bytes
    tabs = {n: tables(sk[n]) for n in names}
    sizes = sk['proband'].hashsizes()
    rng = np.random.default_rng(2)
    sample = np.concatenate((rng.choice(np.unique(r), size=100, replace=False), rng.integers(0, n_reads, size=100)))
    ref = ok.Counttable(k, MEM / 4, 4)
    want = {}
    for ridx, seq in zip(sample.tolist(), synth.unpack_reads(packed['proband'][sample], L)):
        for i in range(nk):
            h = ref.hash(seq[i:i + k])
            ab = tuple(min(int(tabs[n][t][h % sizes[t]]) for t in range(4)) for n in names)
            if ab[0] >= 6 and max(ab[1:]) <= 1:
                want[(ridx, i)] = ab
    sel = np.isin(r, sample)
    got = {(x, y): tuple(z) for x, y, z in zip(r[sel].tolist(), o[sel].tolist(), a[sel].tolist())}
    assert got == want and len(want) > 500
