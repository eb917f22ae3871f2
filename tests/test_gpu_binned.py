"""The partitioned count path (kv_binned.hip: hash+partition, split, LDS-resident slices) must
produce the same table bytes as the scalar oracle -- on its natural workload (millions of
k-mers into megabyte tables), on skewed input that overflows buckets into the spill list, and,
forced through KV_COUNT_PATH=binned, on tiny tables."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def launches(name):
    from kevlar_amd import _lib
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    _lib.load().kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return n.value


@pytest.fixture
def profiled():
    from kevlar_amd import _lib
    lib = _lib.load()
    lib.kv_prof_reset()
    lib.kv_prof_enable(1)
    os.environ['KV_COUNT_PATH'] = 'binned'      # this file is about the one-item-per-k-mer partition (the super-k-mer
    yield                                       # front end that large batches take by default: tests/test_gpu_skm.py)
    lib.kv_prof_enable(0)
    os.environ.pop('KV_COUNT_PATH', None)


def synthetic_reads(n, seed, skew=0):
    from kevlar_amd import synth
    trio = synth.make_trio(400000, seed)
    words = synth.sample_reads_packed(trio['proband'], n, 100, 0.005, seed + 1)
    reads = synth.unpack_reads(words, 100)
    return reads + ['A' * 100] * skew + ['ACGT' * 25] * (skew // 4)


@pytest.mark.parametrize('kind,tablesize', [('Counttable', 3.0e6), ('SmallCounttable', 2.5e6), ('Nodetable', 9.0e6)])
def test_partitioned_count_matches_oracle(hk, ok, profiled, kind, tablesize):
    reads = synthetic_reads(64000, 5)
    dev, ref = getattr(hk, kind)(31, tablesize, 4), getattr(ok, kind)(31, tablesize, 4)
    n_dev = dev.consume_batch(hk.ReadBatch(reads))
    assert launches('k_bin_apply') == 1 and launches('k_consume') == 0     # took the partitioned path
    bases, offs = ok.concat_reads(reads)
    assert n_dev == ok.consume_reads(ref, bases, offs, len(reads))
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert dev.n_occupied() == ref.n_occupied()
    # n_unique_kmers on this path is a linear-counting estimate: close, not exact
    assert abs(dev.n_unique_kmers() - ref.n_unique_kmers()) < 0.02 * ref.n_unique_kmers()
    # a second batch accumulates on top of the first (tables are loaded, not zeroed)
    more = synthetic_reads(64000, 9)
    dev.consume_batch(hk.ReadBatch(more))
    bases, offs = ok.concat_reads(more)
    ok.consume_reads(ref, bases, offs, len(more))
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert dev.n_occupied() == ref.n_occupied()


def test_partitioned_count_skew_goes_through_spill_list(hk, ok, profiled):
    """3000 poly-A reads put 210k copies of one k-mer into one slice: the bucket overflows and the
    remainder is applied from the spill list; counters still saturate at exactly 255."""
    reads = synthetic_reads(62000, 11, skew=3000)
    dev, ref = hk.Counttable(31, 2.2e6, 4), ok.Counttable(31, 2.2e6, 4)
    dev.consume_batch(hk.ReadBatch(reads))
    assert launches('k_bin_apply') == 1 and launches('k_bin_spill') == 1
    bases, offs = ok.concat_reads(reads)
    ok.consume_reads(ref, bases, offs, len(reads))
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert dev.get('A' * 31) == 255 == ref.get('A' * 31)


def test_partitioned_count_band_and_mask(hk, ok, profiled):
    reads = synthetic_reads(70000, 21)
    dmask, rmask = hk.Nodetable(31, 4e6, 4), ok.Nodetable(31, 4e6, 4)
    dmask.consume_batch(hk.ReadBatch(reads[:20000]))
    bases, offs = ok.concat_reads(reads[:20000])
    ok.consume_reads(rmask, bases, offs, 20000)
    bases, offs = ok.concat_reads(reads)
    os.environ['KV_COUNT_PATH'] = 'binned'
    from kevlar_amd import _lib
    _lib.load().kv_prof_reset()          # the small mask batch above legitimately used k_consume
    for nbands, band, mask_args in [(4, 3, None), (0, 0, (0, False)), (2, 0, (1, True))]:
        dev, ref = hk.Counttable(31, 2.5e6, 4), ok.Counttable(31, 2.5e6, 4)
        if mask_args is None:
            n_dev = dev.consume_batch(hk.ReadBatch(reads), nbands, band)
            n_ref = ok.consume_reads(ref, bases, offs, len(reads), nbands, band)
        else:
            n_dev = dev.consume_batch(hk.ReadBatch(reads), nbands, band, dmask, mask_args[0], mask_args[1])
            n_ref = ok.consume_reads(ref, bases, offs, len(reads), nbands, band, rmask, mask_args[0], mask_args[1])
        assert n_dev == n_ref and n_dev > 0
        for t in range(4):
            assert dev.table_bytes(t) == ref.table_bytes(t)
    assert launches('k_consume') == 0


@pytest.mark.parametrize('kind', ['Counttable', 'SmallCounttable', 'Nodetable', 'Countgraph'])
@pytest.mark.parametrize('k,tablesize', [(21, 5e3), (31, 2.1e5)])
def test_partitioned_count_forced_on_small_tables(hk, ok, profiled, kind, k, tablesize):
    """Same cases as test_gpu_sketch.py::test_consume_tables_bit_exact, through the other path
    (one to four slices per table, saturation, short/empty reads)."""
    os.environ['KV_COUNT_PATH'] = 'binned'
    rng = np.random.default_rng(11)
    letters = np.array(list('ACGT'))
    reads = [''.join(letters[rng.integers(0, 4, size=int(rng.integers(10, 151)))]) for _ in range(700)]
    reads += ['A' * 150] * 30 + ['ACGT' * 30] * 5 + ['', 'ACG']
    dev, ref = getattr(hk, kind)(k, tablesize, 4), getattr(ok, kind)(k, tablesize, 4)
    n_dev = dev.consume_batch(hk.ReadBatch(reads))
    assert launches('k_bin_apply') == 1
    bases, offs = ok.concat_reads(reads)
    assert n_dev == ok.consume_reads(ref, bases, offs, len(reads))
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)
    assert dev.n_occupied() == ref.n_occupied()
