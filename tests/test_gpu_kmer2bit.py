"""The per-k-mer kernels that hash from the 2-bit form (kv_kmer2bit_device.h): `k_bin_hash_2bit`, the partitioned count's front end for
batches of equal-length reads, and `k_novel_mark_2bit`, the scan of such batches when there is nothing to deduplicate.  Both must give
what the scalar oracle gives k-mer by k-mer (khmer consume / kevlar/novel.py:123-169) and what the tile kernels they stand in for give:
every k they accept (16 .. 64: one- and two-word keys, every murmur tail length), every storage, bands, masks, reads with bases
outside ACGT, a first read to start from, read lengths that leave a last chunk of one k-mer or end on a word boundary."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KNOBS = ('KV_COUNT_PATH', 'KV_NOVEL_PATH', 'KV_BIN_2BIT', 'KV_NOVEL_2BIT')


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
    for name in KNOBS:
        os.environ.pop(name, None)


class Uniform(object):
    """a batch with the arithmetic layout the 2-bit kernels need (every read the same length), made the way real input makes it: the
    reads as a FASTQ file, parsed and packed on the device (kv_fastq.hip) -- reads with N or lower case keep their flags that way.
    hk.ReadBatch(list of strings) goes through the ragged constructor and never has that layout."""
    def __init__(self, hk, tmp_path, seqs, tag):
        path = str(tmp_path / '{}.fq'.format(tag))
        with open(path, 'w') as fh:
            for i, seq in enumerate(seqs):
                fh.write('@r{}\n{}\n+\n{}\n'.format(i, seq, 'I' * len(seq)))
        self.parser = hk.ReadParser(path)
        self.text = self.parser.text_batch(4 * len(seqs) + 16)        # (the first batch is sized from a guess of 280 bytes per record)
        assert type(self.text).__name__ == 'DeviceTextBatch' and self.text.n == len(seqs)
        self.batch = self.text.batch


def family(genome_len, n, seed, read_len):
    from kevlar_amd import synth
    trio = synth.make_trio(genome_len, seed, inherited_per_mb=400, denovo_per_mb=600)
    out = {}
    for i, name in enumerate(('proband', 'mother', 'father')):
        out[name] = synth.unpack_reads(synth.sample_reads_packed(trio[name], n, read_len, 0.005, seed + 1 + i), read_len)
    return out


@pytest.mark.parametrize('kind,k,read_len', [('Counttable', 31, 100), ('SmallCounttable', 31, 100), ('Nodetable', 31, 100),
                                             ('Counttable', 16, 100), ('Counttable', 17, 37), ('Counttable', 32, 96), ('Counttable', 33, 100),
                                             ('Counttable', 47, 100), ('Counttable', 48, 64), ('Counttable', 51, 100), ('Counttable', 64, 100),
                                             ('Counttable', 31, 41), ('Counttable', 31, 31), ('Counttable', 25, 250)])
def test_two_bit_count_equals_the_oracle_and_the_tile_front_end(hk, ok, prof, tmp_path, kind, k, read_len):
    reads = family(60000, 9000, 3, read_len)['proband']
    uni = Uniform(hk, tmp_path, reads, 'count')
    os.environ['KV_COUNT_PATH'] = 'binned'
    nk = read_len - k + 1
    for nbands, band in ((0, 0), (8, 0), (8, 7), (3, 1)):
        prof.kv_prof_reset()
        dev = getattr(hk, kind)(k, 4e5, 4)
        n_dev = dev.consume_batch(uni.batch, nbands, band)
        assert launches('k_bin_hash_2bit') == 1 and launches('k_bin_hash_direct') == 0 and launches('k_consume') == 0
        ref = getattr(ok, kind)(k, 4e5, 4)
        bases, offs = ok.concat_reads(reads)
        n_ref = ok.consume_reads(ref, bases, offs, len(reads), nbands, band)
        assert n_dev == n_ref and (nbands or n_dev == len(reads) * nk)
        for t in range(4):
            assert dev.table_bytes(t) == ref.table_bytes(t), 'table {} differs from the oracle ({} bands, band {})'.format(t, nbands, band)
        assert dev.n_occupied() == ref.n_occupied()
    os.environ['KV_BIN_2BIT'] = '0'
    old = getattr(hk, kind)(k, 4e5, 4)
    old.consume_batch(uni.batch, 3, 1)
    assert launches('k_bin_hash_direct') == 1
    for t in range(4):
        assert old.table_bytes(t) == dev.table_bytes(t)
    # the packed constructor gives the same layout (bench.py's batches)
    from kevlar_amd import synth
    os.environ.pop('KV_BIN_2BIT', None)
    prof.kv_prof_reset()
    again = getattr(hk, kind)(k, 4e5, 4)
    codes = np.frombuffer(''.join(reads).encode(), dtype=np.uint8).reshape(len(reads), read_len)
    again.consume_batch(hk.ReadBatch.from_packed(synth.pack_codes(np.searchsorted(np.frombuffer(b'ACGT', dtype=np.uint8), codes)), read_len), 3, 1)
    assert launches('k_bin_hash_2bit') == 1
    for t in range(4):
        assert again.table_bytes(t) == dev.table_bytes(t)


def test_two_bit_count_with_a_mask_and_reads_outside_acgt(hk, ok, prof, tmp_path):
    """consume_seqfile_with_mask / _banding_with_mask (kevlar/count.py:43-60) through the 2-bit front end; reads with N and lower case
    are counted as khmer counts them (cleaned to A / upper case)"""
    reads = family(50000, 7000, 8, 100)
    sample = list(reads['proband'])
    sample[3] = sample[3][:40] + 'N' + sample[3][41:]
    sample[11] = sample[11].lower()
    os.environ['KV_COUNT_PATH'] = 'binned'
    mask_dev, mask_ref = hk.Nodetable(31, 2e6, 2), ok.Nodetable(31, 2e6, 2)
    mask_dev.consume_batch(hk.ReadBatch(reads['mother']))
    bases, offs = ok.concat_reads(reads['mother'])
    ok.consume_reads(mask_ref, bases, offs, len(reads['mother']))
    bases, offs = ok.concat_reads(sample)
    uni = Uniform(hk, tmp_path, sample, 'masked')
    for nbands, band, masked in ((0, 0, False), (0, 0, True), (4, 2, False)):
        prof.kv_prof_reset()
        dev, ref = hk.Counttable(31, 3e5, 4), ok.Counttable(31, 3e5, 4)
        n_dev = dev.consume_batch(uni.batch, nbands, band, mask=mask_dev, threshold=1 if masked else 0, consume_masked=masked)
        n_ref = ok.consume_reads(ref, bases, offs, len(sample), nbands, band, mask_ref, 1 if masked else 0, masked)
        assert launches('k_bin_hash_2bit') == 1
        assert n_dev == n_ref > 0
        for t in range(4):
            assert dev.table_bytes(t) == ref.table_bytes(t)
        assert dev.n_occupied() == ref.n_occupied()


@pytest.mark.parametrize('k,read_len', [(31, 100), (51, 100), (16, 100), (64, 100), (32, 96), (33, 41), (25, 250)])
def test_two_bit_scan_equals_the_oracle_and_the_tile_scan(hk, ok, prof, tmp_path, k, read_len):
    reads = family(50000, 8000, 11, read_len)
    sample = list(reads['proband'])
    sample[5] = sample[5][:10] + 'N' + sample[5][11:]            # the scan skips reads with bases outside ACGT (kevlar/novel.py:134-139)
    sample[6] = sample[6].lower()
    dev, ref = {}, {}
    for n, seqs in (('proband', sample), ('mother', reads['mother']), ('father', reads['father'])):
        dev[n], ref[n] = hk.Counttable(k, 3e5, 4), ok.Counttable(k, 3e5, 4)
        dev[n].consume_batch(hk.ReadBatch(seqs))
        bases, offs = ok.concat_reads(seqs)
        ok.consume_reads(ref[n], bases, offs, len(seqs))
    uni = Uniform(hk, tmp_path, sample, 'scan')
    batch = uni.batch
    bases, offs = ok.concat_reads(sample)
    for band_mode, nbands, band, first in ((0, 0, 0, 0), (1, 8, 0, 0), (1, 8, 7, 0), (2, 4, 2, 0), (0, 0, 0, 3000)):
        want, _ = ok.novel_scan([ref['proband']], [ref['mother'], ref['father']], bases, offs, len(sample), k, 5, 1, band_mode=band_mode, nbands=nbands, band=band)
        want = [h for h in want if h[0] >= first]
        got = {}
        for which in ('2bit', 'tiles'):
            prof.kv_prof_reset()
            if which == 'tiles':
                os.environ['KV_NOVEL_PATH'] = 'tiles'
            else:
                os.environ.pop('KV_NOVEL_PATH', None)
            r, o, a, _ = hk.novel_scan([dev['proband']], [dev['mother'], dev['father']], batch, 5, 1, band_mode=band_mode, nbands=nbands, band=band, first_read=first)
            got[which] = [(int(r[i]), int(o[i]), tuple(int(x) for x in a[i])) for i in range(len(r))]
            assert launches('k_novel_mark_2bit' if which == '2bit' else 'k_novel_mark') == 1
            assert launches('k_novel_mark' if which == '2bit' else 'k_novel_mark_2bit') == 0
        os.environ.pop('KV_NOVEL_PATH', None)
        assert got['2bit'] == want and got['tiles'] == want
        if band_mode == 0 and first == 0:
            assert len(want) > 20
    # with an abundance screen the scan keeps the tile kernel (the discard flag follows the reference's evaluation order)
    prof.kv_prof_reset()
    hk.novel_scan([dev['proband']], [dev['mother'], dev['father']], batch, 5, 1, screen=2)
    assert launches('k_novel_mark') == 1 and launches('k_novel_mark_2bit') == 0


def test_ragged_batches_keep_the_tile_kernels(hk, ok, prof):
    """reads of different lengths have no arithmetic layout: the tile kernels take them, with the oracle's result"""
    reads = family(40000, 4000, 5, 100)['proband']
    reads[7] = reads[7][:80]
    os.environ['KV_COUNT_PATH'] = 'binned'
    dev, ref = hk.Counttable(31, 3e5, 4), ok.Counttable(31, 3e5, 4)
    n_dev = dev.consume_batch(hk.ReadBatch(reads))
    bases, offs = ok.concat_reads(reads)
    assert n_dev == ok.consume_reads(ref, bases, offs, len(reads))
    assert launches('k_bin_hash_direct') == 1 and launches('k_bin_hash_2bit') == 0
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)


def test_randomised_parity_of_the_two_bit_kernels(hk):
    """scratch/fuzz_kmer2bit.py, a short fixed-seed run: random k (16..64), read lengths, storages, table sizes, bands (both rules), masks,
    thresholds, reads with bases outside ACGT, a first read; tables byte for byte and hits identical to the oracle, and the 2-bit kernels are
    the ones that ran (200 trials of seed 1 at the end of round 4: no mismatch)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, os.path.join(root, 'scratch', 'fuzz_kmer2bit.py'), '30', '7'], cwd=root, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    assert 'done: 30 trials (seed 7), 0 mismatches' in done.stdout
