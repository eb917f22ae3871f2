"""`kevlar dist` on the GPU (kevlar_amd/dist.py, kv_abundance_distribution) against the reference's own
test expectations (kevlar/tests/test_dist.py: golden sketch file, abundance dictionary, mu/sigma, TSV)
and against the oracle on inputs that span batches and start from a non-empty tracking table."""
import filecmp
import json

import numpy as np
import pandas
import pytest

from conftest import data_file

pytestmark = pytest.mark.gpu

GOLDEN_ABUND = {10: 6, 11: 10, 12: 12, 13: 18, 14: 16, 15: 11, 16: 9, 17: 9, 18: 11, 19: 8, 20: 9, 21: 7, 22: 3}


def test_count_first_pass(hk, tmp_path):
    from kevlar_amd.dist import count_first_pass
    mask = hk.Nodetable.load(data_file('minitrio/mask.nt'))
    counts = hk.Counttable(31, 1e4, 4)
    count_first_pass([data_file('minitrio/trio-proband.fq.gz')], counts, mask)
    out = str(tmp_path / 'first.ct')
    counts.save(out)
    assert filecmp.cmp(data_file('minitrio/trio-proband-mask-counts.ct'), out, shallow=False)


def test_count_second_pass(hk):
    from kevlar_amd.dist import count_second_pass
    counts = hk.Counttable.load(data_file('minitrio/trio-proband-mask-counts.ct'))
    abund = count_second_pass([data_file('minitrio/trio-proband.fq.gz')], counts)
    assert abund == GOLDEN_ABUND


def test_dist(hk):
    import kevlar_amd
    mask = hk.Nodetable.load(data_file('minitrio/mask.nt'))
    mu, sigma, data = kevlar_amd.dist.dist([data_file('minitrio/trio-proband.fq.gz')], mask, memory=4e4)
    assert mu == pytest.approx(15.32558, abs=1e-5)
    assert sigma == pytest.approx(3.280581, abs=1e-6)
    assert list(data['Count'][-5:]) == [11.0, 8.0, 9.0, 7.0, 3.0]


def test_dist_empty(hk):
    import kevlar_amd
    from kevlar_amd.dist import KevlarZeroAbundanceDistError
    mask = hk.Nodetable(31, 1e4, 4)
    mask.consume('GATTACA' * 10)
    mask.consume('A' * 50)
    with pytest.raises(KevlarZeroAbundanceDistError):
        kevlar_amd.dist.dist([data_file('minitrio/trio-proband.fq.gz')], mask, memory=4e4)


def test_main_and_tsv(hk, capsys, tmp_path):
    import kevlar_amd
    tsv = str(tmp_path / 'dist.tsv')
    args = kevlar_amd.cli.parser().parse_args(['dist', '--tsv', tsv, data_file('minitrio/mask.nt'),
                                               data_file('minitrio/trio-proband.fq.gz')])
    kevlar_amd.dist.main(args)
    out, _ = capsys.readouterr()
    js = json.loads(out)
    data = pandas.read_csv(tsv, sep='\t')
    assert list(data['CumulativeCount']) == [15.0, 18.0, 24.0, 44.0, 78.0, 153.0, 222.0, 325.0, 423.0, 515.0,
                                             585.0, 666.0, 756.0, 814.0, 861.0, 888.0, 902.0, 903.0]
    golden = pandas.read_csv(data_file('minitrio/trio-proband-dist.tsv'), sep='\t')
    assert np.array_equal(data.values, golden.values)
    mu = float(np.average(golden['Abundance'], weights=golden['Count']))
    assert js['mu'] == pytest.approx(mu)


def test_abundance_distribution_matches_oracle_across_batches(hk, ok):
    """two batches, the second starting from the tracking state the first left behind; ragged / non-ACGT reads"""
    from kevlar_amd import synth
    trio = synth.make_trio(60000, 17)
    reads = synth.unpack_reads(synth.sample_reads_packed(trio['proband'], 9000, 100, 0.01, 3), 100)
    reads[5] = reads[5][:60] + 'N' + reads[5][61:]
    reads[9] = reads[9][:25]
    reads.append('ACGT' * 12)
    k = 27
    dev_counts, ref_counts = hk.Counttable(k, 30000, 4), ok.Counttable(k, 30000, 4)
    dev_counts.consume_batch(hk.ReadBatch(reads))
    bases, offs = ok.concat_reads(reads)
    ok.consume_reads(ref_counts, bases, offs, len(reads))
    dev_track = hk.Nodetable(k, 1, 1, primes=dev_counts.hashsizes())
    ref_track = ok.Nodetable(k, 1, 1, primes=ref_counts.hashsizes())
    half = len(reads) // 2
    for part in (reads[:half], reads[half:]):
        got = dev_counts.abundance_distribution(hk.ReadBatch(part), dev_track)
        want = [0] * 65536
        import ctypes
        hist = (ctypes.c_uint64 * 65536)()
        for seq in part:
            b = seq.encode()
            ok.lib.kvo_abundance_distribution(ref_counts._h, ref_track._h, b, len(b), hist)
        want = list(hist)
        assert got == want
        assert sum(got) > 1000
        for t in range(4):
            assert dev_track.table_bytes(t) == ref_track.table_bytes(t)


def test_simlike_spanning_kmer_abundances_golden(hk):
    """kevlar/tests/test_simlike.py:82-106 on the device: four sketches counted from files, batched gets"""
    from conftest import check_spanning_kmer_abundances
    check_spanning_kmer_abundances(hk)
