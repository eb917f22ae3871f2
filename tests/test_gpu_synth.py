"""The family kv_synth.hip writes into HBM (kv_reads_generate: what bench.py's cfg4-band workload counts) against its
numpy restatement (kevlar_amd.synth.device_family_reads), bit for bit, and the properties the bench relies on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('sample', [0, 1, 2])
@pytest.mark.parametrize('read_len', [100, 37])
def test_device_reads_equal_the_numpy_restatement(hk, sample, read_len):
    from kevlar_amd import synth
    G, seed, first, n = 3_000_000_000, 42, 123_456_789, 3000
    batch = hk.ReadBatch.generate(G, seed, sample, first, n, read_len)
    wpr = (read_len + 15) // 16
    got = batch.packed_words(0, n * wpr).reshape(n, wpr)
    want = synth.pack_codes(synth.device_family_reads(G, seed, sample, np.arange(first, first + n), read_len))
    assert np.array_equal(got, want)


def test_device_family_counts_like_host_packed_reads(hk, ok):
    """a generated batch is an ordinary batch: its count equals the oracle's count of the same reads as text"""
    from kevlar_amd import synth
    G, seed, n, L, k = 200_000, 7, 4000, 100, 31
    batch = hk.ReadBatch.generate(G, seed, 0, 0, n, L)
    codes = synth.device_family_reads(G, seed, 0, np.arange(n), L)
    seqs = [row.tobytes().decode('ascii') for row in synth.ALPHABET[codes]]
    dev = hk.Counttable(k, 1e5, 4)
    assert dev.consume_batch(batch) == n * (L - k + 1)
    ref = ok.Counttable(k, 1e5, 4)
    bases, offs = ok.concat_reads(seqs)
    ok.consume_reads(ref, bases, offs, n)
    for t in range(4):
        assert dev.table_bytes(t) == ref.table_bytes(t)


def test_device_family_has_the_trio_structure(hk):
    """30x of a small genome: the proband carries k-mers neither parent has (de novo variants), at the expected scale"""
    G, seed, L, k = 1_000_000, 11, 100, 31
    n = G * 30 // L
    sk = {}
    for name, sample in (('proband', 0), ('mother', 1), ('father', 2)):
        sk[name] = hk.Counttable(k, 4e7 / 4, 4)
        sk[name].consume_batch(hk.ReadBatch.generate(G, seed, sample, 0, n, L))
    r, o, a, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], hk.ReadBatch.generate(G, seed, 0, 0, n, L), 6, 1)
    # 200 de novo SNVs per Mb, each covered ~15x on its haplotype, 31 k-mers each, most of them seen >= 6 times
    assert 200 * 31 * 8 < len(r) < 200 * 31 * 16
