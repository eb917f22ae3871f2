"""The packed-read cache: with KEVLAR_PACK_CACHE=1 the first complete pass over a sequence file
leaves FILE.kvpack behind; later opens stream the stored 2-bit words into HBM without inflating or parsing, and can
still reproduce every record's text.  Round trip must be exact: same packed batches (hence same sketches), same record
text, same `novel` output bytes; a stale cache (source changed) must be ignored."""
import gzip
import os
import shutil

import numpy as np
import pytest

from conftest import data_file

pytestmark = pytest.mark.gpu


@pytest.fixture
def caching():
    os.environ['KEVLAR_PACK_CACHE'] = '1'
    yield
    os.environ.pop('KEVLAR_PACK_CACHE', None)


def records(hk, path, batch):
    out = []
    parser = hk.ReadParser(path)
    for tb in parser.text_batches(batch):
        out += [(r.name, r.sequence, r.quality) for r in (tb.record(i) for i in range(tb.n))]
        tb.batch.close()
    return out, parser


def test_cache_round_trip_records_and_sketches(hk, tmp_path, caching):
    rng = np.random.default_rng(4)
    letters = np.array(list('ACGT'))

    def rnd(n):
        return ''.join(letters[rng.integers(0, 4, size=n)])
    fq = str(tmp_path / 'mixed.fq.gz')
    with gzip.open(fq, 'wt') as fh:
        for i in range(150000):
            seq = rnd(int(rng.integers(20, 151)))
            if i % 997 == 0:
                seq = seq[:7] + 'N' + seq[8:]          # characters 2 bits cannot hold
            if i % 1499 == 0:
                seq = seq[:11] + 'acgtn' + seq[16:]
            fh.write('@read{} extra words/{}\n{}\n+\n{}\n'.format(i, i % 2 + 1, seq, ''.join(chr(33 + (j * 7 + i) % 40) for j in range(len(seq)))))
    fa = str(tmp_path / 'contigs.fa')
    with open(fa, 'w') as fh:
        for i, n in enumerate((40, 9000, 100, 25000, 17)):
            seq = rnd(n)
            fh.write('>contig{}\n'.format(i) + '\n'.join(seq[j:j + 70] for j in range(0, n, 70)) + '\n')
    for path, batch in ((fq, 40000), (fa, 3)):
        os.environ['KEVLAR_PACK_CACHE'] = '0'
        plain, parser = records(hk, path, batch)
        assert not parser.from_cache and not os.path.exists(path + '.kvpack')
        ref = hk.Counttable(31, 2e6, 4)
        ref.consume_seqfile(path)
        os.environ['KEVLAR_PACK_CACHE'] = '1'
        first = hk.Counttable(31, 2e6, 4)
        first.consume_seqfile(path)                     # a complete uploading pass: writes the cache
        assert os.path.exists(path + '.kvpack')
        again, parser = records(hk, path, 100000)
        assert parser.from_cache
        assert again == plain
        second = hk.Counttable(31, 2e6, 4)
        second.consume_seqfile(path)                    # from the cache: no inflate, no parsing, no packing kernel
        for t in range(4):
            assert ref.table_bytes(t) == first.table_bytes(t) == second.table_bytes(t)
        assert hk.ReadParser(path).num_reads == 0
    # the source changes: the cache no longer matches and is ignored (then replaced by the next complete pass)
    with gzip.open(fq, 'at') as fh:
        fh.write('@late\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\n' + 'I' * 36 + '\n')
    os.utime(fq, None)
    got, parser = records(hk, fq, 50000)
    assert not parser.from_cache and got[-1][0] == 'late' and len(got) == 150001


def test_novel_output_is_identical_from_the_cache(hk, tmp_path, caching):
    import kevlar_amd
    from test_gpu_pipeline import run_cli
    files = {}
    for name in ('proband', 'mother', 'father'):
        files[name] = str(tmp_path / (name + '.fq.gz'))
        shutil.copy(data_file('synth-cfg1/{}.fq.gz'.format(name)), files[name])
    argv = ['novel', '--ksize', '31', '--memory', '1M', '--case', files['proband'], '--control', files['mother'],
            '--control', files['father'], '--case-min', '6', '--ctrl-max', '1']
    os.environ['KEVLAR_PACK_CACHE'] = '0'
    want, _ = run_cli(argv)
    os.environ['KEVLAR_PACK_CACHE'] = '1'
    first, _ = run_cli(argv)                 # counting the samples writes the caches; the scan of the case already reads its cache
    assert all(os.path.exists(p + '.kvpack') for p in files.values())
    second, _ = run_cli(argv)                # everything from caches
    assert want == first == second and want.count('\n') > 100
