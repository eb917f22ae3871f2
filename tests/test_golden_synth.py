"""BASELINE.json config 1 exactly: the 50 kb / 10x / k=31 trio of the bench's own seeded generator
(kevlar_amd/synth.py), whose FASTQ files were run through the REFERENCE's count / novel / filter / partition drivers
by tests/golden/make_golden_synth.py.  The generator must still produce those files, the oracle must reproduce the
reference's count tables, and (GPU) the product's drivers must reproduce the reference's outputs byte for byte."""
import gzip
import hashlib
import io
import json
import os

import pytest

from conftest import data_file, expected_file


def manifest():
    return json.load(open(expected_file('manifest-synth.json')))


def read_fastq_sequences(path):
    with gzip.open(path, 'rt') as fh:
        return [line.rstrip('\n') for i, line in enumerate(fh) if i % 4 == 1]


def test_generator_reproduces_the_committed_inputs():
    from kevlar_amd import synth
    m = manifest()
    packed = synth.trio_reads_packed(m['genome'], m['coverage'], 100)
    for name, words in packed.items():
        assert synth.unpack_reads(words, 100) == read_fastq_sequences(data_file('synth-cfg1/{}.fq.gz'.format(name))), name


def test_oracle_counts_the_synthetic_samples_like_the_reference_run(ok, tmp_path):
    m = manifest()
    for name in ('proband', 'mother', 'father'):
        seqs = read_fastq_sequences(data_file('synth-cfg1/{}.fq.gz'.format(name)))
        sketch = ok.Counttable(m['ksize'], 1e6 / 4, 4)
        bases, offs = ok.concat_reads(seqs)
        ok.consume_reads(sketch, bases, offs, len(seqs))
        path = str(tmp_path / (name + '.ct'))
        sketch.save(path)
        want = [line for line in m['cases']['count-' + name] if line.startswith('md5 ')][0].split()[1]
        assert hashlib.md5(open(path, 'rb').read()).hexdigest() == want
        assert '{} distinct k-mers stored'.format(sketch.n_unique_kmers()) in m['cases']['count-' + name][0]


@pytest.mark.gpu
def test_drivers_reproduce_the_reference_on_the_synthetic_trio(hk, tmp_path):
    from test_gpu_pipeline import run_cli, summary, load_partitions
    m = manifest()
    files = {n: data_file('synth-cfg1/{}.fq.gz'.format(n)) for n in ('proband', 'mother', 'father')}
    for name, path in files.items():
        ct = str(tmp_path / (name + '.ct'))
        _, log = run_cli(['count', '--ksize', str(m['ksize']), '--memory', m['memory'], ct, path])
        want = m['cases']['count-' + name]
        assert want[0] in log and want[1].rstrip(';') in log
        assert hashlib.md5(open(ct, 'rb').read()).hexdigest() == want[2].split()[1]
    out, log = run_cli(['novel', '--ksize', str(m['ksize']), '--memory', m['memory'], '--case', files['proband'],
                        '--control', files['mother'], '--control', files['father'], '--case-min', '6', '--ctrl-max', '1'])
    assert out == open(expected_file('novel-synth-cfg1.augfastq')).read()
    assert summary(m['cases']['novel-synth-cfg1.augfastq'][-1]) in log
    novel_path = str(tmp_path / 'novel.augfastq')
    open(novel_path, 'w').write(out)
    out, log = run_cli(['filter', '--memory', '500K', '--case-min', '6', '--ctrl-max', '1', '-o', str(tmp_path / 'filtered.augfastq'), novel_path])
    filtered = open(str(tmp_path / 'filtered.augfastq')).read()
    assert filtered == open(expected_file('filter-synth-cfg1.augfastq')).read()
    for needle in ('Processed 52 reads', 'Validated 46 reads'):
        assert needle in log
    out, log = run_cli(['partition', str(tmp_path / 'filtered.augfastq')])
    want = json.load(open(expected_file('partition-synth-cfg1.json')))
    got = load_partitions(out)
    assert sorted(got) == sorted(want['partitions'])
    for pid in got:
        assert sorted(set(s for _, s in got[pid])) == sorted(set(s for _, s in want['partitions'][pid]))
        assert len(got[pid]) == len(want['partitions'][pid])
    assert want['log'][0].split('] ')[-1] in log
