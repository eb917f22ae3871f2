"""`kevlar split` and `kevlar augment` (kevlar/split.py:14-44, kevlar/augment.py:13-45) on the GPU box:
on the reference's own fixtures (kevlar/tests/test_split.py, test_augment.py) and on what this build's novel ->
partition run on the device has just written -- the annotated reads pass through the native augmented-FASTQ codec of the
library on the way."""
import contextlib
import io

import pytest

from conftest import data_file

pytestmark = pytest.mark.gpu


def run_cli(arglist):
    import kevlar_amd
    args = kevlar_amd.cli.parser().parse_args(arglist)
    out, err = io.StringIO(), io.StringIO()
    old = kevlar_amd.logstream
    kevlar_amd.logstream = err
    try:
        with contextlib.redirect_stdout(out):
            kevlar_amd.cli.mains[args.cmd](args)
    finally:
        kevlar_amd.logstream = old
    return out.getvalue(), err.getvalue()


def records(path):
    import kevlar_amd
    return list(kevlar_amd.parse_augmented_fastx(kevlar_amd.open(path, 'r')))


def partitions(path):
    import kevlar_amd
    stream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(path, 'r'))
    return [(pid, reads) for pid, reads in kevlar_amd.parse_partitioned_reads(stream)]


def signature(rec):
    return (rec.name, rec.sequence, tuple((k.offset, rec.ikmerseq(k), tuple(k.abund)) for k in rec.annotations))


def test_split_reference_fixture_on_the_gpu_box(hk, tmp_path):
    """kevlar/tests/test_split.py:34-70"""
    run_cli(['split', data_file('fiveparts.augfastq.gz'), '3', str(tmp_path / 'out')])
    sizes = [[len(reads) for _, reads in partitions(str(tmp_path / 'out.{}.augfastx.gz'.format(i)))] for i in range(3)]
    assert sizes == [[67, 12], [23, 11], [15]]


def test_split_deals_the_device_pipeline_s_partitions(hk, tmp_path):
    """partition (read graph and components on the device) -> split: every partition arrives whole, in input order,
    partition i in file i mod N"""
    parted = str(tmp_path / 'parted.augfastq')
    run_cli(['partition', '--out', parted, data_file('pico-filtered.fq.gz')])
    whole = partitions(parted)
    assert len(whole) >= 5
    nfiles = 3
    run_cli(['split', parted, str(nfiles), str(tmp_path / 'deal')])
    dealt = [partitions(str(tmp_path / 'deal.{}.augfastx'.format(i))) for i in range(nfiles)]
    for i in range(nfiles):
        want = whole[i::nfiles]
        assert [pid for pid, _ in dealt[i]] == [pid for pid, _ in want]
        for (_, got_reads), (_, want_reads) in zip(dealt[i], want):
            assert [signature(r) for r in got_reads] == [signature(r) for r in want_reads]


def test_augment_reference_fixtures_on_the_gpu_box(hk, capsys):
    """kevlar/tests/test_augment.py:31-44: reads that lost their annotation lines get them back, byte for byte"""
    import kevlar_amd
    args = kevlar_amd.cli.parser().parse_args(['augment', data_file('reaugment.augfastq'), data_file('reaugment.fq')])
    kevlar_amd.augment.main(args)
    out, _ = capsys.readouterr()
    assert out == open(data_file('reaugment.out')).read()


def test_augment_restores_the_device_scan_s_annotations(hk, tmp_path):
    """the reads of this build's own `kevlar novel` output, stripped to plain FASTQ and augmented again from the
    annotated file, carry the same k-mers at the same offsets with the same abundances"""
    import kevlar_amd
    novel = str(tmp_path / 'novel.augfastq')
    run_cli(['novel', '--case', data_file('microtrios/trio-na-proband.fq.gz'), '--ksize', '25', '--case-min', '7', '--ctrl-max', '0',
             '--memory', '500K', '--control', data_file('microtrios/trio-na-father.fq.gz'),
             '--control', data_file('microtrios/trio-na-mother.fq.gz'), '--out', novel])
    annotated = records(novel)
    assert len(annotated) > 5 and all(r.annotations for r in annotated)
    bare = str(tmp_path / 'bare.fq')
    with open(bare, 'w') as fh:
        for r in annotated:
            fh.write('@{}\n{}\n+\n{}\n'.format(r.name, r.sequence, r.quality))
    again = str(tmp_path / 'again.augfastq')
    run_cli(['augment', '--out', again, novel, bare])
    assert [signature(r) for r in records(again)] == [signature(r) for r in annotated]
