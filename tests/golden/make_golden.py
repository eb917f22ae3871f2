#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python drivers over the CPU oracle.

Runs only in the build container (needs /root/reference); its outputs under tests/golden/
are committed and are what travels to the GPU box.  What it does:

1. copies /root/reference/kevlar to a scratch directory (never into this repo), builds its
   Cython codec (kevlar/sequence.pyx) there and stubs the imports that are off the
   count -> novel -> filter -> partition path (pysam, intervaltree, screed, the ksw2 /
   fermi-lite C extensions);
2. registers oracle/okhmer.py as the module ``khmer`` (khmer itself -- dib-lab/khmer @
   6c893074, reference Dockerfile:36 -- is not vendored and cannot be installed here);
3. copies the reference's test *data files* for this path into tests/golden/data/
   (fixtures: inputs and expected outputs only, no source text);
4. runs kevlar.count / novel / filter / partition / unband from the reference on those
   inputs and stores their outputs + asserted log lines in tests/golden/expected/.

Usage:  PYTHONHASHSEED=0 python tests/golden/make_golden.py
"""
import contextlib
import gzip
import io
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
REFDATA = os.path.join(REF, 'kevlar', 'tests', 'data')
DATA = os.path.join(HERE, 'data')
EXPECTED = os.path.join(HERE, 'expected')

# reference test data files used by the hot-path tests
FIXTURES = [
    'simple-genome-case-reads.fa.gz', 'simple-genome-ctrl1-reads.fa.gz',
    'simple-genome-ctrl2-reads.fa.gz', 'simple-genome-case.ct', 'simple-genome-ctrl1.ct',
    'simple-genome-ctrl2.ct', 'simple-genome-case-band-2-1.ct',
    'simple-genome-case-band-16-7.ct',
    'test.countgraph', 'test.smallcountgraph', 'test.counttable', 'test.smallcounttable',
    'test.nodegraph', 'test.nodetable', 'test.notasketchtype',
    'bogus-genome/refr.fa', 'bogus-genome/mask.nt', 'bogus-genome/mask-chr1.fa',
    'bogus-genome/mask-chr2.fa',
    'microtrios/trio-li-proband.fq.gz', 'microtrios/trio-li-mother.fq.gz',
    'microtrios/trio-li-father.fq.gz',
    'microtrios/trio-na-proband.fq.gz', 'microtrios/trio-na-mother.fq.gz',
    'microtrios/trio-na-father.fq.gz',
    'trio1/novel_3_1,2.txt', 'trio1/novel_1_1,2.txt',
    'screen-case.fa', 'screen-ctrl.fa', 'ambig.fasta',
    'collect.alpha.txt', 'collect.beta.1.txt', 'collect.beta.2.txt', 'worm.augfasta',
    'example1.augfastq', 'example2.augfastq', 'seqs-mates.augfastq',
    'dup.augfastq', 'dupl-part.augfastq.gz', 'dupl-part-2reads.augfastq.gz',
    'pico-filtered.fq.gz', 'connectivity-1311.augfastq', 'connectivity-1541.augfastq',
    'helium-unband/novel.band1.augfastq.gz', 'helium-unband/novel.band2.augfastq.gz',
    'helium-unband/novel.band3.augfastq.gz', 'helium-unband/novel.band4.augfastq.gz',
    'part-reads-simple.fa',
    # kevlar dist (kevlar/tests/test_dist.py)
    'minitrio/mask.nt', 'minitrio/trio-proband.fq.gz', 'minitrio/trio-proband-mask-counts.ct',
    'minitrio/trio-proband-dist.tsv', 'minitrio/trio-mother.fq.gz', 'minitrio/trio-father.fq.gz', 'minitrio/refr.fa',
    # kevlar split / augment (kevlar/tests/test_split.py, test_augment.py)
    'fiveparts.augfastq.gz', 'snorkel.augfastq', 'snorkel-contig.fasta', 'reaugment.augfastq', 'reaugment.fq',
    'reaugment.out', 'deadbeef.augfastq.gz', 'deadbeef.contig.fa', 'deadbeef.fq.gz', 'part-reads-mixed.fa',
    # progress lines (kevlar/tests/test_progress.py:18-28)
    'progind.txt',
]
# the three trio1 files behind test_novel.py:179-207 are 1.8 MB each: stored gzipped
GZ_FIXTURES = ['trio1/case1.fq', 'trio1/ctrl1.fq', 'trio1/ctrl2.fq',
               # two case samples + two controls: test_novel.py:108-144
               'trio1/case6.fq', 'trio1/case6b.fq', 'trio1/ctrl5.fq', 'trio1/ctrl6.fq']


def copy_fixtures():
    for rel in FIXTURES:
        dst = os.path.join(DATA, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REFDATA, rel), dst)
    for rel in GZ_FIXTURES:
        dst = os.path.join(DATA, rel + '.gz')
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        with open(os.path.join(REFDATA, rel), 'rb') as src, \
                gzip.GzipFile(dst, 'wb', mtime=0) as out:
            shutil.copyfileobj(src, out)


def import_reference():
    scratch = tempfile.mkdtemp(prefix='kevlar-ref-')
    shutil.copytree(os.path.join(REF, 'kevlar'), os.path.join(scratch, 'kevlar'))
    subprocess.check_call(['chmod', '-R', 'u+w', scratch])
    os.remove(os.path.join(scratch, 'kevlar', 'sequence.c'))
    subprocess.check_call(['cythonize', '-i', '-3', 'kevlar/sequence.pyx'], cwd=scratch,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, REPO)
    from oracle import okhmer
    sys.modules['khmer'] = okhmer
    sys.modules['khmer.khmer_args'] = okhmer.khmer_args
    for name in ('pysam', 'intervaltree', 'screed'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['intervaltree'].IntervalTree = object
    align = types.ModuleType('kevlar.alignment')
    align.contig_align = lambda *a, **k: None
    align.align_both_strands = lambda *a, **k: None
    sys.modules['kevlar.alignment'] = align
    asm = types.ModuleType('kevlar.assembly')
    asm.fml_asm = lambda *a, **k: iter(())
    sys.modules['kevlar.assembly'] = asm
    sys.path.insert(0, scratch)
    import kevlar
    return kevlar, scratch


def run_cli(kevlar, arglist):
    """Run one reference subcommand; returns (stdout, stderr-log)."""
    args = kevlar.cli.parser().parse_args(arglist)
    out, err = io.StringIO(), io.StringIO()
    kevlar.logstream = err
    with contextlib.redirect_stdout(out):
        kevlar.cli.mains[args.cmd](args)
    kevlar.logstream = None
    return out.getvalue(), err.getvalue()


def keep_lines(log, needles):
    return [ln.strip() for ln in log.split('\n') if any(n in ln for n in needles)]


def main():
    shutil.rmtree(DATA, ignore_errors=True)
    shutil.rmtree(EXPECTED, ignore_errors=True)
    os.makedirs(EXPECTED)
    copy_fixtures()
    kevlar, scratch = import_reference()
    d = lambda rel: os.path.join(REFDATA, rel)  # noqa: E731
    manifest = {'PYTHONHASHSEED': os.environ.get('PYTHONHASHSEED'), 'cases': {}}
    work = tempfile.mkdtemp(prefix='kevlar-golden-')

    # ---- count: log lines for the five golden .ct files (test_count.py:45-68)
    for name, nb, b in [('case', None, None), ('ctrl1', None, None), ('ctrl2', None, None),
                        ('case', 2, 1), ('case', 16, 7)]:
        arglist = ['count', '--ksize', '25', '--memory', '10K']
        tag = name
        if nb:
            arglist += ['--num-bands', str(nb), '--band', str(b)]
            tag = '{}-band-{}-{}'.format(name, nb, b)
        outct = os.path.join(work, tag + '.ct')
        arglist += [outct, d('simple-genome-{}-reads.fa.gz'.format(name))]
        _, log = run_cli(kevlar, arglist)
        with open(outct, 'rb') as f1, open(d('simple-genome-{}.ct'.format(tag)), 'rb') as f2:
            assert f1.read() == f2.read(), tag
        manifest['cases']['count-' + tag] = keep_lines(log, ['reads processed', 'estimated false'])

    # ---- novel on the micro trios (test_novel.py:80-105 uses banded mode; both stored)
    for trio in ('li', 'na'):
        base = ['novel', '--case', d('microtrios/trio-{}-proband.fq.gz'.format(trio)),
                '--ksize', '25', '--case-min', '7', '--ctrl-max', '0', '--memory', '500K',
                '--control', d('microtrios/trio-{}-father.fq.gz'.format(trio)),
                '--control', d('microtrios/trio-{}-mother.fq.gz'.format(trio))]
        out, log = run_cli(kevlar, base)
        name = 'novel-trio-{}.augfastq'.format(trio)
        with open(os.path.join(EXPECTED, name), 'w') as fh:
            fh.write(out)
        manifest['cases'][name] = keep_lines(log, ['Found', 'reads processed'])
        for band in (1, 2):   # reference quirk mode (novel.py:144-147), per-band files
            out, log = run_cli(kevlar, base + ['--num-bands', '2', '--band', str(band)])
            name = 'novel-trio-{}-refband-2-{}.augfastq'.format(trio, band)
            with open(os.path.join(EXPECTED, name), 'w') as fh:
                fh.write(out)
            manifest['cases'][name] = keep_lines(log, ['Found', 'reads processed'])

    # ---- novel trio1 with --skip-until (test_novel.py:179-207)
    readname = 'bogus-genome-chr1_115_449_0:0:0_0:0:0_1f4/1'
    out, log = run_cli(kevlar, [
        'novel', '--ctrl-max', '0', '--case-min', '6', '--case', d('trio1/case1.fq'),
        '--control', d('trio1/ctrl1.fq'), '--control', d('trio1/ctrl2.fq'),
        '--skip-until', readname])
    with open(os.path.join(EXPECTED, 'novel-trio1-skipuntil.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-trio1-skipuntil.augfastq'] = keep_lines(log, ['Found'])
    assert '29 unique novel kmers in 14 reads' in log
    out, log = run_cli(kevlar, [
        'novel', '--ctrl-max', '0', '--case-min', '6', '--case', d('trio1/case1.fq'),
        '--control', d('trio1/ctrl1.fq'), '--control', d('trio1/ctrl2.fq')])
    with open(os.path.join(EXPECTED, 'novel-trio1.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-trio1.augfastq'] = keep_lines(log, ['Found'])

    # ---- two case samples, two controls, from saved counts (test_novel.py:108-144), and -- through novel() itself, which
    #      the CLI cannot ask for -- two cases and no control at all (novel.py:36-51: an empty control loop)
    tables = []
    for tag, rel in [('case1', 'trio1/case6.fq'), ('case2', 'trio1/case6b.fq'), ('ctrl1', 'trio1/ctrl5.fq'), ('ctrl2', 'trio1/ctrl6.fq')]:
        tables.append(os.path.join(work, 'two-' + tag + '.ct'))
        run_cli(kevlar, ['count', '--ksize', '19', '--memory', '1e7', tables[-1], d(rel)])
    out, log = run_cli(kevlar, ['novel', '--ksize', '19', '--memory', '1e7', '--ctrl-max', '1', '--case-min', '7',
                                '--case', d('trio1/case6.fq'), '--case', d('trio1/case6b.fq'),
                                '--case-counts', tables[0], tables[1], '--control-counts', tables[2], tables[3]])
    assert out.strip() != ''
    with open(os.path.join(EXPECTED, 'novel-trio1-two-cases.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-trio1-two-cases.augfastq'] = keep_lines(log, ['Found', 'counttables'])
    sketches = [kevlar.sketch.load(t) for t in tables[:2]]
    buf = io.StringIO()
    kevlar.logstream = io.StringIO()
    stream = kevlar.multi_file_iter_khmer([d('trio1/case6.fq'), d('trio1/case6b.fq')])
    for rec in kevlar.novel.novel(stream, sketches, [], ksize=19, casemin=12, ctrlmax=0):
        kevlar.print_augmented_fastx(rec, buf)
    manifest['cases']['novel-trio1-two-cases-no-control.augfastq'] = keep_lines(kevlar.logstream.getvalue(), ['Found'])
    kevlar.logstream = None
    assert buf.getvalue().strip() != ''
    with open(os.path.join(EXPECTED, 'novel-trio1-two-cases-no-control.augfastq'), 'w') as fh:
        fh.write(buf.getvalue())

    # ---- novel with abundance screen (test_novel.py:167-176)
    out, log = run_cli(kevlar, ['novel', '--ksize', '25', '--ctrl-max', '1', '--case-min', '8',
                                '--case', d('screen-case.fa'), '--control', d('screen-ctrl.fa'),
                                '--abund-screen', '3'])
    with open(os.path.join(EXPECTED, 'novel-screen.augfasta'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-screen.augfasta'] = keep_lines(log, ['Found'])
    assert '>seq_error' not in out

    # ---- novel from saved counts + ambiguous reads (test_novel.py:264-282)
    out, log = run_cli(kevlar, ['novel', '-k', '25', '--case', d('simple-genome-case-reads.fa.gz'),
                                d('ambig.fasta'), '--case-counts', d('simple-genome-case.ct'),
                                '--control-counts', d('simple-genome-ctrl1.ct'),
                                d('simple-genome-ctrl2.ct')])
    with open(os.path.join(EXPECTED, 'novel-simple-genome.augfasta'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-simple-genome.augfasta'] = keep_lines(log, ['Found', 'counttables'])

    # ---- filter (test_filter.py:75-87 and :44-57)
    out, log = run_cli(kevlar, ['filter', '--mask', d('bogus-genome/mask.nt'), '--memory', '10M',
                                '--max-fpr', '0.001', '--case-min', '6',
                                d('trio1/novel_3_1,2.txt')])
    with open(os.path.join(EXPECTED, 'filter-trio1-masked.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['filter-trio1-masked.augfastq'] = keep_lines(log, ['Processed', 'Validated',
                                                                        'FPR for'])
    assert 'Processed 178 reads' in log and 'Validated 18 reads' in log
    for name, infile, kw in [
            ('filter-trio1-nomask.augfastq', 'trio1/novel_3_1,2.txt', dict(memory=1e7)),
            ('filter-alpha.augfastq', 'collect.alpha.txt', dict(memory=500)),
            ('filter-worm.augfasta', 'worm.augfasta', dict(memory=1000, casemin=5, ctrlmax=0))]:
        buf = io.StringIO()
        kevlar.logstream = io.StringIO()
        for rec in kevlar.filter.filter(d(infile), **kw):
            kevlar.print_augmented_fastx(rec, buf)
        kevlar.logstream = None
        with open(os.path.join(EXPECTED, name), 'w') as fh:
            fh.write(buf.getvalue())

    # ---- partition (test_partition.py:37-154): membership per partition id
    for name, infile, extra in [
            ('partition-dup', 'dup.augfastq', []),
            ('partition-dup-nodedup', 'dup.augfastq', ['--no-dedup']),
            ('partition-pico-minabund5', 'pico-filtered.fq.gz', ['--min-abund', '5']),
            ('partition-pico-default', 'pico-filtered.fq.gz', []),
            ('partition-conn1311', 'connectivity-1311.augfastq', []),
            ('partition-conn1541-nodedup', 'connectivity-1541.augfastq', ['--no-dedup']),
            ('partition-conn1311-strict', 'connectivity-1311.augfastq', ['--strict']),
            ('partition-conn1541-strict-nodedup', 'connectivity-1541.augfastq', ['--strict', '--no-dedup']),
            ('partition-pico-strict', 'pico-filtered.fq.gz', ['--strict'])]:
        out, log = run_cli(kevlar, ['partition'] + extra + [d(infile)])
        parts = {}
        reader = kevlar.parse_augmented_fastx(io.StringIO(out))
        if out.strip():
            for rec in reader:
                pid = kevlar.seqio.partition_id(rec.name)
                parts.setdefault(pid, []).append(
                    [rec.name.rsplit(' kvcc=', 1)[0], kevlar.revcommin(rec.sequence)])
        for pid in parts:
            parts[pid].sort()
        with open(os.path.join(EXPECTED, name + '.json'), 'w') as fh:
            json.dump({'partitions': parts, 'log': keep_lines(log, ['grouped'])}, fh, indent=0,
                      sort_keys=True)

    # ---- read graph edge counts (test_readgraph.py:20-31)
    edges = {}
    for infile in ('connectivity-1311.augfastq', 'connectivity-1541.augfastq'):
        with open(d(infile)) as fh:
            reads = list(kevlar.parse_augmented_fastx(fh))
        rg = kevlar.ReadGraph()
        rg.load(reads)
        rg.populate_edges()
        edges[infile] = {'relaxed': rg.number_of_edges()}
        rg = kevlar.ReadGraph()
        rg.load(reads)
        rg.populate_edges(strict=True)
        edges[infile]['strict'] = rg.number_of_edges()
    manifest['readgraph_edges'] = edges

    # ---- unband (test_unband.py:25-45): order is hash(name)-dependent -> store sorted by name
    instream = kevlar.seqio.afxstream(
        [d('helium-unband/novel.band{}.augfastq.gz'.format(i)) for i in (1, 2, 3, 4)])
    kevlar.logstream = io.StringIO()
    reads = sorted(kevlar.unband.unband(instream, numbatches=16), key=lambda r: r.name)
    kevlar.logstream = None
    buf = io.StringIO()
    for rec in reads:
        kevlar.print_augmented_fastx(rec, buf)
    with open(os.path.join(EXPECTED, 'unband-helium.sorted.augfastq'), 'w') as fh:
        fh.write(buf.getvalue())

    with open(os.path.join(EXPECTED, 'manifest.json'), 'w') as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    shutil.rmtree(scratch, ignore_errors=True)
    shutil.rmtree(work, ignore_errors=True)
    print('golden vectors written to', EXPECTED)


if __name__ == '__main__':
    main()
