"""novel / filter / partition / unband on the HIP path vs golden vectors produced by the
reference's own drivers (tests/golden/make_golden.py) and vs the oracle."""
import contextlib
import io
import json
import os
import re

import numpy as np
import pytest

from conftest import data_file, expected_file

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


def manifest():
    return json.load(open(expected_file('manifest.json')))


def summary(line):
    return re.search(r'Found \d+ instances of \d+ unique novel kmers in \d+ reads', line).group(0)


@pytest.mark.parametrize('trio', ['li', 'na'])
def test_novel_microtrio_bytes(hk, trio):
    """Unbanded `kevlar novel` output is byte-identical to the reference's; so are the count
    log lines of the three samples and the summary line."""
    base = ['novel', '--case', data_file('microtrios/trio-{}-proband.fq.gz'.format(trio)), '--ksize', '25',
            '--case-min', '7', '--ctrl-max', '0', '--memory', '500K',
            '--control', data_file('microtrios/trio-{}-father.fq.gz'.format(trio)),
            '--control', data_file('microtrios/trio-{}-mother.fq.gz'.format(trio))]
    out, log = run_cli(base)
    name = 'novel-trio-{}.augfastq'.format(trio)
    assert out == open(expected_file(name)).read()
    want = manifest()['cases'][name]
    for line in want[:-1]:
        assert line in log
    assert summary(want[-1]) in log


@pytest.mark.parametrize('trio', ['li', 'na'])
@pytest.mark.parametrize('band', [1, 2])
def test_novel_reference_band_quirk_bytes(hk, trio, band):
    """--ref-band-quirk reproduces the reference's per-band files literally
    (kevlar/novel.py:144-147; band 1 of 2 is empty)."""
    base = ['novel', '--case', data_file('microtrios/trio-{}-proband.fq.gz'.format(trio)), '--ksize', '25',
            '--case-min', '7', '--ctrl-max', '0', '--memory', '500K',
            '--control', data_file('microtrios/trio-{}-father.fq.gz'.format(trio)),
            '--control', data_file('microtrios/trio-{}-mother.fq.gz'.format(trio)),
            '--num-bands', '2', '--band', str(band), '--ref-band-quirk']
    out, log = run_cli(base)
    name = 'novel-trio-{}-refband-2-{}.augfastq'.format(trio, band)
    assert out == open(expected_file(name)).read()
    want = manifest()['cases'][name]
    for line in want[:-1]:
        assert line in log
    assert summary(want[-1]) in log


def test_novel_range_banding_union_equals_unbanded(hk):
    """Default banding (hash-range, the count-side rule): every annotation satisfies the
    thresholds (kevlar/tests/test_novel.py:80-105) and, with tables large enough that no
    Count-Min collision matters, the union over bands is the unbanded result."""
    import kevlar_amd
    base = ['novel', '--case', data_file('microtrios/trio-li-proband.fq.gz'), '--ksize', '25',
            '--case-min', '7', '--ctrl-max', '0', '--memory', '50M',
            '--control', data_file('microtrios/trio-li-father.fq.gz'),
            '--control', data_file('microtrios/trio-li-mother.fq.gz')]
    whole, _ = run_cli(base)
    merged = {}
    for band in (1, 2, 3, 4):
        out, _ = run_cli(base + ['--num-bands', '4', '--band', str(band)])
        for line in out.split('\n'):
            if line.endswith('#') and not line.startswith('#mateseq'):
                m = re.search(r'(\d+) (\d+) (\d+)#$', line)
                assert int(m.group(1)) >= 7 and m.group(2) == '0' and m.group(3) == '0', line
        for rec in kevlar_amd.parse_augmented_fastx(io.StringIO(out)):
            if rec is None:
                continue
            merged.setdefault(rec.name, set()).update((k.offset, k.abund) for k in rec.annotations)
    want = {rec.name: set((k.offset, k.abund) for k in rec.annotations)
            for rec in kevlar_amd.parse_augmented_fastx(io.StringIO(whole))}
    assert merged == want and len(want) > 0


def test_novel_skip_until(hk):
    """kevlar/tests/test_novel.py:179-207."""
    readname = 'bogus-genome-chr1_115_449_0:0:0_0:0:0_1f4/1'
    base = ['novel', '--ctrl-max', '0', '--case-min', '6', '--case', data_file('trio1/case1.fq.gz'),
            '--control', data_file('trio1/ctrl1.fq.gz'), '--control', data_file('trio1/ctrl2.fq.gz')]
    out, log = run_cli(base + ['--skip-until', readname])
    assert 'Found read bogus-genome-chr1_115_449_0:0:0_0:0:0_1f4/1 (skipped 1001 reads)' in log
    assert '29 unique novel kmers in 14 reads' in log
    assert out == open(expected_file('novel-trio1-skipuntil.augfastq')).read()
    out, log = run_cli(base)
    assert out == open(expected_file('novel-trio1.augfastq')).read()
    assert 'Found 209 instances of 29 unique novel kmers in 18 reads' in log
    out, log = run_cli(base + ['--skip-until', 'BOGUSREADNAME'])
    assert 'Found read' not in log and '(skipped ' not in log
    assert 'Found 0 instances of 0 unique novel kmers in 0 reads' in log


def test_novel_abund_screen(hk):
    """kevlar/tests/test_novel.py:167-176."""
    out, log = run_cli(['novel', '--ksize', '25', '--ctrl-max', '1', '--case-min', '8', '--case',
                        data_file('screen-case.fa'), '--control', data_file('screen-ctrl.fa'),
                        '--abund-screen', '3'])
    assert '>seq_error' not in out
    assert out == open(expected_file('novel-screen.augfasta')).read()
    assert summary(manifest()['cases']['novel-screen.augfasta'][0]) in log


def test_novel_load_counts_and_ambiguous_reads(hk):
    """kevlar/tests/test_novel.py:264-282: saved sketches + a case file with non-ACGT reads."""
    out, log = run_cli(['novel', '-k', '25', '--case', data_file('simple-genome-case-reads.fa.gz'),
                        data_file('ambig.fasta'), '--case-counts', data_file('simple-genome-case.ct'),
                        '--control-counts', data_file('simple-genome-ctrl1.ct'),
                        data_file('simple-genome-ctrl2.ct')])
    assert 'counttables for 2 sample(s) provided' in log
    assert out == open(expected_file('novel-simple-genome.augfasta')).read()


def test_novel_save_counts(hk, tmp_path):
    """kevlar/tests/test_novel.py:210-262."""
    import filecmp
    outdir = str(tmp_path)
    for ind in ('father', 'mother', 'proband'):
        run_cli(['count', '--ksize', '27', '--memory', '500K', '{}/{}.ct'.format(outdir, ind),
                 data_file('microtrios/trio-na-{}.fq.gz'.format(ind))])
    _, log = run_cli(['novel', '--ksize', '27', '--out', outdir + '/novel.augfastq.gz',
                      '--save-case-counts', outdir + '/kid.ct', '--save-ctrl-counts', outdir + '/mom.ct',
                      outdir + '/dad.ct', '--case', data_file('microtrios/trio-na-proband.fq.gz'),
                      '--control', data_file('microtrios/trio-na-mother.fq.gz'),
                      '--control', data_file('microtrios/trio-na-father.fq.gz'), '--memory', '500K'])
    for a, b in zip(('father', 'mother', 'proband'), ('dad', 'mom', 'kid')):
        assert filecmp.cmp('{}/{}.ct'.format(outdir, a), '{}/{}.ct'.format(outdir, b), shallow=False)
    _, log = run_cli(['novel', '--ksize', '27', '--out', outdir + '/novel2.augfastq.gz',
                      '--save-case-counts', outdir + '/kid2.ct', '--save-ctrl-counts', outdir + '/mom2.ct',
                      outdir + '/dad2.ct', outdir + '/sib2.ct', '--case', data_file('microtrios/trio-na-proband.fq.gz'),
                      '--control', data_file('microtrios/trio-na-mother.fq.gz'),
                      '--control', data_file('microtrios/trio-na-father.fq.gz'), '--memory', '500K'])
    assert 'stubbornly refusing to save k-mer counts' in log


def test_novel_api_errors(hk):
    import kevlar_amd
    with pytest.raises(ValueError, match='Must specify `numbands` and `band` together'):
        list(kevlar_amd.novel.novel(None, [], [], numbands=4))
    with pytest.raises(ValueError, match='Must specify `numbands` and `band` together'):
        list(kevlar_amd.novel.novel(None, [], [], band=0))
    with pytest.raises(ValueError, match='`band` must be a value between 0 and 3'):
        list(kevlar_amd.novel.novel(None, [], [], numbands=4, band=-1))
    args = kevlar_amd.cli.parser().parse_args(['novel', '--case', 'case1.fq', '--control', 'cntl1.fq', '--band', '1'])
    with pytest.raises(ValueError, match='Must specify --num-bands and --band together'):
        kevlar_amd.novel.main(args)


def test_novel_scan_vs_oracle_multi_sample_k51(hk, ok):
    """Config-5 shape in miniature: 1 case + 3 controls, k=51 (3 murmur blocks + 3-byte tail),
    with and without the abundance screen, both band modes."""
    rng = np.random.default_rng(3)
    letters = np.array(list('ACGT'))
    genome = rng.integers(0, 4, size=6000)

    def sample(g, n, err=0.004):
        out = []
        for s in rng.integers(0, len(g) - 120, size=n):
            r = g[s:s + 120].copy()
            mut = rng.random(120) < err
            r[mut] = (r[mut] + rng.integers(1, 4, size=int(mut.sum()))) % 4
            out.append(''.join(letters[r]))
        return out

    kid = genome.copy()
    kid[[1500, 3000, 4500]] = (kid[[1500, 3000, 4500]] + 1) % 4
    samples = [sample(kid, 2500)] + [sample(genome, 2500) for _ in range(3)]
    dev = [hk.Counttable(51, 3e5, 4) for _ in samples]
    ref = [ok.Counttable(51, 3e5, 4) for _ in samples]
    for d, r, seqs in zip(dev, ref, samples):
        d.consume_batch(hk.ReadBatch(seqs))
        bases, offs = ok.concat_reads(seqs)
        ok.consume_reads(r, bases, offs, len(seqs))
    batch = hk.ReadBatch(samples[0])
    bases, offs = ok.concat_reads(samples[0])
    for screen, band_mode, nbands, band in [(0, 0, 0, 0), (2, 0, 0, 0), (0, 1, 4, 1), (0, 2, 4, 2), (3, 1, 2, 0)]:
        r, o, a, disc = hk.novel_scan(dev[:1], dev[1:], batch, 5, 1, screen=screen, band_mode=band_mode,
                                      nbands=nbands, band=band)
        got = [(int(r[i]), int(o[i]), tuple(int(x) for x in a[i])) for i in range(len(r))]
        want, status = ok.novel_scan(ref[:1], ref[1:], bases, offs, len(samples[0]), 51, 5, 1, screen=screen,
                                     band_mode=band_mode, nbands=nbands, band=band)
        assert got == want
        assert sorted(disc.tolist()) == [i for i, s in enumerate(status) if s == 2]
        if screen == 0 and band_mode == 0:
            assert len(got) > 0


def test_filter_golden(hk):
    """kevlar/tests/test_filter.py:27-87."""
    import kevlar_amd
    out, log = run_cli(['filter', '--mask', data_file('bogus-genome/mask.nt'), '--memory', '10M', '--max-fpr',
                        '0.001', '--case-min', '6', data_file('trio1/novel_3_1,2.txt')])
    assert 'Processed 178 reads' in log and 'Validated 18 reads' in log
    assert 'FPR for re-computed k-mer counts: 0.000' in log
    assert out == open(expected_file('filter-trio1-masked.augfastq')).read()
    for name, infile, kw, nreads in [
            ('filter-trio1-nomask.augfastq', 'trio1/novel_3_1,2.txt', dict(memory=1e7), None),
            ('filter-alpha.augfastq', 'collect.alpha.txt', dict(memory=500), 8),
            ('filter-worm.augfasta', 'worm.augfasta', dict(memory=1000, casemin=5, ctrlmax=0), 5)]:
        buf = io.StringIO()
        recs = list(kevlar_amd.filter.filter(data_file(infile), **kw))
        for rec in recs:
            kevlar_amd.print_augmented_fastx(rec, buf)
        assert buf.getvalue() == open(expected_file(name)).read()
        if nreads is not None:
            assert len(recs) == nreads
    # (424, 5782) with no mask / (13, 171) with the genome mask: kevlar/tests/test_filter.py:44-57
    for mask, nkmers, ninst in [(None, 424, 5782), (kevlar_amd.sketch.load(data_file('bogus-genome/mask.nt')), 13, 171)]:
        ikmers = {}
        for read in kevlar_amd.filter.filter(data_file('trio1/novel_3_1,2.txt'), memory=1e7, mask=mask):
            for ikmer in read.annotations:
                key = kevlar_amd.revcommin(read.ikmerseq(ikmer))
                ikmers[key] = ikmers.get(key, 0) + 1
        assert (len(ikmers), sum(ikmers.values())) == (nkmers, ninst)


def load_partitions(text):
    import kevlar_amd
    parts = {}
    if text.strip():
        for rec in kevlar_amd.parse_augmented_fastx(io.StringIO(text)):
            pid = kevlar_amd.seqio.partition_id(rec.name)
            parts.setdefault(pid, []).append([rec.name.rsplit(' kvcc=', 1)[0], kevlar_amd.revcommin(rec.sequence)])
    return parts


@pytest.mark.parametrize('name,infile,extra', [
    ('partition-dup', 'dup.augfastq', []),
    ('partition-dup-nodedup', 'dup.augfastq', ['--no-dedup']),
    ('partition-pico-minabund5', 'pico-filtered.fq.gz', ['--min-abund', '5']),
    ('partition-pico-default', 'pico-filtered.fq.gz', []),
    ('partition-conn1311', 'connectivity-1311.augfastq', []),
    ('partition-conn1541-nodedup', 'connectivity-1541.augfastq', ['--no-dedup']),
])
def test_partition_golden(hk, name, infile, extra):
    """Partition ids and membership vs the reference (relation P3 of SURVEY.md 8(a): same
    numbering; per partition the same set of canonical sequences, and the same names up to
    identical-sequence duplicates when dedup is on)."""
    out, log = run_cli(['partition'] + extra + [data_file(infile)])
    want = json.load(open(expected_file(name + '.json')))
    got = load_partitions(out)
    assert sorted(got) == sorted(want['partitions'])
    for pid in got:
        assert sorted(set(s for _, s in got[pid])) == sorted(set(s for _, s in want['partitions'][pid]))
        assert len(got[pid]) == len(want['partitions'][pid])
        if '--no-dedup' in extra:
            assert sorted(got[pid]) == sorted(want['partitions'][pid])
    assert want['log'][0].split('] ')[-1] in log


def test_partition_known_answers(hk, tmp_path):
    """kevlar/tests/test_partition.py:37-154."""
    import kevlar_amd
    out, log = run_cli(['partition', '--split', str(tmp_path / 'dedup'), data_file('dup.augfastq')])
    assert 'grouped 16 reads into 1 connected components' in log
    recs = list(kevlar_amd.parse_augmented_fastx(kevlar_amd.open(str(tmp_path / 'dedup.cc1.augfastq.gz'), 'r')))
    assert len(recs) == 16
    out, log = run_cli(['partition', '--no-dedup', data_file('dup.augfastq')])
    assert 'grouped 18 reads into 1 connected components' in log
    stream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(data_file('dupl-part.augfastq.gz'), 'r'))
    assert len(list(kevlar_amd.partition.partition(stream, minabund=5))) == 0
    stream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(data_file('dupl-part-2reads.augfastq.gz'), 'r'))
    assert len(list(kevlar_amd.partition.partition(stream, minabund=5, dedup=False))) == 0
    stream = kevlar_amd.parse_augmented_fastx(kevlar_amd.open(data_file('pico-filtered.fq.gz'), 'r'))
    assert len(list(kevlar_amd.partition.partition(stream, minabund=6))) == 10
    out, log = run_cli(['partition', '--min-abund', '5', data_file('pico-filtered.fq.gz')])
    assert len(set(re.findall(r'kvcc=\d+', out))) == 10


def test_readgraph_edges(hk):
    """kevlar/tests/test_readgraph.py:20-31 (relaxed mode)."""
    import kevlar_amd
    want = manifest()['readgraph_edges']
    for infile, edges in [('connectivity-1311.augfastq', 30), ('connectivity-1541.augfastq', 31)]:
        with open(data_file(infile)) as fh:
            reads = list(kevlar_amd.parse_augmented_fastx(fh))
        rg = kevlar_amd.ReadGraph()
        rg.load(reads)
        rg.populate_edges()
        assert rg.number_of_edges() == want[infile]['relaxed']
        assert rg.number_of_edges() == pytest.approx(edges, 1)


def test_readgraph_components_random_vs_host_union_find(hk):
    """Randomised check of the device union-find against a plain dict/set restatement of
    kevlar/readgraph.py:43-84,104-125 (min/max abundance filter included)."""
    rng = np.random.default_rng(99)
    letters = np.array(list('ACGT'))
    k = 19
    genome = ''.join(letters[rng.integers(0, 4, size=3000)])
    rc = lambda s: s[::-1].translate(str.maketrans('ACGT', 'TGCA'))  # noqa: E731
    reads, ann_read, ann_off = [], [], []
    for i in range(400):
        s = int(rng.integers(0, len(genome) - 80))
        seq = genome[s:s + 80]
        if rng.random() < 0.5:
            seq = rc(seq)
        reads.append(seq)
        for off in rng.choice(80 - k + 1, size=int(rng.integers(0, 4)), replace=False):
            ann_read.append(i)
            ann_off.append(int(off))
    node_of_read = np.arange(len(reads), dtype=np.uint32)
    node_of_read[350:] = node_of_read[:50]           # duplicate names share a node
    n_nodes = 350
    batch = hk.ReadBatch(reads)
    for minab, maxab in [(0, 0), (2, 0), (2, 3), (0, 2)]:
        labels, nedges = hk.readgraph_components(batch, k, ann_read, ann_off, node_of_read, n_nodes, minab, maxab,
                                                 want_edges=True)
        groups = {}
        for r, o in zip(ann_read, ann_off):
            km = reads[r][o:o + k]
            groups.setdefault(min(km, rc(km)), set()).add(int(node_of_read[r]))
        parent = list(range(n_nodes))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        edges = set()
        for nodes in groups.values():
            if (minab and len(nodes) < minab) or (maxab and len(nodes) > maxab):
                continue
            nodes = sorted(nodes)
            for a in nodes:
                for b in nodes:
                    if a < b:
                        edges.add((a, b))
                ra, rb = find(nodes[0]), find(a)
                if ra != rb:
                    parent[max(ra, rb)] = min(ra, rb)
        want = {}
        for x in range(n_nodes):
            want.setdefault(find(x), []).append(x)
        want_labels = np.empty(n_nodes, dtype=np.uint32)
        for members in want.values():
            want_labels[members] = min(members)
        assert np.array_equal(labels, want_labels)
        assert nedges == len(edges)


def test_unband_golden(hk):
    """kevlar/tests/test_unband.py:25-45 (host-side merge; order is name-sorted here)."""
    import kevlar_amd
    infiles = [data_file('helium-unband/novel.band{}.augfastq.gz'.format(i)) for i in (1, 2, 3, 4)]
    reads = sorted(kevlar_amd.unband.unband(kevlar_amd.seqio.afxstream(infiles), numbatches=16), key=lambda r: r.name)
    assert len(reads) == 135
    some = [r for r in reads if r.name == 'seq1_haplo1_285110_285519_1:0:0_0:0:0_2dbcd/1'][0]
    assert len(some.annotations) == 75
    buf = io.StringIO()
    for rec in reads:
        kevlar_amd.print_augmented_fastx(rec, buf)
    assert buf.getvalue() == open(expected_file('unband-helium.sorted.augfastq')).read()


def test_novel_tally_counts_kmers_in_front_of_a_screen_trip(hk, ok, kevlar_log):
    """With --abund-screen the reference adds an interesting k-mer to its set of unique novel k-mers as soon as it
    finds it, and only later -- at the first k-mer of the read whose case abundance is under the screen -- drops the
    read (kevlar/novel.py:152-164).  The closing line therefore counts k-mers of reads that are not reported."""
    import kevlar_amd
    from kevlar_amd import synth
    trio = synth.make_trio(30000, 5, denovo_per_mb=2000)
    reads = {n: synth.unpack_reads(synth.sample_reads_packed(trio[n], 9000, 100, 0.01, 77 + i), 100)
             for i, n in enumerate(('proband', 'mother', 'father'))}
    k, casemin, ctrlmax, screen = 25, 6, 1, 3
    dev, ref = {}, {}
    for n in reads:
        dev[n], ref[n] = hk.Counttable(k, 2e5, 4), ok.Counttable(k, 2e5, 4)
        dev[n].consume_batch(hk.ReadBatch(reads[n]))
        bases, offs = ok.concat_reads(reads[n])
        ok.consume_reads(ref[n], bases, offs, len(reads[n]))
    # the reference's loop, k-mer by k-mer, over the oracle's sketches
    unique, instances, nreads, dropped_with_kmers = set(), 0, 0, 0
    for seq in reads['proband']:
        found, discard = [], False
        for i in range(len(seq) - k + 1):
            kmer = seq[i:i + k]
            a = ref['proband'].get(kmer)
            if a < casemin:
                if a < screen:
                    discard = True
                    break
                continue
            if ref['mother'].get(kmer) > ctrlmax or ref['father'].get(kmer) > ctrlmax:
                continue
            found.append(kmer)
            unique.add(kevlar_amd.revcommin(kmer))
        if discard:
            dropped_with_kmers += 1 if found else 0
        elif found:
            nreads += 1
            instances += len(found)
    assert dropped_with_kmers > 0, 'the input must contain discarded reads that hold interesting k-mers'

    class Rec(object):
        def __init__(self, i, s):
            self.name, self.sequence, self.quality = 'r{}'.format(i), s, None
    stream = [Rec(i, s) for i, s in enumerate(reads['proband'])]
    out = list(kevlar_amd.novel.novel(stream, [dev['proband']], [dev['mother'], dev['father']], ksize=k, abundscreen=screen,
                                      casemin=casemin, ctrlmax=ctrlmax))
    assert len(out) == nreads and sum(len(r.annotations) for r in out) == instances
    assert 'Found {:d} instances of {:d} unique novel kmers in {:d} reads'.format(instances, len(unique), nreads) in kevlar_log.getvalue()


def test_partition_gml(hk, tmp_path):
    """`partition --gml`: nodes are the reads, edges the pairs that share a retained k-mer (the reference's own call
    raises a NameError before it writes anything, kevlar/partition.py:38-39)"""
    import networkx
    import kevlar_amd
    gml = str(tmp_path / 'graph.gml')
    out, log = run_cli(['partition', '--gml', gml, data_file('connectivity-1311.augfastq')])
    assert '[kevlar] graph written to ' + gml in log
    graph = networkx.read_gml(gml)
    with open(data_file('connectivity-1311.augfastq')) as fh:
        reads = list(kevlar_amd.parse_augmented_fastx(fh))
    assert sorted(graph.nodes) == sorted(r.name for r in reads)
    assert graph.number_of_edges() == 30                # kevlar/tests/test_readgraph.py:20-31
    rg = kevlar_amd.ReadGraph()
    rg.load(reads)
    rg.populate_edges()
    assert graph.number_of_edges() == rg.number_of_edges()
    assert len(list(networkx.connected_components(graph))) == len(rg.connected_components())


def test_hash_positions_equals_hashing_the_kmer_text(hk):
    """kv_hash_positions (k-mers addressed as (read, offset) in a packed batch) against kv_hash_kmers on the same
    k-mers as text, for both hash families, long reads included"""
    rng = np.random.default_rng(5)
    letters = np.array(list('ACGT'))
    seqs = [''.join(letters[rng.integers(0, 4, size=n)]) for n in (100, 31, 9000, 64, 250, 20000)]
    batch = hk.ReadBatch(seqs)
    for cls, k in ((hk.Counttable, 31), (hk.Counttable, 63), (hk.Counttable, 200), (hk.Countgraph, 25)):
        sketch = cls(k, 1e4, 2)
        reads, offs, kmers = [], [], []
        for ridx, seq in enumerate(seqs):
            if len(seq) < k:
                continue
            for off in sorted(set(rng.integers(0, len(seq) - k + 1, size=40).tolist() + [0, len(seq) - k])):
                reads.append(ridx); offs.append(off); kmers.append(seq[off:off + k])
        assert sketch.hash_positions(batch, reads, offs).tolist() == sketch.hash_kmers(kmers).tolist()
    with pytest.raises(Exception, match='does not lie inside its read'):
        hk.Counttable(31, 1e4, 2).hash_positions(batch, [1], [5])


def test_randomised_parity_against_the_oracle(hk):
    """scratch/fuzz_parity.py, a short fixed-seed run: random sketch kinds, k, read shapes, bands, count and scan paths;
    tables byte for byte and hits identical to the oracle"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, os.path.join(root, 'scratch', 'fuzz_parity.py'), '24', '5'], cwd=root, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-2000:]
    assert '24 trials, 0 failures' in done.stdout


def test_hit_arrays_outlive_their_handle(hk):
    """the hit arrays are views into pinned memory that is recycled when the kv_hits handle goes: every view derived
    from them -- np.asarray() included -- must keep the handle alive (bench.py's band replay once read recycled memory)"""
    import gc
    from kevlar_amd import synth
    trio = synth.make_trio(150000, 5)
    reads = {n: synth.unpack_reads(synth.sample_reads_packed(trio[n], 40000, 100, 0.005, 9 + i), 100)
             for i, n in enumerate(('proband', 'mother', 'father'))}
    sk = {n: hk.Counttable(31, 1.5e6, 4) for n in reads}
    for n in reads:
        sk[n].consume_batch(hk.ReadBatch(reads[n]))
    batch = hk.ReadBatch(reads['proband'])
    r, o, a, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batch, 6, 1)
    keep = (np.array(r), np.array(o), np.array(a))                  # real copies
    views = (np.asarray(r), np.asarray(o)[::1], np.asarray(a).reshape(-1, 3))
    del r, o, a
    gc.collect()
    for _ in range(6):                                               # later scans take blocks from the same pool
        hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batch, 1 + _, 2)
    assert len(keep[0]) > 100
    for got, want in zip(views, keep):
        assert np.array_equal(got, want)


def test_novel_tally_counts_a_kmer_seen_on_both_paths_once(hk):
    """`N unique novel kmers` (kevlar/novel.py:161-162) when one run tallies through the record path and the native
    text path: a k-mer (either strand) that both paths saw is one k-mer"""
    import kevlar_amd
    from kevlar_amd.novel import _Tally
    sketch = hk.Counttable(21, 1e5, 4)
    a, b, c = 'GATTACAGATTACAGATTACA', 'CCTGATATCCGGAATCTTAGC', 'ACGTACGTACGTACGTACGTT'
    tally = _Tally(sketch)
    tally.kmers.update([a, kevlar_amd.revcom(a), b])
    tally.hashes.append(sketch.hash_kmers([kevlar_amd.revcom(b), c]))
    tally.instances, tally.reads = 5, 2
    assert 'Found 5 instances of 3 unique novel kmers in 2 reads' in tally.line(0.0)
    only_text = _Tally(sketch)
    only_text.hashes.append(sketch.hash_kmers([a, kevlar_amd.revcom(a), c]))
    assert 'of 2 unique novel kmers' in only_text.line(0.0)
    only_records = _Tally()
    only_records.kmers.update([a, kevlar_amd.revcom(a), c])
    assert 'of 2 unique novel kmers' in only_records.line(0.0)


def test_device_argsort_is_numpys_stable_argsort(hk):
    """kv_argsort_u64 / kv_argsort_rows (the sorts of partition's host half on the device's radix sort) give numpy's stable order:
    integer keys with many ties, names of every width around the eight-byte passes, NUL padding, bytes above 127"""
    import ctypes
    from kevlar_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for n, hi in ((0, 10), (1, 10), (1000, 7), (300000, 1 << 40), (300000, 50), (70000, 1 << 63)):
        keys = rng.integers(0, hi, size=n, dtype=np.uint64) if hi < (1 << 63) else rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        order = np.empty(n, dtype=np.uint32)
        _lib.check(lib.kv_argsort_u64(ctypes.c_void_p(keys.ctypes.data), n, ctypes.c_void_p(order.ctypes.data)))
        assert np.array_equal(order, np.argsort(keys, kind='stable'))
    for width in (1, 3, 7, 8, 9, 16, 23):
        n = 120000
        alphabet = np.frombuffer(b'ab\x00\xfe/0', dtype=np.uint8)          # few symbols: long common prefixes, many equal names
        raw = alphabet[rng.integers(0, len(alphabet), size=(n, width))]
        cut = rng.integers(1, width + 1, size=n)
        raw[np.arange(width)[None, :] >= cut[:, None]] = 0                # names of different lengths, NUL padded
        rows = np.ascontiguousarray(raw).view('S{}'.format(width)).reshape(-1)
        order = np.empty(n, dtype=np.uint32)
        _lib.check(lib.kv_argsort_rows(ctypes.c_void_p(rows.ctypes.data), n, width, ctypes.c_void_p(order.ctypes.data)))
        # numpy compares 'S' values with trailing NULs stripped, which is what NUL padding + byte order gives as well
        assert np.array_equal(order, np.argsort(rows, kind='stable')), width


def test_partition_orders_through_the_device_sort_exactly_as_through_numpy(hk, monkeypatch):
    """assemble_partitions with its three sorts on the device (threshold lowered) against the same call with numpy's sorts"""
    import random
    import kevlar_amd
    from kevlar_amd import partition
    rng = random.Random(21)
    n = 30000
    pool = [''.join(rng.choice('ACGT') for _ in range(rng.choice([40, 40, 55]))) for _ in range(n // 3)]
    names = ['read{}'.format(rng.randrange(n)) if rng.random() < 0.2 else 'r{}/{}'.format(i, rng.randrange(3)) for i in range(n)]
    seqs = [kevlar_amd.revcom(s) if rng.random() < 0.4 else s for s in (rng.choice(pool) for _ in range(n))]
    nb = ''.join(names).encode(); no = np.cumsum([0] + [len(x) for x in names]).astype(np.uint64)
    sb = ''.join(seqs).encode(); so = np.cumsum([0] + [len(x) for x in seqs]).astype(np.uint64)
    label_rng = np.random.default_rng(3)
    labels_cache = {}

    def component_of(node_of_read, n_nodes):
        if n_nodes not in labels_cache:
            labels_cache[n_nodes] = label_rng.integers(0, n // 7, size=n_nodes).astype(np.uint32)
        return labels_cache[n_nodes]
    monkeypatch.setenv('KV_HOST_SORT', '1')
    want = partition.assemble_partitions(nb, no, sb, so, component_of, 2, True)
    monkeypatch.delenv('KV_HOST_SORT')
    monkeypatch.setattr(partition, '_DEVICE_SORT_MIN', 16)
    calls = []
    real = partition._argsort
    monkeypatch.setattr(partition, '_argsort', lambda keys: (calls.append(len(keys)), real(keys))[1])
    got = partition.assemble_partitions(nb, no, sb, so, component_of, 2, True)
    assert len(calls) == 3 and len(want[0]) > 1000
    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
