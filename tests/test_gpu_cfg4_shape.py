"""BASELINE.json config 4 as ONE of its eight GPUs sees it (bench.py --workload cfg4-band), at 1/150 of the size and
against the oracle: a 20 Mb genome at 30x, the reads of every sample written into HBM by kv_reads_generate in 48 batches of
0.625x coverage each (exactly the batch shape of the 3 Gb run: 18.75 M reads on 3 Gb), band 0 of 8 counted into band sketches
of config 4's bins-per-base, the proband scanned batch by batch under the hash-range band rule -- then `kevlar filter` and
`kevlar partition` on the band's annotated reads.

What is compared with what:
* the reads: the packed words the device generator wrote, a spread sample of them against its independent numpy restatement
  (kevlar_amd.synth.device_family_reads; tests/test_gpu_synth.py holds the two against each other more thoroughly);
* count: every byte of the three band sketches and n_occupied against the oracle's banded count of the same reads
  (kvo_consume_reads_mt_banded = kvo_consume's band test, kevlar/count.py:62-66, on the host cores);
* scan: every hit (read, offset, abundances) against the oracle's scan loop over all reads of the proband (kevlar/novel.py:123-169
  restated, band rule of the count);
* filter: the validated reads, their annotations and recounted abundances against a literal restatement of kevlar/filter.py:15-82
  over an oracle Counttable;
* partition: partition numbers and membership against a dict / set restatement of kevlar/readgraph.py:43-161 +
  kevlar/partition.py:15-55.
It also pins what the batches' shape is there to exercise: a batch of 0.6x coverage has nothing to deduplicate, so the
super-k-mer front end must decline once per sketch (not once per batch), and the scan must not cut a batch into super-k-mers
that it then throws away."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G, COVERAGE, L, K = 20_000_000, 30, 100, 31
NBANDS, BAND = 8, 0
PER_BATCH = 125_000                       # 0.625x of 20 Mb: bench.py's 18.75 M reads on 3 Gb
MEM_BAND = 64e9 * (G / 3e9) / NBANDS      # config 4's 64 GB per sample for 3 Gb, split over 8 bands
SEED = 42
CASE_MIN, CTRL_MAX = 6, 1
NAMES = ('proband', 'mother', 'father')


def host_cores():
    from test_gpu_fullsize import host_cores as hc
    return hc()


@pytest.fixture(scope='module')
def family(hk):
    n_reads = G * COVERAGE // L
    firsts = list(range(0, n_reads, PER_BATCH))
    batches = {name: [hk.ReadBatch.generate(G, SEED, si, lo, min(PER_BATCH, n_reads - lo), L) for lo in firsts]
               for si, name in enumerate(NAMES)}
    return n_reads, firsts, batches


@pytest.fixture(scope='module')
def counted(hk, family):
    """the band sketches and the band's hits, the way bench.py's cfg4-band step produces them; launches by profile scope"""
    from test_gpu_fullsize import Profiled
    n_reads, firsts, batches = family
    assert len(firsts) == 48
    T = 4
    sk = {name: hk.Counttable(K, MEM_BAND / T, T) for name in NAMES}
    seen = {}
    with Profiled(hk) as prof:
        kmers = 0
        for name in ('mother', 'father', 'proband'):
            sk[name].clear()
            for b in batches[name]:
                kmers += sk[name].consume_batch(b, NBANDS, BAND)
        for scope in ('k_skm_emit', 'k_skm_count', 'k_consume', 'k_bin_hash_direct', 'k_bin_hash_2bit', 'k_bin_split', 'k_bin_apply'):
            seen[scope] = prof.count(scope)
    with Profiled(hk) as prof:
        rs, os_, as_ = [], [], []
        for first, b in zip(firsts, batches['proband']):
            r, o, a, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], b, CASE_MIN, CTRL_MAX, band_mode=1, nbands=NBANDS, band=BAND)
            rs.append(np.asarray(r, dtype=np.uint32) + np.uint32(first)); os_.append(np.asarray(o, dtype=np.uint32)); as_.append(np.asarray(a, dtype=np.uint8))
        for scope in ('k_skm_novel', 'k_skm_novel_list', 'k_novel_mark', 'k_novel_mark_2bit', 'k_skm_emit'):
            seen['scan:' + scope] = prof.count(scope)
    hits = (np.concatenate(rs), np.concatenate(os_), np.concatenate(as_))
    return sk, kmers, hits, seen


@pytest.fixture(scope='module')
def oracle_side(ok, family):
    """the same reads as text (unpacked from the words the device wrote), the oracle's banded count and scan of them"""
    from test_gpu_fullsize import ascii_block
    n_reads, firsts, batches = family
    wpr = (L + 15) // 16
    cores = host_cores()
    ref, keep = {}, None
    words_of = {}
    for name in NAMES:
        words = np.concatenate([b.packed_words(0, b.n_reads * wpr).reshape(b.n_reads, wpr) for b in batches[name]])
        words_of[name] = words
        bases, offs_p, offs = ascii_block(words, L)
        ref[name] = ok.Counttable(K, MEM_BAND / 4, 4)
        ok.consume_reads_mt_banded(ref[name], bases, offs_p, n_reads, cores, NBANDS, BAND)
        if name == 'proband':
            keep = (bases, offs_p, offs)
    want = ok.novel_scan_mt([ref['proband']], [ref['mother'], ref['father']], keep[0], keep[1], n_reads, K, CASE_MIN, CTRL_MAX, cores,
                            band_mode=1, nbands=NBANDS, band=BAND)
    return ref, want, words_of


def test_generated_reads_are_the_numpy_restatement(family, oracle_side):
    from kevlar_amd import synth
    n_reads, firsts, batches = family
    _ref, _want, words_of = oracle_side
    rng = np.random.default_rng(4)
    for si, name in enumerate(NAMES):
        # reads around every batch boundary and a random spread
        idx = np.unique(np.concatenate((rng.integers(0, n_reads, size=3000), np.asarray(firsts), np.asarray(firsts[1:]) - 1, [n_reads - 1])))
        want = synth.pack_codes(synth.device_family_reads(G, SEED, si, idx, L))
        assert np.array_equal(words_of[name][idx], want), name


def test_band_sketches_equal_the_oracle(hk, counted, oracle_side, family):
    sk, kmers, _hits, seen = counted
    ref, _want, _words = oracle_side
    n_reads = family[0]
    share = kmers / float(3 * n_reads * (L - K + 1))
    assert abs(share * NBANDS - 1.0) < 0.01, 'a band holds 1/N of the hash space'
    for name in NAMES:
        assert sk[name].hashsizes() == ref[name].hashsizes()
        for t in range(4):
            got = np.frombuffer(sk[name].table_bytes(t), dtype=np.uint8)
            exp = np.frombuffer(ref[name].table_bytes(t), dtype=np.uint8)
            assert np.array_equal(got, exp), '{} table {} differs from the oracle'.format(name, t)
        assert sk[name].n_occupied() == ref[name].n_occupied()
    # a 0.6x batch has nothing to deduplicate: the super-k-mer count declines, and the sketch remembers -- it is tried at most once
    # per sketch, not 48 times
    assert seen['k_skm_emit'] <= len(NAMES), seen
    assert seen['k_consume'] + seen['k_bin_hash_direct'] + seen['k_bin_hash_2bit'] >= 3 * 48 - len(NAMES), seen


def test_band_hits_equal_the_oracle(counted, oracle_side):
    _sk, _kmers, hits, seen = counted
    _ref, want, _words = oracle_side
    r, o, a = hits
    wr, wo, wa = want
    assert len(wr) > 20_000
    assert len(r) == len(wr)
    assert np.array_equal(r, wr) and np.array_equal(o, wo.astype(np.uint32)) and np.array_equal(a, wa)
    # the scan of a batch that cannot be deduplicated goes straight to the tile scan: cutting it into super-k-mers first, running
    # into the tables' capacity and scanning again (round 3: 48 launches of each per step) may happen once, not per batch
    assert seen['scan:k_novel_mark'] + seen['scan:k_novel_mark_2bit'] >= 47, seen
    assert seen['scan:k_skm_novel'] <= 1 and seen['scan:k_skm_emit'] <= 1, seen


@pytest.fixture(scope='module')
def downstream(hk, counted, tmp_path_factory):
    """the band's annotated reads as a file, `kevlar filter` and `kevlar partition` through the CLI"""
    import kevlar_amd
    import bench
    from kevlar_amd import synth
    _sk, _kmers, hits, _seen = counted
    tmp = tmp_path_factory.mktemp('band')
    ann = bench.band_annotated_reads(hits, G, SEED, L, K, 3, synth)
    novel_file, filtered_file, part_file = (str(tmp / f) for f in ('band.novel.augfastq', 'band.filtered.augfastq', 'band.part.augfastq'))
    with open(novel_file, 'wb') as fh:
        fh.write(ann.format(np.arange(ann.n, dtype=np.uint64)))
    log = io.StringIO()
    old, kevlar_amd.logstream = kevlar_amd.logstream, log
    try:
        for argv in (['filter', '--memory', '20M', '--case-min', str(CASE_MIN), '--ctrl-max', str(CTRL_MAX), '-o', filtered_file, novel_file],
                     ['partition', '-o', part_file, filtered_file]):
            args = kevlar_amd.cli.parser().parse_args(argv)
            kevlar_amd.cli.mains[args.cmd](args)
    finally:
        kevlar_amd.logstream = old
    return novel_file, filtered_file, part_file, log.getvalue()


def load(path):
    import kevlar_amd
    with open(path) as fh:
        return list(kevlar_amd.parse_augmented_fastx(fh))


def test_filter_of_the_band_equals_the_reference_loop(ok, downstream):
    """kevlar/filter.py:15-82 literally, over an oracle Counttable: first pass adds every annotated k-mer occurrence, second pass keeps
    the annotations whose recount reaches case-min and whose control abundances stay within ctrl-max, with the recount as case
    abundance; reads left without annotations go; order kept"""
    novel_file, filtered_file, _part, log = downstream
    reads = load(novel_file)
    assert len(reads) > 5000
    counts = ok.Counttable(K, 20e6 / 4, 4)
    for rec in reads:
        for ik in rec.annotations:
            counts.add(rec.ikmerseq(ik))
    want = []
    for rec in reads:
        kept = []
        for ik in rec.annotations:
            again = counts.get(rec.ikmerseq(ik))
            if again < CASE_MIN or any(x > CTRL_MAX for x in ik.abund[1:]):
                continue
            kept.append((ik.offset, (again,) + tuple(ik.abund[1:])))
        if kept:
            want.append((rec.name, rec.sequence, kept))
    got = [(rec.name, rec.sequence, [(ik.offset, tuple(ik.abund)) for ik in rec.annotations]) for rec in load(filtered_file)]
    assert got == want
    assert 'Processed {:d} reads'.format(len(reads)) in log and 'Validated {:d} reads'.format(len(want)) in log
    assert 0 < len(want) <= len(reads)


def test_partition_of_the_band_equals_the_reference_loop(downstream):
    """kevlar/readgraph.py:43-161 + kevlar/partition.py:15-55 as dicts and sets: reads that share an interesting k-mer (up to reverse
    complement) are connected; components largest first (ties: by their sorted names, descending), singletons dropped, one read per
    canonical sequence, numbered from 1"""
    import kevlar_amd
    _novel, filtered_file, part_file, log = downstream
    reads = load(filtered_file)
    holder, by_kmer = {}, {}
    for i, rec in enumerate(reads):
        holder[rec.name] = i
        for ik in rec.annotations:
            by_kmer.setdefault(kevlar_amd.revcommin(rec.ikmerseq(ik)), set()).add(rec.name)
    parent = {name: name for name in holder}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for names in by_kmer.values():
        names = sorted(names)
        for other in names[1:]:
            ra, rb = find(names[0]), find(other)
            if ra != rb:
                parent[rb] = ra
    groups = {}
    for name in holder:
        groups.setdefault(find(name), []).append(name)
    keyed = sorted(((len(m), sorted(m)) for m in groups.values() if len(m) >= 2), reverse=True)
    want, num = [], 0
    for _size, members in keyed:
        seen, kept = set(), []
        for name in members:
            canon = kevlar_amd.revcommin(reads[holder[name]].sequence)
            if canon not in seen:
                seen.add(canon)
                kept.append(name)
        num += 1
        want += [(name, num) for name in kept]
    got = [(rec.name.rsplit(' kvcc=', 1)[0], kevlar_amd.seqio.partition_id(rec.name)) for rec in load(part_file)]
    got = [(name, int(pid)) for name, pid in got]
    assert len(keyed) > 100
    # membership and numbering; inside a partition the product writes the members in name order as the loop above does
    assert got == want
    # the closing line counts the reads WRITTEN, i.e. after dedup (kevlar/partition.py:67-80: numreads += len(part) over the
    # partitions as yielded; its own known answer for dup.augfastq is 16 reads with dedup, 18 without)
    assert 'grouped {:d} reads into {:d} connected components'.format(len(want), num) in log
    assert len(want) < sum(size for size, _members in keyed), 'the input should hold duplicate sequences, or dedup is not exercised'
