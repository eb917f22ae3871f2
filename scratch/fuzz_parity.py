"""Randomised device-vs-oracle parity: random sketch kind, k, table size, read lengths, error rate, Ns, banding, mask,
count path (super-k-mer / plain partition / atomic) and scan path; tables byte for byte, hits (read, offset, abundances)
identical.  python scratch/fuzz_parity.py [trials] [seed]   (the oracle is test infrastructure: this is a test tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kevlar_amd import _lib, khmer as hk, synth
from oracle import okhmer as ok
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ['Counttable', 'SmallCounttable', 'Nodetable', 'Countgraph', 'SmallCountgraph', 'Nodegraph']
fails = 0
for trial in range(trials):
    kind = KINDS[rng.integers(0, len(KINDS))] if rng.random() < 0.5 else 'Counttable'
    graph = kind.endswith('graph')
    k = int(rng.integers(9, 33)) if graph else int(rng.choice([int(rng.integers(9, 100)), 16, 17, 31, 32, 33, 51, 63, 64, 65]))
    n = int(rng.choice([300, 5000, 60000]))
    L = int(rng.choice([k + int(rng.integers(0, 40)), 100, 151, 250]))
    trio = synth.make_trio(int(rng.choice([3000, 40000, 200000])), int(rng.integers(0, 1 << 30)))
    reads = {}
    for i, name in enumerate(('proband', 'mother', 'father')):
        seqs = synth.unpack_reads(synth.sample_reads_packed(trio[name], n, L, float(rng.choice([0.0, 0.005, 0.03])), int(rng.integers(0, 1 << 30))), L)
        for _ in range(int(rng.integers(0, 6))):                     # ragged lengths, Ns, lower case, short reads
            j = int(rng.integers(0, n)); what = rng.integers(0, 4)
            if what == 0: seqs[j] = seqs[j][:int(rng.integers(0, L))]
            elif what == 1 and L > 3: p = int(rng.integers(0, L - 1)); seqs[j] = seqs[j][:p] + 'N' + seqs[j][p + 1:]
            elif what == 2: seqs[j] = seqs[j].lower()
            else: seqs[j] = seqs[j] + 'ACGT' * int(rng.integers(1, 80))
        reads[name] = seqs
    mem = float(rng.choice([2e4, 1e6, 8e6]))
    nbands = int(rng.choice([0, 0, 2, 5])); band = int(rng.integers(0, nbands)) if nbands else 0
    cpath = [None, 'skm', 'binned', 'atomic'][rng.integers(0, 4)]
    npath = [None, 'skm', 'tiles'][rng.integers(0, 3)]
    env = {}
    if cpath: env['KV_COUNT_PATH'] = cpath
    if npath: env['KV_NOVEL_PATH'] = npath
    desc = 'trial {} {} k={} n={} L={} mem={:g} bands={}/{} count={} scan={}'.format(trial, kind, k, n, L, mem, band, nbands, cpath, npath)
    os.environ.update(env)
    try:
        dev, ref = {}, {}
        for name, seqs in reads.items():
            dev[name] = getattr(hk, kind)(k, mem / 4, 4)
            ref[name] = getattr(ok, kind)(k, mem / 4, 4)
            nk = dev[name].consume_batch(hk.ReadBatch(seqs), nbands, band)
            bases, offs = ok.concat_reads(seqs)
            nk_ref = ok.consume_reads(ref[name], bases, offs, len(seqs), nbands, band)
            assert nk == nk_ref, (desc, 'k-mers counted', nk, nk_ref)
            for t in range(4):
                assert dev[name].table_bytes(t) == ref[name].table_bytes(t), (desc, name, 'table', t)
            assert dev[name].n_occupied() == ref[name].n_occupied(), (desc, 'occupied')
        case_min, ctrl_max = int(rng.integers(1, 8)), int(rng.integers(0, 3))
        # who is a case and who a control (kevlar/novel.py:36-51 loops over any number of either; the reference's own two-case run is
        # kevlar/tests/test_novel.py:108-144): one or two cases, two / one / no controls
        split = [(['proband'], ['mother', 'father']), (['proband'], ['mother', 'father']), (['proband', 'mother'], ['father']),
                 (['proband', 'father'], []), (['proband'], []), (['proband'], ['father'])][rng.integers(0, 6)]
        desc += ' cases={} ctrls={}'.format(len(split[0]), len(split[1]))
        batch = hk.ReadBatch(reads['proband'])
        r, o, a, _ = hk.novel_scan([dev[s] for s in split[0]], [dev[s] for s in split[1]], batch, case_min, ctrl_max,
                                   band_mode=1 if nbands else 0, nbands=nbands, band=band)
        bases, offs = ok.concat_reads(reads['proband'])
        hits, _ = ok.novel_scan([ref[s] for s in split[0]], [ref[s] for s in split[1]], bases, offs, len(reads['proband']), k, case_min, ctrl_max,
                                band_mode=1 if nbands else 0, nbands=nbands, band=band, cap=max(1 << 20, 4 * n * max(1, L - k + 1)))
        got = list(zip(r.tolist(), o.tolist(), map(tuple, a.tolist())))
        assert got == [(h[0], h[1], tuple(h[2])) for h in hits], (desc, 'hits', len(got), len(hits))
        print('ok  ', desc, len(got), 'hits', flush=True)
    except Exception as exc:
        fails += 1
        print('FAIL', desc, repr(exc)[:300], flush=True)
    finally:
        for key in env: os.environ.pop(key, None)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
