"""Randomised parity of the round-3 paths: equal-length reads handed over as packed words (the wave-per-group S1 kernel), the
case sample counted last with the scan hint, the scan of the same batch from the count pass's distinct list -- against the oracle.
Random kind, k, read length, coverage, error rate, bucket size, bands, forced table misses, thresholds.
python scratch/fuzz_list.py [trials] [seed]   (the oracle is test infrastructure: this is a test tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kevlar_amd import _lib, khmer as hk, synth
from oracle import okhmer as ok
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ['Counttable', 'SmallCounttable', 'Nodetable']
fails = 0
os.environ['KV_COUNT_PATH'] = 'skm'; os.environ['KV_NOVEL_PATH'] = 'skm'; os.environ['KV_SKM_DL'] = '1'
for trial in range(trials):
    kind = KINDS[rng.integers(0, 3)] if rng.random() < 0.4 else 'Counttable'
    k = int(rng.choice([int(rng.integers(16, 65)), 16, 25, 31, 32, 33, 51, 64]))
    L = int(rng.choice([k + int(rng.integers(0, 60)), 100, 150, 151, 250]))
    glen = int(rng.choice([20000, 100000, 400000]))
    n = int(min(120000, max(500, glen * float(rng.choice([3, 12, 30])) / L)))
    trio = synth.make_trio(glen, int(rng.integers(0, 1 << 30)), inherited_per_mb=400, denovo_per_mb=400)
    names = ('proband', 'mother', 'father')
    words = {s: synth.sample_reads_packed(trio[s], n, L, float(rng.choice([0.0, 0.005, 0.03])), int(rng.integers(0, 1 << 30))) for s in names}
    reads = {s: synth.unpack_reads(words[s], L) for s in names}
    mem = float(rng.choice([1e6, 8e6]))
    nbands = int(rng.choice([0, 0, 0, 3])); band = int(rng.integers(0, nbands)) if nbands else 0
    env = {'KV_SKM_BUCKET_KMERS': str(int(rng.choice([1024, 2048, 4096, 8192])))}
    if rng.random() < 0.3: env['KV_SKM_FORCE_LOOSE'] = '1'
    if rng.random() < 0.2: env['KV_SKM_CAP_PCT'] = '40'
    case_min, ctrl_max = (1, 0) if kind == 'Nodetable' else (int(rng.integers(2, 8)), int(rng.integers(0, 3)))
    desc = 'trial {} {} k={} n={} L={} genome={} mem={:g} bands={}/{} {} case_min={} ctrl_max={}'.format(trial, kind, k, n, L, glen, mem, band, nbands, env, case_min, ctrl_max)
    os.environ.update(env)
    try:
        dev = {s: getattr(hk, kind)(k, mem / 4, 4) for s in names}
        ref = {s: getattr(ok, kind)(k, mem / 4, 4) for s in names}
        # one or two case samples, two / one / no controls (kevlar/novel.py:36-51); the scanned batch -- the proband's -- is counted last
        cases, ctrls = [(['proband'], ['mother', 'father']), (['proband'], ['mother', 'father']), (['proband', 'mother'], ['father']),
                        (['mother', 'proband'], ['father']), (['proband', 'father'], []), (['proband'], [])][rng.integers(0, 6)]
        desc += ' cases={} ctrls={}'.format(','.join(cases), ','.join(ctrls))
        for s in cases:
            dev[s].expect_scan()
        batches = {s: hk.ReadBatch.from_packed(words[s], L) for s in names}
        for s in ('mother', 'father', 'proband'):
            nk = dev[s].consume_batch(batches[s], nbands, band)
            bases, offs = ok.concat_reads(reads[s])
            nk_ref = ok.consume_reads(ref[s], bases, offs, n, nbands, band)
            assert nk == nk_ref, ('k-mers counted', s, nk, nk_ref)
            for t in range(4):
                assert dev[s].table_bytes(t) == ref[s].table_bytes(t), (s, 'table', t)
        r, o, a, _ = hk.novel_scan([dev[s] for s in cases], [dev[s] for s in ctrls], batches['proband'], case_min, ctrl_max,
                                   band_mode=1 if nbands else 0, nbands=nbands, band=band)
        bases, offs = ok.concat_reads(reads['proband'])
        hits, _ = ok.novel_scan([ref[s] for s in cases], [ref[s] for s in ctrls], bases, offs, n, k, case_min, ctrl_max,
                                band_mode=1 if nbands else 0, nbands=nbands, band=band, cap=max(1 << 20, 4 * n * max(1, L - k + 1)))
        got = list(zip(r.tolist(), o.tolist(), map(tuple, a.tolist())))
        assert got == [(h[0], h[1], tuple(h[2])) for h in hits], ('hits', len(got), len(hits))
        print('ok  ', desc, len(got), 'hits', flush=True)
    except Exception as exc:
        fails += 1
        print('FAIL', desc, repr(exc)[:300], flush=True)
    finally:
        for key in env: os.environ.pop(key, None)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
