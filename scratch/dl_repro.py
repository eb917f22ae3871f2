"""the list scan against the walking scan on a batch of N reads per sample, several times over"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kevlar_amd import _lib, khmer as hk, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
packed = synth.trio_reads_packed(25_000_000, 30, 100)
names = ('mother', 'father', 'proband')
batches = {s: hk.ReadBatch.from_packed(packed[s][:n], 100) for s in names}
ref = None
for rep in range(6):
    hint = rep != 0
    sk = {s: hk.Counttable(31, 2e9 / 4, 4) for s in names}        # fresh: the buckets are sized without knowing the distinct share, some k-mers miss the LDS tables
    sk['proband'].expect_scan(hint)
    for s in names:
        sk[s].clear(); sk[s].consume_batch(batches[s])
    r, o, a, _ = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1)
    got = (np.array(r), np.array(o), np.array(a))
    if ref is None:
        ref = got
        print('walk:', len(r), 'hits')
        continue
    same = len(got[0]) == len(ref[0]) and all(np.array_equal(x, y) for x, y in zip(got, ref))
    print('list rep', rep, len(r), 'hits', 'same' if same else 'DIFFERENT')
    if not same:
        A = set(zip(ref[0].tolist(), ref[1].tolist())); B = set(zip(got[0].tolist(), got[1].tolist()))
        print('  only walk:', sorted(A - B)[:8], len(A - B), ' only list:', sorted(B - A)[:8], len(B - A))
