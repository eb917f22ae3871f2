"""Randomised device-vs-host ingest parity: random record shapes (lengths 0..2000, names, CRLF, missing final newline,
non-ACGT characters), uncompressed, BGZF or ordinary gzip (one or several members, random chunk size of the block search) at random levels, random batch sizes and text budgets; batches, record
text and count tables must agree.  python scratch/fuzz_ingest.py [trials] [seed]"""
import os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
from kevlar_amd import _lib, bgzf, khmer as hk
from test_gpu_ingest import batches_of
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
alphabet = np.frombuffer(b'ACGT', dtype=np.uint8)
fails = 0
with tempfile.TemporaryDirectory() as tmp:
    for trial in range(trials):
        n = int(rng.choice([1, 7, 500, 20000]))
        style = rng.integers(0, 4)
        recs = []
        for i in range(n):
            L = int(rng.choice([0, 1, 24, 25, 26, 100, 151, int(rng.integers(0, 2000))])) if style else 100
            seq = alphabet[rng.integers(0, 4, L)].tobytes().decode()
            if L and rng.random() < 0.02:
                p = int(rng.integers(0, L)); seq = seq[:p] + rng.choice(list('NnacgtRY-')) + seq[p + 1:]
            qual = bytes(rng.integers(33, 75, L, dtype=np.uint8)).decode('latin-1')
            name = 'r{} {}'.format(i, 'x' * int(rng.integers(0, 30))) if rng.random() < 0.9 else ''
            eol = '\r\n' if style == 3 else '\n'
            recs.append('@' + name + eol + seq + eol + '+' + (name if rng.random() < 0.1 else '') + eol + qual + eol)
        text = ''.join(recs)
        if rng.random() < 0.3:
            text = text.rstrip('\r\n')
        kind = str(rng.choice(['plain', 'bgzf', 'gzip', 'gzip']))
        path = os.path.join(tmp, 't{}.fq'.format(trial) + ('' if kind == 'plain' else '.gz'))
        if kind == 'plain':
            with open(path, 'w', newline='') as fh: fh.write(text)
        elif kind == 'gzip':
            import gzip
            raw = text.encode('latin-1')
            cuts = sorted(set([0, len(raw)] + [int(c) for c in rng.integers(0, len(raw) + 1, int(rng.integers(0, 3)))]))
            with open(path, 'wb') as fh:
                for a, b in zip(cuts, cuts[1:]):
                    fh.write(gzip.compress(raw[a:b], int(rng.choice([1, 4, 6, 9]))))
        else:
            with bgzf.BgzfWriter(path, level=int(rng.choice([0, 1, 6, 9]))) as fh: fh.write(text)
        batch = int(rng.choice([1, 3, 1000, 100000]))
        env = {'KV_INGEST_TEXT_MB': '1'} if rng.random() < 0.5 else {}
        if kind == 'gzip' and rng.random() < 0.5:
            env['KV_GUNZIP_CHUNK_KB'] = str(rng.choice([1, 2, 4, 64]))
        desc = 'trial {} {} n={} style={} batch={} {}'.format(trial, kind, n, style, batch, env)
        try:
            host = batches_of(hk, path, batch, {'KV_INGEST': 'host'})
            dev = batches_of(hk, path, batch, env)
            assert host[4] == dev[4] == n and sum(dev[0]) == n and max(dev[0]) <= batch, (desc, host[4], dev[4])
            assert host[2] == dev[2], (desc, 'count tables')
            if host[0] == dev[0]:
                assert host[1] == dev[1], (desc, 'records')
            print('ok  ', desc, set(dev[3]), flush=True)
        except Exception as exc:
            fails += 1
            print('FAIL', desc, repr(exc)[:300], flush=True)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
