"""native augmented-FASTQ load, one pass against pieces parsed side by side: python scratch/augfastx_load_rate.py"""
import numpy as np, time, sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rng=np.random.default_rng(1)
recs=[]
n=120000
for i in range(n):
    seq=''.join('ACGT'[c] for c in rng.integers(0,4,100))
    qual='F'*100
    lines=['@read%d kvcc=1\n%s\n+\n%s\n'%(i,seq,qual)]
    st=int(rng.integers(0,40))
    for o in range(st, st+19):
        lines.append(' '*o+seq[o:o+31]+' '*10+'12 0 0#\n')
    recs.append(''.join(lines))
open('/tmp/big.augfastq','w').write(''.join(recs))
import os; print(os.path.getsize('/tmp/big.augfastq')>>20,'MB')
import os
from kevlar_amd import _lib
lib = _lib.load()
for threads in ('1', '4', '16'):
    os.environ['KV_AUGFASTX_THREADS'] = threads
    best = 9
    for rep in range(3):
        h = ctypes.c_void_p(); t = time.perf_counter()
        _lib.check(lib.kv_augfastx_load(b'/tmp/big.augfastq', ctypes.byref(h)))
        best = min(best, time.perf_counter() - t); lib.kv_augfastx_free(h)
    print('KV_AUGFASTX_THREADS={}: load {:.3f} s'.format(threads, best))
