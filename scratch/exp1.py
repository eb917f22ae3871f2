import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
trio = synth.make_trio(25_000_000, 42)
n = 3_000_000
words = synth.sample_reads_packed(trio['proband'], n, L, 0.005, 1001)
batch = hk.ReadBatch.from_packed(words, L)
def t(f, reps=3):
    f(); lib.kv_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    lib.kv_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
nk = n * 70
for mem in (8e7, 2e9):
    sk = hk.Counttable(k, mem / 4, 4)
    ms = t(lambda: sk.consume_batch(batch))
    print('consume mem=%g: %.2f ms  %.2f Gkmer/s' % (mem, ms, nk / ms / 1e6))
    ms = t(lambda: sk.consume_batch(batch, 1 << 20, 0))
    print('hash-only (band rejects all) mem=%g: %.2f ms  %.2f Gkmer/s' % (mem, ms, nk / ms / 1e6))
    nt = hk.Nodetable(k, mem / 4 * 8, 4)
    ms = t(lambda: nt.consume_batch(batch))
    print('nodetable consume (atomicOr) mem=%g: %.2f ms  %.2f Gkmer/s' % (mem, ms, nk / ms / 1e6))
    ms = t(lambda: hk.novel_scan([sk], [sk, sk], batch, 6, 1, band_mode=1, nbands=1 << 20, band=0))
    print('novel hash-only mem=%g: %.2f ms' % (mem, ms))
    t0 = time.perf_counter(); r = hk.novel_scan([sk], [sk, sk], batch, 6, 1); dt = time.perf_counter() - t0
    print('novel self-vs-self: %.2f ms total, hits %d' % (dt * 1e3, len(r[0])))
