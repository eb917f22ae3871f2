import sys, time, ctypes, os
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
packed = synth.trio_reads_packed(25_000_000, 30, L)
names = ('proband', 'mother', 'father')
batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
sk = {n: hk.Counttable(k, 5e8, 4) for n in names}
def prof_all():
    buf = ctypes.create_string_buffer(4096); lib.kv_prof_names(buf, 4096); out = {}
    for name in buf.value.decode().split(','):
        if not name: continue
        ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); out[name] = (round(ms.value / max(1, c.value), 2), c.value)
    return out
def seq(ns=names):
    for n in ns:
        sk[n].clear(); sk[n].consume_batch(batches[n])
def conc(ns=names):
    def job(n):
        def f():
            t0 = time.perf_counter(); sk[n].clear(); t1 = time.perf_counter(); r = sk[n].consume_batch(batches[n]); return (round((t1 - t0) * 1e3, 1), round((time.perf_counter() - t1) * 1e3, 1))
        return f
    return hk.run_concurrently([job(n) for n in ns])
for name, fn in (('sequential', seq), ('concurrent3', conc), ('concurrent2', lambda: conc(names[:2]))):
    fn(); fn()
    lib.kv_synchronize()
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(2): r = fn()
    lib.kv_synchronize()
    dt = (time.perf_counter() - t0) / 2 * 1e3
    lib.kv_prof_enable(0)
    print(name, '%.1f ms' % dt, r, prof_all(), flush=True)
