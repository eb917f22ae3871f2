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
def seq():
    for n in names:
        sk[n].clear(); sk[n].consume_batch(batches[n])
def conc():
    def job(n):
        def f():
            sk[n].clear(); return sk[n].consume_batch(batches[n])
        return f
    hk.run_concurrently([job(n) for n in names])
for name, fn in (('sequential', seq), ('concurrent', conc), ('sequential', seq), ('concurrent', conc)):
    fn()
    lib.kv_synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    lib.kv_synchronize()
    print(name, '%.1f ms per 3-sample count' % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
ref = [sk['proband'].table_bytes(0)[:1000]]
