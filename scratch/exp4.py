import sys, time, ctypes, os
sys.path.insert(0, '.')
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
packed = synth.trio_reads_packed(25_000_000, 30, L)
names = ('proband', 'mother', 'father')
batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
def prof(name):
    ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); return ms.value / max(1, c.value)
for label, cls, size, cmin, cmax in (('byte tables 4x500MB', hk.Counttable, 2e9 / 4, 6, 1), ('bit tables 4x62.5MB', hk.Nodetable, 2e9 / 4, 1, 0)):
    sk = {n: cls(k, size, 4) for n in names}
    for n in names:
        sk[n].consume_batch(batches[n])
    hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], cmin, cmax)
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    for _ in range(2):
        r = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], cmin, cmax)
    print(label, 'k_novel_mark %.2f ms, hits %d' % (prof('k_novel_mark'), len(r[0])))
    lib.kv_prof_enable(0)
    del sk
