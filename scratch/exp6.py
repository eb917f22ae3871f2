import sys, time, ctypes
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
packed = synth.trio_reads_packed(25_000_000, 30, L)
names = ('proband', 'mother', 'father')
batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
def prof_all():
    buf = ctypes.create_string_buffer(4096); lib.kv_prof_names(buf, 4096); out = {}
    for name in buf.value.decode().split(','):
        if not name: continue
        ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); out[name] = round(ms.value, 2)
    return out
for N in (1, 2, 4, 8):
    sk = {n: hk.Counttable(k, 2e9 / N / 4, 4) for n in names}
    nb, band = (N, 0) if N > 1 else (0, 0)
    def step():
        for n in names:
            sk[n].clear(); sk[n].consume_batch(batches[n], nb, band)
        return hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1, band_mode=1 if N > 1 else 0, nbands=nb, band=band)
    step(); lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(3): r = step()
    lib.kv_synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    lib.kv_prof_enable(0)
    print('N=%d per-rank step %.1f ms hits %d kernels(3 steps) %s' % (N, dt, len(r[0]), prof_all()))
    del sk
