import sys, time, ctypes, os
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
packed = synth.trio_reads_packed(25_000_000, 30, L)
b = hk.ReadBatch.from_packed(packed['proband'], L)
def prof_all():
    buf = ctypes.create_string_buffer(4096); lib.kv_prof_names(buf, 4096); out = {}
    for name in buf.value.decode().split(','):
        if not name: continue
        ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); out[name] = round(ms.value / max(1, c.value), 2)
    return out
ref = None
for mode, dbg in (('x', ''), ('x', '')):
    sk = hk.Counttable(k, 5e8, 4)
    sk.consume_batch(b); sk.clear()
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(4):
        sk.clear(); sk.consume_batch(b)
    dt = (time.perf_counter() - t0) / 4 * 1e3
    lib.kv_prof_enable(0)
    occ = sk.n_occupied()
    if ref is None: ref = [sk.table_bytes(t) for t in range(4)]
    same = all(sk.table_bytes(t) == ref[t] for t in range(4))
    print('dbg', dbg, 'direct=%s  %.1f ms/sample  occ %d same=%s  %s' % (mode, dt, occ, same, prof_all()), flush=True)
