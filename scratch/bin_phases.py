"""Timing dissection of stage C (k_bin_apply): KV_BIN_DEBUG skips parts of it (results are wrong then; only HIP-event
times are read).  gpurun -- python scratch/bin_phases.py 0 1 2 4 6"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load()
packed = synth.trio_reads_packed(25_000_000, 30, 100)
batch = hk.ReadBatch.from_packed(packed['proband'], 100)
sk = hk.Counttable(31, 2e9 / 4, 4)


def prof(name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(1, n.value)


for dbg in [int(x) for x in (sys.argv[1:] or ['0', '1', '2', '4', '6'])]:
    os.environ['KV_BIN_DEBUG'] = str(dbg)
    for rep in range(4):
        if rep == 1:
            lib.kv_prof_reset(); lib.kv_prof_enable(1)
        sk.clear(); sk.consume_batch(batch)
    lib.kv_prof_enable(0)
    print(dbg, {k: round(prof(k), 3) for k in ('k_skm_count', 'k_bin_split_w', 'k_bin_apply_w')}, flush=True)
