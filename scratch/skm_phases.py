"""Timing dissection of the super-k-mer kernels: KV_SKM_DEBUG skips parts of them (results are wrong then;
only the HIP-event times are read).  gpurun -- python scratch/skm_phases.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__
__graft_entry__.build()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load()
packed = synth.trio_reads_packed(25_000_000, 30, 100)
names = ('mother', 'father', 'proband')
batches = {n: hk.ReadBatch.from_packed(packed[n], 100) for n in names}
sk = {n: hk.Counttable(31, 2e9 / 4, 4) for n in names}


def prof(name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(1, n.value)


for dbg in [int(x) for x in (sys.argv[1:] or ['0', '1', '2', '4', '8', '16', '32'])]:
    os.environ['KV_SKM_DEBUG'] = str(dbg)
    for rep in range(3):
        if rep == 1:
            lib.kv_prof_reset(); lib.kv_prof_enable(1)
        for n in names:
            sk[n].clear(); sk[n].consume_batch(batches[n])
        try:
            hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1)
        except Exception as exc:
            print('scan failed', exc)
    lib.kv_prof_enable(0)
    print(dbg, {k: round(prof(k), 3) for k in ('k_skm_emit', 'k_skm_split', 'k_skm_count', 'k_bin_split_w', 'k_bin_apply_w', 'k_skm_novel')}, flush=True)
