"""Timing dissection of the list scan (k_skm_novel_list): KV_SKM_DEBUG 4 skips the evaluation, 8 the marking pass (results are wrong then)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load()
packed = synth.trio_reads_packed(25_000_000, 30, 100)
names = ('mother', 'father', 'proband')
batches = {n: hk.ReadBatch.from_packed(packed[n], 100) for n in names}
sk = {n: hk.Counttable(31, 2e9 / 4, 4) for n in names}
sk['proband'].expect_scan()


def prof(name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value / max(1, n.value)


for dbg in [int(x) for x in (sys.argv[1:] or ['0', '4', '8', '12'])]:
    for rep in range(3):
        if rep == 1:
            lib.kv_prof_reset(); lib.kv_prof_enable(1)
        os.environ['KV_SKM_DEBUG'] = '0'
        for n in names:
            sk[n].clear(); sk[n].consume_batch(batches[n])
        os.environ['KV_SKM_DEBUG'] = str(dbg)
        # the debug word travels in the bucket geometry, which the count built: set it for the scan through the environment of the
        # NEXT build only -- so rebuild-free scans read KV_SKM_SCAN_DEBUG instead
        os.environ['KV_SKM_SCAN_DEBUG'] = str(dbg)
        try:
            r = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1)
        except Exception as exc:
            print('scan failed', exc)
    lib.kv_prof_enable(0)
    print(dbg, {k: round(prof(k), 3) for k in ('k_skm_count', 'k_skm_novel_list', 'k_skm_novel', 'k_skm_loose_novel', 'k_tile_hits', 'k_novel_emit')}, len(r[0]), flush=True)
