import sys, time, ctypes, os
sys.path.insert(0, '.')
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
trio = synth.make_trio(25_000_000, 42)
n = 7_500_000
words = synth.sample_reads_packed(trio['proband'], n, L, 0.005, 1001)
batch = hk.ReadBatch.from_packed(words, L)
sk = hk.Counttable(k, 2e9 / 4, 4)
def prof(name):
    ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); return ms.value / max(1, c.value)
for dbg in ('0', '1', '2', '4', '8', '12'):
    os.environ['KV_BIN_DEBUG'] = dbg
    sk.clear(); sk.consume_batch(batch)
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    for _ in range(2):
        sk.clear(); sk.consume_batch(batch)
    print('debug', dbg, 'k_bin_hash %.2f ms  split %.2f  apply %.2f' % (prof('k_bin_hash'), prof('k_bin_split'), prof('k_bin_apply')))
    lib.kv_prof_enable(0)
