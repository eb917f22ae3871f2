"""k_skm_emit alone on one config-2 sample: python scratch/s1_time.py [reps]  (KV_LIB_PATH / KV_SKM_* pick the variant); prints the
best and the median HIP-event time of the S1 launch over `reps` counts"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k = int(os.environ.get('S1_K', '31'))
words = synth.sample_reads_packed(synth.make_trio(25_000_000, 42)['mother'], 7_500_000, 100, 0.005, 1002)
batch = hk.ReadBatch.from_packed(words, 100)
sk = hk.Counttable(k, 5e8, 4)
ms, n = ctypes.c_double(), ctypes.c_uint64()
times = {name: [] for name in ('k_skm_emit', 'k_skm_split', 'k_skm_count')}
for rep in range(reps + 1):
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    try:
        sk.clear(); sk.consume_batch(batch)
    except Exception as exc:          # (dissection builds leave broken batches behind)
        pass
    lib.kv_prof_enable(0)
    for name in times:
        lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
        if rep: times[name].append(ms.value)
print(os.environ.get('KV_LIB_PATH', 'default'), {k_: os.environ[k_] for k_ in os.environ if k_.startswith('KV_SKM')},
      {name: (round(min(v), 3), round(float(np.median(v)), 3)) for name, v in times.items()}, flush=True)
