import sys, ctypes
sys.path.insert(0, '.')
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
lib.kv_prof_enable(1)
trio = synth.make_trio(400000, 5)
words = synth.sample_reads_packed(trio['proband'], 64000, 100, 0.005, 6)
batch = hk.ReadBatch.from_packed(words, 100)
dev = hk.Counttable(31, 3.0e6, 4)
n = dev.consume_batch(batch)
print('n', n, 'err:', _lib.last_error())
buf = ctypes.create_string_buffer(1024); lib.kv_prof_names(buf, 1024); print(buf.value)
for name in buf.value.decode().split(','):
    ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); print(name, ms.value, c.value)
