import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ['KV_SKM_VERBOSE'] = '1'
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load()
packed = synth.trio_reads_packed(25_000_000, 30, 100)
b = hk.ReadBatch.from_packed(packed['mother'], 100)
sk = hk.Counttable(31, 5e8, 4)
for rep in range(3):
    sk.clear(); sk.consume_batch(b)
