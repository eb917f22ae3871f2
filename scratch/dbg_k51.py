import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['KV_SKM_VERBOSE'] = '1'
import __graft_entry__; __graft_entry__.build()
from kevlar_amd import _lib, khmer as hk, synth
from oracle import okhmer as ok
packed = synth.trio_reads_packed(25_000_000, 30, 100)
b = hk.ReadBatch.from_packed(packed['mother'], 100)
sk = hk.Counttable(51, 5e8, 4)
print('kmers', sk.consume_batch(b))
print(open('/sys/fs/cgroup/cpu.max').read() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'no cpu.max', os.cpu_count(), len(os.sched_getaffinity(0)))
n = 100000
seqs = synth.unpack_reads(packed['mother'][:n], 100)
bases, offs = ok.concat_reads(seqs)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    s = ok.Counttable(31, 5e8, 4)
    t = time.time(); ok.consume_reads_mt(s, bases, offs, n, th); dt = time.time() - t
    print(th, 'threads', round(n / dt), 'reads/s')
