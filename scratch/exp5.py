import sys, time, os
sys.path.insert(0, '.')
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
trio = synth.make_trio(25_000_000, 42)
n = 3_000_000
words = synth.sample_reads_packed(trio['proband'], n, L, 0.005, 1001)
batch = hk.ReadBatch.from_packed(words, L)
os.environ['KV_COUNT_PATH'] = 'atomic'
def t(f, reps=3):
    f(); lib.kv_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    lib.kv_synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
nk = n * 70
for kk in (31, 51):
    sk = hk.Counttable(kk, 2e9 / 4, 4)
    for mode in ('roll', 'lds'):
        if mode == 'lds': os.environ['KV_NO_ROLL'] = '1'
        else: os.environ.pop('KV_NO_ROLL', None)
        ms = t(lambda: sk.consume_batch(batch, 1 << 20, 0))
        print('k=%d %s hash-only: %.2f ms  %.1f Gkmer/s' % (kk, mode, ms, n * (L - kk + 1) / ms / 1e6))
