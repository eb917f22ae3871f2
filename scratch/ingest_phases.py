"""kernel times of the device ingest on a FASTQ file: python scratch/ingest_phases.py [reads] [bgzf|gzip|plain]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kevlar_amd import _lib, bgzf, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
L = 100
words = synth.trio_reads_packed(25_000_000, 30, L)['proband'][:n]
rng = np.random.default_rng(12)
tag = b'@proband_'
rec = np.empty((n, len(tag) + 8 + 1 + L + 3 + L + 1), dtype=np.uint8)
col = 0
rec[:, :len(tag)] = np.frombuffer(tag, dtype=np.uint8); col += len(tag)
digits = np.arange(n, dtype=np.int64)
for d in range(8):
    rec[:, col + 7 - d] = 48 + digits % 10
    digits //= 10
col += 8; rec[:, col] = 10; col += 1
for j in range(L):
    rec[:, col + j] = np.frombuffer(b'ACGT', dtype=np.uint8)[(words[:, j >> 4] >> np.uint32(2 * (j & 15))) & np.uint32(3)]
col += L
rec[:, col:col + 3] = np.frombuffer(b'\n+\n', dtype=np.uint8); col += 3
rec[:, col:col + L] = np.frombuffer(b'F:,#', dtype=np.uint8)[rng.choice(4, size=(n, L), p=[0.9, 0.06, 0.03, 0.01])]; col += L
rec[:, col] = 10
kind = sys.argv[2] if len(sys.argv) > 2 else 'bgzf'
path = '/tmp/phases.fq' if kind == 'plain' else '/tmp/phases.fq.gz'
if kind == 'bgzf':
    bgzf.write_file(path, rec.tobytes(), level=int(os.environ.get('BGZF_LEVEL', '4')), threads=16)
elif kind == 'gzip':
    import bench
    bench.write_gzip(path, rec.tobytes(), level=4, threads=16)
else:
    with open(path, 'wb') as fh:
        fh.write(rec.tobytes())
print('file', os.path.getsize(path) >> 20, 'MB for', rec.nbytes >> 20, 'MB of text')
for rep in range(3):
    lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t0 = time.perf_counter()
    parser, got = hk.ReadParser(path), 0
    while True:
        b = parser.take_batch(hk.BATCH_READS)
        if b is None:
            break
        got += b.n_reads; b.close()
    dt = time.perf_counter() - t0
    lib.kv_prof_enable(0)
    assert got == n
print('{:.1f} ms wall = {:.1f} M reads/s'.format(dt * 1e3, n / dt / 1e6))
buf = ctypes.create_string_buffer(8192); lib.kv_prof_names(buf, 8192)
for name in buf.value.decode().split(','):
    ms, nl = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(nl))
    print('    {:20s} {:8.3f} ms {:4d} launches'.format(name, ms.value, nl.value))
