"""scratch: kv_gunzip_host against zlib on a few shapes of gzip stream; prints rates"""
import ctypes, gzip, io, sys, time, zlib
import numpy as np
sys.path.insert(0, '.')
from kevlar_amd import _lib
sys.path.insert(0, 'tests')
from test_gpu_ingest import fastq_text

lib = _lib.load()

def gunzip(image, segment=0, cap=None):
    cap = cap or len(gzip.decompress(image)) + 64
    out = ctypes.create_string_buffer(cap)
    n = ctypes.c_uint64(); ms = ctypes.c_double(); stats = (ctypes.c_uint64 * 4)()
    rc = lib.kv_gunzip_host(image, len(image), out, cap, segment, ctypes.byref(n), stats, ctypes.byref(ms))
    if rc != 0:
        return rc, _lib.last_error() if hasattr(_lib, 'last_error') else None, list(stats), ms.value
    return out.raw[:n.value], list(stats), ms.value

def check(name, image, segment=0):
    want = gzip.decompress(image)
    t = time.time()
    got = gunzip(image, segment)
    dt = time.time() - t
    if isinstance(got[0], int):
        print(name, 'rc', got)
        return
    ok = got[0] == want
    print('{:28s} {} text {:>10d} comp {:>10d} seg {:>9d} stats {} device {:.2f} ms wall {:.1f} ms'.format(name, 'ok ' if ok else 'BAD', len(want), len(image), segment, got[1], got[2], dt * 1e3))
    if not ok:
        a = np.frombuffer(got[0][:len(want)], dtype=np.uint8); b = np.frombuffer(want[:len(got[0])], dtype=np.uint8)
        bad = np.flatnonzero(a[:len(b)] != b[:len(a)])
        print('   lengths', len(got[0]), len(want), 'first diffs', bad[:10], 'n', len(bad))

text = fastq_text(40000, 7)
for level in (1, 6, 9):
    img = gzip.compress(text, compresslevel=level)
    check('fastq level %d' % level, img)
    check('fastq level %d' % level, img, 1 << 20)
    check('fastq level %d' % level, img, 200000)
check('two members', gzip.compress(text[:3000000], 6) + gzip.compress(text[3000000:], 4), 1 << 20)
buf = io.BytesIO()
with gzip.GzipFile('some_name.fq', 'wb', 6, buf) as fh:
    fh.write(text[:2000000])
check('with a file name', buf.getvalue())
check('stored', gzip.compress(text[:300000], 0))
check('all A', gzip.compress(b'A' * 5000000, 6))
rng = np.random.default_rng(1)
check('random bytes', gzip.compress(bytes(rng.integers(0, 256, 3000000, dtype=np.uint8)), 6))
check('tiny', gzip.compress(b'@r\nACGT\n+\nIIII\n', 6))
check('empty', gzip.compress(b'', 6))
big = fastq_text(400000, 9)
img = gzip.compress(big, 6)
for _ in range(2):
    check('fastq 400k reads', img)
check('fastq 400k reads', img, 8 << 20)

# ---- through ReadParser: the same batches as the host parser
import os, tempfile
from kevlar_amd import khmer
d = tempfile.mkdtemp()
path = os.path.join(d, 'reads.fq.gz')
with open(path, 'wb') as fh:
    fh.write(img)
def run(env):
    for k in ('KV_INGEST', 'KV_GUNZIP'):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = time.time()
    parser = khmer.ReadParser(path)
    n = 0
    digest = 0
    while True:
        batch = parser.take_batch(1 << 23)
        if batch is None:
            break
        n += batch.n_reads
        batch.close()
    return n, time.time() - t
for env in ({}, {}, {'KV_GUNZIP': 'host'}, {'KV_INGEST': 'host'}):
    n, dt = run(env)
    print(env, n, 'reads', '%.1f ms' % (dt * 1e3), '%.2f M reads/s' % (n / dt / 1e6))
