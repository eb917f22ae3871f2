"""Randomised kv_gunzip_host against zlib: data of random character (FASTQ-like, low entropy, runs, noise, mixtures), random
level / strategy (default, filtered, Huffman only, RLE, fixed codes), random sync / full flushes inside the stream, one or
several members, random segment size and block-search chunk size.  python scratch/fuzz_gunzip.py [trials] [seed]"""
import ctypes, os, struct, sys, zlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
from kevlar_amd import _lib
from test_gpu_ingest import device_gunzip, fastq_text
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
fastq = fastq_text(60000, 77)


def some_data(simple=False):
    kind = int(rng.integers(0, 5 if simple else 6))
    n = int(rng.choice([100, 70000, 300000] if simple else [0, 1, 100, 70000, 300000, 3000000, 12000000]))
    if kind == 0:
        at = int(rng.integers(0, max(1, len(fastq) - n)))
        return fastq[at:at + n], 'fastq'
    if kind == 1:
        return bytes(rng.integers(0, 256, n, dtype=np.uint8)), 'noise'
    if kind == 2:
        return bytes(rng.choice(np.frombuffer(b'ACGT\n', dtype=np.uint8), n, p=[.3, .2, .2, .29, .01])), 'bases'
    if kind == 3:
        runs = rng.integers(1, 400, max(1, n // 100))
        vals = rng.integers(65, 70, len(runs), dtype=np.uint8)
        return np.repeat(vals, runs)[:n].tobytes(), 'runs'
    if kind == 4:
        word = bytes(rng.integers(97, 123, int(rng.integers(1, 40000)), dtype=np.uint8))
        return (word * (n // max(1, len(word)) + 1))[:n], 'periodic'
    parts = []
    while sum(map(len, parts)) < n:
        parts.append(some_data(True)[0][:int(rng.integers(1, 200000))])
    return b''.join(parts)[:n], 'mixture'


def member(data):
    level = int(rng.choice([1, 2, 4, 6, 9]))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    z = zlib.compressobj(level, zlib.DEFLATED, -15, int(rng.choice([1, 8, 9])), strategy)
    out, at = [], 0
    while at < len(data):
        step = int(rng.choice([len(data), 1000, 65536, 1 << 20]))
        out.append(z.compress(data[at:at + step]))
        at += step
        if at < len(data) and rng.random() < 0.3:
            out.append(z.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH]))))
    out.append(z.flush(zlib.Z_FINISH))
    head = b'\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03'
    if rng.random() < 0.3:
        head = b'\x1f\x8b\x08\x08\x00\x00\x00\x00\x00\x03' + b'name.fq\x00'
    return head + b''.join(out) + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff), (level, strategy)


fails = 0
for trial in range(trials):
    pieces, what = [], []
    for _ in range(int(rng.choice([1, 1, 1, 2, 3]))):
        data, kind = some_data()
        image, how = member(data)
        pieces.append((data, image)); what.append((kind, len(data)) + how)
    image = b''.join(p[1] for p in pieces)
    want = b''.join(p[0] for p in pieces)
    segment = int(rng.choice([0, 0, 100000, 1 << 20, 1 << 24]))
    chunk = rng.choice([None, '1', '4', '64'])
    if chunk: os.environ['KV_GUNZIP_CHUNK_KB'] = chunk
    else: os.environ.pop('KV_GUNZIP_CHUNK_KB', None)
    desc = 'trial {} {} segment={} chunk={}'.format(trial, what, segment, chunk)
    try:
        got, stats = device_gunzip(image, segment, cap=len(want) + 64)
        assert got == want, 'text differs (lengths {} {})'.format(len(got), len(want))
        print('ok  ', desc, stats, flush=True)
    except ValueError as exc:
        # the decoder may decline (no block start for megabytes: Huffman-only or fixed-code streams are one block); never a wrong answer
        # (both are KV_ERR_TYPE: kv_fastx_next then reads the file through zlib from the same text offset)
        declined = 'no DEFLATE block start' in str(exc) or 'too many stretches needed decoding again' in str(exc)
        print('decl' if declined else 'FAIL', desc, repr(exc)[:200], flush=True)
        fails += 0 if declined else 1
    except Exception as exc:
        fails += 1
        print('FAIL', desc, repr(exc)[:300], flush=True)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
