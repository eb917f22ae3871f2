"""device rate on one big ordinary gzip stream of FASTQ, per kernel: python scratch/gunzip_rate.py [reads] [level]
(quality strings of real reads are far more regular than the uniform ones of tests/test_gpu_ingest.fastq_text: `binned`
draws them from 8 values in runs, closer to what a sequencer writes)"""
import ctypes, gzip, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from kevlar_amd import _lib, synth
from test_gpu_ingest import device_gunzip
lib = _lib.load(); _lib.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def text_of(n, binned):
    trio = synth.make_trio(1000000, 3)
    packed = synth.sample_reads_packed(trio['proband'], n, 100, 0.005, 4)
    seqs = np.frombuffer(''.join(synth.unpack_reads(packed, 100)).encode('ascii'), dtype=np.uint8).reshape(n, 100)
    rng = np.random.default_rng(5)
    if binned:
        levels = np.frombuffer(b'#,5:AFI?', dtype=np.uint8)
        runs = rng.integers(0, 8, size=(n, 10))
        quals = levels[np.repeat(runs, 10, axis=1)]
    else:
        quals = rng.integers(33, 74, size=(n, 100), dtype=np.uint8)
    names = ['@read{}/1 sample=proband\n'.format(i).encode('ascii') for i in range(n)]
    width = max(len(x) for x in names)
    rec = np.full((n, width + 100 + 3 + 100 + 1), 0, dtype=np.uint8)
    out = bytearray()
    for i in range(n):
        out += names[i]; out += seqs[i].tobytes(); out += b'\n+\n'; out += quals[i].tobytes(); out += b'\n'
    return bytes(out)


for binned in (False, True):
    text = text_of(n, binned)
    t0 = time.time(); image = gzip.compress(text, level); t_c = time.time() - t0
    t0 = time.time(); ref = gzip.decompress(image); t_cpu = time.time() - t0
    print('{} qualities: text {:.1f} MB -> {:.1f} MB (ratio {:.2f}); zlib inflate on one core {:.0f} MB/s'.format(
        'binned' if binned else 'uniform', len(text) / 1e6, len(image) / 1e6, len(text) / len(image), len(text) / t_cpu / 1e6))
    for rep in range(3):
        out = ctypes.create_string_buffer(len(text) + 65)
        nb, ms, stats = ctypes.c_uint64(), ctypes.c_double(), (ctypes.c_uint64 * 4)()
        lib.kv_prof_enable(1); lib.kv_prof_reset()
        _lib.check(lib.kv_gunzip_host(image, len(image), out, len(text) + 64, 0, ctypes.byref(nb), stats, ctypes.byref(ms)))
        assert out.raw[:nb.value] == text
        per = {}
        for name in ('k_gz_find', 'k_gz_decode', 'k_gz_tails', 'k_gz_scan', 'k_gz_resolve', 'k_gz_crc'):
            kms, cnt = ctypes.c_double(), ctypes.c_uint64()
            lib.kv_prof_get(name.encode(), ctypes.byref(kms), ctypes.byref(cnt))
            per[name] = round(kms.value, 2)
        lib.kv_prof_enable(0)
        print('  ', per)
        print('  device: {:.2f} ms = {:.2f} GB/s of text ({:.1f} M reads/s); passes, stretches, dropped, again = {}'.format(
            ms.value, len(text) / ms.value / 1e6, n / ms.value / 1e3, list(stats)))
