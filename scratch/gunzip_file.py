"""kv_gunzip_host on a file: python scratch/gunzip_file.py FILE.gz  (stats + per-kernel times)"""
import ctypes, gzip, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kevlar_amd import _lib
lib = _lib.load(); _lib.require_device()
image = open(sys.argv[1], 'rb').read()
text = gzip.decompress(image)
for rep in range(2):
    out = ctypes.create_string_buffer(len(text) + 65)
    nb, ms, stats = ctypes.c_uint64(), ctypes.c_double(), (ctypes.c_uint64 * 4)()
    lib.kv_prof_enable(1); lib.kv_prof_reset()
    _lib.check(lib.kv_gunzip_host(image, len(image), out, len(text) + 64, 0, ctypes.byref(nb), stats, ctypes.byref(ms)))
    per = {}
    for name in ('k_gz_find', 'k_gz_decode', 'k_gz_tails', 'k_gz_scan', 'k_gz_resolve', 'k_gz_crc'):
        kms, cnt = ctypes.c_double(), ctypes.c_uint64()
        lib.kv_prof_get(name.encode(), ctypes.byref(kms), ctypes.byref(cnt))
        per[name] = (round(kms.value, 2), cnt.value)
    lib.kv_prof_enable(0)
    print(out.raw[:nb.value] == text, len(image), len(text), list(stats), round(ms.value, 2), per)
