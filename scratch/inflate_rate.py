"""device inflate rate on a big BGZF FASTQ image: python scratch/inflate_rate.py [reads]"""
import ctypes, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from kevlar_amd import _lib, bgzf
from test_gpu_ingest import fastq_text, device_inflate
_lib.load(); _lib.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
text = fastq_text(n, 1)
t0 = time.time()
image = b''.join(bgzf.member(text[i:i + bgzf.BLOCK_TEXT], 6) for i in range(0, len(text), bgzf.BLOCK_TEXT)) + bgzf._EOF
print('text {:.1f} MB -> {:.1f} MB compressed ({:.1f} s in python zlib)'.format(len(text) / 1e6, len(image) / 1e6, time.time() - t0))
t0 = time.time(); ref = zlib.decompress(image, 31) if False else None
import gzip
t0 = time.time(); ref = gzip.decompress(image); t_cpu = time.time() - t0
for rep in range(3):
    got, members, ms = device_inflate(image)
    print('device: {} members in {:.2f} ms = {:.2f} GB/s of text ({:.1f} M reads/s); zlib one core {:.0f} MB/s'.format(
        members, ms, len(text) / ms / 1e6, n / ms / 1e3, len(text) / t_cpu / 1e6))
assert got == text
