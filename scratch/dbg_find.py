import sys, gzip, zlib
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_gpu_ingest import device_gunzip, fastq_text
rng=np.random.default_rng(1)
bases=bytes(rng.choice(np.frombuffer(b'ACGT\n', dtype=np.uint8), 3000000, p=[.3, .2, .2, .29, .01]))
for name, data in (('bases', bases), ('fastq', fastq_text(40000, 3))):
    for lv in (2, 6):
        img=gzip.compress(data, lv)
        got, stats = device_gunzip(img)
        print(name, lv, len(img), got==data, stats)
