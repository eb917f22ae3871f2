import sys, gzip, zlib, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_gpu_ingest import device_gunzip, fastq_text
text = fastq_text(40000, 31)
for lv in (1, 6):
    img = gzip.compress(text, lv)
    try:
        got, stats = device_gunzip(img)
        print(lv, len(img), got == text, stats)
    except Exception as e:
        print(lv, 'ERR', e)
