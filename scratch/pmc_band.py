# two batches per sample of config 4's band shape (3 Gb genome, 75 M device-generated reads per batch, band 0 of 8, 8 GB band sketches)
# counted and scanned once, for rocprofv3 --pmc passes (PMC_SCRIPT=scratch/pmc_band.py scratch/pmc_skm.sh)
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk
lib = _lib.load(); _lib.require_device()
G, L, k, per = 3_000_000_000, 100, 31, 75_000_000
names = ('mother', 'father', 'proband')
batches = {n: [hk.ReadBatch.generate(G, 42, (si + 1) % 3, lo, per, L) for lo in (0, per)] for si, n in enumerate(names)}
sk = {n: hk.Counttable(k, 8e9 / 4, 4) for n in names}
for n in names:
    for b in batches[n]:
        sk[n].consume_batch(b, 8, 0)
hits = 0
for b in batches['proband']:
    hits += len(hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], b, 6, 1, band_mode=1, nbands=8, band=0)[0])
print(hits)
