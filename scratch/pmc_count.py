# one count of each sample + one novel scan of config 2, for rocprofv3 --pmc passes (scratch/pmc_skm.sh)
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, int(os.environ.get('PMC_K', '31'))
packed = synth.trio_reads_packed(25_000_000, 30, L)
names = ('mother', 'father', 'proband')
batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
sk = {n: hk.Counttable(k, 5e8, 4) for n in names}
sk['proband'].expect_scan(steady=True)        # as bench.py does: the distinct list from the first batch on (the list scan is what it times)
for n in names: sk[n].consume_batch(batches[n])
r = hk.novel_scan([sk['proband']], [sk['mother'], sk['father']], batches['proband'], 6, 1)
print(len(r[0]))
