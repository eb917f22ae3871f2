"""kv_readgraph_components at config-4 scale: 1 M reads, 10 M annotations (DESIGN.md section 8 / round-1 verdict item 7).

100 k variant loci, ten reads of 100 bp over each (random start within +-30 bp), every read annotated with the ten
k-mers (k=31) that span its locus' variant position: reads of one locus share k-mers, so the expected answer is
100 k components of ten reads.  Checks the labels against that and times the call; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (profiles/r2_readgraph/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__; __graft_entry__.build()
from kevlar_amd import _lib, khmer as hk
lib = _lib.load(); _lib.require_device()
rng = np.random.default_rng(3)
K, L, LOCI, COPIES, ANN = 31, 100, 100_000, 10, 10
genome = rng.integers(0, 4, size=LOCI * 400, dtype=np.uint8)
centre = np.arange(LOCI, dtype=np.int64) * 400 + 200
start = np.repeat(centre, COPIES) - 50 + rng.integers(-15, 16, size=LOCI * COPIES)
n_reads = len(start)
idx = start[:, None] + np.arange(L)[None, :]
codes = genome[idx]
# pack 2 bits per base, 16 bases per word
wpr = (L + 15) // 16
pad = np.zeros((n_reads, wpr * 16), dtype=np.uint32); pad[:, :L] = codes
words = (pad.reshape(n_reads, wpr, 16) << (2 * np.arange(16, dtype=np.uint32))[None, None, :]).sum(axis=2, dtype=np.uint32)
batch = hk.ReadBatch.from_packed(words, L)
# the ten k-mers ending at / after the locus centre: genome positions centre-30+1 .. centre-30+10 -> read offsets
gpos = np.repeat(centre, COPIES)[:, None] - 30 + np.arange(1, ANN + 1)[None, :]
off = (gpos - start[:, None]).astype(np.uint32)
assert off.min() >= 0 and off.max() + K <= L
ann_read = np.repeat(np.arange(n_reads, dtype=np.uint32), ANN)
ann_off = off.reshape(-1)
node = np.arange(n_reads, dtype=np.uint32)
for rep in range(3):
    t0 = time.perf_counter()
    out = hk.readgraph_components(batch, K, ann_read, ann_off, node, n_reads, 2, 200, want_edges=(rep == 2))
    labels, nedges = out if rep == 2 else (out, None)
    dt = time.perf_counter() - t0
    ncomp = len(np.unique(labels))
    print('run', rep, '%.1f ms' % (dt * 1e3), 'components', ncomp, 'edges', nedges, flush=True)
want = np.repeat(np.arange(LOCI, dtype=np.uint32) * COPIES, COPIES)
assert np.array_equal(labels, want), 'every locus must be one component labelled by its first read'
print('reads', n_reads, 'annotations', len(ann_read), 'ok')
