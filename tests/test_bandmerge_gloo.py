"""world_size-2 gloo test of the multi-GPU band merge (kevlar_amd/bandmerge.py) on CPU.

Each rank plays one band: it counts and scans its hash range with the ORACLE (there is no GPU
here), then the product's merge code all-reduces the band masks and all-gathers the hits.
With tables large enough that Count-Min collisions play no role, the merged result must equal
the unbanded scan."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def make_samples():
    rng = np.random.default_rng(12)
    letters = np.array(list('ACGT'))
    genome = rng.integers(0, 4, size=4000)
    kid = genome.copy()
    kid[[1000, 2500]] = (kid[[1000, 2500]] + 1) % 4

    def sample(g, n):
        return [''.join(letters[g[s:s + 80]]) for s in rng.integers(0, len(g) - 80, size=n)]
    return [sample(kid, 1500), sample(genome, 1500), sample(genome, 1500)]


def scan(ok, samples, k, nbands, band):
    sketches = [ok.Counttable(k, 4e6, 4) for _ in samples]
    for sk, seqs in zip(sketches, samples):
        bases, offs = ok.concat_reads(seqs)
        ok.consume_reads(sk, bases, offs, len(seqs), nbands, band)
    bases, offs = ok.concat_reads(samples[0])
    hits, _ = ok.novel_scan(sketches[:1], sketches[1:], bases, offs, len(samples[0]), k, 6, 0,
                            band_mode=1 if nbands else 0, nbands=nbands, band=band)
    return hits


def worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from kevlar_amd import bandmerge
    from oracle import okhmer as ok
    k, L = 25, 80
    stride = L - k + 1
    samples = make_samples()
    hits = scan(ok, samples, k, world, rank)
    read = np.array([h[0] for h in hits], dtype=np.uint32)
    off = np.array([h[1] for h in hits], dtype=np.uint32)
    abund = np.array([h[2] for h in hits], dtype=np.uint8).reshape(len(hits), 3)
    mask = torch.zeros((len(samples[0]) * stride + 31) // 32, dtype=torch.int32)
    bits = read.astype(np.int64) * stride + off
    words = mask.numpy().view(np.uint32)
    np.bitwise_or.at(words, bits >> 5, (np.uint32(1) << (bits & 31).astype(np.uint32)))
    bandmerge.allreduce_mask(mask)
    mread, moff = bandmerge.mask_to_hits(mask, stride)
    gread, goff, gabund = bandmerge.allgather_hits(read, off, abund, torch.device('cpu'))
    if rank == 0:
        want = scan(ok, samples, k, 0, 0)
        assert len(want) > 0 and 0 < len(hits) < len(want)
        assert list(zip(mread.tolist(), moff.tolist())) == [(r, o) for r, o, _ in want]
        got = [(int(r), int(o), tuple(int(x) for x in a)) for r, o, a in zip(gread, goff, gabund)]
        assert got == want
        open(result_file, 'w').write('ok {}'.format(len(want)))
    dist.barrier()
    dist.destroy_process_group()


def test_band_merge_two_ranks_gloo(tmp_path):
    result = str(tmp_path / 'result.txt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(2, port, result), nprocs=2, join=True)
    assert open(result).read().startswith('ok')
