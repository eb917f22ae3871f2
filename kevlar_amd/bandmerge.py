"""Multi-GPU merge of a k-mer-banded novel scan (one band per rank, one rank per GPU).

kevlar's banding (docs/banding.rst; kevlar/count.py:62-66) splits the 64-bit hash space into N
ranges.  Rank b counts and scans only band b, so its sketches are 1/N of the memory and its
hits are the interesting k-mers of band b.  The reference gathers the per-band outputs with a
file merge (`kevlar unband`, kevlar/unband.py:41-77); here the gather is two collectives over
RCCL/xGMI (gloo in the CPU tests):

  1. all-reduce(SUM) of the per-band bit masks -- one bit per (read, k-mer offset); bands are
     disjoint, so the sum of 0/1 words is their OR;
  2. all-gather of the sparse hits (read, offset, abundances), which only the owning band knows.

torch.distributed is plumbing here: the tensors are plain int32/uint8 buffers.
"""
import numpy as np
import torch
import torch.distributed as dist


def allreduce_mask(mask, group=None):
    """In-place OR of disjoint per-band masks (int32 words)."""
    dist.all_reduce(mask, op=dist.ReduceOp.SUM, group=group)
    return mask


def mask_to_hits(mask, stride):
    """(read, offset) arrays of the set bits, in (read, offset) order."""
    words = mask.detach().cpu().numpy().view(np.uint32)
    nz = np.flatnonzero(words)
    if len(nz) == 0:
        return np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint32)
    bits = ((words[nz][:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool)
    idx = (nz[:, None].astype(np.int64) * 32 + np.arange(32, dtype=np.int64)[None, :])[bits]
    return (idx // stride).astype(np.uint32), (idx % stride).astype(np.uint32)


def allgather_hits(read, offset, abund, device, group=None):
    """Concatenate every rank's hits and return them sorted by (read, offset) as numpy arrays."""
    world = dist.get_world_size(group)
    S = abund.shape[1] if abund.ndim == 2 else 0
    n = torch.tensor([len(read)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    rec = torch.zeros((cap, 2 + S), dtype=torch.int64, device=device)
    if len(read):
        rec[:len(read), 0] = torch.from_numpy(read.astype(np.int64)).to(device)
        rec[:len(read), 1] = torch.from_numpy(offset.astype(np.int64)).to(device)
        if S:
            rec[:len(read), 2:] = torch.from_numpy(abund.astype(np.int64)).to(device)
    parts = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(parts, rec, group=group)
    rows = torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0).cpu().numpy()
    order = np.lexsort((rows[:, 1], rows[:, 0]))
    rows = rows[order]
    return rows[:, 0].astype(np.uint32), rows[:, 1].astype(np.uint32), rows[:, 2:].astype(np.uint8)
