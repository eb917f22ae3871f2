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
    """Concatenate every rank's hits and return them sorted by (read, offset) as numpy arrays.

    One padded all-gather of packed rows: 4 B read, 4 B offset, S abundance bytes (S <= 16)."""
    world = dist.get_world_size(group)
    S = abund.shape[1] if abund.ndim == 2 else 0
    n = torch.tensor([len(read)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    width = 8 + S
    rows = np.zeros((cap, width), dtype=np.uint8)
    if len(read):
        rows[:len(read), 0:4] = np.ascontiguousarray(read, dtype='<u4').view(np.uint8).reshape(-1, 4)
        rows[:len(read), 4:8] = np.ascontiguousarray(offset, dtype='<u4').view(np.uint8).reshape(-1, 4)
        if S:
            rows[:len(read), 8:] = abund
    rec = torch.from_numpy(rows).to(device)
    parts = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(parts, rec, group=group)
    merged = torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0).cpu().numpy()
    mread = np.ascontiguousarray(merged[:, 0:4]).view('<u4').reshape(-1)
    moff = np.ascontiguousarray(merged[:, 4:8]).view('<u4').reshape(-1)
    order = np.lexsort((moff, mread))
    return mread[order].astype(np.uint32), moff[order].astype(np.uint32), merged[order][:, 8:].astype(np.uint8)


def allgather_hits_device(read, offset, abund, device, group=None, staged=False):
    """allgather_hits for ranks that own a GPU: the rows travel as (tag = read << 16 | offset, abundances),
    are gathered on the device and sorted there (kv_hits_from_tagged) -- a host lexsort of a few million
    hits would cost more than the whole banded scan.  `staged` = gloo transport (host-staged)."""
    from kevlar_amd import khmer as hk
    from kevlar_amd import shardrun
    S = abund.shape[1]
    n = len(read)
    # the tag packs a 16-bit offset under the read index, and kv_hits_from_tagged returns the read as u32: hits that
    # do not fit (reads of 64 kb and more, batches beyond 2^32 reads) take the host merge instead of aliasing
    # (decided together: every rank must take the same sequence of collectives)
    wide = torch.tensor([1 if n and (int(np.max(offset)) >= 1 << 16 or int(np.max(read)) >= 1 << 32) else 0],
                        dtype=torch.int64, device=torch.device('cpu') if staged else device)
    dist.all_reduce(wide, op=dist.ReduceOp.MAX, group=group)
    if int(wide.item()):
        return allgather_hits(read, offset, abund, torch.device('cpu') if staged else device, group)
    tags = (np.asarray(read).astype(np.int64) << 16) | np.asarray(offset).astype(np.int64)
    d_tags = torch.from_numpy(tags).to(device)
    d_abund = torch.from_numpy(np.ascontiguousarray(abund)).to(device) if n else torch.zeros((0, S), dtype=torch.uint8, device=device)
    all_tags, total = shardrun.gather_rows(d_tags, n, -1, group, staged)
    all_abund, _ = shardrun.gather_rows(d_abund, n, 0, group, staged)
    torch.cuda.synchronize()
    return hk.hits_from_tagged(all_tags.data_ptr(), all_abund.data_ptr(), all_tags.shape[0], total, S)
