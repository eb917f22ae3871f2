"""world_size 2 and 3 gloo tests (CPU) of the read-sharded exchange (kevlar_amd/shardrun.py).

There is no GPU here, so each rank hashes ITS shard of the reads with the oracle, fills the
per-destination send blocks the way kv_route_hashes does, and the product's transport code
(exchange_rows / gather_rows, host-staged as under gloo on the GPU box) moves them.  What a rank
receives must be exactly the k-mers of its band over ALL reads: counting them reproduces the banded
sketch byte for byte."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def make_reads():
    rng = np.random.default_rng(5)
    letters = np.array(list('ACGT'))
    genome = rng.integers(0, 4, size=3000)
    return [''.join(letters[genome[s:s + 60]]) for s in rng.integers(0, len(genome) - 60, size=801)]


def worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from kevlar_amd import shardrun
    from oracle import okhmer as ok
    k = 21
    reads = make_reads()
    lo, hi = shardrun.shard_bounds(len(reads), world, rank)
    hasher = ok.Counttable(k, 1000, 1)
    bs = (2 ** 64 - 1) // world
    blocks = [[] for _ in range(world)]
    for r in range(lo, hi):
        for i, h in enumerate(hasher.get_kmer_hashes(reads[r])):
            blocks[min(int(h) // bs, world - 1)].append((int(h), (r << 16) | i))
    counts = [len(b) for b in blocks]
    flat = [item for b in blocks for item in b]          # destination 0's items, then destination 1's, ...
    send = torch.zeros((max(1, len(flat)), 2), dtype=torch.int64)
    if flat:
        send[:len(flat)] = torch.from_numpy(np.array(flat, dtype=np.uint64).view(np.int64).reshape(len(flat), 2))
    recv, recv_counts = shardrun.exchange_rows(send, counts, staged=True)
    assert recv.shape == (sum(recv_counts), 2)
    got = recv.numpy().view(np.uint64)
    # (1) exactly the band's k-mers of ALL reads, each with its (read, offset) tag
    expect = []
    for r, seq in enumerate(reads):
        for i, h in enumerate(hasher.get_kmer_hashes(seq)):
            if min(int(h) // bs, world - 1) == rank:
                expect.append((int(h), (r << 16) | i))
    assert sorted(map(tuple, got.tolist())) == sorted(expect)
    # blocks arrive in source-rank order: tags of source s lie in its read range
    pos = 0
    for src, n in enumerate(recv_counts):
        slo, shi = shardrun.shard_bounds(len(reads), world, src)
        rd = got[pos:pos + n, 1] >> np.uint64(16)
        assert ((rd >= slo) & (rd < shi)).all()
        pos += n
    # (2) counting them reproduces the banded sketch
    banded = ok.Counttable(k, 50000, 4)
    bases, offs = ok.concat_reads(reads)
    ok.consume_reads(banded, bases, offs, len(reads), world, rank)
    routed = ok.Counttable(k, 50000, 4)
    for h in got[:, 0]:
        routed.add(int(h))
    for t in range(4):
        assert routed.table_bytes(t) == banded.table_bytes(t)
    # (3) the padded all-gather of ragged per-rank rows
    mine = torch.arange(rank + 2, dtype=torch.int64) + 100 * rank
    rows, total = shardrun.gather_rows(torch.cat([mine, torch.full((5,), 77, dtype=torch.int64)]), rank + 2, -1, staged=True)
    assert total == sum(r + 2 for r in range(world))
    valid = rows[rows != -1]
    assert valid.tolist() == [100 * r + j for r in range(world) for j in range(r + 2)]
    # (4) a rank that cannot produce its part says so inside the size exchange: every rank raises, none is left in a collective
    for decliner in range(world):
        try:
            shardrun.exchange_rows_async(send, None if rank == decliner else counts, staged=True)
            raised = False
        except shardrun.PeerDeclined:
            raised = True
        assert raised
    recv2, recv_counts2 = shardrun.exchange_rows(send, counts, staged=True)       # and the next exchange is in step again
    assert recv_counts2 == recv_counts and torch.equal(recv2, recv)
    # (5) slabs cut at split points every rank knows (the records of the minimizer layout): source r sends (r + 1) * (d + 1) words
    # to destination d; blocking, asynchronous (two in flight, waited for in the other order) and host-staged give the same slabs
    def slab(src, dst, salt):
        return [1000 * src + 10 * dst + salt + j for j in range((src + 1) * (dst + 1))]
    in_splits = [(rank + 1) * (d + 1) for d in range(world)]
    out_splits = [(src + 1) * (rank + 1) for src in range(world)]
    for staged in (True, False):
        send_a = torch.tensor([v for d in range(world) for v in slab(rank, d, 1)], dtype=torch.int64)
        send_b = torch.tensor([v for d in range(world) for v in slab(rank, d, 2)], dtype=torch.int64)
        want_a = [v for src in range(world) for v in slab(src, rank, 1)]
        want_b = [v for src in range(world) for v in slab(src, rank, 2)]
        assert shardrun.exchange_slabs(send_a, in_splits, out_splits, staged=staged).tolist() == want_a
        first = shardrun.exchange_slabs_async(send_a, in_splits, out_splits, staged=staged)
        second = shardrun.exchange_slabs_async(send_b, in_splits, out_splits, staged=staged)
        assert second.wait().tolist() == want_b and first.wait().tolist() == want_a
    dist.barrier()
    dist.destroy_process_group()
    with open(result_file + str(rank), 'w') as fh:
        fh.write('ok')


@pytest.mark.parametrize('world', [2, 3])
def test_exchange_by_band(tmp_path, world):
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    result = str(tmp_path / 'done')
    mp.spawn(worker, args=(world, port, result), nprocs=world, join=True)
    for r in range(world):
        assert open(result + str(r)).read() == 'ok'
