"""One rank of tests/test_gpu_shard.py::test_sharded_trio_ranks_share_one_gpu (not a test module)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def against_the_oracle(reads, names, k, mem, world, rank, sharded, hits):
    """the sharded result against the ORACLE itself, not only the banded device path (a mistake common to both device paths would
    pass the checks in main()): every band of every sample counted on the host in one pass over the reads
    (kvo_consume_reads_mt_allbands = `kevlar count --num-bands N --band b` for every b, kevlar/count.py:62-66), the merged hits of
    its all-band scan (kevlar/novel.py:123-169 per band, then kevlar/unband.py:41-77).  SHARD_ORACLE=0 skips it (the decline cases
    repeat layouts that the plain cases already hold against the oracle)."""
    from oracle import okhmer as ok
    by_band = {}
    for n in names:
        bases, offs = ok.concat_reads(reads[n])
        by_band[n] = [ok.Counttable(k, mem / world / 4, 4) for _ in range(world)]
        assert ok.consume_reads_mt_allbands(by_band[n], bases, offs, len(reads[n]), 2) == sum(max(0, len(s) - k + 1) for s in reads[n])
        mine = by_band[n][rank]
        assert sharded[n].hashsizes() == mine.hashsizes()
        for t in range(4):
            assert sharded[n].table_bytes(t) == mine.table_bytes(t), ('oracle', n, t)
        assert sharded[n].n_occupied() == mine.n_occupied()
    bases, offs = ok.concat_reads(reads['proband'])
    wr, wo, wa, _wb = ok.novel_scan_mt_allbands([[by_band['proband'][b]] for b in range(world)],
                                                [[by_band['mother'][b], by_band['father'][b]] for b in range(world)],
                                                bases, offs, len(reads['proband']), k, 6, 1, 2)
    r, o, a = hits
    assert len(wr) == len(r) and np.array_equal(r, wr) and np.array_equal(o, wo) and np.array_equal(a, wa), 'hits differ from the oracle'


def main():
    torch.cuda.init()
    torch.cuda.set_device(0)
    backend = os.environ.get('SHARD_BACKEND', 'gloo')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    from kevlar_amd import _lib, bandmerge, khmer as hk, shardrun, synth
    _lib.load()
    _lib.require_device()
    k, mem = int(os.environ.get('SHARD_K', '31')), 4.0e6
    L = int(os.environ.get('SHARD_L', '100'))
    trio = synth.make_trio(250000, 33 + k + L)
    names = ('proband', 'mother', 'father')
    reads = {n: synth.unpack_reads(synth.sample_reads_packed(trio[n], 5000100 // L, L, 0.005, 7 + i), L)
             for i, n in enumerate(names)}
    reads['proband'][3] = reads['proband'][3][:30] + 'N' + reads['proband'][3][31:]
    for i in range(11, len(reads['proband']), 97):         # one read in a hundred is skipped by the scan and still counted: some of them hold novel k-mers
        reads['proband'][i] = reads['proband'][i][:60] + 'N' + reads['proband'][i][61:]

    if os.environ.get('SHARD_RAGGED'):                      # one shorter read in that rank's shard of the mother's reads: its cut has no
        lo, _ = shardrun.shard_bounds(len(reads['mother']), world, int(os.environ['SHARD_RAGGED']))     # 16-byte records, the sample falls back
        reads['mother'][lo + 5] = reads['mother'][lo + 5][:-7]

    run = shardrun.ShardedTrio(k, hk.Counttable)
    sharded = {n: hk.Counttable(k, mem / world / 4, 4) for n in names}
    # (the case sample last: what its combine leaves on the device is what the owners answer the scan from)
    for n in (names[1:] + names[:1] if os.environ.get('SHARD_MINIMIZER') == '1' else names):
        lo, hi = shardrun.shard_bounds(len(reads[n]), world, rank)
        counted = run.count_sample(sharded[n], hk.ReadBatch(reads[n][lo:hi]), lo, keep_for_scan=(n == 'proband'),
                                   distinct=os.environ.get('SHARD_DISTINCT') == '1',
                                   minimizer=(len(reads[n]), L) if os.environ.get('SHARD_MINIMIZER') == '1' else None)
        total = torch.tensor([counted], dtype=torch.int64)
        if backend == 'nccl':
            total = total.cuda()
        dist.all_reduce(total)
        assert int(total.item()) == sum(max(0, len(s) - k + 1) for s in reads[n]), n
    if os.environ.get('SHARD_MINIMIZER') == '1' and os.environ.get('SHARD_SCAN', 'owner') == 'owner':
        # the owners of the minimizer buckets answer the scan (a sample that fell back to `distinct` pairs: every rank its own shard)
        lo, hi = shardrun.shard_bounds(len(reads['proband']), world, rank)
        r, o, a = run.scan_minimizer([sharded['proband']], [sharded['mother'], sharded['father']], 6, 1,
                                     hk.ReadBatch(reads['proband'][lo:hi]), lo)
        if not os.environ.get('KV_MEX_TEST_DECLINE', '').startswith('owner') and not os.environ.get('SHARD_RAGGED'):        # (an owner that cannot answer: the scan falls back, the layout did not; a control that went as pairs: the case sample did not)
            assert (getattr(run, 'scan_fallbacks', 0) == 0) == (getattr(run, 'fallbacks', 0) == 0), (getattr(run, 'scan_fallbacks', 0), getattr(run, 'fallbacks', 0))
    elif os.environ.get('SHARD_DISTINCT') == '1' or os.environ.get('SHARD_MINIMIZER') == '1':
        lo, hi = shardrun.shard_bounds(len(reads['proband']), world, rank)
        r, o, a = run.scan_distinct([sharded['proband']], [sharded['mother'], sharded['father']], 6, 1,
                                    hk.ReadBatch(reads['proband'][lo:hi]), lo)
    else:
        r, o, a = run.scan([sharded['proband']], [sharded['mother'], sharded['father']], 6, 1)

    # the banded run of the same trio, band = rank, every read hashed here
    full = {n: hk.ReadBatch(reads[n]) for n in names}
    banded = {n: hk.Counttable(k, mem / world / 4, 4) for n in names}
    for n in names:
        banded[n].consume_batch(full[n], world, rank)
        for t in range(4):
            assert sharded[n].table_bytes(t) == banded[n].table_bytes(t), (n, t)
        assert sharded[n].n_occupied() == banded[n].n_occupied()
    br, bo, ba, _ = hk.novel_scan([banded['proband']], [banded['mother'], banded['father']], full['proband'], 6, 1,
                                  band_mode=1, nbands=world, band=rank)
    mr, mo, ma = bandmerge.allgather_hits_device(br, bo, ba, torch.device('cuda', 0), staged=(backend != 'nccl'))
    if backend != 'nccl':
        hr, ho, ha = bandmerge.allgather_hits(br, bo, ba, torch.device('cpu'))     # the host merge agrees with the device merge
        assert np.array_equal(hr, mr) and np.array_equal(ho, mo) and np.array_equal(ha, ma)
    assert len(mr) > 50
    assert np.array_equal(r, mr) and np.array_equal(o, mo) and np.array_equal(a, ma)

    if os.environ.get('SHARD_ORACLE', '1') == '1':
        against_the_oracle(reads, names, k, mem, world, rank, sharded, (r, o, a))
    dist.barrier()
    dist.destroy_process_group()
    print('shard worker ok: rank {} of {}, {} hits, {} fallbacks, {} scan fallbacks, {} unexpected, own failures {}'.format(
        rank, world, len(r), run.fallbacks, run.scan_fallbacks, run.unexpected_failures, sorted(run.fallback_reasons.items())))


if __name__ == '__main__':
    main()
