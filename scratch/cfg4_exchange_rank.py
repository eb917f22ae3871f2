"""BASELINE.json configs[3] (3 Gb trio, 30x, k = 31, 8 GPUs) in the EXCHANGE layout, replayed on one GPU at true size: what one rank of
eight computes, and -- by playing every bucket owner in turn -- band 0's sketches, which must equal the banded count's.

Per sample: every shard (1/8 of the reads, generated on the device) is cut into 16-byte super-k-mer records under the sample's plan
(kv_mex_emit_pack); what each shard holds of an owner's minimizer buckets is set aside as that owner's received records; every owner
combines its buckets at the sample's full coverage (kv_mex_route -- in passes, the buckets of a 63 G-k-mer sample being eight times the
LDS table: SkmGeom::passes) and hands band 0 its (hash, occurrences) pairs, which band 0's sketch adds (kv_consume_hashes_weighted).
Timed: rank 0's own cut, owner 0's combine, and band 0 adding as many pairs as it receives from all owners.  Checked: band 0's tables
against a banded count of all reads (kv_consume with band 0 of 8: what bench.py --workload cfg4-band counts).

    gpurun -- python scratch/cfg4_exchange_rank.py [genome_mb] [world]       (defaults 3000, 8; 250 is the quick check)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__
__graft_entry__.build_product()
from kevlar_amd import _lib, khmer as hk, shardrun


def main():
    genome_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    k, L, T, cov, seed = 31, 100, 4, 30.0, 42
    memory = 64e9 * genome_mb / 3000.0
    genome_len = int(genome_mb * 1e6)
    n_reads = int(genome_len * cov / L)
    nk = L - k + 1
    names = ('proband', 'mother', 'father')
    dev = torch.device('cuda', 0)
    lib = _lib.load()
    plan = hk.mex_plan(hk.Counttable, k, n_reads, L, world, short=True)
    recw, nwg1 = int(plan.recw), int(plan.nwg1)
    print('{} Mb, {} reads per sample, {} ranks; plan: {} x {} buckets, {} writers, {}-byte records, segments of {} records ({:.1f} GB per shard)'.format(
        genome_mb, n_reads, world, int(plan.C1), int(plan.F2), nwg1, 8 * recw, int(plan.cap1), int(plan.seg_words) * 8 / 1e9), flush=True)
    band0 = {n: hk.Counttable(k, memory / world / T, T) for n in names}
    bounds = [shardrun.shard_bounds(n_reads, world, r) for r in range(world)]
    width = [int(plan.c_lo[d + 1]) - int(plan.c_lo[d]) for d in range(world)]
    if os.environ.get('CFG4_PROF'):
        lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t_cut = t_combine = t_add = 0.0
    sent_records = sent_pairs = 0
    pairs_cap = int(n_reads * nk * float(os.environ.get('CFG4_PAIRS_FRAC', '0.205')) / world) + (1 << 22)      # (a fifth of the occurrences are distinct at 30x; the library sizes its own
                                                                    # staging from this number, generously)
    group = int(os.environ.get('CFG4_OWNERS_AT_ONCE', '2'))      # owners whose records are held at once (every shard is cut once per group)
    for si, n in enumerate(names):
        arrived_all = 0
        for d0 in range(0, world, group):
            owners = list(range(d0, min(world, d0 + group)))
            # every shard's cut; what it holds for these owners is set aside
            hk.scratch_trim()                                # (the owner's combine of a sample this size leaves ~200 GB of working buffers)
            got_rec = {d: [] for d in owners}
            got_cnt = {d: [] for d in owners}
            seg = torch.empty(int(plan.seg_words), dtype=torch.int64, device=dev)
            cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=dev)
            out = torch.empty(int(plan.seg_words) // 2 + 4096, dtype=torch.int64, device=dev)
            for r, (lo, hi) in enumerate(bounds):
                batch = hk.ReadBatch.generate(genome_len, seed, si, lo, hi - lo, L)
                if r == 0 and d0 == 0:                       # (once untimed: the stream arena grows on the first call)
                    hk.mex_emit_pack(batch, plan, lo, seg.data_ptr(), cnt.data_ptr(), out.data_ptr(), out.shape[0])
                lib.kv_synchronize()
                ta = time.perf_counter()
                per_dest, fitted = hk.mex_emit_pack(batch, plan, lo, seg.data_ptr(), cnt.data_ptr(), out.data_ptr(), out.shape[0])
                lib.kv_synchronize()
                if r == 0 and d0 == 0:
                    t_cut += time.perf_counter() - ta
                    sent_records += sum(per_dest) - per_dest[0]
                assert fitted, 'the packed records of shard {} did not fit half the segments'.format(r)
                first = 0
                for d in range(world):
                    if d in got_rec:
                        got_rec[d].append(out[first * recw:(first + per_dest[d]) * recw].clone())
                        got_cnt[d].append(cnt[int(plan.c_lo[d]) * nwg1:int(plan.c_lo[d + 1]) * nwg1].clone())
                    first += per_dest[d]
                del batch
            # these owners' combines; band 0 adds what each sends it
            del seg, out, cnt
            torch.cuda.empty_cache()
            pairs = torch.empty((pairs_cap, 2), dtype=torch.int64, device=dev)
            for d in owners:
                rs = torch.cat(got_rec[d]); rc = torch.cat(got_cnt[d])
                got_rec[d] = got_cnt[d] = None
                if d == 0:                                   # (once untimed, as above)
                    hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, pairs.data_ptr(), pairs_cap, compact=True)
                lib.kv_synchronize()
                ta = time.perf_counter()
                counts, arrived = hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, pairs.data_ptr(), pairs_cap, compact=True)
                lib.kv_synchronize()
                tb = time.perf_counter()
                arrived_all += arrived
                del rs, rc
                torch.cuda.empty_cache()
                if d == 0:
                    t_combine += tb - ta
                    sent_pairs += sum(counts) - counts[0]
                    # (a band owner receives about what a bucket owner sends: band 0 adding ALL of owner 0's pairs into a scratch sketch is
                    # the size of its real work; only the pairs of band 0 go into the sketch that is checked)
                    scratch = hk.Counttable(k, memory / world / T, T)
                    chunk = 1 << 28                            # (pairs per call: the partitioned add stages four items a pair, twice)
                    for rep in range(2):
                        scratch.clear()
                        lib.kv_synchronize()
                        tc = time.perf_counter()
                        for c0 in range(0, sum(counts), chunk):
                            scratch.consume_hashes_weighted(pairs.data_ptr() + c0 * 16, min(chunk, sum(counts) - c0))
                        lib.kv_synchronize()
                        if rep == 1:
                            t_add += time.perf_counter() - tc
                    del scratch
                for c0 in range(0, counts[0], 1 << 29):
                    band0[n].consume_hashes_weighted(pairs.data_ptr() + c0 * 16, min(1 << 29, counts[0] - c0))
            del pairs
            torch.cuda.empty_cache()
        assert arrived_all == n_reads * nk, (arrived_all, n_reads * nk)
        print('{}: cut, combined by {} owners, band 0 added'.format(n, world), flush=True)
    print('one rank, three samples: its cut {:.3f} s, its buckets combined {:.3f} s, its band\'s pairs added {:.3f} s -> {:.3f} s before the scan; '
          'it sends {:.1f} GB of records and {:.1f} GB of pairs'.format(t_cut, t_combine, t_add, t_cut + t_combine + t_add, sent_records * recw * 8 / 1e9, sent_pairs * 16 / 1e9), flush=True)
    if os.environ.get('CFG4_PROF'):
        import ctypes
        lib.kv_prof_enable(0)
        buf = ctypes.create_string_buffer(8192)
        lib.kv_prof_names(buf, 8192)
        for name in buf.value.decode().split(','):
            ms, nl = ctypes.c_double(), ctypes.c_uint64()
            lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(nl))
            print('    {:24s} {:10.1f} ms {:6d} launches (all owners, all shards)'.format(name, ms.value, nl.value))
    hk.scratch_trim()
    # band 0 the banded way
    per_batch = 18_750_000
    for si, n in enumerate(names):
        ref = hk.Counttable(k, memory / world / T, T)
        for lo in range(0, n_reads, per_batch):
            ref.consume_batch(hk.ReadBatch.generate(genome_len, seed, si, lo, min(per_batch, n_reads - lo), L), world, 0)
        for t in range(T):
            assert ref.table_bytes(t) == band0[n].table_bytes(t), (n, t)
        del ref
        print('{}: band 0 of the exchange layout equals the banded count, table for table'.format(n), flush=True)


if __name__ == '__main__':
    main()
