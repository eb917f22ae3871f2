"""BASELINE.json configs[3] (3 Gb trio, 30x, k = 31, 8 GPUs) in the EXCHANGE layout, replayed on one GPU at true size: what one rank of
eight computes, and -- by playing every bucket owner in turn -- band 0's sketches, which must equal the banded count's.

Per sample: every shard (1/8 of the reads, generated on the device) is cut into 16-byte super-k-mer records under the sample's plan
(kv_mex_emit_pack); what each shard holds of an owner's minimizer buckets is set aside as that owner's received records; every owner
combines its buckets at the sample's full coverage (kv_mex_route -- in passes, the buckets of a 63 G-k-mer sample being eight times the
LDS table: SkmGeom::passes) and hands band 0 its (hash, occurrences) pairs, which band 0's sketch adds (kv_consume_hashes_weighted).
Timed: rank 0's own cut, owner 0's combine, and band 0 adding as many pairs as it receives from all owners.  Checked: band 0's tables
against a banded count of all reads (kv_consume with band 0 of 8: what bench.py --workload cfg4-band counts).

CFG4_SCAN=1 adds the SCAN of that layout at the same size (round 6; kevlar/novel.py:123-169 answered per band, then
kevlar/unband.py:41-77): band 0 judges the distinct case k-mers it received (kv_novel_scan_distinct over every owner's pairs of
band 0), and then every bucket owner in turn -- its buckets combined again with keep_scan -- answers for the occurrences in its
buckets against the gathered set (kv_mex_scan_set).  The set at hand is band 0's interesting hashes; it is padded with seven shifted
copies (hashes of the other bands' ranges that no k-mer has) so that set size, probe cost and gathered bytes are those of eight
bands.  Timed: band 0's judging and owner 0's answer.  Checked: the eight owners' hits together are the hits of the banded scan of
band 0 over all reads (bench.py --workload cfg4-band: 19575611 of them at 3 Gb), read for read, offset for offset, abundances included.

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


def scan_leg(lib, dev, band0, names, case_items, case_n, k, L, T, genome_len, seed, n_reads, world, bounds, pairs_cap):
    """the scan of the exchange layout at this size: see the module's text (CFG4_SCAN=1); returns the owners' hits (read, offset, abund) sorted"""
    S = len(names)
    cap = max(min(case_n, 1 << 26), 1)
    hashes = torch.empty(cap, dtype=torch.int64, device=dev)
    abund = torch.empty((cap, S), dtype=torch.uint8, device=dev)
    t_judge = None
    for rep in range(2):                                    # (the second call: buffers grown, kernels loaded)
        lib.kv_synchronize()
        t0 = time.perf_counter()
        n_mine = hk.novel_scan_distinct([band0['proband']], [band0['mother'], band0['father']], case_items.data_ptr(), case_n, 6, 1,
                                        hashes.data_ptr(), abund.data_ptr(), cap)
        lib.kv_synchronize()
        t_judge = time.perf_counter() - t0
    del case_items
    torch.cuda.empty_cache()
    # the gathered set: band 0's interesting hashes + the same number in every other band's range
    bs = (2 ** 64 - 1) // world
    mine_h, mine_a = hashes[:n_mine].clone(), abund[:n_mine].clone()
    assert bool(((mine_h.cpu().numpy().view(np.uint64)) < np.uint64(bs)).all()), 'band 0 judged a hash outside its range'
    parts = [mine_h]
    for b in range(1, world):
        shift = (b * bs) & (2 ** 64 - 1)
        parts.append(mine_h + (shift - 2 ** 64 if shift >= 2 ** 63 else shift))     # (64-bit wrap-around: a hash of band b's range)
    set_h = torch.cat(parts)
    set_a = mine_a.repeat(world, 1)
    n_set = int(set_h.shape[0])
    del hashes, abund
    print('scan: band 0 judged {} distinct case k-mers in {:.3f} s: {} interesting; the gathered set holds {} ({:.1f} MB arrive at every rank)'.format(
        case_n, t_judge, n_mine, n_set, (n_set - n_mine) * (8 + S) / 1e6), flush=True)
    # every owner in turn: its buckets combined again (keep_scan), then its answer
    plan = hk.mex_plan(hk.Counttable, k, n_reads, L, world, short=False)      # records with read positions: the sample the scan is answered from
    recw, nwg1 = int(plan.recw), int(plan.nwg1)
    print('scan: the case sample cut again with {}-byte records ({:.1f} GB per shard), combined by one owner at a time'.format(8 * recw, int(plan.seg_words) * 8 / 1e9), flush=True)
    hit_cap = max(1 << 16, 64 * n_set // world)
    found, t_owner, t_route, n_owner = [], [], [], []
    by_shard = False          # an owner could not answer: EVERY rank then looks its own shard up in the set (the ranks agree on that in one small
                              # all-gather, ShardedTrio.scan_minimizer) -- what owners found and what shards found do not add up to the hits
    d = -1
    while d + 1 < world:
        d += 1
        if by_shard:
            lo, hi = bounds[d]
            batch = hk.ReadBatch.generate(genome_len, seed, 0, lo, hi - lo, L)
            best = None
            for rep in range(2):
                lib.kv_synchronize()
                t0 = time.perf_counter()
                r, o, a = hk.novel_scan_set(batch, hk.Counttable, k, S, set_h.data_ptr(), set_a.data_ptr(), n_set)
                lib.kv_synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            found.append((np.asarray(r).astype(np.int64) + lo, np.asarray(o).copy(), np.asarray(a).copy()))
            t_owner.append(best); n_owner.append(len(r))
            del batch
            print('scan: rank {}: its shard of {} reads against the set {:.3f} s, {} hits ({:.1f} MB leave it)'.format(d, hi - lo, best, len(r), len(r) * (8 + S) / 1e6), flush=True)
            continue
        hk.scratch_trim()
        seg = torch.empty(int(plan.seg_words), dtype=torch.int64, device=dev)
        cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=dev)
        out = torch.empty(int(plan.seg_words) // 2 + 4096, dtype=torch.int64, device=dev)
        got_rec, got_cnt = [], []
        for r, (lo, hi) in enumerate(bounds):
            batch = hk.ReadBatch.generate(genome_len, seed, 0, lo, hi - lo, L)
            per_dest, fitted = hk.mex_emit_pack(batch, plan, lo, seg.data_ptr(), cnt.data_ptr(), out.data_ptr(), out.shape[0])
            assert fitted, 'the packed records of shard {} did not fit half the segments'.format(r)
            first = sum(per_dest[:d])
            got_rec.append(out[first * recw:(first + per_dest[d]) * recw].clone())
            got_cnt.append(cnt[int(plan.c_lo[d]) * nwg1:int(plan.c_lo[d + 1]) * nwg1].clone())
            del batch
        del seg, out, cnt
        torch.cuda.empty_cache()
        rs = torch.cat(got_rec); rc = torch.cat(got_cnt)
        del got_rec, got_cnt
        torch.cuda.empty_cache()
        pairs = torch.empty((pairs_cap, 2), dtype=torch.int64, device=dev)
        best = None
        for rep in range(2):                                # (the second call: the arenas a trim gave back have grown again)
            lib.kv_synchronize()
            t0 = time.perf_counter()
            hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, pairs.data_ptr(), pairs_cap, compact=True, keep_scan=True)
            lib.kv_synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        t_route.append(best)
        del pairs
        torch.cuda.empty_cache()
        tags = torch.empty(hit_cap, dtype=torch.int64, device=dev)
        rows = torch.empty((hit_cap, S), dtype=torch.uint8, device=dev)
        best = None
        try:
            if os.environ.get('CFG4_SCAN_FORCE_FALLBACK') and d == 1:
                raise _lib.KvCapacityError('forced by CFG4_SCAN_FORCE_FALLBACK')
            for rep in range(2):
                lib.kv_synchronize()
                t0 = time.perf_counter()
                n_hits = hk.mex_scan_set(hk.Counttable, k, S, set_h.data_ptr(), set_a.data_ptr(), n_set, tags.data_ptr(), rows.data_ptr(), hit_cap)
                lib.kv_synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            r, o, a = hk.hits_from_tagged(tags.data_ptr(), rows.data_ptr(), n_hits, n_hits, S)
            found.append((np.asarray(r).astype(np.int64), np.asarray(o).copy(), np.asarray(a).copy()))
        except _lib.KvCapacityError as exc:
            print('scan: owner {} cannot answer ({}): every rank scans its own shard against the set instead'.format(d, exc), flush=True)
            del tags, rows, rs, rc
            hk.scratch_trim()
            torch.cuda.empty_cache()
            by_shard, d = True, -1
            found, t_owner, n_owner = [], [], []
            continue
        t_owner.append(best); n_owner.append(n_hits)
        del tags, rows, rs, rc
        torch.cuda.empty_cache()
        print('scan: owner {}: buckets combined with their distinct list in {:.3f} s, answer {:.3f} s, {} hits ({:.1f} MB leave it)'.format(
            d, t_route[-1], best, n_hits, n_hits * (8 + S) / 1e6), flush=True)
    hk.scratch_trim()
    hr, ho, ha = np.concatenate([f[0] for f in found]), np.concatenate([f[1] for f in found]), np.concatenate([f[2] for f in found])
    order = np.lexsort((ho, hr))
    hr, ho, ha = hr[order], ho[order], ha[order]
    line = ('one rank\'s scan: judging its band\'s {} distinct k-mers {:.3f} s + ' + ('its shard against the set' if by_shard else 'answering for its buckets') +
            ' {:.3f} s (rank 0; the eight: {:.3f}-{:.3f}) = {:.3f} s; it receives {:.1f} MB of the set and sends {:.1f} MB of hits; {} hits in all')
    print(line.format(case_n, t_judge, t_owner[0], min(t_owner), max(t_owner), t_judge + t_owner[0], (n_set - n_mine) * (8 + S) / 1e6,
                      n_owner[0] * (8 + S) / 1e6 * (world - 1) / world, len(hr)), flush=True)
    return np.asarray(hr).copy(), np.asarray(ho).copy(), np.asarray(ha).copy()


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
    want_scan = os.environ.get('CFG4_SCAN') == '1'
    case_items, case_n = None, 0          # CFG4_SCAN: every (hash, occurrences) pair band 0 receives of the case sample, as its owner keeps them
    for si, n in enumerate(names):
        arrived_all = 0
        if want_scan and n == 'proband':
            case_items = torch.empty((pairs_cap, 2), dtype=torch.int64, device=dev)
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
                if want_scan and n == 'proband':
                    assert case_n + counts[0] <= case_items.shape[0], 'band 0 received more pairs of the case sample than CFG4_PAIRS_FRAC allows for'
                    case_items[case_n:case_n + counts[0]] = pairs[:counts[0]]
                    case_n += counts[0]
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
    owners_hits = None
    if want_scan:
        owners_hits = scan_leg(lib, dev, band0, names, case_items, case_n, k, L, T, genome_len, seed, n_reads, world, bounds, pairs_cap)
        case_items = None
        hk.scratch_trim()
    # band 0 the banded way
    per_batch = 18_750_000
    refs = {}
    for si, n in enumerate(names):
        ref = hk.Counttable(k, memory / world / T, T)
        for lo in range(0, n_reads, per_batch):
            ref.consume_batch(hk.ReadBatch.generate(genome_len, seed, si, lo, min(per_batch, n_reads - lo), L), world, 0)
        for t in range(T):
            assert ref.table_bytes(t) == band0[n].table_bytes(t), (n, t)
        if want_scan:
            refs[n] = ref
        del ref
        print('{}: band 0 of the exchange layout equals the banded count, table for table'.format(n), flush=True)
    if want_scan:
        # the banded scan of band 0 over all reads, batch by batch (what bench.py --workload cfg4-band scans), against the owners' hits
        rr, oo, aa = [], [], []
        lib.kv_synchronize()
        t0 = time.perf_counter()
        for lo in range(0, n_reads, per_batch):
            r, o, a, _ = hk.novel_scan([refs['proband']], [refs['mother'], refs['father']], hk.ReadBatch.generate(genome_len, seed, 0, lo, min(per_batch, n_reads - lo), L),
                                       6, 1, band_mode=1, nbands=world, band=0)
            rr.append(np.asarray(r).astype(np.int64) + lo); oo.append(np.asarray(o).copy()); aa.append(np.asarray(a).copy())
        t_banded = time.perf_counter() - t0
        rr, oo, aa = np.concatenate(rr), np.concatenate(oo), np.concatenate(aa)
        hr, ho, ha = owners_hits
        assert len(hr) == len(rr), ('hits', len(hr), len(rr))
        assert np.array_equal(np.asarray(hr).astype(np.int64), rr) and np.array_equal(ho, oo) and np.array_equal(ha, aa), 'the owners\' hits differ from the banded scan of band 0'
        print('scan: the {} owners\' hits together are the {} hits of the banded scan of band 0 over all reads ({:.2f} s with the reads generated batch by batch), '
              'read for read, offset for offset, abundances included'.format(world, len(rr), t_banded), flush=True)


if __name__ == '__main__':
    main()
