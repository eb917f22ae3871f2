"""What ONE rank of an N-GPU exchange run computes per step of config 2, measured on this one GPU.

Rank 0's work is replayed exactly: it routes its own shard of every sample (timed), and it counts / scans what
the N shards send to band 0 -- those items are produced here by routing every shard and keeping destination 0's
block (not timed: on the real node the other ranks do that, at the same time).  The exchange itself is not
measured (no second GPU); its volume per rank is printed.  usage: exchange_rank_cost.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from kevlar_amd import _lib, khmer as hk, shardrun, synth


def main():
    worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
    lib = _lib.load()
    _lib.require_device()
    k, L, mem, T = 31, 100, 2e9, 4
    packed = synth.trio_reads_packed(25_000_000, 30, L)
    names = tuple(packed)
    n_reads = packed['proband'].shape[0]
    dev = torch.device('cuda', 0)
    modes = [m for m in os.environ.get('RANK_COST_MODES', 'minimizer,distinct,plain').split(',') if m]
    owner_scan = os.environ.get('RANK_COST_SCAN', 'owner') == 'owner'      # minimizer layout: the bucket owners answer the scan (shard: every rank hashes its shard again)
    for mode_name in modes:
        distinct = mode_name in ('distinct', 'minimizer')
        minimizer = mode_name == 'minimizer'
        for world in worlds:
            sk = {n: hk.Counttable(k, mem / world / T, T) for n in names}
            shards = {n: [hk.ReadBatch.from_packed(packed[n][lo:hi], L) for lo, hi in
                          (shardrun.shard_bounds(n_reads, world, r) for r in range(world))] for n in names}
            nkm = shards['proband'][0].num_kmers(k)
            words = 2
            send = torch.empty((nkm + nkm // 4 + (1 << 20), 2), dtype=torch.int64, device=dev)
            recv_count, recv_tagged, sent = {}, None, 0

            def route(batch, base, mode):
                if mode == 'distinct':
                    return hk.route_distinct(batch, hk.Counttable, k, world, send.data_ptr(), send.shape[0])
                return hk.route_hashes(batch, hk.Counttable, k, world, base, mode == 'tagged', send.data_ptr(), send.shape[0])

            # what band 0 receives (other ranks' routing: not timed)
            order = list(names[1:]) + [names[0]] if distinct else list(names)
            mex_recv0 = {}
            recv_travel = {}
            short_pairs = os.environ.get('KV_MEX_PAIRS', '16') == '9'
            if minimizer:
                # every shard cut into records; bucket owner d combines what the N shards hold of its buckets; band 0 receives
                # the owners' pairs of band 0.  Rank 0's own part (its shard's emit, its buckets' combine) is timed below.
                # (16-byte records for the samples nobody scans from their records: RANK_COST_SHORT=0 keeps 24 bytes everywhere)
                want_short = os.environ.get('RANK_COST_SHORT', '1') != '0'
                plans = {n: hk.mex_plan(hk.Counttable, k, n_reads, L, world, short=want_short and not (owner_scan and n == 'proband')) for n in names}
                for n in names:
                    plan = plans[n]
                    per_bucket = int(plan.nwg1) * int(plan.cap1) * int(plan.recw)
                    segs, cnts = [], []
                    for r in range(world):
                        lo, _ = shardrun.shard_bounds(n_reads, world, r)
                        seg = torch.empty(int(plan.seg_words), dtype=torch.int64, device=dev)
                        cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=dev)
                        hk.mex_emit(shards[n][r], plan, lo, seg.data_ptr(), cnt.data_ptr())
                        segs.append(seg); cnts.append(cnt)
                    blocks = []
                    for d in range(world):
                        c0, c1 = int(plan.c_lo[d]), int(plan.c_lo[d + 1])
                        rs = torch.cat([sg_[c0 * per_bucket:c1 * per_bucket] for sg_ in segs])
                        rc = torch.cat([cn[c0 * int(plan.nwg1):c1 * int(plan.nwg1)] for cn in cnts])
                        if d == 0:
                            mex_recv0[n] = (rs, rc)
                        c, _ = hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, send.data_ptr(), send.shape[0])
                        blocks.append(send.view(-1)[:c[0] * 2].clone().view(-1, 2))
                    recv_count[n] = torch.cat(blocks)
                    # (as they arrive: every owner's block in the 9-byte travelling form, kv_pairs_pack)
                    tw, ww = [], []
                    for blk in blocks:
                        nb = int(blk.shape[0])
                        tbuf = torch.empty(nb + nb // 8 + 16, dtype=torch.int64, device=dev)
                        w = hk.pairs_pack(blk.data_ptr() if nb else 0, [nb], tbuf.data_ptr(), tbuf.shape[0])
                        tw.append(tbuf[:w[0]]); ww.append(w[0])
                    recv_travel[n] = (torch.cat(tw), ww)
                    del segs, cnts
                big = max(int(p_.seg_words) for p_ in plans.values())
                my_seg = torch.empty(big, dtype=torch.int64, device=dev)
                my_cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=dev)
                my_packed = torch.empty(big, dtype=torch.int64, device=dev)
                travel = torch.empty(send.shape[0] + send.shape[0] // 8 + 64, dtype=torch.int64, device=dev)
                unpacked = torch.empty((max(int(v.shape[0]) for v in recv_count.values()) + 8, 2), dtype=torch.int64, device=dev)
            for n in ([] if minimizer else names):
                blocks = []
                for r in range(world):
                    c = route(shards[n][r], 0, 'distinct' if distinct else 'plain')
                    w = 2 if distinct else 1
                    blocks.append(send.view(-1)[:c[0] * w].clone().view(-1, w))
                recv_count[n] = torch.cat(blocks)
            if not distinct:
                blocks = []
                for r in range(world):
                    lo, _ = shardrun.shard_bounds(n_reads, world, r)
                    c = route(shards['proband'][r], lo, 'tagged')
                    blocks.append(send[:c[0]].clone())
                recv_tagged = torch.cat(blocks)
            else:
                # the interesting k-mers of the OTHER bands (their owners' work): evaluated here against full-size sketches
                full = {n: hk.Counttable(k, mem / T, T) for n in names}
                whole = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
                for n in names:
                    full[n].consume_batch(whole[n])
                fr, fo, fa, _ = hk.novel_scan([full['proband']], [full[n] for n in names[1:]], whole['proband'], 6, 1)
                fh = full['proband'].hash_positions(whole['proband'], fr, fo)
                fh, first = np.unique(fh, return_index=True)
                bs = (2 ** 64 - 1) // world
                other = fh >= np.uint64(bs)             # band 0's own come from this rank's scan below
                others_hash = torch.from_numpy(fh[other].view(np.int64)).to(dev)
                others_abund = torch.from_numpy(np.ascontiguousarray(np.asarray(fa)[first][other])).to(dev)
                del full, whole
                torch.cuda.empty_cache()
            torch.cuda.synchronize()
            best = None
            for rep in range(4):
                lib.kv_prof_reset()
                lib.kv_prof_enable(1 if os.environ.get('RANK_COST_PROF') else 0)
                for n in names:
                    sk[n].clear()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                t_route = t_count = 0.0
                out_bytes = 0
                for n in order:
                    ta = time.perf_counter()
                    if minimizer:
                        plan = plans[n]
                        per_dest, fitted = hk.mex_emit_pack(shards[n][0], plan, 0, my_seg.data_ptr(), my_cnt.data_ptr(), my_packed.data_ptr(), my_packed.shape[0])
                        assert fitted
                        rs, rc = mex_recv0[n]
                        c, _ = hk.mex_route(plan, 0, rs.data_ptr(), rc.data_ptr(), world, send.data_ptr(), send.shape[0], keep_scan=(owner_scan and n == 'proband'))
                        if short_pairs:
                            wc = hk.pairs_pack(send.data_ptr(), c, travel.data_ptr(), travel.shape[0])
                            out_bytes += (sum(wc) - wc[0]) * 8
                        else:
                            out_bytes += (sum(c) - c[0]) * 16
                        out_bytes += (sum(per_dest) - per_dest[0]) * int(plan.recw) * 8 + int(plan.cnt_entries) * 4 * (world - 1) // world
                    elif distinct:
                        c = route(shards[n][0], 0, 'distinct')
                        out_bytes += (sum(c) - c[0]) * 16
                    elif n == 'proband':
                        c = route(shards[n][0], 0, 'tagged')
                        out_bytes += (sum(c) - c[0]) * 16
                    else:
                        c = route(shards[n][0], 0, 'plain')
                        out_bytes += (sum(c) - c[0]) * 8
                    tb = time.perf_counter()
                    if distinct and minimizer and short_pairs:
                        tbuf, ww = recv_travel[n]
                        per_src, _ = hk.pairs_unpack(tbuf.data_ptr(), ww, unpacked.data_ptr(), unpacked.shape[0])
                        sk[n].consume_hashes_weighted(unpacked.data_ptr(), sum(per_src))
                    elif distinct:
                        items = recv_count[n]
                        sk[n].consume_hashes_weighted(items.data_ptr(), items.shape[0])
                    elif n == 'proband':
                        sk[n].consume_hashes(recv_tagged.data_ptr(), recv_tagged.shape[0], 2)
                    else:
                        items = recv_count[n]
                        sk[n].consume_hashes(items.data_ptr(), items.shape[0], 1)
                    tc = time.perf_counter()
                    t_route += tb - ta
                    t_count += tc - tb
                t1 = time.perf_counter()
                if distinct:
                    items = recv_count['proband']
                    cap = items.shape[0]
                    tags = torch.empty(cap, dtype=torch.int64, device=dev)
                    abund = torch.empty((cap, 3), dtype=torch.uint8, device=dev)
                    nh = hk.novel_scan_distinct([sk['proband']], [sk[n] for n in names[1:]], items.data_ptr(), cap, 6, 1,
                                                tags.data_ptr(), abund.data_ptr(), cap)
                    out_bytes += nh * 11 * (world - 1)
                    set_h = torch.cat([tags[:nh], others_hash])
                    set_a = torch.cat([abund[:nh], others_abund])
                    torch.cuda.synchronize()
                    if minimizer and owner_scan:
                        # this rank's minimizer buckets against the gathered set (kv_mex_scan_set); the sort of all ranks' hits that follows
                        # the all-gather is the same work in either layout of the scan and is not timed in either
                        hit_cap = max(1 << 16, 64 * int(set_h.shape[0]) // world)
                        htags = torch.empty(hit_cap, dtype=torch.int64, device=dev)
                        hrows = torch.empty((hit_cap, 3), dtype=torch.uint8, device=dev)
                        nh = hk.mex_scan_set(hk.Counttable, k, 3, set_h.data_ptr(), set_a.data_ptr(), set_h.shape[0], htags.data_ptr(), hrows.data_ptr(), hit_cap)
                    else:
                        r, o, a = hk.novel_scan_set(shards['proband'][0], hk.Counttable, k, 3, set_h.data_ptr(), set_a.data_ptr(), set_h.shape[0])
                        nh = len(r)
                else:
                    cap = recv_tagged.shape[0]
                    tags = torch.empty(min(cap, 1 << 26), dtype=torch.int64, device=dev)
                    abund = torch.empty((min(cap, 1 << 26), 3), dtype=torch.uint8, device=dev)
                    nh = hk.novel_scan_hashes([sk['proband']], [sk[n] for n in names[1:]], recv_tagged.data_ptr(), cap, 6, 1,
                                              tags.data_ptr(), abund.data_ptr(), tags.shape[0])
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                res = dict(route=t_route * 1e3, count=t_count * 1e3, scan=(t2 - t1) * 1e3, total=(t2 - t0) * 1e3, out_mb=out_bytes / 1e6,
                           items=sum(int(v.shape[0]) for v in recv_count.values()), hits=nh)
                if os.environ.get('RANK_COST_PROF') and rep == 3:
                    import ctypes
                    buf = ctypes.create_string_buffer(8192)
                    lib.kv_prof_names(buf, 8192)
                    for name in buf.value.decode().split(','):
                        ms, nl = ctypes.c_double(), ctypes.c_uint64()
                        lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(nl))
                        print('    {:24s} {:8.3f} ms {:4d} launches'.format(name, ms.value, nl.value))
                if best is None or res['total'] < best['total']:
                    best = res
            print('N={} items={}: per-rank route {route:.2f} ms, count {count:.2f} ms, scan {scan:.2f} ms, total {total:.2f} ms; '
                  'sends {out_mb:.0f} MB; counts {items} items; {hits} hits (set: this shard; plain: this band)'.format(world, mode_name, **best), flush=True)
            del sk, shards, recv_count, recv_tagged, send, mex_recv0, recv_travel
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
