"""Randomised parity of the minimizer-sharded exchange's count side, all ranks emulated in one process: for random k, read length, number
of ranks and shard sizes, every shard is cut into super-k-mer records (kv_mex_emit_pack) under a classic plan and under a short-record
plan (kv_mex_plan_short, where the shape has one), every owner combines what the shards hold of its buckets (kv_mex_route; from the
segments as cut or from their packed form, as it travels), every band owner adds the pairs it is sent (kv_consume_hashes_weighted) --
and band b's sketch must equal band b of a banded count of all reads, either way.  The scan side (round 6): under the classic plan every
owner combines its buckets again with keep_scan and answers for them against a random set of the sample's k-mer hashes (kv_mex_scan_set)
-- with the distinct list in stretches per workgroup or as the pool of chunks (KV_MEX_DL_POOL), the buckets in one pass or several
(KV_MEX_PASSES) -- and the owners' hits together must be the hits of the whole sample against that set (kv_novel_scan_set), abundances
included.  python scratch/fuzz_mex.py [trials] [seed]     (KV_TUNING=1: the switches above are tuning switches)"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch
from kevlar_amd import _lib, khmer as hk, shardrun, synth
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
dev = torch.device('cuda', 0)
fails = 0
for trial in range(trials):
    k = int(rng.choice([31, 31, 31, 25, 33, 51]))
    world = int(rng.choice([1, 2, 3, 5, 8]))
    L = int(rng.choice([60, 100, 100, 151, 250]))
    n = int(rng.choice([3000, 20000, 90000, 5]))              # (5: ranks without reads when world = 8)
    text = bool(rng.random() < 0.5)                     # shards handed over as text (with an N somewhere) or as packed words
    mem = float(rng.choice([4e5, 4e6]))
    knobs = {}
    if rng.random() < 0.5:
        knobs['KV_MEX_DL_POOL'] = '1'
    if rng.random() < 0.5:
        knobs['KV_MEX_PASSES'] = str(rng.choice(['2', '3', '4', '7']))
    desc = 'trial {} k={} world={} n={} L={} {} {}'.format(trial, k, world, n, L, 'text' if text else 'packed', knobs)
    for name in ('KV_MEX_DL_POOL', 'KV_MEX_PASSES'):
        os.environ.pop(name, None)
    os.environ.update(knobs)
    try:
        trio = synth.make_trio(int(rng.choice([8000, 60000])), int(rng.integers(0, 1 << 30)))
        packed = synth.sample_reads_packed(trio['mother'], n, L, 0.005, int(rng.integers(0, 1 << 30)))
        cuts = [shardrun.shard_bounds(n, world, r) for r in range(world)]
        if text:
            seqs = synth.unpack_reads(packed, L)
            j = int(rng.integers(0, n)); seqs[j] = seqs[j][:L // 2] + 'N' + seqs[j][L // 2 + 1:]
            shards = [hk.ReadBatch(seqs[lo:hi]) for lo, hi in cuts]
            whole = hk.ReadBatch(seqs)
        else:
            shards = [hk.ReadBatch.from_packed(packed[lo:hi], L) for lo, hi in cuts]
            whole = hk.ReadBatch.from_packed(packed, L)
        banded = [hk.Counttable(k, mem / world / 4, 4) for _ in range(world)]
        for b in range(world):
            banded[b].consume_batch(whole, world, b)
        nk_all = whole.num_kmers(k)
        for short in (False, True):
            plan = hk.mex_plan(hk.Counttable, k, n, L, world, short=short)
            is_short = bool(int(plan.flags) & 1)
            if short and not is_short:
                continue
            recw, nwg1 = int(plan.recw), int(plan.nwg1)
            from_packed_form = bool(rng.random() < 0.5)
            segs, cnts, outs, per_dests = [], [], [], []
            for r, (lo, hi) in enumerate(cuts):
                seg = torch.empty(int(plan.seg_words), dtype=torch.int64, device=dev)
                cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=dev)
                out = torch.empty(int(plan.seg_words), dtype=torch.int64, device=dev)
                per_dest, fitted = hk.mex_emit_pack(shards[r], plan, lo, seg.data_ptr(), cnt.data_ptr(), out.data_ptr(), out.shape[0])
                assert fitted
                segs.append(seg); cnts.append(cnt); outs.append(out); per_dests.append(per_dest)
            got = [hk.Counttable(k, mem / world / 4, 4) for _ in range(world)]
            pairs_buf = torch.empty((nk_all + 1024, 2), dtype=torch.int64, device=dev)
            arrived_all = 0
            per_bucket = nwg1 * int(plan.cap1) * recw
            sampled, received = [], []
            for d in range(world):
                c0, c1 = int(plan.c_lo[d]), int(plan.c_lo[d + 1])
                rc = torch.cat([cn[c0 * nwg1:c1 * nwg1] for cn in cnts])
                if from_packed_form:
                    parts = []
                    for r in range(world):
                        first = sum(per_dests[r][:d]) * recw
                        parts.append(outs[r][first:first + per_dests[r][d] * recw])
                    rs = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=dev)
                    if rs.numel() == 0:
                        rs = torch.zeros(8, dtype=torch.int64, device=dev)
                else:
                    rs = torch.cat([sg[c0 * per_bucket:c1 * per_bucket] for sg in segs])
                counts, arrived = hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, pairs_buf.data_ptr(), pairs_buf.shape[0], compact=from_packed_form)
                arrived_all += arrived
                if not is_short:
                    received.append((rs, rc))
                    n_pairs = sum(counts)
                    if n_pairs:
                        take = torch.from_numpy(rng.integers(0, n_pairs, size=max(1, n_pairs // 7))).to(dev)
                        sampled.append(pairs_buf[:n_pairs, 0][take].clone())
                off = 0
                for b in range(world):
                    if counts[b]:
                        got[b].consume_hashes_weighted(pairs_buf.data_ptr() + off * 16, counts[b])
                    off += counts[b]
            assert arrived_all == nk_all, (arrived_all, nk_all)
            for b in range(world):
                for t in range(4):
                    assert got[b].table_bytes(t) == banded[b].table_bytes(t), (short, from_packed_form, b, t)
            if not is_short and sampled and nk_all:
                # the scan side: the owners' answers against a set of the sample's own hashes
                S = 3
                set_h = torch.unique(torch.cat(sampled))
                set_h = set_h[set_h != -1]                  # (~0 is the set's padding value)
                n_set = int(set_h.shape[0])
                set_a = torch.from_numpy(rng.integers(0, 256, size=(max(n_set, 1), S), dtype=np.uint8)).to(dev)
                wr, wo, wa = hk.novel_scan_set(whole, hk.Counttable, k, S, set_h.data_ptr(), set_a.data_ptr(), n_set)
                tags = torch.empty(nk_all + 1024, dtype=torch.int64, device=dev)
                rows = torch.empty((nk_all + 1024, S), dtype=torch.uint8, device=dev)
                found, answered = [], True
                for d in range(world):
                    rs, rc = received[d]
                    hk.mex_route(plan, d, rs.data_ptr(), rc.data_ptr(), world, pairs_buf.data_ptr(), pairs_buf.shape[0], compact=from_packed_form, keep_scan=True)
                    try:
                        n_hits = hk.mex_scan_set(hk.Counttable, k, S, set_h.data_ptr(), set_a.data_ptr(), n_set, tags.data_ptr(), rows.data_ptr(), tags.shape[0])
                    except _lib.KvCapacityError as exc:
                        answered = False                    # (an owner that cannot answer: every rank would scan its shard -- nothing to compare here)
                        desc += ' [owner {} cannot answer: {}]'.format(d, str(exc)[:60])
                        break
                    r, o, a = hk.hits_from_tagged(tags.data_ptr(), rows.data_ptr(), n_hits, n_hits, S)
                    found.append((np.asarray(r).astype(np.int64), np.asarray(o).copy(), np.asarray(a).copy().reshape(-1, S)))
                if answered:
                    hr = np.concatenate([f[0] for f in found]); ho = np.concatenate([f[1] for f in found]); ha = np.concatenate([f[2] for f in found])
                    keep = ~np.isin(hr, whole.flagged_reads())           # (the scan skips reads with a byte outside ACGT; their k-mers were counted)
                    hr, ho, ha = hr[keep], ho[keep], ha[keep]
                    order = np.lexsort((ho, hr))
                    hr, ho, ha = hr[order], ho[order], ha[order]
                    assert len(hr) == len(wr), ('scan hits', len(hr), len(wr), knobs)
                    assert np.array_equal(hr, np.asarray(wr).astype(np.int64)) and np.array_equal(ho, wo) and np.array_equal(ha, np.asarray(wa).reshape(-1, S)), ('scan hits differ', knobs)
                    desc += ' [{} hits of {} set members]'.format(len(hr), n_set)
        print('ok   ' + desc, flush=True)
    except Exception as exc:
        fails += 1
        print('FAIL ' + desc + ': ' + repr(exc)[:300], flush=True)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
