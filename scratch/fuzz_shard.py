"""Randomised parity of the read-sharded multi-GPU primitives, all ranks emulated in one process: for random k, sketch
kind, number of ranks, shard sizes and item format, the sketches a band owner builds from what the shards route to it
must equal band b of a banded count of all reads, and the hits of the exchange scans (tagged items, or distinct pairs
+ set lookup) the merged banded scan.  python scratch/fuzz_shard.py [trials] [seed]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import torch
from kevlar_amd import _lib, khmer as hk, shardrun, synth
_lib.load(); _lib.require_device()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
dev = torch.device('cuda', 0)
fails = 0
for trial in range(trials):
    kind = str(rng.choice(['Counttable', 'Counttable', 'SmallCounttable', 'Nodetable', 'Countgraph']))
    k = int(rng.integers(12, 33)) if kind == 'Countgraph' else int(rng.choice([16, 21, 31, 32, 33, 51, 64, 70]))
    world = int(rng.choice([1, 2, 3, 5, 8]))
    n = int(rng.choice([600, 9000, 40000])); L = int(rng.choice([100, 151]))
    distinct = bool(rng.random() < 0.5)
    path = str(rng.choice(['skm', 'plain'])) if distinct else None
    cls = getattr(hk, kind)
    mem = float(rng.choice([4e5, 4e6]))
    desc = 'trial {} {} k={} world={} n={} L={} items={} {}'.format(trial, kind, k, world, n, L, 'distinct' if distinct else 'plain', path)
    env = {'KV_ROUTE_PATH': path, 'KV_NOVEL_PATH': 'skm' if path == 'skm' else 'tiles'} if path else {}
    os.environ.update(env)
    try:
        trio = synth.make_trio(int(rng.choice([5000, 60000])), int(rng.integers(0, 1 << 30)))
        names = ('proband', 'mother', 'father')
        reads = {}
        for name in names:
            seqs = synth.unpack_reads(synth.sample_reads_packed(trio[name], n, L, 0.005, int(rng.integers(0, 1 << 30))), L)
            j = int(rng.integers(0, n)); seqs[j] = seqs[j][:L // 2] + 'N' + seqs[j][L // 2 + 1:]
            reads[name] = seqs
        cuts = {name: [shardrun.shard_bounds(n, world, r) for r in range(world)] for name in names}
        sharded = {name: [cls(k, mem / world / 4, 4) for _ in range(world)] for name in names}        # [band]
        case_items = [[] for _ in range(world)]
        shard_batches = [hk.ReadBatch(reads['proband'][lo:hi]) for lo, hi in cuts['proband']]
        for name in names:
            for r, (lo, hi) in enumerate(cuts[name]):
                batch = shard_batches[r] if name == 'proband' else hk.ReadBatch(reads[name][lo:hi])
                nk = max(batch.num_kmers(k), 1)
                send = torch.zeros((nk, 2), dtype=torch.int64, device=dev)
                tagged = name == 'proband' and not distinct
                if distinct:
                    counts = hk.route_distinct(batch, cls, k, world, send.data_ptr(), nk)
                else:
                    send = torch.zeros((nk, 2 if tagged else 1), dtype=torch.int64, device=dev)
                    counts = hk.route_hashes(batch, cls, k, world, lo, tagged, send.data_ptr(), nk)
                starts = np.concatenate(([0], np.cumsum(counts)))
                for b in range(world):
                    block = send[int(starts[b]):int(starts[b + 1])]
                    if counts[b]:
                        if distinct:
                            sharded[name][b].consume_hashes_weighted(block.data_ptr(), counts[b])
                        else:
                            sharded[name][b].consume_hashes(block.data_ptr(), counts[b], block.shape[1])
                    if name == 'proband':
                        case_items[b].append(block.clone())
        full = {name: hk.ReadBatch(reads[name]) for name in names}
        want_r, want_o, want_a = [], [], []
        hit_sets_h, hit_sets_a, tagged_hits = [], [], []
        for b in range(world):
            banded = {name: cls(k, mem / world / 4, 4) for name in names}
            for name in names:
                banded[name].consume_batch(full[name], world, b)
                for t in range(4):
                    assert banded[name].table_bytes(t) == sharded[name][b].table_bytes(t), (desc, name, 'band', b, 'table', t)
            r, o, a, _ = hk.novel_scan([banded['proband']], [banded['mother'], banded['father']], full['proband'], 5, 1, band_mode=1, nbands=world, band=b)
            want_r.append(np.array(r)); want_o.append(np.array(o)); want_a.append(np.array(a))
            items = torch.cat(case_items[b]) if case_items[b] else torch.zeros((0, 2), dtype=torch.int64, device=dev)
            cap = max(items.shape[0], 1)
            out_h = torch.empty(cap, dtype=torch.int64, device=dev); out_a = torch.empty((cap, 3), dtype=torch.uint8, device=dev)
            cases, ctrls = [sharded['proband'][b]], [sharded['mother'][b], sharded['father'][b]]
            if distinct:
                nh = hk.novel_scan_distinct(cases, ctrls, items.data_ptr(), items.shape[0], 5, 1, out_h.data_ptr(), out_a.data_ptr(), cap) if items.shape[0] else 0
                hit_sets_h.append(out_h[:nh].clone()); hit_sets_a.append(out_a[:nh].clone())
            else:
                nh = hk.novel_scan_hashes(cases, ctrls, items.data_ptr(), items.shape[0], 5, 1, out_h.data_ptr(), out_a.data_ptr(), cap) if items.shape[0] else 0
                tagged_hits.append((out_h[:nh].clone(), out_a[:nh].clone()))
        want = np.lexsort((np.concatenate(want_o), np.concatenate(want_r)))
        wr, wo, wa = np.concatenate(want_r)[want], np.concatenate(want_o)[want], np.concatenate(want_a)[want]
        if distinct:
            set_h, set_a = torch.cat(hit_sets_h), torch.cat(hit_sets_a)
            torch.cuda.synchronize()
            got_r, got_o, got_a = [], [], []
            for r, (lo, hi) in enumerate(cuts['proband']):
                rr, oo, aa = hk.novel_scan_set(shard_batches[r], cls, k, 3, set_h.data_ptr() if len(set_h) else 0, set_a.data_ptr() if len(set_h) else 0, len(set_h))
                got_r.append(np.array(rr, dtype=np.int64) + lo); got_o.append(np.array(oo)); got_a.append(np.array(aa))
            gr, go, ga = np.concatenate(got_r), np.concatenate(got_o), np.concatenate(got_a)
        else:
            tags = torch.cat([t for t, _ in tagged_hits]); ab = torch.cat([a for _, a in tagged_hits])
            torch.cuda.synchronize()
            gr, go, ga = hk.hits_from_tagged(tags.data_ptr(), ab.data_ptr(), tags.shape[0], tags.shape[0], 3) if tags.shape[0] else (np.zeros(0), np.zeros(0), np.zeros((0, 3)))
        assert np.array_equal(gr, wr) and np.array_equal(go, wo) and np.array_equal(ga, wa), (desc, 'hits', len(gr), len(wr))
        print('ok  ', desc, len(wr), 'hits', flush=True)
    except Exception as exc:
        fails += 1
        print('FAIL', desc, repr(exc)[:400], flush=True)
    finally:
        for key in env: os.environ.pop(key, None)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
