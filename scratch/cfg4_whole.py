"""BASELINE.json configs[3] -- the 3 Gb trio, 30x, k = 31, EIGHT k-mer bands -- as a whole job on ONE GPU, every k-mer hashed once.

`bench.py --workload cfg4-band` times what one of the eight GPUs does (its band of every sample: all 189 G k-mers hashed to keep an eighth);
eight such passes are the whole job on one GPU (8 x 3.25 s = 26 s, 105 M reads/s).  Here the eight bands' sketches are resident together
(8 x 3 x 8 GB = 192 GB of the 288) and a batch is hashed ONCE: kv_route_hashes writes its hashes grouped by band, every band's sketch adds
its group (kv_consume_hashes); the scan does the same with (hash, tag) pairs and kv_novel_scan_hashes.  The reads are generated batch by
batch on the device (kv_reads_generate) -- they cannot stay resident beside 192 GB of tables -- and that is inside the time reported.

    gpurun -- python scratch/cfg4_whole.py [genome_mb] [bands]        (defaults 3000, 8; 250 + 8 is the quick check)

Prints seconds per phase, whole-job reads/s, and per band the hit checksum bench.py prints for that band (band 0 of the true size:
19575611:b2768c24ac4eafe6)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__
__graft_entry__.build_product()
from kevlar_amd import _lib, khmer as hk


def checksum(r, o, a):
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(r, dtype='<u4').tobytes())
    h.update(np.ascontiguousarray(o, dtype='<u4').tobytes())
    h.update(np.ascontiguousarray(a, dtype=np.uint8).tobytes())
    return '{}:{}'.format(len(r), h.hexdigest()[:16])


def main():
    genome_mb = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    k, L, T, cov, seed = 31, 100, 4, 30.0, 42
    memory = 64e9 * genome_mb / 3000.0
    per_batch = 18_750_000
    genome_len = int(genome_mb * 1e6)
    n_reads = int(genome_len * cov / L)
    nk = L - k + 1
    names = ('proband', 'mother', 'father')
    streams = int(os.environ.get('CFG4_STREAMS', '3'))
    dev = torch.device('cuda', 0)
    lib = _lib.load()
    t0 = time.time()
    sk = {n: [hk.Counttable(k, memory / B / T, T) for _ in range(B)] for n in names}
    lib.kv_synchronize()
    print('{} Mb, {} reads per sample, {} bands, {:.1f} GB of sketches ({:.1f} s to make)'.format(genome_mb, n_reads, B, 3 * memory / 1e9, time.time() - t0), flush=True)
    cap = per_batch * nk
    firsts = list(range(0, n_reads, per_batch))
    t_gen = [0.0]

    def count_sample(si, n):
        send = torch.empty(cap, dtype=torch.int64, device=dev)
        added = 0
        for lo in firsts:
            ta = time.perf_counter()
            batch = hk.ReadBatch.generate(genome_len, seed, si, lo, min(per_batch, n_reads - lo), L)
            t_gen[0] += time.perf_counter() - ta
            counts = hk.route_hashes(batch, hk.Counttable, k, B, lo, False, send.data_ptr(), cap)
            off = 0
            for b in range(B):
                if counts[b]:
                    added += sk[n][b].consume_hashes(send.data_ptr() + off * 8, counts[b], 1)
                off += counts[b]
            del batch
        return added

    if os.environ.get('CFG4_PROF'):
        lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t1 = time.time()
    order = [(1, 'mother'), (2, 'father'), (0, 'proband')]
    if streams > 1:
        added = hk.run_concurrently([(lambda si=si, n=n: count_sample(si, n)) for si, n in order])
    else:
        added = [count_sample(si, n) for si, n in order]
    lib.kv_synchronize()
    t_count = time.time() - t1
    assert all(a == n_reads * nk for a in added), (added, n_reads * nk)
    print('count: {:.2f} s ({} k-mers per sample added; {:.2f} s of it generating reads, summed over the streams)'.format(t_count, added[0], t_gen[0]), flush=True)

    # the scan: the proband's batches once more, as (hash, tag) pairs grouped by band
    t2 = time.time()
    hit_cap = max(1 << 20, int(n_reads * 0.03))
    tags = [torch.empty(hit_cap, dtype=torch.int64, device=dev) for _ in range(B)]
    abund = [torch.empty((hit_cap, 3), dtype=torch.uint8, device=dev) for _ in range(B)]
    nhit = [0] * B
    send = torch.empty((cap, 2), dtype=torch.int64, device=dev)
    gen0 = t_gen[0]
    for lo in firsts:
        ta = time.perf_counter()
        batch = hk.ReadBatch.generate(genome_len, seed, 0, lo, min(per_batch, n_reads - lo), L)
        t_gen[0] += time.perf_counter() - ta
        counts = hk.route_hashes(batch, hk.Counttable, k, B, lo, True, send.data_ptr(), cap)
        off = 0
        for b in range(B):
            if counts[b]:
                room = hit_cap - nhit[b]
                got = hk.novel_scan_hashes([sk['proband'][b]], [sk['mother'][b], sk['father'][b]], send.data_ptr() + off * 16, counts[b], 6, 1,
                                           tags[b].data_ptr() + nhit[b] * 8, abund[b].data_ptr() + nhit[b] * 3, room)
                assert got <= room, 'hit buffer of band {} too small'.format(b)
                nhit[b] += got
            off += counts[b]
        del batch
    lib.kv_synchronize()
    t_scan = time.time() - t2
    print('scan: {:.2f} s ({:.2f} s of it generating reads)'.format(t_scan, t_gen[0] - gen0), flush=True)
    t3 = time.time()
    sums = []
    for b in range(B):
        r, o, a = hk.hits_from_tagged(tags[b].data_ptr(), abund[b].data_ptr(), nhit[b], nhit[b], 3)
        sums.append(checksum(r, o, a))
    t_sort = time.time() - t3
    if os.environ.get('CFG4_PROF'):
        import ctypes
        lib.kv_prof_enable(0)
        buf = ctypes.create_string_buffer(8192)
        lib.kv_prof_names(buf, 8192)
        for name in buf.value.decode().split(','):
            ms, nl = ctypes.c_double(), ctypes.c_uint64()
            lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(nl))
            print('    {:24s} {:10.1f} ms {:6d} launches'.format(name, ms.value, nl.value))
    total = t_count + t_scan + t_sort
    free, whole = torch.cuda.mem_get_info(0)
    print('hits in (read, offset) order per band: {:.2f} s'.format(t_sort))
    print('whole job: {:.2f} s = {:.1f} M reads/s through count + novel (3 x {} reads); HBM in use at the end {:.0f} GB'.format(
        total, 3 * n_reads / total / 1e6, n_reads, (whole - free) / 1e9))
    for b in range(B):
        print('  band {}: {}'.format(b, sums[b]))
    if os.environ.get('CFG4_CHECK') is not None:
        # one band again the way bench.py --workload cfg4-band computes it: every batch hashed for that band alone
        cb = int(os.environ['CFG4_CHECK'])
        ref = {n: hk.Counttable(k, memory / B / T, T) for n in names}
        rs, os_, as_ = [], [], []
        for si, n in order:
            for lo in firsts:
                ref[n].consume_batch(hk.ReadBatch.generate(genome_len, seed, si, lo, min(per_batch, n_reads - lo), L), B, cb)
            for t in range(T):
                assert ref[n].table_bytes(t) == sk[n][cb].table_bytes(t), (n, t)
        for lo in firsts:
            batch = hk.ReadBatch.generate(genome_len, seed, 0, lo, min(per_batch, n_reads - lo), L)
            r, o, a, _ = hk.novel_scan([ref['proband']], [ref['mother'], ref['father']], batch, 6, 1, band_mode=1, nbands=B, band=cb)
            rs.append(np.asarray(r, dtype=np.uint32) + np.uint32(lo)); os_.append(o); as_.append(a)
        want = checksum(np.concatenate(rs), np.concatenate(os_), np.concatenate(as_))
        print('check band {}: tables equal; banded scan {} {}'.format(cb, want, 'ok' if want == sums[cb] else 'MISMATCH'))
        assert want == sums[cb]


if __name__ == '__main__':
    main()
