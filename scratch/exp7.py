# per-rank compute of the read-sharded mode at N ranks, emulated on one GPU (no comm)
import sys, time, ctypes
sys.path.insert(0, '.')
import torch
torch.cuda.init()
from kevlar_amd import _lib, khmer as hk, synth
lib = _lib.load(); _lib.require_device()
L, k = 100, 31
packed = synth.trio_reads_packed(25_000_000, 30, L)
names = ('proband', 'mother', 'father')
n_reads = packed['proband'].shape[0]
def prof_all():
    buf = ctypes.create_string_buffer(4096); lib.kv_prof_names(buf, 4096); out = {}
    for name in buf.value.decode().split(','):
        if not name: continue
        ms, c = ctypes.c_double(), ctypes.c_uint64(); lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(c)); out[name] = round(ms.value, 2)
    return out
for N in (8, 4, 2):
    # what rank 0 receives: destination-0 blocks of a full routing
    recv = {}
    for n in names:
        w = 2 if n == 'proband' else 1
        parts = []
        CH = n_reads // 8
        for c in range(8):
            part = hk.ReadBatch.from_packed(packed[n][c * CH:(c + 1) * CH if c < 7 else n_reads], L)
            nk = part.num_kmers(k)
            send = torch.empty((nk, w), dtype=torch.int64, device='cuda')
            counts = hk.route_hashes(part, hk.Counttable, k, N, c * CH, w == 2, send.data_ptr(), send.shape[0])
            parts.append(send[:counts[0]].clone())
            del send, part
        recv[n] = torch.cat(parts)
        del parts
    torch.cuda.synchronize()
    shard = {n: hk.ReadBatch.from_packed(packed[n][:n_reads // N], L) for n in names}
    nk_s = shard['proband'].num_kmers(k)
    sendbuf = {w: torch.empty((nk_s, w), dtype=torch.int64, device='cuda') for w in (1, 2)}
    sk = {n: hk.Counttable(k, 2e9 / N / 4, 4) for n in names}
    tags = torch.empty(1 << 24, dtype=torch.int64, device='cuda'); abund = torch.empty((1 << 24, 3), dtype=torch.uint8, device='cuda')
    def step():
        t = {}
        t0 = time.perf_counter()
        for n in names:
            w = 2 if n == 'proband' else 1
            hk.route_hashes(shard[n], hk.Counttable, k, N, 0, w == 2, sendbuf[w].data_ptr(), nk_s)
        t['route'] = time.perf_counter() - t0; t0 = time.perf_counter()
        for n in names:
            sk[n].clear(); sk[n].consume_hashes(recv[n].data_ptr(), recv[n].shape[0], recv[n].shape[1])
        t['count'] = time.perf_counter() - t0; t0 = time.perf_counter()
        nh = hk.novel_scan_hashes([sk['proband']], [sk['mother'], sk['father']], recv['proband'].data_ptr(), recv['proband'].shape[0], 6, 1, tags.data_ptr(), abund.data_ptr(), 1 << 24)
        t['scan'] = time.perf_counter() - t0; t0 = time.perf_counter()
        r, o, a = hk.hits_from_tagged(tags.data_ptr(), abund.data_ptr(), nh, nh, 3)
        t['sort'] = time.perf_counter() - t0
        return t, nh
    step(); lib.kv_prof_reset(); lib.kv_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(3): t, nh = step()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    lib.kv_prof_enable(0)
    print('N=%d per-rank compute %.1f ms hits %d wall(ms) %s kernels(3 steps) %s' % (N, dt, nh, {a: round(b * 1e3, 1) for a, b in t.items()}, prof_all()), flush=True)
    del sk, shard, sendbuf, recv
