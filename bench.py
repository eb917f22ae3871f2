#!/usr/bin/env python3
"""bench.py -- reads/sec through count + novel on a synthetic family (BASELINE.json metric).

A "step" is one full pass of the hot path over one synthetic family already resident in HBM as
2-bit packed reads: zero the sketches, `count` every sample (kv_consume), then the fused `novel`
scan of the proband against its controls (kv_novel_scan).
    value = (reads of all samples) / (time per step), whole job.

--workload (BASELINE.json configs):
    cfg2        25 Mb trio, 30x, k=31, 2 GB sketch per sample  (the configuration the metric is quoted on; default)
    cfg5        proband + 3 controls, k=51, 30x, 25 Mb          (multi-control test, 128-bit keys / three murmur blocks)
    cfg1        50 kb trio, 10x, k=31, 1 MB sketch              (the reference's own CPU-runnable case: plumbing)
    cfg4-proxy  250 Mb trio, 30x, k=31, 8 GB sketch per sample, reads in batches of 18.75 M: one band's share of
                config 4 (3 Gb, 8 bands) on one GPU -- proves the batching and the HBM budget, not the 8-GPU run
    cfg4-band   config 4 at true scale as one of its eight GPUs sees it: 3 Gb genome, 900 M reads per sample (written into
                HBM by kv_reads_generate: 25 GB per sample), band 0 of 8 with 8 GB sketches, count + novel, then `filter` and
                `partition` of the band's annotated reads; per-stage seconds and peak HBM on the line
N > 1 : configs[2]: the same trio, kevlar's k-mer banding with band b on GPU b (1/N of the hash space and of
        the table memory per GPU).  Total work is fixed -> "scaling": "strong".
        --multi banded   : the reference's layout -- every GPU streams all reads and keeps its band; then one RCCL
                           all-gather of the per-band hits, sorted on the device.
        --multi exchange : every GPU hashes 1/N of the reads once and one RCCL all-to-all delivers each hash to its
                           band's owner (kevlar_amd/shardrun.py); sketches and hits are identical to the banded
                           run's.  Default from 4 GPUs up.  A failure in either layout ends the job non-zero:
                           there is no silent switch of layouts.
        A multi-GPU line validates itself: `ranks_seen` and `kmers_per_rank` come from RCCL collectives, and
        `hits_checksum` (the merged hits) must equal `replay_checksum` -- rank 0 replaying the N bands one after
        the other on its own GPU after the timed region -- which is also what a single-GPU run prints as
        `banded_checksums[N]`.  (A banded run is not bit-identical to the unbanded one: every band has its own
        Count-Min tables, hence its own collisions; it is bit-identical to kevlar run band by band.)

Also on the line: `roofline` (algorithmic bytes / live HIP-event time of the dominant stage), `cpu_baseline`
(the C oracle on the host cores: one core and all cores), `end_to_end` (FASTQ files on disk -> annotated reads
through the CLI drivers, host parse and PCIe included).
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    'cfg2': dict(genome_mb=25.0, coverage=30.0, ksize=31, memory=2e9, controls=2, batch_reads=0,
                 label='BASELINE.json configs[1]: synthetic 25 Mb trio, 30x, k=31, single band'),
    'cfg5': dict(genome_mb=25.0, coverage=30.0, ksize=51, memory=2e9, controls=3, batch_reads=0,
                 label='BASELINE.json configs[4]: proband + 3 controls, k=51, 30x, 25 Mb'),
    'cfg1': dict(genome_mb=0.05, coverage=10.0, ksize=31, memory=1e6, controls=2, batch_reads=0,
                 label='BASELINE.json configs[0]: 50 kb trio, 10x, k=31 (reference CPU plumbing case)'),
    'cfg4-proxy': dict(genome_mb=250.0, coverage=30.0, ksize=31, memory=8e9, controls=2, batch_reads=18_750_000,
                       label='one band\'s share of BASELINE.json configs[3] on one GPU: 250 Mb trio, 30x, k=31, 8 GB sketch '
                             'per sample, reads streamed in batches of 18.75 M (four per sample)'),
    'cfg4-band': dict(genome_mb=3000.0, coverage=30.0, ksize=31, memory=64e9, controls=2, batch_reads=75_000_000, bands=8, band=0,
                      device_generated=True,
                      label='BASELINE.json configs[3] as ONE of its 8 GPUs sees it: 3 Gb trio, 30x, k=31, band 0 of 8 (8 GB sketch per '
                            'sample on this GPU), all 900 M reads of every sample streamed in batches of 75 M (12 per sample: a batch streams the 8 GB of tables once; generated on the '
                            'device, resident in HBM), then filter and partition of the band\'s annotated reads'),
}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=None, help='timed steps (default: 20, or 2 for the config-4 workloads whose step takes seconds)')
    p.add_argument('--warmup', type=int, default=None, help='untimed steps first (default: 5, or 1): the first steps of a process allocate the '
                                                            'gigabyte arenas and load the kernels, a fresh box takes a few more to settle')
    p.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    p.add_argument('--batch-reads', type=int, default=None, help='reads per batch (default: the workload\'s; 0 = whole samples)')
    p.add_argument('--genome-mb', type=float, default=None)
    p.add_argument('--coverage', type=float, default=None)
    p.add_argument('--read-len', type=int, default=100)
    p.add_argument('--ksize', type=int, default=None)
    p.add_argument('--memory', type=float, default=None, help='sketch bytes per sample (all bands together)')
    p.add_argument('--case-min', type=int, default=6)
    p.add_argument('--ctrl-max', type=int, default=1)
    p.add_argument('--cpu-reads', type=int, default=100000, help='reads per sample for the one-core CPU baseline leg')
    p.add_argument('--cpu-reads-mt', type=int, default=1000000, help='reads per sample for the all-cores CPU baseline leg')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--e2e-reads', type=int, default=2000000, help='reads per sample written as FASTQ for the end-to-end leg')
    p.add_argument('--no-e2e', action='store_true')
    p.add_argument('--no-replay', action='store_true', help='skip the banded replays behind banded_checksums / replay_checksum')
    p.add_argument('--exchange-items', default='auto', choices=['auto', 'minimizer', 'distinct', 'plain'],
                   help='--multi exchange: distinct = (hash, occurrences) pairs of each rank\'s deduplicated shard (kv_route_distinct), '
                        'the scan answered as an all-gathered set of interesting k-mers (kv_novel_scan_set); plain = one hash per '
                        'k-mer, the case sample as (hash, tag) pairs scanned by the band owners; auto = distinct up to 4 GPUs '
                        '(measured per-rank compute, scratch/exchange_rank_cost.py: 29.6 / 20.8 / 13.8 ms at 2 / 4 / 8 ranks against '
                        '43.0 / 21.4 / 11.3 ms plain)')
    p.add_argument('--exchange-scan', default='owner', choices=['owner', 'shard'],
                   help='minimizer layout: who answers the scan -- the owners of the minimizer buckets from their combined buckets, or every rank by hashing its shard again')
    p.add_argument('--multi', default='auto', choices=['auto', 'exchange', 'banded'],
                   help='N>1: exchange = shard the reads, hash once, all-to-all the hashes by band (kevlar_amd/shardrun.py); '
                        'banded = every rank streams all reads and keeps its band; auto = exchange from 4 GPUs up')
    p.add_argument('--merge', default='mask', choices=['hits', 'mask'],
                   help='multi-GPU merge: mask (default) = north_star\'s all-reduce of the per-owner interesting-k-mer bit masks (a band, a '
                        'set of minimizer buckets, a shard: disjoint findings, so the sum is the OR) inside the timed step, asserted equal to '
                        'the gathered hits, besides the all-gather of the hits that carries their abundances; hits = the all-gather alone')
    p.add_argument('--eager-hits', action='store_true',
                   help='N=1: every scan waits for its hit arrays to reach the host before the step goes on (default: the copy of step i runs '
                        'beside the counts of step i + 1; all hits are fetched either way)')
    p.add_argument('--count-streams', type=int, default=3,
                   help='N=1: count the samples concurrently on this many HIP streams, one host thread each (the samples are '
                        'independent; kernels bound by different units overlap and no stream waits for another\'s host round '
                        'trips: 39-40 ms against 43-46 ms per step at config 2).  Per-kernel HIP-event durations then include '
                        'time sharing; 1 keeps the launches back to back for a clean per-kernel attribution')
    p.add_argument('--bands', type=int, default=None, help='one GPU plays ONE band of a run banded this many ways (what a rank of --multi banded computes); --band picks it')
    p.add_argument('--band', type=int, default=0)
    p.add_argument('--no-downstream', action='store_true', help='cfg4-band: count and scan only (no annotated reads, filter, partition)')
    p.add_argument('--traffic', default='live', choices=['live', 'file', 'none'],
                   help='roofline.traffic (HBM bytes of the dominant stage): live = measured by two rocprofv3 --pmc child passes of this '
                        'command before the timed run (cfg2, one GPU; ~1.5 min); file = the committed profiles/r*_final file if it '
                        'measured these kernel sources; none')
    p.add_argument('--launch-check', action='store_true', help='only start the ranks and let them meet (no GPU work)')
    p.add_argument('--backend', default='nccl', help='nccl (= RCCL) is what the driver runs; gloo lets two ranks share one GPU in tests')
    args = p.parse_args()
    heavy = args.workload in ('cfg4-band', 'cfg4-proxy')
    if args.steps is None:
        args.steps = 2 if heavy else 20
    if args.warmup is None:
        args.warmup = 1 if heavy else 5
    return args


LIVE_PMC = None      # per-kernel HBM bytes of one step, measured by live_traffic() of this very run

# which stage a kernel belongs to, by the start of its name.  The library's profile scopes (kv_prof_*: HIP-event times) and the
# profiler's kernel names (rocprofv3: PMC bytes) are two name spaces -- the scope "k_skm_emit" times the kernel k_skm_emit_wave,
# "k_skm_split" times k_skm_split_sorted -- so both are grouped by the SAME prefixes, never matched name by name.
STAGE_PREFIXES = {
    'count': ('k_bin_', 'k_route_', 'k_skm_emit', 'k_skm_split', 'k_skm_count', 'k_skm_loose_count', 'k_skm_forward_flag', 'k_consume',
              'k_mex_', 'k_skm_route', 'k_skm_loose_route', 'k_add_hashes'),                       # (the exchange layouts' cut / pack / combine: the count's front end there)
    'novel': ('k_novel_', 'k_skm_novel', 'k_skm_loose_novel', 'k_tile_', 'k_ab_fill', 'k_hit_abund', 'k_case_bits', 'k_skm_set_hits',
              'k_skm_loose_set_hits', 'k_set_insert'),
}


def stage_of(name):
    for stage, prefixes in STAGE_PREFIXES.items():
        if name.startswith(prefixes):
            return stage
    return None


def stage_traffic(pmc_kernels, stage, scope_ms=None, min_share=0.01):
    """HBM bytes per step of one stage from a PMC table keyed by KERNEL name (profiles/summarise.reduce_pmc) -> (total, by kernel).
    scope_ms: {profile scope: ms} of the same stage; a scope that holds more than min_share of the stage's time and has no
    kernel of its own in the PMC table (no kernel name starts with the scope's name, nor the other way round) raises: a kernel
    that was renamed, or that the counter pass did not see, must not drop out of the sum silently."""
    by_kernel = {name: int(rec['hbm_bytes_per_step']) for name, rec in pmc_kernels.items() if stage_of(name) == stage}
    if scope_ms:
        total_ms = sum(scope_ms.values())
        for scope, ms in scope_ms.items():
            if total_ms > 0 and ms > min_share * total_ms and not any(k.startswith(scope) or scope.startswith(k) for k in by_kernel):
                raise RuntimeError('roofline.traffic: the profile scope {} holds {:.1f} % of the {} stage and no kernel in the PMC table '
                                   'matches it (kernels seen: {})'.format(scope, 100.0 * ms / total_ms, stage, sorted(by_kernel)))
    return sum(by_kernel.values()), by_kernel


def live_traffic(args):
    """Two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass), one
    step each with the samples back to back, BEFORE this process initialises the GPU; returns the reduced counters or
    None (no rocprofv3, a failing pass, a time-out: the bench line then says so and goes on)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if not rocprof:
        return None
    sys.path.insert(0, os.path.join(ROOT, 'profiles'))
    import summarise
    tmp = tempfile.mkdtemp(prefix='kv_pmc_')
    env = dict(os.environ, TMPDIR=tmp)
    child = [sys.executable, os.path.abspath(__file__), '--workload', args.workload, '--steps', '1', '--warmup', '0', '--no-cpu-baseline',
             '--no-e2e', '--no-replay', '--count-streams', '1', '--traffic', 'none']
    try:
        for counter, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
            res = subprocess.run([rocprof, '--pmc', counter, '--output-format', 'csv', '-d', os.path.join(tmp, sub), '--'] + child,
                                 cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420)
            if res.returncode != 0:
                return None
        rec = summarise.reduce_pmc(tmp)
        rec['source'] = 'live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two passes of one step of this command on this GPU, before the timed run'
        return rec if rec['kernels'] else None
    except (subprocess.TimeoutExpired, OSError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(n):
    """Run this very command under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port) and return its exit code.  The children are new processes: this one never initialises HIP."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def launch_check(args, rank, world):
    """--launch-check: the ranks meet, exchange their ids over the chosen backend and leave; no GPU work (the CPU
    test of the launcher, tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(args.backend if args.backend != 'nccl' or torch.cuda.is_available() else 'gloo')
    mine = torch.tensor([rank], dtype=torch.int64)
    if dist.get_backend() == 'nccl':
        mine = mine.cuda(int(os.environ.get('LOCAL_RANK', '0')) % max(1, torch.cuda.device_count()))
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    if rank == 0:
        print(json.dumps({'launch_check': True, 'n_gpus': world, 'backend': dist.get_backend(),
                          'ranks_seen': sorted(int(t.item()) for t in seen)}))
    dist.destroy_process_group()


def prof(lib, name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def hits_checksum(r, o, a):
    import numpy as np
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(r, dtype='<u4').tobytes())
    h.update(np.ascontiguousarray(o, dtype='<u4').tobytes())
    h.update(np.ascontiguousarray(a, dtype=np.uint8).tobytes())
    return '{}:{}'.format(len(r), h.hexdigest()[:16])


def main():
    args = parse_args()
    wl = dict(WORKLOADS[args.workload])
    for key in ('genome_mb', 'coverage', 'ksize', 'memory', 'batch_reads'):
        if getattr(args, key) is not None:
            wl[key] = getattr(args, key)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, BEFORE this process
        # has touched the GPU (nothing above imports torch or the library); relay rank 0's line and the exit code
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus {} was started with WORLD_SIZE={}'.format(args.gpus, world))
    if args.launch_check:
        return launch_check(args, rank, world)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # (ranks started by somebody else's torch.distributed.run: see self_launch)
    global LIVE_PMC
    if world == 1 and args.gpus == 1 and args.traffic == 'live' and args.workload == 'cfg2':
        LIVE_PMC = live_traffic(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__
    __graft_entry__.build_product()          # the product only: the oracle is built (and imported) by the cpu_baseline leg alone
    from kevlar_amd import _lib, khmer as hk, synth

    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)
    os.environ['LOCAL_RANK'] = str(dev_index)      # kevlar_amd._lib binds the library to the same device
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(args.backend)
    coll_device = torch.device('cuda', dev_index) if args.backend == 'nccl' else torch.device('cpu')
    lib = _lib.load()
    _lib.require_device()

    # ---- synthetic family (same seeds on every rank), packed on the host, uploaded once
    L, k = args.read_len, int(wl['ksize'])
    genome_len = int(wl['genome_mb'] * 1e6)
    t0 = time.time()
    on_device = bool(wl.get('device_generated'))
    GEN_SEED = 42
    if on_device:
        assert world == 1, '{} is a one-GPU workload'.format(args.workload)
        packed = None
        names = ('proband', 'mother', 'father')
        n_reads = int(genome_len * wl['coverage'] / L)
    else:
        packed = synth.trio_reads_packed(genome_len, wl['coverage'], L, extra_controls=wl['controls'] - 2)
        names = tuple(packed)                           # proband first, then the controls
        n_reads = packed['proband'].shape[0]
    controls = names[1:]
    multi = args.multi if args.multi != 'auto' else ('exchange' if world >= 4 else 'banded')
    if args.exchange_items == 'auto':
        # minimizer: super-k-mer records go to the owner of their minimizer bucket, who deduplicates at the sample's full
        # coverage (needs 16 <= k <= 64); distinct: every rank deduplicates its own shard (little to combine at 1/8 of the
        # coverage); plain: one hash per k-mer
        args.exchange_items = 'minimizer' if 16 <= k <= 64 else ('distinct' if world <= 4 else 'plain')
    exchange = world > 1 and multi == 'exchange'
    by_minimizer_layout = exchange and args.exchange_items == 'minimizer'
    per_batch = int(wl['batch_reads']) or n_reads
    upload_s = None
    if exchange:
        from kevlar_amd import shardrun
        bounds = {n: shardrun.shard_bounds(n_reads, world, rank) for n in names}
        batches = {n: [hk.ReadBatch.from_packed(packed[n][bounds[n][0]:bounds[n][1]], L)] for n in names}
        run = shardrun.ShardedTrio(k, hk.Counttable)
    elif on_device:
        t_up = time.time()
        batches = {n: [hk.ReadBatch.generate(genome_len, GEN_SEED, si, lo, min(per_batch, n_reads - lo), L) for lo in range(0, n_reads, per_batch)]
                   for si, n in enumerate(names)}
        lib.kv_synchronize()
        upload_s = time.time() - t_up
    else:
        t_up = time.time()
        batches = {n: [hk.ReadBatch.from_packed(packed[n][lo:lo + per_batch], L) for lo in range(0, n_reads, per_batch)] for n in names}
        lib.kv_synchronize()
        upload_s = time.time() - t_up
    batch_first = [lo for lo in range(0, n_reads, per_batch)]
    gen_s = time.time() - t0
    nk = L - k + 1
    T = 4
    S = len(names)
    if args.bands:
        wl = dict(wl, bands=int(args.bands), band=int(args.band))
    solo_bands, solo_band = int(wl.get('bands', 0)), int(wl.get('band', 0))       # a single GPU playing one band of a banded run
    mem_per_gpu = wl['memory'] / max(1, world) / max(1, solo_bands)
    hbm_low = [torch.cuda.mem_get_info(dev_index)[0]]

    def note_hbm():
        hbm_low[0] = min(hbm_low[0], torch.cuda.mem_get_info(dev_index)[0])

    def make_sketches(memory):
        sk = {n: hk.Counttable(k, memory / T, T) for n in names}
        if len(batches['proband']) == 1:
            # what `kevlar novel` does for a case sample that is one batch (kevlar_amd/count.py): the count keeps the batch's distinct
            # k-mers with their hashes, the scan that follows evaluates from that list
            # (a stream's first batch gets no list unless asked -- the allocation does not pay for a one-shot run; the bench measures
            # the steady state, and its one-step counter passes must take the path the timed steps take)
            sk['proband'].expect_scan(steady=True)
        return sk

    sketches = make_sketches(mem_per_gpu)
    wall = {'count': 0.0, 'novel': 0.0, 'merge': 0.0}

    # --merge mask: one bit per (read, k-mer offset) of the proband, set by this rank's scan for the k-mers of its band
    band_mask = None
    if args.merge == 'mask' and world > 1 and len(batch_first) == 1 and n_reads * nk < (1 << 36):      # (whole samples, one batch; up to 8 GB of mask)
        band_mask = torch.zeros((n_reads * nk + 31) // 32, dtype=torch.int32, device=torch.device('cuda', dev_index))
        if exchange:
            run.band_mask = (band_mask, nk)         # every scan of the exchange layout sets this rank's findings and all-reduces (ShardedTrio._merge_mask)

    class PendingHits(object):
        """the hits of a one-batch scan whose arrays are still on their way to the host (hk.novel_scan(lazy=True)): the next step's
        counts run meanwhile; unpacks like the (read, offset, abundances) triple it stands for"""
        def __init__(self, lazy):
            self.lazy = lazy

        def __iter__(self):
            return iter(self.lazy.arrays()[:3])

    def scan_batches(sk, band_mode, nbands, band):
        rs, os_, as_ = [], [], []
        if len(batch_first) == 1 and world == 1 and band_mask is None and sk is sketches and not args.eager_hits:
            # one batch, one GPU: the scan returns when its kernels are done and the hit arrays (25 MB at config 2, half a millisecond of
            # PCIe) travel while the next step counts; every step's hits are fetched all the same -- the fence that ends the timed
            # region waits for the last copy, the checksum is taken from those arrays (--eager-hits: wait inside the step, as before)
            return PendingHits(hk.novel_scan([sk['proband']], [sk[n] for n in controls], batches['proband'][0], args.case_min, args.ctrl_max,
                                             band_mode=band_mode, nbands=nbands, band=band, lazy=True))
        many = len(batch_first) > 1 and band_mask is None and world == 1 and args.count_streams > 1
        if many:
            # the batches of a case sample are independent: scanned side by side on the streams the counts used (every scan is a
            # few kernels between host round trips -- mask, hit count, hit list -- that another batch's kernels fill)
            def job(batch):
                return lambda: hk.novel_scan([sk['proband']], [sk[n] for n in controls], batch, args.case_min, args.ctrl_max,
                                             band_mode=band_mode, nbands=nbands, band=band)
            found = []
            todo = list(batches['proband'])
            for lo in range(0, len(todo), args.count_streams):
                found += hk.run_concurrently([job(b) for b in todo[lo:lo + args.count_streams]])
            for first, (r, o, a, _) in zip(batch_first, found):
                rs.append(r if first == 0 else np.asarray(r, dtype=np.uint32) + np.uint32(first)); os_.append(o); as_.append(a)
        for first, batch in zip(batch_first, [] if many else batches['proband']):
            extra = {}
            if band_mask is not None and sk is sketches:
                band_mask.zero_()
                torch.cuda.synchronize()
                extra = dict(mask_ptr=band_mask.data_ptr(), mask_stride=nk)
            r, o, a, _ = hk.novel_scan([sk['proband']], [sk[n] for n in controls], batch, args.case_min, args.ctrl_max,
                                       band_mode=band_mode, nbands=nbands, band=band, **extra)
            # (a sample that is one batch: its read indices are the batch's -- no 9 MB copy just to add zero, a millisecond per step)
            rs.append(r if first == 0 else np.asarray(r, dtype=np.uint32) + np.uint32(first)); os_.append(o); as_.append(a)
        if len(rs) == 1:
            return rs[0], os_[0], as_[0]
        return np.concatenate(rs), np.concatenate(os_), np.concatenate(as_)

    def count_and_scan(sk, nbands, band):
        """count every sample (controls first, the case sample last: its super-k-mer buckets are still in place when
        the scan starts), then scan the case sample"""
        kmers = 0
        t_a = time.perf_counter()
        order = list(controls) + ['proband']
        if args.count_streams > 1 and not exchange:           # (a rank of a banded run counts its band of the three samples side by side as well)
            def job(n):
                def count_one():
                    sk[n].clear()
                    return sum(sk[n].consume_batch(b, nbands, band) for b in batches[n])
                return count_one
            for lo in range(0, len(order), args.count_streams):
                kmers += sum(hk.run_concurrently([job(n) for n in order[lo:lo + args.count_streams]]))
        else:
            for n in order:
                sk[n].clear()
                for b in batches[n]:
                    kmers += sk[n].consume_batch(b, nbands, band)
        t_b = time.perf_counter()
        found = scan_batches(sk, 1 if nbands else 0, nbands, band)
        t_c = time.perf_counter()
        wall['count'] += t_b - t_a
        wall['novel'] += t_c - t_b
        return kmers, found

    def step_banded():
        kmers, found = count_and_scan(sketches, world if world > 1 else solo_bands, rank if world > 1 else solo_band)
        if world == 1:
            note_hbm()
            return kmers, found
        r, o, a = found
        note_hbm()
        t_c = time.perf_counter()
        if world > 1:
            # every band's hits to every rank, sorted on the device (the per-band bit mask that `kevlar unband`
            # would OR together carries no information beyond the hits, so it is not exchanged here)
            from kevlar_amd import bandmerge
            if band_mask is not None:
                # bands are disjoint: the sum of the 0/1 words is their OR (docs/banding.rst; kevlar/unband.py:41-77)
                torch.cuda.synchronize()
                if args.backend == 'nccl':
                    bandmerge.allreduce_mask(band_mask)
                else:
                    staged_mask = band_mask.cpu()
                    bandmerge.allreduce_mask(staged_mask)
                    band_mask.copy_(staged_mask)
                torch.cuda.synchronize()
            r, o, a = bandmerge.allgather_hits_device(r, o, a, torch.device('cuda', dev_index), staged=(args.backend != 'nccl'))
        wall['merge'] += time.perf_counter() - t_c
        return kmers, (r, o, a)

    def step_exchange():
        # route + exchange of sample i+1 overlap the count of sample i (RCCL runs on its own stream)
        t_a = time.perf_counter()
        for n in names:
            sketches[n].clear()
        kmers = 0
        distinct = args.exchange_items in ('distinct', 'minimizer')
        by_minimizer = args.exchange_items == 'minimizer'

        def begin(n_):
            if by_minimizer:
                return run.start_minimizer(batches[n_][0], bounds[n_][0], n_reads, L)
            return run.start(batches[n_][0], bounds[n_][0], not distinct and n_ == names[0], distinct=distinct)
        if distinct:
            # counts travel as (hash, occurrences) pairs of each rank's deduplicated shard; the case sample goes last, so
            # its bucketed shard is still resident when the scan looks its k-mers up in the set of interesting ones
            order = list(controls) + [names[0]]
        else:
            order = list(names)         # the case sample travels as (hash, tag) pairs, counted and scanned by the owners
        if by_minimizer:
            # three collectives per sample -- records to the owners of their minimizer buckets, pairs to the band owners, in between the
            # owner's combine -- interleaved so that a sample's records travel while the next sample's shard is cut, and its pairs
            # while the next sample's records are combined (RCCL runs on its own stream; every rank keeps this order)
            def cut(n_):
                # (records without read positions for every sample but the one the owners answer the scan from)
                return run.cut_minimizer(batches[n_][0], bounds[n_][0], n_reads, L, short=not (n_ == names[0] and args.exchange_scan == 'owner'))
            cuts = {0: cut(order[0])}
            flying = None
            for i, n in enumerate(order):
                if i + 1 < len(order):
                    cuts[i + 1] = cut(order[i + 1])
                ex = run.combine_minimizer(cuts.pop(i), keep_scan=(n == names[0] and args.exchange_scan == 'owner'))
                if flying is not None:
                    kmers += run.finish(flying[0], sketches[flying[1]], keep_for_scan=(flying[1] == names[0]))
                flying = (ex, n)
            kmers += run.finish(flying[0], sketches[flying[1]], keep_for_scan=(flying[1] == names[0]))
        else:
            pending = begin(order[0])
            for i, n in enumerate(order):
                nn = order[i + 1] if i + 1 < len(order) else None
                nxt = begin(nn) if nn else None
                kmers += run.finish(pending, sketches[n], keep_for_scan=(n == names[0]))
                pending = nxt
        t_b = time.perf_counter()
        cases, ctrls = [sketches['proband']], [sketches[n] for n in controls]
        if by_minimizer and args.exchange_scan == 'owner':
            # the owners of the minimizer buckets answer (they hold every occurrence with its position, and its hash); a sample that
            # fell back to `distinct` pairs is scanned shard by shard, decided collectively inside
            r, o, a = run.scan_minimizer(cases, ctrls, args.case_min, args.ctrl_max, batches[names[0]][0], bounds[names[0]][0])
        elif distinct:
            r, o, a = run.scan_distinct(cases, ctrls, args.case_min, args.ctrl_max, batches[names[0]][0], bounds[names[0]][0])
        else:
            r, o, a = run.scan(cases, ctrls, args.case_min, args.ctrl_max)
        t_c = time.perf_counter()
        wall['count'] += t_b - t_a
        wall['novel'] += t_c - t_b
        return kmers, (r, o, a)

    step = step_exchange if exchange else step_banded

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib.kv_prof_reset()
    lib.kv_prof_enable(1)
    for key in wall:
        wall[key] = 0.0
    if exchange:
        for key in run.timing:
            run.timing[key] = 0.0
    if world > 1:
        from kevlar_amd import shardrun as _sr
        _sr.SENT['bytes'] = 0
    fence()
    clocks = ClockWatch(dev_index) if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kmers, hits = step()
    fence()
    elapsed = time.perf_counter() - t0
    clocks = clocks.stop() if clocks is not None else None
    lib.kv_prof_enable(0)
    phases = None
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # where a rank's step goes, host clock around each phase, slowest rank per phase (a bad scaling point then says
        # whether routing, the all-to-all, the owner's count, the scan or the gather is to blame).  exchange: route =
        # hashing / deduplicating the shard, exchange = issuing the all-to-all + waiting for it, count = the owner's
        # adds, scan = evaluating (and, with distinct items, the set lookup), gather = the collectives behind the hits
        mine = dict(run.timing) if exchange else {'count': wall['count'], 'scan': wall['novel'], 'gather': wall['merge']}
        keys = sorted(mine)
        both = torch.tensor([mine[key] for key in keys] + [-mine[key] for key in keys], dtype=torch.float64, device=coll_device)
        dist.all_reduce(both, op=dist.ReduceOp.MAX)
        both = [float(v) for v in both.cpu()]
        phases = {'max_over_ranks': {key: round(both[i] / args.steps * 1e3, 3) for i, key in enumerate(keys)},
                  'min_over_ranks': {key: round(-both[len(keys) + i] / args.steps * 1e3, 3) for i, key in enumerate(keys)},
                  'unit': 'ms per step'}
        # what a scaling point is made of, so that a bad one explains itself without a second run: the kernels a rank ran per step
        # (summed HIP-event time, slowest rank), what it handed to collectives for other ranks, and how that compares with the
        # single-GPU step of the committed round profile (a reference from another run, named as such)
        names_buf = ctypes.create_string_buffer(8192)
        lib.kv_prof_names(names_buf, 8192)
        mine_kernel_ms = sum(prof(lib, nm)[0] for nm in names_buf.value.decode().split(',') if nm) / args.steps
        sent = float(_sr.SENT['bytes']) / args.steps
        agg = torch.tensor([mine_kernel_ms, sent, -mine_kernel_ms], dtype=torch.float64, device=coll_device)
        dist.all_reduce(agg, op=dist.ReduceOp.MAX)
        agg = [float(v) for v in agg.cpu()]
        ref_ms, ref_src = None, None
        for round_dir in sorted((d for d in os.listdir(os.path.join(ROOT, 'profiles')) if d.endswith('_final')), reverse=True):
            f = os.path.join(ROOT, 'profiles', round_dir, 'bench.json')
            if os.path.exists(f) and args.workload == 'cfg2':
                try:
                    ref_ms, ref_src = json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'], os.path.relpath(f, ROOT)
                except (ValueError, KeyError, IndexError):
                    pass
                break
        phases['projection'] = {
            'per_rank_kernel_ms_per_step': {'slowest_rank': round(agg[0], 3), 'fastest_rank': round(-agg[2], 3)},
            'per_rank_host_and_wait_ms_per_step': round(elapsed / args.steps * 1e3 - agg[0], 3),
            'bytes_sent_per_rank_per_step_max': int(agg[1]),
            'xgmi_ms_at_350_GBps': round(agg[1] / 350e9 * 1e3, 3),
            'single_gpu_reference_ms_per_step': ref_ms, 'single_gpu_reference': ref_src,
            'speedup_vs_reference': round(ref_ms / (elapsed / args.steps * 1e3), 3) if ref_ms else None,
            'speedup_if_only_kernels_counted': round(ref_ms / agg[0], 3) if ref_ms and agg[0] > 0 else None,
        }

    # ---- cheap end-to-end sanity on the timed result (parity proper lives in tests/)
    r, o, a = hits                      # (a PendingHits unpacks into its arrays: they arrived before the fence above returned)
    nhits = len(r)
    assert nhits > 0, 'the synthetic family must yield interesting k-mers'
    assert (a[:, 0] >= args.case_min).all() and (a[:, 1:] <= args.ctrl_max).all()
    selfcheck = {'hits_checksum': hits_checksum(r, o, a)}
    if os.environ.get('BENCH_DUMP_HITS') and rank == 0:
        np.savez_compressed(os.path.join(os.environ['BENCH_DUMP_HITS'], 'step_world{}.npz'.format(world)), r=np.asarray(r), o=np.asarray(o), a=np.asarray(a))
    if band_mask is not None:
        from kevlar_amd import bandmerge
        mr, mo = bandmerge.mask_to_hits(band_mask, nk)
        selfcheck['mask_allreduce_equals_gathered_hits'] = bool(np.array_equal(mr, np.asarray(r, dtype=np.uint32)) and
                                                                np.array_equal(mo, np.asarray(o, dtype=np.uint32)))
        assert selfcheck['mask_allreduce_equals_gathered_hits'], 'the all-reduced band masks and the gathered hits disagree'
    if world == 1 and solo_bands:
        selfcheck['kmers_in_band'] = int(kmers)
        selfcheck['kmers_in_band_share'] = round(kmers / float(S * n_reads * nk), 5)
        assert abs(selfcheck['kmers_in_band_share'] * solo_bands - 1.0) < 0.01, 'a band holds 1/N of the hash space'
    elif world == 1:
        assert kmers == S * n_reads * nk
    else:
        # every rank reports through RCCL: who took part, and how many k-mers each one counted
        mine = torch.tensor([rank, kmers], dtype=torch.int64, device=coll_device)
        table = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(table, mine)
        table = [t.cpu().tolist() for t in table]
        selfcheck['ranks_seen'] = sorted(int(t[0]) for t in table)
        selfcheck['kmers_per_rank'] = [int(t[1]) for t in sorted(table)]
        assert selfcheck['ranks_seen'] == list(range(world)), 'a rank is missing from the collective'
        assert sum(selfcheck['kmers_per_rank']) == S * n_reads * nk, 'every k-mer of the family must be counted by exactly one rank'
        if exchange:
            # how the exchange layout ran on rank 0 (the decisions are collective: the same on every rank): samples that fell back from the
            # minimizer layout to `distinct` pairs, scans the bucket owners could not answer (all steps, warm-up included)
            selfcheck['exchange'] = {'items': args.exchange_items, 'scan': args.exchange_scan if by_minimizer_layout else ('set' if args.exchange_items == 'distinct' else 'owners of the bands'),
                                     'layout_fallbacks': int(run.fallbacks), 'scan_fallbacks': int(run.scan_fallbacks),
                                     # rank 0's own failures behind them, by reason; argument errors (a mis-wired call, not skew or memory) counted apart
                                     'own_failures': dict(run.fallback_reasons), 'unexpected_failures': int(run.unexpected_failures)}

    def replay_bands(nbands):
        """the nbands-band configuration one band after the other on this GPU: what kevlar does band by band"""
        if exchange:
            full = {n: [hk.ReadBatch.from_packed(packed[n], L)] for n in names}
        else:
            full = batches
        sk = make_sketches(wl['memory'] / nbands)
        parts = []
        for band in range(nbands):
            for n in list(controls) + ['proband']:
                sk[n].clear()
                for b in full[n]:
                    sk[n].consume_batch(b, nbands, band)
            rs = [hk.novel_scan([sk['proband']], [sk[n] for n in controls], b, args.case_min, args.ctrl_max,
                                band_mode=1, nbands=nbands, band=band) for b in full['proband']]
            for first, (rr, oo, aa, _) in zip(batch_first, rs):
                parts.append((np.asarray(rr, dtype=np.uint32) + np.uint32(first), np.asarray(oo), np.asarray(aa)))
        rr = np.concatenate([p[0] for p in parts]); oo = np.concatenate([p[1] for p in parts])
        aa = np.concatenate([p[2] for p in parts]) if parts else np.zeros((0, S), dtype=np.uint8)
        order = np.lexsort((oo, rr))
        if os.environ.get('BENCH_DUMP_HITS'):          # debugging aid: the arrays behind a checksum
            np.savez_compressed(os.path.join(os.environ['BENCH_DUMP_HITS'], 'replay_{}_of_{}.npz'.format(nbands, world)),
                                r=rr[order], o=oo[order], a=aa[order])
        return hits_checksum(rr[order], oo[order], aa[order])

    if not args.no_replay and args.workload in ('cfg2', 'cfg1', 'cfg5'):
        saved_wall = dict(wall)
        if world == 1:
            selfcheck['banded_checksums'] = {str(nb): replay_bands(nb) for nb in (2, 4, 8)}
        elif rank == 0:
            selfcheck['replay_checksum'] = replay_bands(world)
            selfcheck['replay_matches'] = selfcheck['replay_checksum'] == selfcheck['hits_checksum']
            assert selfcheck['replay_matches'], 'merged multi-GPU hits {} differ from the band-by-band replay on one GPU {}'.format(
                selfcheck['hits_checksum'], selfcheck['replay_checksum'])
        wall.update(saved_wall)

    ms_step = elapsed / args.steps * 1e3
    total_reads = S * n_reads
    value = total_reads / (elapsed / args.steps)

    # ---- roofline: algorithmic bytes (A_count = L/4 + 2 T nk, A_novel = L/4 + T S nk per read;
    # with N bands only 1/N of the k-mers reach this GPU's tables) / live HIP-event time
    frac_band = 1.0 / max(1, world) / max(1, solo_bands)
    a_count = n_reads * (L / 4.0 + 2 * T * nk * frac_band)
    a_novel = n_reads * (L / 4.0 + T * S * nk * frac_band)
    buf = ctypes.create_string_buffer(8192)
    lib.kv_prof_names(buf, 8192)
    times = {name: prof(lib, name) for name in buf.value.decode().split(',') if name}
    groups = {st: [n_ for n_ in times if stage_of(n_) == st] for st in STAGE_PREFIXES}
    stage_ms = {st: sum(times[n_][0] for n_ in groups[st]) / args.steps for st in groups}       # per step
    # a kernel that ran in the timed steps, holds a measurable share of them and belongs to no stage would drop out of the stage's
    # time AND of its PMC bytes without anybody noticing (stage_traffic's guard only sees scopes that are in a stage already)
    timed_ms = sum(v[0] for n_, v in times.items() if n_.startswith('k_'))
    unstaged = {n_: round(v[0] / args.steps, 4) for n_, v in times.items()
                if n_.startswith('k_') and stage_of(n_) is None and timed_ms > 0 and v[0] > 0.01 * timed_ms}
    if unstaged and world == 1:        # (N > 1: reported on the line instead -- a scaling run must not die of bookkeeping)
        raise RuntimeError('profile scopes {} hold more than 1 % of the timed kernels each and are in no stage of STAGE_PREFIXES'.format(unstaged))
    kernel_sum_ms = dict(stage_ms)
    concurrent = world == 1 and args.count_streams > 1
    if concurrent:
        # the samples' kernels overlap: their durations add up to more than the time the stage took; the stage's time is
        # the host clock around it (its last kernel is awaited inside)
        stage_ms['count'] = min(stage_ms['count'], wall['count'] / args.steps * 1e3)
    stage_alg = {'count': a_count * S, 'novel': a_novel}
    stage = max(stage_ms, key=lambda st: stage_ms[st])
    dominant = max(groups[stage], key=lambda n_: times[n_][0]) if groups[stage] else None
    achieved = stage_alg[stage] / (stage_ms[stage] * 1e-3) / 1e9 if stage_ms[stage] > 0 else 0.0
    # HBM bytes of the stage from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, separate passes, FETCH doubled for the wide
    # coalesced streams as the guide prescribes, profiles/summarise.py): measured by THIS run before it touched the GPU
    # (live_traffic, below), else from a committed file -- but only one that measured these very kernel sources
    traffic, traffic_source = None, None
    pmc = LIVE_PMC
    if pmc is None and world == 1 and args.workload == 'cfg2' and args.traffic != 'none':
        sys.path.insert(0, os.path.join(ROOT, 'profiles'))
        import summarise
        for round_dir in sorted((d for d in os.listdir(os.path.join(ROOT, 'profiles')) if d.endswith('_final')), reverse=True):
            f = os.path.join(ROOT, 'profiles', round_dir, 'pmc_hbm_bytes.json')
            if os.path.exists(f):
                rec = json.load(open(f))
                if rec.get('kernel_sources_sha16') == summarise.kernel_sources_sha(ROOT):
                    pmc = rec
                    pmc['source'] = '{} ({})'.format(os.path.relpath(f, ROOT), rec.get('collected', 'rocprofv3 --pmc'))
                else:
                    traffic_source = 'none: {} measured other kernel sources than this checkout'.format(os.path.relpath(f, ROOT))
                break
    traffic_by_kernel = None
    if pmc is not None:
        tot, traffic_by_kernel = stage_traffic(pmc.get('kernels', {}), stage, {n_: times[n_][0] for n_ in groups[stage]})
        if tot:
            traffic = int(tot)
            traffic_source = pmc.get('source')
    dom_ms, dom_launches = times[dominant] if dominant else (0.0, 0)
    roofline = {
        'bound': 'hbm', 'stage': stage, 'kernel': dominant,
        'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'traffic_source': traffic_source,
        'traffic_by_kernel': traffic_by_kernel,
        'traffic_over_algorithmic': round(traffic / stage_alg[stage], 3) if traffic else None,
        'stage_ms_per_step': round(stage_ms[stage], 4), 'algorithmic_bytes_per_step_of_stage': int(stage_alg[stage]),
        'kernel_avg_launch_ms': round(dom_ms / max(1, dom_launches), 4), 'kernel_launches': int(dom_launches),
        'kernels_in_no_stage_ms_per_step': unstaged or None,
        'whole_step': {'algorithmic_bytes': int(a_count * S + a_novel),
                       'achieved': round((a_count * S + a_novel) / (ms_step * 1e-3) / 1e9, 2),
                       'frac': round((a_count * S + a_novel) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        'stage_kernel_sum_ms_per_step': round(kernel_sum_ms[stage], 4),
        'note': ('the samples are counted concurrently on {} streams: achieved = algorithmic bytes of the stage in one step / the '
                 'stage\'s wall time (its kernels\' HIP-event durations, which include time sharing, add up to '
                 'stage_kernel_sum_ms_per_step; --count-streams 1 runs them back to back); '.format(args.count_streams) if concurrent else
                 'achieved = algorithmic bytes of the stage in one step / summed HIP-event duration of its kernels; ') + 'the '
                'stage is one logical kernel split over launches (super-k-mer cut, bucket split, per-distinct-k-mer hash, '
                'bin split, LDS-resident apply); it is bound by integer VALU issue (two murmur3 per distinct k-mer, 2-bit '
                'k-mer extraction) and LDS atomics, not by HBM: see DESIGN.md section 4',
        'kernels_ms_per_step': {name: round(times[name][0] / args.steps, 4) for name in sorted(times)},
        'host_wall_ms_per_step': {key: round(val / args.steps * 1e3, 3) for key, val in wall.items()},
    }

    cpu = e2e = None
    downstream = None
    if on_device and not args.no_downstream:
        # the step's working buffers (staging for three batches side by side) go back first: `filter` and `partition` are processes of
        # their own in the workflow and allocate their own sketches; the step's peak is noted before
        note_hbm()
        hk.scratch_trim()
        downstream = band_downstream(args, wl, hits, genome_len, GEN_SEED, L, k, S, synth)
        note_hbm()
        props = torch.cuda.get_device_properties(dev_index)
        downstream['hbm_peak_gb'] = round((props.total_memory - hbm_low[0]) / 1e9, 1)
        downstream['hbm_resident_reads_gb'] = round(sum(b.device_bytes() for n in names for b in batches[n]) / 1e9, 1)
    if rank == 0 and world == 1 and not on_device:
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args, wl, packed, names, synth)
        if not args.no_e2e:
            e2e = end_to_end(args, wl, packed, names, synth)

    if rank == 0:
        out = {
            'metric': 'reads/sec through count+novel ({} samples, k={})'.format(S, k),
            'value': round(value, 1), 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u64 hash / u8 counters', 'data': 'synthetic',
            'config': {
                'workload': '{} [{}]: {:g} Mb genome, {:g}x, {} bp reads, k={}, {} reads/sample x {} samples, '
                            '{:g} GB Count-Min sketch per sample ({} tables), case-min {}, ctrl-max {}'.format(
                                args.workload, wl['label'], wl['genome_mb'], wl['coverage'], L, k, n_reads, S,
                                wl['memory'] / 1e9, T, args.case_min, args.ctrl_max),
                'parallelism': 'single band' if world == 1 else (
                    '{} k-mer bands, 1 per GPU; reads sharded, {} exchanged by band (all-to-all); {}'.format(
                        world, {'distinct': 'distinct (hash, occurrences) pairs of each shard',
                                'minimizer': 'super-k-mer records first exchanged by minimizer bucket, then the bucket owners\' distinct '
                                             '(hash, occurrences) pairs'}.get(args.exchange_items, 'hashes'),
                        'per-owner bit masks of the interesting k-mer occurrences all-reduced (asserted equal to the gathered hits), hits '
                        'all-gathered for their abundances' if band_mask is not None else 'hits all-gathered (--merge hits: no bit-mask all-reduce)') if exchange else
                    ('{} k-mer bands, 1 per GPU; every rank streams all reads; per-band bit masks all-reduced (asserted equal to the '
                     'gathered hits), hits all-gathered for their abundances' if band_mask is not None else
                     '{} k-mer bands, 1 per GPU; every rank streams all reads; hits all-gathered (the per-band bit mask of '
                     'north_star carries nothing beyond them; --merge mask adds its all-reduce)').format(world)),
                'read_batches_per_sample': len(batch_first), 'count_streams': args.count_streams,
                'interesting_kmer_instances': nhits, 'host_generate_pack_upload_s': round(gen_s, 1),
                'packed_reads_upload_s': round(upload_s, 3) if upload_s is not None else None,
                'device': '{} ({} CUs)'.format(torch.cuda.get_device_properties(dev_index).name,
                                               torch.cuda.get_device_properties(dev_index).multi_processor_count),
            },
            # shader clock and socket power sampled beside the timed steps (sysfs; null where the box hides them): box-to-box spread of
            # ms_per_step reads off these; `knobs`: every registered KV_* switch set in this run's environment (kv_knobs_describe)
            'clocks': clocks,
            'knobs': _knobs_active(),
            'selfcheck': selfcheck,
            'downstream': downstream,
            'phases': phases,
            'roofline': roofline,
            'cpu_baseline': cpu,
            'end_to_end': e2e,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _knobs_active():
    from kevlar_amd import _lib
    return _lib.knobs_active()


class ClockWatch(object):
    """The GPU's shader clock (MHz) and power (W) while the timed steps run, read from the amdgpu hwmon files of the card every
    20 ms on a thread of its own (what scratch/clock_watch.sh did with rocm-smi beside a long run).  stop() -> {'sclk_mhz': [min,
    median, max], 'power_w': [...], 'samples': n} or None when the files are not there / not readable."""

    def __init__(self, dev_index, period=0.02):
        import glob
        import threading
        self.freq, self.power = None, None
        # the card this process computes on, by its PCI address (a box shows every card of its node in sysfs, whichever one the
        # process was given: the first collection of round 6 read an idle neighbour's 97 MHz)
        cards = []
        try:
            import torch
            pr = torch.cuda.get_device_properties(dev_index)
            addr = '{:04x}:{:02x}:{:02x}.0'.format(pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            cards = sorted(glob.glob('/sys/bus/pci/devices/{}/hwmon/hwmon*'.format(addr)))
        except Exception:
            cards = []
        cards = [c for c in cards if os.path.exists(os.path.join(c, 'freq1_input'))]
        if cards:
            hw = cards[0]
            self.freq = os.path.join(hw, 'freq1_input')
            for name in ('power1_input', 'power1_average'):
                if os.path.exists(os.path.join(hw, name)):
                    self.power = os.path.join(hw, name)
                    break
        self.samples = []
        self.done = threading.Event()
        self.period = period
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return int(fh.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def _run(self):
        while self.freq and not self.done.is_set():
            self.samples.append((self._read(self.freq), self._read(self.power) if self.power else None))
            self.done.wait(self.period)

    def stop(self):
        self.done.set()
        self.thread.join()
        f = sorted(s[0] / 1e6 for s in self.samples if s[0])
        p = sorted(s[1] / 1e6 for s in self.samples if s[1])
        if not f:
            return None

        def three(v):
            return [round(v[0]), round(v[len(v) // 2]), round(v[-1])] if v else None
        return {'sclk_mhz': three(f), 'power_w': three(p), 'samples': len(f), 'source': os.path.dirname(self.freq)}


def band_annotated_reads(hits, genome_len, seed, L, k, S, synth):
    """The hits of a scan over device-generated proband reads as an AnnotatedReads: the text of the reads that hold a hit comes
    from the generator's numpy restatement (kevlar_amd.synth.device_family_reads: only those reads; the others never leave HBM),
    names are read<index>, qualities a constant."""
    import numpy as np
    from kevlar_amd.annotated import AnnotatedReads
    r, o, a = (np.asarray(x) for x in hits)
    ur, first = np.unique(r, return_index=True)
    n = len(ur)
    seqs = np.empty((n, L), dtype=np.uint8)
    for lo in range(0, n, 200000):
        seqs[lo:lo + 200000] = synth.ALPHABET[synth.device_family_reads(genome_len, seed, 0, ur[lo:lo + 200000], L)]
    name_w = 14
    names = np.empty((n, name_w), dtype=np.uint8)
    names[:, :4] = np.frombuffer(b'read', dtype=np.uint8)
    digits = ur.astype(np.int64)
    for d in range(name_w - 4):
        names[:, name_w - 1 - d] = 48 + digits % 10
        digits //= 10
    ann = AnnotatedReads.__new__(AnnotatedReads)
    ann.n = n
    ann.names, ann.name_offs = names.tobytes(), np.arange(n + 1, dtype=np.uint64) * np.uint64(name_w)
    ann.seqs, ann.seq_offs = seqs.tobytes(), np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    ann.quals, ann.qual_offs = b'I' * (n * L), np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    ann.is_fastq = np.ones(n, dtype=np.uint8)
    ann.first = np.concatenate((first, [len(r)])).astype(np.uint64)
    ann.offset = o.astype(np.uint32)
    ann.abund = a.astype(np.int32).reshape(len(r), S)
    ann.mate_record = np.zeros(0, dtype=np.uint32)
    ann.mates, ann.mate_offs = b'', np.zeros(1, dtype=np.uint64)
    ann.ksize, ann.nsamples = k, S
    ann._finish()
    return ann


def band_downstream(args, wl, hits, genome_len, seed, L, k, S, synth):
    """What follows the scan in the workflow (kevlar/workflows/mark-I/Snakefile:236-309), on the band's hits: the annotated
    reads written as an augmented FASTQ file, `kevlar filter` on it, `kevlar partition` on what survives."""
    import io
    import tempfile
    import numpy as np
    import kevlar_amd
    r = np.asarray(hits[0])
    t0 = time.perf_counter()
    ann = band_annotated_reads(hits, genome_len, seed, L, k, S, synth)
    n = ann.n
    t_text = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix='kv_band_')
    novel_file, filtered_file, part_file = (os.path.join(tmp, f) for f in ('band.novel.augfastq', 'band.filtered.augfastq', 'band.part.augfastq'))
    with kevlar_amd.open_sink(novel_file) as sink:
        ann.format_to(sink, np.arange(n, dtype=np.uint64))          # rendered on the host cores, written as it is rendered
    t1 = time.perf_counter()
    log = io.StringIO()
    old_log, kevlar_amd.logstream = kevlar_amd.logstream, log

    def run(argv):
        a_ = kevlar_amd.cli.parser().parse_args(argv)
        t = time.perf_counter()
        kevlar_amd.cli.mains[a_.cmd](a_)
        return time.perf_counter() - t
    try:
        t_filter = run(['filter', '--memory', '2G', '--case-min', str(args.case_min), '--ctrl-max', str(args.ctrl_max), '-o', filtered_file, novel_file])
        t_part = run(['partition', '-o', part_file, filtered_file])
        if os.environ.get('KV_E2E_PROFILE'):           # where the seconds of filter / partition go at this scale (stderr)
            import cProfile, pstats
            for argv in (['filter', '--memory', '2G', '--case-min', str(args.case_min), '--ctrl-max', str(args.ctrl_max), '-o', filtered_file + '.2', novel_file],
                         ['partition', '-o', part_file + '.2', filtered_file]):
                prof = cProfile.Profile(); t_prof = time.perf_counter()
                prof.enable(); run(argv); prof.disable()
                st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(16)
                sys.stderr.write('[downstream profile] {} {:.2f} s\n{}\n'.format(argv[0], time.perf_counter() - t_prof, st.getvalue()[:4000]))
                os.remove(argv[-2])
    finally:
        kevlar_amd.logstream = old_log
    said = [ln.split('] ', 1)[-1] for ln in log.getvalue().splitlines() if 'Validated' in ln or 'grouped' in ln or 'Processed' in ln]
    out = {'annotated_reads': int(n), 'interesting_kmer_instances': int(len(r)), 'regenerate_read_text_numpy_s': round(t_text - t0, 2),
           'write_annotated_reads_s': round(t1 - t_text, 2),
           'annotated_reads_mb': os.path.getsize(novel_file) >> 20, 'filter_s': round(t_filter, 2), 'partition_s': round(t_part, 2),
           'log': said[-4:]}
    for f in (novel_file, filtered_file, part_file):
        if os.path.exists(f):
            os.remove(f)
    os.rmdir(tmp)
    return out


def cpu_baseline(args, wl, packed, names, synth):
    """The C oracle (oracle/kvoracle.c) on the host cores, full-size sketches, count of every sample then the novel
    scan of the proband: (i) one core, the scalar loop; (ii) all cores the way kevlar drives khmer
    (kevlar/count.py:41-76): threads share one sketch and add with atomic saturating increments."""
    from oracle import okhmer as ok
    k = int(wl['ksize'])
    controls = names[1:]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                            # a container's CPU quota, not the host's core count, is what it can use
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(quota) // int(period)))
    except (OSError, ValueError):
        pass

    def leg(n, threads):
        n = min(n, packed['proband'].shape[0])
        data = {name: ok.concat_reads(synth.unpack_reads(packed[name][:n], args.read_len)) for name in names}
        sk = {name: ok.Counttable(k, wl['memory'] / 4, 4) for name in names}
        t0 = time.perf_counter()
        for name in names:
            if threads == 1:
                ok.consume_reads(sk[name], data[name][0], data[name][1], n)
            else:
                ok.consume_reads_mt(sk[name], data[name][0], data[name][1], n, threads)
        if threads == 1:
            ok.novel_scan([sk['proband']], [sk[c] for c in controls], data['proband'][0], data['proband'][1], n, k,
                          args.case_min, args.ctrl_max)
        else:
            ok.novel_scan_count_mt([sk['proband']], [sk[c] for c in controls], data['proband'][0], data['proband'][1], n, k,
                                   args.case_min, args.ctrl_max, threads)
        dt = time.perf_counter() - t0
        return len(names) * n / dt, n, dt

    one_v, one_n, one_dt = leg(args.cpu_reads, 1)
    out = {'value': round(one_v, 1), 'unit': 'reads/s', 'cores': 1, 'kind': 'port', 'nproc': cores,
           'sample': 'first {} reads of each of the {} samples, full-size sketches, {:.1f} s'.format(one_n, len(names), one_dt)}
    if cores > 1:
        # a short calibration leg first, then a sample sized to ~20 s of wall clock (at most --cpu-reads-mt reads per
        # sample): large enough that the full-size sketches are not empty, bounded so that the default run stays short
        mt_v, mt_n, mt_dt = leg(args.cpu_reads, cores)
        n_big = int(min(args.cpu_reads_mt, mt_v * 20.0 / len(names)))
        if n_big >= 2 * mt_n:
            mt_v, mt_n, mt_dt = leg(n_big, cores)
        out = {'value': round(mt_v, 1), 'unit': 'reads/s', 'cores': cores, 'kind': 'port', 'nproc': cores,
               'sample': 'first {} reads of each of the {} samples, full-size sketches, {} threads on one sketch with atomic '
                         'saturating adds (kevlar/count.py:41-76), {:.1f} s'.format(mt_n, len(names), cores, mt_dt),
               'one_core': out}
    return out


def end_to_end(args, wl, packed, names, synth):
    """FASTQ files on local disk -> `kevlar count` per sample -> `kevlar novel` (annotated reads written), through
    the CLI drivers of kevlar_amd: host parse, 2-bit packing, PCIe, kernels and text output all inside the clock."""
    import io
    import shutil
    import tempfile
    import numpy as np
    import kevlar_amd
    n = min(args.e2e_reads, packed['proband'].shape[0])
    k = int(wl['ksize'])
    tmp = tempfile.mkdtemp(prefix='kv_e2e_')
    try:
        from kevlar_amd import bgzf
        # records as one byte matrix (fixed-width names), binned qualities as current instruments write them: mostly one
        # value, a few lower ones
        rng = np.random.default_rng(12)
        L = args.read_len
        workers = max(1, len(os.sched_getaffinity(0)))
        for name in names:
            tag = '@{}_'.format(name).encode('ascii')
            rec = np.empty((n, len(tag) + 8 + 1 + L + 3 + L + 1), dtype=np.uint8)
            col = 0
            rec[:, :len(tag)] = np.frombuffer(tag, dtype=np.uint8); col += len(tag)
            digits = np.arange(n, dtype=np.int64)
            for d in range(8):
                rec[:, col + 7 - d] = 48 + digits % 10
                digits //= 10
            col += 8
            rec[:, col] = 10; col += 1
            words = packed[name][:n]
            for j in range(L):
                rec[:, col + j] = np.frombuffer(b'ACGT', dtype=np.uint8)[(words[:, j >> 4] >> np.uint32(2 * (j & 15))) & np.uint32(3)]
            col += L
            rec[:, col:col + 3] = np.frombuffer(b'\n+\n', dtype=np.uint8); col += 3
            rec[:, col:col + L] = np.frombuffer(b'F:,#', dtype=np.uint8)[rng.choice(4, size=(n, L), p=[0.9, 0.06, 0.03, 0.01])]; col += L
            rec[:, col] = 10
            text = rec.tobytes()
            del rec
            with open(os.path.join(tmp, name + '.fq'), 'wb') as fh:
                fh.write(text)
            bgzf.write_file(os.path.join(tmp, name + '.bgzf.fq.gz'), text, level=4, threads=workers)   # what kevlar_amd.open(..., 'w') writes
            write_gzip(os.path.join(tmp, name + '.fq.gz'), text, level=4, threads=workers)             # what gzip / pigz write
            del text
        os.sync()       # the 2.5 GB just written are on their way to the disk: let that finish before anything is timed
        saved, kevlar_amd.logstream = kevlar_amd.logstream, io.StringIO()
        mem = '{:d}'.format(int(wl['memory']))

        def novel_run(suffix):
            argv = ['novel', '--ksize', str(k), '--memory', mem, '--threads', '2', '--case', os.path.join(tmp, 'proband' + suffix)]
            for c in names[1:]:
                argv += ['--control', os.path.join(tmp, c + suffix)]
            argv += ['--case-min', str(args.case_min), '--ctrl-max', str(args.ctrl_max), '-o', os.path.join(tmp, 'novel' + suffix + '.augfastq')]
            t0 = time.perf_counter()
            a = kevlar_amd.cli.parser().parse_args(argv)
            kevlar_amd.cli.mains[a.cmd](a)
            return time.perf_counter() - t0
        def stage(argv):
            t0 = time.perf_counter()
            a = kevlar_amd.cli.parser().parse_args(argv)
            kevlar_amd.cli.mains[a.cmd](a)
            return time.perf_counter() - t0
        try:
            # twice each, the better run counts: the first touches cold files, grows the device buffers and pays for lazily
            # loaded kernels
            runs = {sfx: [novel_run(sfx), novel_run(sfx)] for sfx in ('.fq', '.bgzf.fq.gz', '.fq.gz')}
            if os.environ.get('KV_E2E_PROFILE'):
                sys.stderr.write('[e2e runs] {}\n'.format({sfx: [round(t, 3) for t in ts] for sfx, ts in runs.items()}))
                import cProfile, pstats, threading, kevlar_amd.count as kc, kevlar_amd.khmer as kh
                spans, lock = [], threading.Lock()

                def timed(owner, name):
                    inner = getattr(owner, name)

                    def outer(*a, **k):
                        t0 = time.perf_counter()
                        try:
                            return inner(*a, **k)
                        finally:
                            with lock:
                                spans.append((name, round(time.perf_counter() - t0, 4), threading.current_thread().name))
                    setattr(owner, name, outer)
                    return inner
                saved_fns = [(kc, 'allocate', timed(kc, 'allocate')), (kc, '_count_file', timed(kc, '_count_file')), (kc, 'estimate_fpr', timed(kc, 'estimate_fpr')),
                             (kh.ReadParser, '__init__', timed(kh.ReadParser, '__init__')), (kh.ReadParser, 'text_batch', timed(kh.ReadParser, 'text_batch')),
                             (kh.ReadParser, 'take_batch', timed(kh.ReadParser, 'take_batch')), (kh.Counttable, 'consume_batch', timed(kh.Counttable, 'consume_batch')),
                             (kh.Counttable, 'n_unique_kmers', timed(kh.Counttable, 'n_unique_kmers'))]
                prof = cProfile.Profile(); prof.enable(); t_third = novel_run('.fq'); prof.disable()
                for owner, name, inner in saved_fns:
                    setattr(owner, name, inner)
                sys.stderr.write('[e2e spans] {}\n'.format(spans))
                st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('cumtime').print_stats(22)
                sys.stderr.write('[e2e third plain run] {:.3f} s\n{}\n'.format(t_third, st.getvalue()[:4500]))
            dt_plain, dt_bgzf, dt_gzip = min(runs['.fq']), min(runs['.bgzf.fq.gz']), min(runs['.fq.gz'])
            novel_out = os.path.join(tmp, 'novel.bgzf.fq.gz.augfastq')
            dt_filter = stage(['filter', '--memory', '50M', '--case-min', str(args.case_min), '--ctrl-max', str(args.ctrl_max),
                               '-o', os.path.join(tmp, 'filtered.augfastq'), novel_out])
            dt_partition = stage(['partition', '-o', os.path.join(tmp, 'partitioned.augfastq'), os.path.join(tmp, 'filtered.augfastq')])
            if os.environ.get('KV_E2E_PROFILE'):       # where a one-shot filter / partition spends its time (stderr)
                import cProfile, pstats
                for argv in (['filter', '--memory', '50M', '--case-min', str(args.case_min), '--ctrl-max', str(args.ctrl_max),
                              '-o', os.path.join(tmp, 'filtered2.augfastq'), novel_out],
                             ['partition', '-o', os.path.join(tmp, 'partitioned2.augfastq'), os.path.join(tmp, 'filtered.augfastq')]):
                    prof = cProfile.Profile(); t0 = time.perf_counter()
                    prof.enable(); stage(argv); prof.disable()
                    st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(12)
                    sys.stderr.write('[e2e profile] second {} {:.3f} s (first: filter {:.3f}, partition {:.3f})\n{}\n'.format(
                        argv[0], time.perf_counter() - t0, dt_filter, dt_partition, st.getvalue()[:3000]))
        finally:
            log_text = kevlar_amd.logstream.getvalue()
            kevlar_amd.logstream = saved
        grouped = [line for line in log_text.split('\n') if 'grouped' in line]
        def same_output(x, y, what):
            with open(os.path.join(tmp, x)) as a, open(os.path.join(tmp, y)) as b:
                ta, tb = a.read(), b.read()
            if ta != tb:
                la, lb = ta.split('\n'), tb.split('\n')
                first = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), min(len(la), len(lb)))
                raise AssertionError('{} input must give the same annotated reads: {} against {} lines, first difference at line {}: {!r} / {!r}'.format(
                    what, len(la), len(lb), first, la[first:first + 1], lb[first:first + 1]))
        same_output('novel.fq.augfastq', 'novel.bgzf.fq.gz.augfastq', 'plain and blocked-gzip')
        same_output('novel.fq.augfastq', 'novel.fq.gz.augfastq', 'plain and gzip')
        ingest = ingest_rates(os.path.join(tmp, 'proband.fq'), os.path.join(tmp, 'proband.bgzf.fq.gz'), n)
        return {'value': round(len(names) * n / dt_plain, 1), 'unit': 'reads/s',
                'from_bgzf_fastq_gz': round(len(names) * n / dt_bgzf, 1),
                'from_gzip_fastq_gz': round(len(names) * n / dt_gzip, 1),
                'whole_path': {'reads_per_s': round(len(names) * n / (dt_bgzf + dt_filter + dt_partition), 1),
                               'novel_s': round(dt_bgzf, 3), 'filter_s': round(dt_filter, 3), 'partition_s': round(dt_partition, 3),
                               'result': grouped[-1].split('] ')[-1] if grouped else None,
                               'what': 'BGZF .fq.gz files -> kevlar novel (counts the samples) -> kevlar filter -> kevlar partition, annotated '
                                       'reads on disk between the stages'},
                'ingest_reads_per_s': ingest,
                'sample': '{} reads per sample as FASTQ on local disk ({} MB each plain, {} MB blocked gzip); one `kevlar novel --case ... '
                          '--control ...` run: every sample parsed, packed and counted, the case sample parsed again and scanned, annotated '
                          'reads written (better of two runs): {:.2f} s from plain FASTQ (uploaded as text, split and packed on the GPU), {:.2f} s from BGZF .fq.gz '
                          '(inflated, split and packed on the GPU), {:.2f} s from ordinary gzip (one DEFLATE stream per file, {} MB, inflated on the '
                          'GPU in parallel stretches); identical output'.format(n, os.path.getsize(os.path.join(tmp, 'proband.fq')) >> 20,
                                                               os.path.getsize(os.path.join(tmp, 'proband.bgzf.fq.gz')) >> 20, dt_plain, dt_bgzf,
                                                               dt_gzip, os.path.getsize(os.path.join(tmp, 'proband.fq.gz')) >> 20)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def write_gzip(path, text, level=4, threads=1, piece=1 << 20):
    """`text` as ONE gzip member holding one DEFLATE stream, compressed on `threads` cores the way pigz does it: pieces
    deflated independently, all but the last ended by a sync flush, joined end to end (what most pipelines hand over
    as .fq.gz; kevlar_amd.open(..., 'w') writes BGZF instead)"""
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    view = memoryview(text)
    cuts = list(range(0, len(view), piece)) or [0]

    def deflate(at):
        z = zlib.compressobj(level, zlib.DEFLATED, -15)
        last = at + piece >= len(view)
        return z.compress(view[at:at + piece]) + z.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)
    with ThreadPoolExecutor(max(1, threads)) as pool, open(path, 'wb') as out:
        out.write(b'\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03')
        for part in pool.map(deflate, cuts):
            out.write(part)
        out.write(struct.pack('<II', zlib.crc32(view) & 0xffffffff, len(view) & 0xffffffff))


def ingest_rates(fastq, bgzf_gz, n):
    """reads/s from a file on disk to 2-bit packed batches in HBM (the reader alone, no count): plain FASTQ (text uploaded,
    records split and packed on the GPU: kv_fastq.hip), ordinary gzip -- one DEFLATE stream -- inflated on the GPU in
    parallel stretches (kv_gunzip.hip) and, for comparison, by zlib on one host core (KV_GUNZIP=host), blocked gzip
    inflated on the GPU one member per wavefront (kv_inflate.hip), and the packed-read cache a first pass leaves behind
    (KEVLAR_PACK_CACHE=1)"""
    from kevlar_amd import khmer as hk

    def drain(path):
        t0 = time.perf_counter()
        parser, got = hk.ReadParser(path), 0
        while True:
            batch = parser.take_batch(hk.BATCH_READS)
            if batch is None:
                break
            got += batch.n_reads
            batch.close()
        assert got == n
        return round(n / (time.perf_counter() - t0), 1)
    gz = fastq + '.gz'                   # written by end_to_end: one gzip stream
    out = {'fastq': drain(fastq), 'fastq_gz_on_device': max(drain(gz), drain(gz)), 'fastq_bgzf_gz_on_device': max(drain(bgzf_gz), drain(bgzf_gz))}
    os.environ['KV_GUNZIP'] = 'host'
    try:
        out['fastq_gz_host_zlib'] = drain(gz)
    finally:
        os.environ.pop('KV_GUNZIP', None)
    os.environ['KEVLAR_PACK_CACHE'] = '1'
    try:
        out['fastq_gz_first_pass_writing_cache'] = drain(gz)
        out['fastq_gz_from_packed_cache'] = drain(gz)
    finally:
        os.environ.pop('KEVLAR_PACK_CACHE', None)
    return out


if __name__ == '__main__':
    main()
