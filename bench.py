#!/usr/bin/env python3
"""bench.py -- reads/sec through count + novel on a synthetic trio (BASELINE.json metric).

A "step" is one full pass of the hot path over one synthetic trio already resident in HBM
as 2-bit packed reads: zero the three sketches, `count` all three samples (kv_consume), then
the fused `novel` scan of the proband against both parents (kv_novel_scan).
    value = (reads of all three samples) / (time per step), whole job.

N = 1 : workload = BASELINE.json configs[1] (25 Mb genome, 30x, k=31, 2 GB sketch per sample).
N > 1 : configs[2]: the same trio, kevlar's k-mer banding with band b on GPU b (1/N of the hash
        space and of the table memory per GPU).  Total work is fixed -> "scaling": "strong".
        --multi banded   : the reference's layout -- every GPU streams and hashes all reads, keeps
                           its band; then one RCCL all-gather of the per-band hits, sorted on the
                           device.
        --multi exchange : every GPU hashes 1/N of the reads once and one RCCL all-to-all delivers
                           each hash to its band's owner (kevlar_amd/shardrun.py); sketches and
                           hits are identical to the banded run's.  Default from 4 GPUs up (with 2
                           GPUs the single xGMI link between them makes the exchange slower than
                           the replicated hashing).

Also reported: `roofline` for the dominant kernel (algorithmic bytes / live HIP-event time,
see DESIGN.md) and `cpu_baseline` (the C oracle on one host core, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--genome-mb', type=float, default=25.0)
    p.add_argument('--coverage', type=float, default=30.0)
    p.add_argument('--read-len', type=int, default=100)
    p.add_argument('--ksize', type=int, default=31)
    p.add_argument('--memory', type=float, default=2e9, help='sketch bytes per sample (all bands together)')
    p.add_argument('--case-min', type=int, default=6)
    p.add_argument('--ctrl-max', type=int, default=1)
    p.add_argument('--cpu-reads', type=int, default=150000, help='reads per sample for the CPU baseline leg')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--multi', default='auto', choices=['auto', 'exchange', 'banded'],
                   help='N>1: exchange = shard the reads, hash once, all-to-all the hashes by band (kevlar_amd/shardrun.py); '
                        'banded = every rank hashes all reads and keeps its band; auto = exchange from 4 GPUs up')
    p.add_argument('--count-streams', type=int, default=1,
                   help='N=1: count the three samples concurrently on this many HIP streams.  3 is ~6 %% faster (the '
                        'hashing stage of one sample overlaps the LDS/HBM-bound stages of another) but per-kernel HIP-event '
                        'durations then include time sharing, so the default keeps the launches back to back')
    p.add_argument('--backend', default='nccl', help='nccl (= RCCL) is what the driver runs; gloo lets two ranks share one GPU in tests')
    return p.parse_args()


def prof(lib, name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node {}'.format(args.gpus))
    import torch
    import torch.distributed as dist

    import __graft_entry__
    __graft_entry__.build()
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

    # ---- synthetic trio (same seeds on every rank), packed on the host, uploaded once
    L, k = args.read_len, args.ksize
    genome_len = int(args.genome_mb * 1e6)
    t0 = time.time()
    packed = synth.trio_reads_packed(genome_len, args.coverage, L)
    upload_s = None
    names = ('proband', 'mother', 'father')
    n_reads = packed['proband'].shape[0]
    multi = args.multi if args.multi != 'auto' else ('exchange' if world >= 4 else 'banded')
    exchange = world > 1 and multi == 'exchange'
    if exchange:
        from kevlar_amd import shardrun
        bounds = {n: shardrun.shard_bounds(n_reads, world, rank) for n in names}
        batches = {n: hk.ReadBatch.from_packed(packed[n][bounds[n][0]:bounds[n][1]], L) for n in names}
        run = shardrun.ShardedTrio(k, hk.Counttable)
    else:
        t_up = time.time()
        batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
        lib.kv_synchronize()
        upload_s = time.time() - t_up
    gen_s = time.time() - t0
    nk = L - k + 1
    T = 4
    nbands = world if world > 1 else 0
    band = rank
    mem_per_gpu = args.memory / max(1, world)
    sketches = {n: hk.Counttable(k, mem_per_gpu / T, T) for n in names}

    wall = {'count': 0.0, 'novel': 0.0, 'merge': 0.0}

    def step_exchange():
        # route + exchange of sample i+1 overlap the count of sample i (RCCL runs on its own stream)
        t_a = time.perf_counter()
        for n in names:
            sketches[n].clear()
        kmers = 0
        pending = run.start(batches[names[0]], bounds[names[0]][0], True)
        for i, n in enumerate(names):
            nxt = run.start(batches[names[i + 1]], bounds[names[i + 1]][0], False) if i + 1 < len(names) else None
            kmers += run.finish(pending, sketches[n], keep_for_scan=(i == 0))
            pending = nxt
        t_b = time.perf_counter()
        r, o, a = run.scan([sketches['proband']], [sketches['mother'], sketches['father']], args.case_min, args.ctrl_max)
        t_c = time.perf_counter()
        wall['count'] += t_b - t_a
        wall['novel'] += t_c - t_b
        return kmers, len(r), (r, o, a)

    def step_banded():
        kmers = 0
        t_a = time.perf_counter()
        if world == 1 and args.count_streams > 1:
            # the samples are independent: their counts run on separate HIP streams (host threads), so the
            # ALU-bound hashing stage of one overlaps the LDS/HBM-bound stages of another
            def job(n):
                def count_one():
                    sketches[n].clear()
                    return sketches[n].consume_batch(batches[n], nbands, band)
                return count_one
            for lo in range(0, len(names), args.count_streams):
                kmers += sum(hk.run_concurrently([job(n) for n in names[lo:lo + args.count_streams]]))
        else:
            # controls first, the case sample last: its super-k-mer buckets are still in place when the scan starts
            for n in names[1:] + names[:1]:
                sketches[n].clear()
                kmers += sketches[n].consume_batch(batches[n], nbands, band)
        t_b = time.perf_counter()
        r, o, a, _ = hk.novel_scan(
            [sketches['proband']], [sketches['mother'], sketches['father']], batches['proband'],
            args.case_min, args.ctrl_max, band_mode=1 if world > 1 else 0, nbands=nbands, band=band)
        nhits = len(r)
        t_c = time.perf_counter()
        if world > 1:
            # every band's hits to every rank, sorted on the device (the per-band bit mask that `kevlar unband`
            # would OR together carries no information beyond the hits, so it is not exchanged here)
            from kevlar_amd import bandmerge
            r, o, a = bandmerge.allgather_hits_device(r, o, a, torch.device('cuda', dev_index), staged=(args.backend != 'nccl'))
            nhits = len(r)
        wall['count'] += t_b - t_a
        wall['novel'] += t_c - t_b
        wall['merge'] += time.perf_counter() - t_c
        return kmers, nhits, (r, o, a)

    step = step_exchange if exchange else step_banded

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if exchange:
        # the exchange path has only ever run its RCCL transport with one rank (a one-GPU pool): try one step and,
        # if any rank fails, let every rank fall back to the banded layout instead of losing the measurement
        ok = 1
        try:
            step()
        except Exception as exc:   # noqa: BLE001
            ok = 0
            print('[bench] rank {}: exchange mode failed ({}: {}); falling back to banded'.format(
                rank, type(exc).__name__, exc), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int64, device=coll_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            exchange = False
            multi = 'banded'
            run = None
            batches = {n: hk.ReadBatch.from_packed(packed[n], L) for n in names}
            step = step_banded
    for _ in range(args.warmup):
        step()
    lib.kv_prof_reset()
    lib.kv_prof_enable(1)
    for key in wall:
        wall[key] = 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kmers, nhits, hits = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.kv_prof_enable(0)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- cheap end-to-end sanity on the timed result (parity proper lives in tests/)
    r, o, a = hits
    assert nhits > 0, 'synthetic trio must yield interesting k-mers'
    assert (a[:, 0] >= args.case_min).all() and (a[:, 1:] <= args.ctrl_max).all()
    if world == 1:
        assert kmers == 3 * n_reads * nk
    elif exchange:
        tot = torch.tensor([kmers], dtype=torch.int64, device=coll_device)
        dist.all_reduce(tot)
        assert int(tot.item()) == 3 * n_reads * nk, 'every k-mer of the trio must be counted by exactly one rank' 

    ms_step = elapsed / args.steps * 1e3
    total_reads = 3 * n_reads
    value = total_reads / (elapsed / args.steps)

    # ---- roofline of the dominant kernel: algorithmic bytes per launch / avg HIP-event time
    # (per-read figures from SURVEY.md 8(d); with N bands only 1/N of the k-mers reach this GPU's tables)
    frac_band = 1.0 / max(1, world)
    a_count = n_reads * (L / 4.0 + 2 * T * nk * frac_band)
    a_novel = n_reads * (L / 4.0 + T * 3 * nk * frac_band)
    buf = ctypes.create_string_buffer(4096)
    lib.kv_prof_names(buf, 4096)
    times = {name: prof(lib, name) for name in buf.value.decode().split(',') if name}
    # the count is one logical kernel split over k_bin_* launches (or k_consume on the atomic path)
    count_prefixes = ('k_bin_', 'k_route_', 'k_skm_emit', 'k_skm_split', 'k_skm_count', 'k_skm_loose_count')
    novel_prefixes = ('k_novel_', 'k_skm_novel', 'k_skm_loose_novel', 'k_tile_')
    groups = {'count': [n_ for n_ in times if n_.startswith(count_prefixes) or n_ == 'k_consume'],
              'novel': [n_ for n_ in times if n_.startswith(novel_prefixes)]}
    alg = {}
    for name in groups['count']:
        alg[name] = a_count
    for name in groups['novel']:
        alg[name] = a_novel
    dominant = max(alg, key=lambda n_: times[n_][0])
    ms_tot, launches = times[dominant]
    avg_ms = ms_tot / max(1, launches)
    stage = 'count' if dominant in groups['count'] else 'novel'
    stage_ms = sum(times[n_][0] for n_ in groups[stage]) / max(1, launches)
    achieved = alg[dominant] / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
    traffic = None
    pmc_file = os.path.join(ROOT, 'profiles', 'r1_final', 'pmc_hbm_bytes.json')
    if world == 1 and os.path.exists(pmc_file) and (args.genome_mb, args.coverage, k, args.memory) == (25.0, 30.0, 31, 2e9):
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same workload (profiles/README.md):
        # FETCH_SIZE doubled for the wide coalesced streams of the k_bin_* kernels, as the guide prescribes
        pmc = json.load(open(pmc_file)).get(dominant)
        if pmc:
            fetch = pmc.get('FETCH_SIZE_KB_per_launch_avg', 0.0) * (2.0 if dominant.startswith('k_bin_') else 1.0)
            traffic = int((fetch + pmc.get('WRITE_SIZE_KB_per_launch_avg', 0.0)) * 1024)
    roofline = {
        'bound': 'hbm', 'kernel': dominant, 'stage': stage,
        'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
        'avg_launch_ms': round(avg_ms, 4), 'stage_ms_per_sample': round(stage_ms, 4), 'launches': int(launches),
        'algorithmic_bytes_per_launch': int(alg[dominant]),
        'note': 'achieved = algorithmic bytes of one sample / summed duration of all kernels of that stage; the dominant '
                'kernel hashes (two murmur3 per k-mer) and is VALU-issue-bound per the SQ counters (DESIGN.md 4.1), '
                'so the HBM roofline is an upper bound it cannot approach',
        'kernels_ms_per_step': {name: round(times[name][0] / args.steps, 4) for name in sorted(times)},
        'host_wall_ms_per_step': {key: round(val / args.steps * 1e3, 3) for key, val in wall.items()},
    }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, packed, names, synth)

    if rank == 0:
        out = {
            'metric': 'reads/sec through count+novel (trio, k={})'.format(k),
            'value': round(value, 1), 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 3), 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'u64 hash / u8 counters', 'data': 'synthetic',
            'config': {
                'workload': 'synthetic {:g} Mb trio, {:g}x, {} bp reads, k={}, {} reads/sample, '
                            '{:g} GB Count-Min sketch per sample ({} tables), case-min {}, ctrl-max {}'.format(
                                args.genome_mb, args.coverage, L, k, n_reads, args.memory / 1e9, T,
                                args.case_min, args.ctrl_max),
                'parallelism': 'single band' if world == 1 else ('{} k-mer bands, 1 per GPU; reads sharded, hashes exchanged by band (all-to-all)'.format(world) if exchange else '{} k-mer bands, 1 per GPU; every rank hashes all reads'.format(world)),
                'interesting_kmer_instances': nhits, 'host_generate_pack_upload_s': round(gen_s, 1),
                'packed_reads_upload_s': round(upload_s, 3) if upload_s is not None else None,
                'device': '{} ({} CUs)'.format(torch.cuda.get_device_properties(dev_index).name,
                                               torch.cuda.get_device_properties(dev_index).multi_processor_count),
            },
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, packed, names, synth):
    """The C oracle (oracle/kvoracle.c, one core) on the first --cpu-reads reads of each sample,
    with full-size sketches: count x3 then the novel scan of the proband sample."""
    from oracle import okhmer as ok
    n = min(args.cpu_reads, packed['proband'].shape[0])
    data = {}
    for name in names:
        seqs = synth.unpack_reads(packed[name][:n], args.read_len)
        data[name] = ok.concat_reads(seqs)
    sk = {name: ok.Counttable(args.ksize, args.memory / 4, 4) for name in names}
    t0 = time.perf_counter()
    for name in names:
        ok.consume_reads(sk[name], data[name][0], data[name][1], n)
    hits, _ = ok.novel_scan([sk['proband']], [sk['mother'], sk['father']], data['proband'][0], data['proband'][1],
                            n, args.ksize, args.case_min, args.ctrl_max)
    dt = time.perf_counter() - t0
    return {'value': round(3 * n / dt, 1), 'unit': 'reads/s', 'cores': 1, 'kind': 'port',
            'sample': 'first {} reads of each of the 3 samples, full-size sketches, {:.1f} s'.format(n, dt)}


if __name__ == '__main__':
    main()
