"""Read-sharded trio run over the GPUs of one node: hash once, exchange by band.

kevlar's way to split a trio over N workers is k-mer banding (docs/banding.rst,
kevlar/count.py:62-66, kevlar/novel.py:144-147): worker b owns the hash range of band b -- but
every worker still reads and hashes every read.  On one node that replication is most of a banded
rank's time.  Here rank r holds reads [r*n/N, (r+1)*n/N) of every sample,
hashes them once (kv_route_hashes; or kv_route_distinct, which first combines the repeats inside
the shard into (hash, occurrences) pairs), and one all-to-all over RCCL/xGMI delivers each hash to
the rank that owns its band; the owner counts what it receives (kv_consume_hashes[_weighted]) into
the very sketch the banded run would have built, and scans the case k-mers it received
(kv_novel_scan_hashes).  Hits are all-gathered and sorted back into (read, offset) order.

torch is plumbing: it owns the exchange buffers and moves them; all arithmetic is in the HIP
library.  With the gloo backend (CPU tests, or several ranks sharing one GPU) the buffers are
staged through host memory.

Stream invariant: the library works on its own HIP stream, torch and RCCL on theirs, and the
buffers come from torch's caching allocator without record_stream.  That is safe because every
hand-over is fenced by the HOST: each kv_* entry point used here returns only after
hipStreamSynchronize on the library stream (see include/kvsketch.h), and _Exchange.wait()
synchronises torch's stream before the library reads a received buffer.  A buffer is therefore
never in use on one stream while the other side (or the allocator) touches it.  Anyone making a
kv_* call asynchronous must add record_stream / wait_stream here first.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from kevlar_amd import _lib, khmer as hk


# payload bytes this rank has handed to collectives for OTHER ranks (what crosses xGMI from here): bench.py reads and resets it
SENT = {'bytes': 0}


def shard_bounds(n_reads, world, rank):
    """Reads [lo, hi) of a sample that rank `rank` hashes."""
    return (n_reads * rank) // world, (n_reads * (rank + 1)) // world


class PeerDeclined(Exception):
    """A rank could not produce its part of an exchange (a buffer of the minimizer-sharded layout overflowed: bucket skew).  It says so
    INSIDE a collective every rank takes part in anyway -- a negative count in the size exchange, a marker in its slab of segment
    counts -- so every rank learns it at the same point and all of them take the same fallback; no rank ever waits in a collective the
    others have left."""


def _local_failures():
    """What a rank's own part of an exchange may raise -- a library error (a buffer too small: KvCapacityError; out of device
    memory or any other HIP failure: KvError; a shard the call does not take -- reads of unequal length under a short-record plan,
    say -- which _lib.check() raises as ValueError, or OSError), torch running out of memory for an exchange buffer -- and what must
    therefore never leave a rank between two collectives: the rank says so inside the next collective instead (PeerDeclined above).
    (An argument that is wrong on every rank alike is refused again by the layout the ranks fall back to, on all of them.)"""
    from kevlar_amd._lib import KvCapacityError, KvError
    return (KvCapacityError, KvError, MemoryError, torch.cuda.OutOfMemoryError, ValueError, OSError)


_SAID = set()


def _note_failure(run, where, exc):
    """A rank's own part of an exchange failed and the ranks are about to agree on a fallback: keep the exception on the handle
    (run.last_failure), count it by reason (run.fallback_reasons: tests and bench.py assert on them; run.unexpected_failures counts
    the argument errors -- a mis-wired caller, which would otherwise only show as slower numbers) and say so on stderr once per
    process and reason, whatever the verbosity."""
    import sys
    from kevlar_amd._lib import KvArgError
    reason = '{}: {}'.format(where, type(exc).__name__)
    run.last_failure = exc
    run.fallback_reasons[reason] = run.fallback_reasons.get(reason, 0) + 1
    if isinstance(exc, KvArgError):
        run.unexpected_failures += 1
    if reason not in _SAID or _lib.knob('KV_MEX_VERBOSE'):
        _SAID.add(reason)
        print('[kevlar_amd.shardrun] rank {} declines ({}{}): {}'.format(run.rank, reason, ', UNEXPECTED -- a caller\'s mistake' if isinstance(exc, KvArgError) else '', exc),
              file=sys.stderr, flush=True)


def _test_failure(point, rank):
    """tests: KV_MEX_TEST_DECLINE='<point>:<rank>' makes that rank fail at that point the way the real thing would -- 'emit-oom'
    (no memory for the packed records), 'route-hip' (the library reports a HIP error while combining), 'scan-fail' (the owner's
    scan of its distinct case k-mers fails), 'owner-hip' (the bucket owner's look-up fails); 'emit' / 'route' are the plain
    capacity declines handled where they occur"""
    if _lib.knob('KV_MEX_TEST_DECLINE', '') != '{}:{}'.format(point, rank):
        return
    if point == 'emit-oom':
        raise MemoryError('forced by KV_MEX_TEST_DECLINE')
    from kevlar_amd._lib import KV_ERR_HIP, KvError
    raise KvError(KV_ERR_HIP, 'forced by KV_MEX_TEST_DECLINE')


class _Exchange(object):
    """An all-to-all in flight: wait() returns the received rows."""

    def __init__(self, work, recv, recv_counts, keep):
        self.work, self.recv, self.recv_counts, self._keep = work, recv, recv_counts, keep

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        # the HIP library runs on its own stream: order by the host, but wait for THIS exchange only
        # (a later one may still be in flight on RCCL's stream)
        if self.recv.is_cuda:
            torch.cuda.current_stream().synchronize()
        self._keep = None
        return self.recv


def exchange_rows_async(send, counts, group=None, staged=False, form=0):
    """All-to-all of variable-length row blocks.
    form: a small number every rank must bring alike (which wire format its rows are in); it travels beside the counts, and ranks
    that disagree raise PeerDeclined together, before anything is sent.

    send: tensor [rows, ...] holding the block for rank 0, then rank 1, ... back to back (counts[d] rows for
    rank d), exactly as kv_route_hashes leaves them.  The received blocks (from rank 0, 1, ...) come from the
    returned handle's wait(); `send` must stay untouched until then.  `staged` moves the data through host
    memory (gloo cannot transport device tensors) and completes before returning."""
    world = dist.get_world_size(group)
    assert counts is None or len(counts) == world
    coll_dev = torch.device('cpu') if staged else send.device
    mine = torch.tensor(([-1] * world if counts is None else list(counts)) + [int(form)], dtype=torch.int64, device=coll_dev)     # None: this rank declines
    table = torch.empty(world * (world + 1), dtype=torch.int64, device=coll_dev)
    dist.all_gather_into_tensor(table, mine, group=group)
    table = table.view(world, world + 1).cpu()
    forms, table = table[:, world], table[:, :world]
    if bool((table < 0).any()):
        raise PeerDeclined('ranks {} declined'.format([r for r in range(world) if bool((table[r] < 0).any())]))
    if bool((forms != forms[0]).any()):
        raise PeerDeclined('the ranks bring different wire formats: {}'.format([int(f) for f in forms]))
    rank = dist.get_rank(group)
    recv_counts = [int(table[src, rank]) for src in range(world)]
    packed = send[:sum(counts)]
    SENT['bytes'] += (sum(counts) - counts[rank]) * send.element_size() * int(np.prod(send.shape[1:], dtype=np.int64)) + 8 * (world - 1)
    recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if staged:
        src_host = packed.cpu()
        dst_host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(dst_host, src_host, recv_counts, list(counts), group=group)
        recv.copy_(dst_host)
        if recv.is_cuda:
            torch.cuda.synchronize()
        return _Exchange(None, recv, recv_counts, None)
    work = dist.all_to_all_single(recv, packed, recv_counts, list(counts), group=group, async_op=True)
    return _Exchange(work, recv, recv_counts, send)


def exchange_rows(send, counts, group=None, staged=False):
    """Blocking form of exchange_rows_async: returns (recv, recv_counts)."""
    ex = exchange_rows_async(send, counts, group, staged)
    return ex.wait(), ex.recv_counts


def exchange_slabs(send, in_splits, out_splits, group=None, staged=False):
    """All-to-all of a flat tensor cut at fixed split points known to every rank (no size exchange): returns what arrived,
    source after source.  Blocking."""
    recv = torch.empty(sum(out_splits), dtype=send.dtype, device=send.device)
    SENT['bytes'] += (sum(in_splits) - in_splits[dist.get_rank(group)]) * send.element_size()
    if staged:
        host = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(host, send.cpu(), list(out_splits), list(in_splits), group=group)
        recv.copy_(host)
        if recv.is_cuda:
            torch.cuda.synchronize()
    else:
        dist.all_to_all_single(recv, send, list(out_splits), list(in_splits), group=group)
        if recv.is_cuda:
            torch.cuda.current_stream().synchronize()
    return recv


def exchange_slabs_async(send, in_splits, out_splits, group=None, staged=False):
    """exchange_slabs that returns while the slabs travel (RCCL; the staged form has delivered when it returns): wait() on the
    handle gives what arrived.  `send` must stay untouched until then."""
    if staged:
        return _Exchange(None, exchange_slabs(send, in_splits, out_splits, group, True), None, None)
    recv = torch.empty(sum(out_splits), dtype=send.dtype, device=send.device)
    SENT['bytes'] += (sum(in_splits) - in_splits[dist.get_rank(group)]) * send.element_size()
    work = dist.all_to_all_single(recv, send, list(out_splits), list(in_splits), group=group, async_op=True)
    return _Exchange(work, recv, None, send)


class _Cut(object):
    """A sample whose shard is cut and whose records are on their way to the owners of their minimizer buckets
    (ShardedTrio.cut_minimizer); combine_minimizer() takes it from there.  fallback: the sample travels as `distinct`
    pairs instead (a rank declined) and this is that exchange."""

    def __init__(self, plan=None, got_cnt=None, records=None, fallback=None, batch=None, base=0):
        self.plan, self.got_cnt, self.records, self.fallback, self.batch, self.base = plan, got_cnt, records, fallback, batch, base


def gather_rows(rows, n_valid, fill, group=None, staged=False):
    """All-gather of each rank's first n_valid rows, padded with `fill` to the longest; returns
    (gathered [world * longest, ...], total valid)."""
    world = dist.get_world_size(group)
    coll_dev = torch.device('cpu') if staged else rows.device
    mine = torch.tensor([n_valid], dtype=torch.int64, device=coll_dev)
    sizes = torch.empty(world, dtype=torch.int64, device=coll_dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = [int(v) for v in sizes.cpu()]
    if min(sizes) < 0:          # a rank could not produce its rows: every rank learns it here, nobody enters the gather
        raise PeerDeclined('ranks {} declined'.format([r for r in range(world) if sizes[r] < 0]))
    SENT['bytes'] += (world - 1) * (n_valid * rows.element_size() * int(np.prod(rows.shape[1:], dtype=np.int64)) + 8)
    longest = max(max(sizes), 1)
    padded = torch.full((longest,) + tuple(rows.shape[1:]), fill, dtype=rows.dtype, device=rows.device)
    padded[:n_valid] = rows[:n_valid]
    out = torch.empty((world * longest,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    if staged:
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host_out, padded.cpu(), group=group)
        out.copy_(host_out)
    else:
        dist.all_gather_into_tensor(out, padded, group=group)
    return out, sum(sizes)


class ShardedTrio(object):
    """One rank of a read-sharded count + novel run.  Every rank calls the same methods in the
    same order (they contain collectives).

    sketch_cls: kevlar_amd.khmer sketch class; the rank's sketches are ordinary sketches of 1/N of
    the memory, i.e. exactly band `rank` of an N-band run."""

    def __init__(self, ksize, sketch_cls=hk.Counttable, group=None, staged=None, device=None):
        self.ksize = int(ksize)
        self.sketch_cls = sketch_cls
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = dist.get_backend(group)
        self.staged = (backend != 'nccl') if staged is None else bool(staged)
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self._send = {}          # words per item -> free send buffers [cap, words] (one per exchange in flight)
        self.case_items = None   # (hash, tag) or (hash, occurrences) pairs of the case k-mers this rank owns
        self.case_items_weighted = False
        self.timing = {'route': 0.0, 'exchange': 0.0, 'count': 0.0, 'scan': 0.0, 'gather': 0.0}
        self.fallbacks, self.scan_fallbacks = 0, 0      # agreed fallbacks this rank took part in (its own failure or a peer's)
        self.fallback_reasons = {}                      # this rank's OWN failures behind them: 'where: ExceptionType' -> count (_note_failure)
        self.unexpected_failures = 0                    # ... of which argument errors (a caller's mistake, not skew or memory)
        self.last_failure = None
        # north_star's merge: (int32 words on the device, bits per read) or None.  When set, every scan_*() also builds this rank's
        # bit mask of the interesting k-mer occurrences IT found -- the owners' findings are disjoint (a band, a set of minimizer
        # buckets, a shard of the reads) -- and all-reduces it (SUM of disjoint 0/1 bits = their OR: docs/banding.rst,
        # kevlar/unband.py:41-77); the caller holds the result against the gathered hits (bandmerge.mask_to_hits).
        self.band_mask = None

    def _send_buffer(self, cap, words):
        """A send buffer nobody is using: an exchange in flight keeps its own until finish()."""
        free = self._send.setdefault(words, [])
        for i, buf in enumerate(free):
            if buf.shape[0] >= cap:
                return free.pop(i)
        if cap * words * 8 >= (1 << 30):
            # torch allocates beside the library: the table buffers the library keeps for future sketches are worth less than this
            from kevlar_amd import _lib
            _lib.load().kv_table_cache_trim()
        return torch.empty((cap, words), dtype=torch.int64, device=self.device)

    def start_minimizer(self, batch, read_index_base, n_reads_global, read_len, keep_scan=False):
        """start() for the minimizer-sharded layout: the shard is cut into super-k-mer records (kv_mex_emit), the records go
        to the rank that owns their minimizer bucket (first all-to-all; fixed split points, so no sizes are exchanged), that
        rank combines every occurrence of a k-mer -- from whichever shard -- at the sample's full coverage (kv_mex_route),
        and the (hash, occurrences) pairs go on to the band owners exactly as start(distinct=True) sends them.  A shard of
        1/8 of the reads has little to combine on its own (49 % of its k-mers are distinct against 20 % of the sample's); this
        way a rank hashes 1/N of the sample's DISTINCT k-mers.  = combine_minimizer(cut_minimizer(...)); callers with several
        samples interleave the two halves so that one sample's records travel while the next one's shard is cut."""
        return self.combine_minimizer(self.cut_minimizer(batch, read_index_base, n_reads_global, read_len, short=not keep_scan), keep_scan)

    def cut_minimizer(self, batch, read_index_base, n_reads_global, read_len, short=False):
        """First half of start_minimizer(): cut the shard, pack what was cut, tell every owner how much is coming (the slab of
        segment counts: blocking, small) and START the all-to-all of the records.  Returns a _Cut for combine_minimizer().
        Every rank calls the halves of every sample in the same order (they are collectives).  short (every rank alike): the
        sample is not the one combine_minimizer(keep_scan) keeps, so its records travel without read positions where the
        plan has such records (16 bytes instead of 24: hk.mex_plan); a rank whose shard cannot be cut that way (reads of
        unequal length) declines like one whose segments overflowed, and the sample travels as pairs."""
        t0 = time.perf_counter()
        forced = _lib.knob('KV_MEX_TEST_DECLINE', '')          # tests: 'emit:RANK' / 'route:RANK' makes that rank decline there
        plan = hk.mex_plan(self.sketch_cls, self.ksize, n_reads_global, read_len, self.world, short=short)
        seg = torch.empty(int(plan.seg_words), dtype=torch.int64, device=self.device)
        cnt = torch.empty(int(plan.cnt_entries), dtype=torch.int32, device=self.device)
        emitted = forced != 'emit:{}'.format(self.rank)
        # only the filled part of the segments travels (the segments' capacity is twice the expected fill): the cut and the
        # packing are one call and one wait; the counts go first (fixed split points), and say how many records every source sends
        per_dest, packed = None, None
        if emitted:
            try:
                _test_failure('emit-oom', self.rank)
                packed = torch.empty(int(plan.seg_words) // 2 + 1024, dtype=torch.int64, device=self.device)
                per_dest, fitted = hk.mex_emit_pack(batch, plan, read_index_base, seg.data_ptr(), cnt.data_ptr(), packed.data_ptr(), packed.shape[0])
                if not fitted:                              # fuller than expected: a buffer of the segments' full size always fits
                    packed = torch.empty(int(plan.seg_words), dtype=torch.int64, device=self.device)
                    per_dest = hk.mex_pack(plan, seg.data_ptr(), cnt.data_ptr(), packed.data_ptr())
            except _local_failures() as e:                  # records outside their exchange segment (minimizer skew), no memory for the packed
                emitted, packed = False, None               # copy, a HIP error, a shard a short-record plan cannot cut: the peers must hear of it, in the slab below
                _note_failure(self, 'cut', e)
        if not emitted:
            cnt.fill_(-1)                                   # the marker every destination finds in this rank's slab of counts
        t1 = time.perf_counter()
        recw = int(plan.recw)
        width = [int(plan.c_lo[d + 1]) - int(plan.c_lo[d]) for d in range(self.world)]
        mine = width[self.rank] * int(plan.nwg1)
        got_cnt = exchange_slabs(cnt, [w * int(plan.nwg1) for w in width], [mine] * self.world, self.group, self.staged)
        per_src = got_cnt.view(self.world, mine)
        summary = torch.cat([per_src.clamp(min=0, max=int(plan.cap1)).sum(dim=1, dtype=torch.int64), (per_src < 0).any(dim=1).to(torch.int64)]).cpu()
        if bool(summary[self.world:].any()):
            # a rank's cut did not fit its exchange segments; every rank has just seen its marker: the sample travels as the
            # (hash, occurrences) pairs of each rank's own deduplicated shard instead (what arrives at the band owners is the same)
            del seg, cnt, got_cnt
            self.fallbacks += 1
            self.timing['route'] += time.perf_counter() - t0
            return _Cut(fallback=self.start(batch, read_index_base, False, distinct=True))
        from_src = [int(v) for v in summary[:self.world]]
        records = exchange_slabs_async(packed[:sum(per_dest) * recw], [n * recw for n in per_dest], [n * recw for n in from_src], self.group, self.staged)
        records.packed = packed                             # (the view above is what travels: the buffer stays until wait())
        del seg, cnt
        self.timing['route'] += t1 - t0
        self.timing['exchange'] += time.perf_counter() - t1
        return _Cut(plan, got_cnt, records, None, batch, read_index_base)

    def combine_minimizer(self, cut, keep_scan=False):
        """Second half of start_minimizer(): wait for the records, combine what the N shards hold of this rank's buckets and start
        the exchange of the (hash, occurrences) pairs; returns the handle finish() takes.  keep_scan (the case sample, combined
        last): the combined buckets stay on the device for scan_minimizer()."""
        self.owner_can_scan = False
        if cut.fallback is not None:
            return cut.fallback
        forced = _lib.knob('KV_MEX_TEST_DECLINE', '')
        plan, got_cnt = cut.plan, cut.got_cnt
        t1 = time.perf_counter()
        got_seg = cut.records.wait()
        cut.records = None
        t2 = time.perf_counter()
        share = int(plan.n_kmers_global) // self.world
        cap = share + share // 4 + (1 << 20)
        send, counts = None, None
        if forced != 'route:{}'.format(self.rank):
            try:
                _test_failure('route-hip', self.rank)
                send = self._send_buffer(cap, 2)
                counts, _ = hk.mex_route(plan, self.rank, got_seg.data_ptr(), got_cnt.data_ptr(), self.world, send.data_ptr(), send.shape[0], compact=True,
                                         keep_scan=keep_scan)
            except _local_failures() as e:                  # more k-mers in this rank's buckets than its pair buffer holds (bucket skew), the
                counts = None                               # stream arena or the distinct list out of memory, no room for the pair buffer
                _note_failure(self, 'combine', e)
        del got_seg, got_cnt
        cut.got_cnt = None
        # KV_MEX_PAIRS=9 (every rank alike): the pairs travel in 9 bytes each -- the hash, the occurrences as a byte saturated at 255 (no
        # counter holds more), the block's exact total in its head word (kv_pairs_pack).  A quarter fewer bytes on the links for a pass
        # over the pairs on either side: 646 instead of 875 MB and 6.56 instead of 5.88 ms per rank of config 2 at N = 8 -- which of the
        # two is cheaper is a property of the links nobody has measured, so 16 bytes stay the default
        # The form is this rank's word in the size exchange (exchange_rows_async(form=)): ranks that disagree -- the switch set on some
        # of them -- all learn it there and fall back together, nobody decodes one form as the other.
        travelling, wcounts = None, None
        form = 9 if _lib.knob('KV_MEX_PAIRS', '16') == '9' else 16
        if counts is not None and form == 9:
            try:
                n_pairs = sum(counts)
                travelling = torch.empty(n_pairs + n_pairs // 8 + 2 * self.world + 8, dtype=torch.int64, device=self.device)
                wcounts = hk.pairs_pack(send.data_ptr(), counts, travelling.data_ptr(), travelling.shape[0])
            except _local_failures() as e:
                counts, travelling, wcounts = None, None, None
                _note_failure(self, 'pairs-pack', e)
        t3 = time.perf_counter()
        try:
            if send is None:                                # (nothing travels from a rank that declines: any tensor carries its "-1")
                send = torch.empty((1, 2), dtype=torch.int64, device=self.device)
            if travelling is not None:
                ex = exchange_rows_async(travelling, wcounts, self.group, self.staged, form=form)
                ex.travelling = travelling                  # (kept until wait())
                ex.packed_pairs = True
            else:
                ex = exchange_rows_async(send, counts, self.group, self.staged, form=form)      # counts None: this rank declines, inside the size exchange
        except PeerDeclined:
            if send.shape[0] > 1:
                self._send[2].append(send)
            self.fallbacks += 1
            self.timing['route'] += t3 - t2
            self.timing['exchange'] += (t2 - t1) + (time.perf_counter() - t3)
            return self.start(cut.batch, cut.base, False, distinct=True)
        ex.send_buffer = send
        ex.weighted = True
        self.owner_can_scan = bool(keep_scan)
        self.timing['route'] += t3 - t2
        self.timing['exchange'] += (t2 - t1) + (time.perf_counter() - t3)
        return ex

    def start(self, batch, read_index_base, with_tags, distinct=False):
        """Hash this rank's shard of a sample and start delivering every hash to its band's owner.
        Returns a handle for finish(); the next sample's start() may run while the exchange flies.
        with_tags: items are (hash, tag) pairs, what scan() needs of a case sample.
        distinct: items are (hash, occurrences in the shard) pairs, one per distinct k-mer of a
        super-k-mer bucket (kv_route_distinct) -- fewer items to send and to count; not with tags."""
        assert not (with_tags and distinct)
        words = 2 if (with_tags or distinct) else 1
        cap = max(batch.num_kmers(self.ksize), 1)
        send = self._send_buffer(cap, words)
        t0 = time.perf_counter()
        if distinct:
            counts = hk.route_distinct(batch, self.sketch_cls, self.ksize, self.world, send.data_ptr(), send.shape[0])
        else:
            counts = hk.route_hashes(batch, self.sketch_cls, self.ksize, self.world, read_index_base, with_tags,
                                     send.data_ptr(), send.shape[0])
        t1 = time.perf_counter()
        ex = exchange_rows_async(send, counts, self.group, self.staged)
        ex.send_buffer = send
        ex.weighted = bool(distinct)
        self.timing['route'] += t1 - t0
        self.timing['exchange'] += time.perf_counter() - t1
        return ex

    def finish(self, ex, sketch, keep_for_scan=False):
        """Wait for the exchange and count what arrived into `sketch` (= band `rank`); sketch None
        only receives.  keep_for_scan: these are the (hash, tag) pairs of a case sample -- keep them
        for scan().  Returns the number of k-mer occurrences counted on this rank."""
        t0 = time.perf_counter()
        recv = ex.wait()
        self._send[ex.send_buffer.shape[1]].append(ex.send_buffer)     # delivered: the buffer is free again
        t1 = time.perf_counter()
        occurrences = None
        if getattr(ex, 'packed_pairs', False):
            # 9-byte pairs (combine_minimizer): back into the 16-byte form the kernels read; what they stand for comes from the blocks' heads
            # (a failure here -- no memory for the unpacked pairs, a HIP error -- comes after the exchange, with the next collective
            # ahead: the ranks agree right here, in one small all-reduce of this opt-in path, and stop together)
            words = list(ex.recv_counts)
            failed = None
            try:
                _test_failure('unpack-fail', self.rank)
                pairs = torch.empty((max(1, sum(max(0, (w - 1) * 8 // 9) for w in words)), 2), dtype=torch.int64, device=recv.device)
                per_src, occurrences = hk.pairs_unpack(recv.data_ptr(), words, pairs.data_ptr(), pairs.shape[0])
                recv = pairs[:sum(per_src)]
            except _local_failures() as e:
                failed = e
                _note_failure(self, 'pairs-unpack', e)
            ex.travelling = None
            flag = torch.tensor([1 if failed is not None else 0], dtype=torch.int64, device=torch.device('cpu') if self.staged else self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if int(flag.item()):
                self._scan_declined('unpacking the 9-byte pairs', failed, 'a rank could not unpack what it received')
        n = recv.shape[0]
        if sketch is None:
            n = 0
        elif n and ex.weighted:
            n = sketch.consume_hashes_weighted(recv.data_ptr(), n)
            if occurrences is not None:
                n = occurrences
        elif n:
            sketch.consume_hashes(recv.data_ptr(), n, recv.shape[1])
        self.timing['exchange'] += t1 - t0
        self.timing['count'] += time.perf_counter() - t1
        if keep_for_scan:
            self.case_items = recv
            self.case_items_weighted = ex.weighted
        return n

    def count_sample(self, sketch, batch, read_index_base=0, keep_for_scan=False, distinct=False, minimizer=None):
        """start() + finish() for one sample: `batch` is this rank's shard of its reads (global index of
        its first read = read_index_base).  A case sample kept `distinct` is scanned with
        scan_distinct(), a tagged one with scan().  minimizer = (reads of the whole sample, read length): the
        minimizer-sharded layout (start_minimizer); what arrives is what `distinct` delivers."""
        if minimizer is not None:
            return self.finish(self.start_minimizer(batch, read_index_base, int(minimizer[0]), int(minimizer[1]), keep_scan=keep_for_scan), sketch, keep_for_scan)
        if not distinct:
            return self.finish(self.start(batch, read_index_base, keep_for_scan), sketch, keep_for_scan)
        return self.finish(self.start(batch, read_index_base, False, distinct=True), sketch, keep_for_scan)

    def _owners_scan(self, what, call):
        """A band owner's part of a scan (the library call `call`) whose rows every rank is about to gather: a failure here -- more hits
        than the buffer holds, no memory -- must not leave this rank alone outside the gather the others are entering.  Returns
        (rows written, None) or (-1, the error): -1 travels as this rank's row count, gather_rows() raises PeerDeclined on EVERY rank
        at the same point, and _scan_declined() turns that into this rank's own error (or, on the others, a clear one)."""
        try:
            _test_failure('scan-fail', self.rank)
            return call(), None
        except _local_failures() as exc:
            return -1, exc

    def _merge_mask(self, reads_t, offs_t):
        """this rank's findings -- device tensors of global read index and k-mer offset -- as bits of self.band_mask, then the
        all-reduce over the ranks (RCCL on the device; gloo through the host)"""
        if self.band_mask is None:
            return
        from kevlar_amd import bandmerge
        mask, stride = self.band_mask
        mask.zero_()
        if reads_t.numel():
            idx = reads_t.to(torch.int64) * int(stride) + offs_t.to(torch.int64)
            bit = torch.bitwise_left_shift(torch.ones_like(idx), idx & 31).to(torch.int32)       # (bit 31 wraps to the sign bit: the words are bit patterns)
            mask.index_add_(0, idx >> 5, bit)                                                    # (an occurrence is found once: distinct bits, the sum is their OR)
        torch.cuda.synchronize()
        if self.staged:
            host = mask.cpu()
            bandmerge.allreduce_mask(host, self.group)
            mask.copy_(host)
        else:
            bandmerge.allreduce_mask(mask, self.group)
        torch.cuda.synchronize()

    def _merge_mask_np(self, reads, offs):
        if self.band_mask is not None:
            self._merge_mask(torch.from_numpy(np.ascontiguousarray(reads, dtype=np.int64)).to(self.device),
                             torch.from_numpy(np.ascontiguousarray(offs).astype(np.int64)).to(self.device))

    def _merge_mask_tags(self, tags, n, skip=None):
        """_merge_mask for findings held as tags (read << 16 | offset), minus the reads in `skip` (numpy, sorted)"""
        if self.band_mask is None:
            return
        t = tags[:max(int(n), 0)]
        reads_t, offs_t = t >> 16, t & 0xffff
        if skip is not None and len(skip) and reads_t.numel():
            keep = ~torch.isin(reads_t, torch.from_numpy(np.asarray(skip, dtype=np.int64)).to(reads_t.device))
            reads_t, offs_t = reads_t[keep], offs_t[keep]
        self._merge_mask(reads_t, offs_t)

    @staticmethod
    def _scan_declined(what, err, declined):
        if err is not None:
            raise err
        raise RuntimeError('{}: {} -- its error is in that rank\'s log; every rank stops here, together'.format(what, declined))

    def scan(self, cases, controls, case_min, ctrl_max):
        """kmer_is_interesting() over the case k-mers this rank owns, then gather: every rank returns
        the complete (read, offset, abund[n, S]) hit arrays in (read, offset) order."""
        assert self.case_items is not None and not self.case_items_weighted, 'count_sample(..., keep_for_scan=True) first'
        S = len(cases) + len(controls)
        items = self.case_items
        n = items.shape[0]
        t0 = time.perf_counter()
        cap = max(min(n, 1 << 26), 1)
        tags = torch.empty(cap, dtype=torch.int64, device=self.device)
        abund = torch.empty((cap, S), dtype=torch.uint8, device=self.device)
        n_hits, err = self._owners_scan('scan', lambda: hk.novel_scan_hashes(cases, controls, items.data_ptr(), n, case_min, ctrl_max,
                                                                              tags.data_ptr(), abund.data_ptr(), cap) if n else 0)
        t1 = time.perf_counter()
        try:
            all_tags, total = gather_rows(tags, n_hits, -1, self.group, self.staged)
        except PeerDeclined as declined:
            self._scan_declined('scan', err, declined)
        all_abund, _ = gather_rows(abund, n_hits, 0, self.group, self.staged)
        torch.cuda.synchronize()
        self._merge_mask_tags(tags, n_hits)                # (k-mers of reads the scan skips never leave kv_novel_scan_hashes: no flagged tag among these)
        r, o, a = hk.hits_from_tagged(all_tags.data_ptr(), all_abund.data_ptr(), all_tags.shape[0], total, S)
        self.timing['scan'] += t1 - t0
        self.timing['gather'] += time.perf_counter() - t1
        return r, o, a

    def scan_distinct(self, cases, controls, case_min, ctrl_max, batch, read_index_base):
        """The scan when the case sample travelled as (hash, occurrences) pairs: no tag says where a k-mer came
        from, so the answer goes back as a set.  Every band owner tests the distinct case k-mers it received, once
        each; the interesting ones -- a few hundred thousand hashes with their abundances -- are all-gathered, and
        every rank looks the k-mers of its own shard (`batch`, first read = read_index_base) up in that set
        (kv_novel_scan_set).  The shards' hits, all-gathered in rank order, are the complete hit arrays in
        (read, offset) order, as scan() returns them."""
        from kevlar_amd import bandmerge
        assert self.case_items is not None and self.case_items_weighted, 'count_sample(..., keep_for_scan=True, distinct=True) first'
        S = len(cases) + len(controls)
        items = self.case_items
        n = items.shape[0]
        t0 = time.perf_counter()
        cap = max(min(n, 1 << 26), 1)
        hashes = torch.empty(cap, dtype=torch.int64, device=self.device)
        abund = torch.empty((cap, S), dtype=torch.uint8, device=self.device)
        n_mine, err = self._owners_scan('scan', lambda: hk.novel_scan_distinct(cases, controls, items.data_ptr(), n, case_min, ctrl_max,
                                                                               hashes.data_ptr(), abund.data_ptr(), cap) if n else 0)
        t1 = time.perf_counter()
        try:
            all_hashes, _ = gather_rows(hashes, n_mine, -1, self.group, self.staged)
        except PeerDeclined as declined:
            self._scan_declined('scan of the distinct case k-mers', err, declined)
        all_abund, _ = gather_rows(abund, n_mine, 0, self.group, self.staged)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r, o, a = hk.novel_scan_set(batch, self.sketch_cls, self.ksize, S, all_hashes.data_ptr(), all_abund.data_ptr(),
                                    all_hashes.shape[0])
        t3 = time.perf_counter()
        self._merge_mask_np(np.asarray(r).astype(np.int64) + int(read_index_base), o)
        r, o, a = bandmerge.allgather_hits_device(np.asarray(r).astype(np.int64) + int(read_index_base), o, a, self.device,
                                                  self.group, self.staged)
        self.timing['scan'] += (t1 - t0) + (t3 - t2)
        self.timing['gather'] += (t2 - t1) + (time.perf_counter() - t3)
        return r, o, a


    def scan_minimizer(self, cases, controls, case_min, ctrl_max, batch, read_index_base):
        """The scan when the case sample went through the minimizer layout and was combined last with keep_scan: the band owners
        judge the distinct k-mers they received (as in scan_distinct), the interesting hashes are all-gathered, and then the OWNERS
        OF THE MINIMIZER BUCKETS answer -- every occurrence of a k-mer sits in their combined buckets with its global (read, offset),
        and they kept its hash (kv_mex_scan_set) -- instead of every rank hashing its whole shard again.  The hits are gathered as
        (tag, abundances) rows and sorted like scan()'s; hits on reads the scan skips (bytes outside ACGT: every rank contributes
        the indices of its shard's) are dropped.  Every rank must be able to answer, or none does: the ranks agree in one small
        all-gather and otherwise take scan_distinct()'s second half."""
        assert self.case_items is not None and self.case_items_weighted, 'count_sample(..., keep_for_scan=True) through the minimizer layout first'
        from kevlar_amd._lib import KvCapacityError
        S = len(cases) + len(controls)
        items = self.case_items
        n = items.shape[0]
        t0 = time.perf_counter()
        cap = max(min(n, 1 << 26), 1)
        hashes = torch.empty(cap, dtype=torch.int64, device=self.device)
        abund = torch.empty((cap, S), dtype=torch.uint8, device=self.device)
        n_mine, err = self._owners_scan('scan', lambda: hk.novel_scan_distinct(cases, controls, items.data_ptr(), n, case_min, ctrl_max,
                                                                               hashes.data_ptr(), abund.data_ptr(), cap) if n else 0)
        t1 = time.perf_counter()
        try:
            all_hashes, _ = gather_rows(hashes, n_mine, -1, self.group, self.staged)
        except PeerDeclined as declined:
            self._scan_declined('scan of the distinct case k-mers', err, declined)
        all_abund, _ = gather_rows(abund, n_mine, 0, self.group, self.staged)
        flagged = batch.flagged_reads() + int(read_index_base)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        # this rank's buckets against the set; how many hits there can be is not known in advance: a buffer sized by the set, again if short
        n_hits, tags, rows = -1, None, None
        if getattr(self, 'owner_can_scan', False):
            hit_cap = max(1 << 16, 64 * int(all_hashes.shape[0]) // max(1, self.world))
            for _ in range(2):
                try:
                    _test_failure('owner-hip', self.rank)
                    tags = torch.empty(hit_cap, dtype=torch.int64, device=self.device)
                    rows = torch.empty((hit_cap, S), dtype=torch.uint8, device=self.device)
                    n_hits = hk.mex_scan_set(self.sketch_cls, self.ksize, S, all_hashes.data_ptr(), all_abund.data_ptr(), all_hashes.shape[0],
                                             tags.data_ptr(), rows.data_ptr(), hit_cap)
                    break
                except KvCapacityError as exc:
                    n_hits = -1
                    if 'exceed the buffer' not in str(exc):
                        break
                    hit_cap *= 16
                except _local_failures() as exc:            # no memory for the rows, a HIP error: this owner cannot answer; the ranks agree below
                    n_hits = -1
                    _note_failure(self, 'owner-scan', exc)
                    break
        t3 = time.perf_counter()
        coll_dev = torch.device('cpu') if self.staged else self.device
        mine = torch.tensor([n_hits, len(flagged)], dtype=torch.int64, device=coll_dev)
        table = torch.empty(2 * self.world, dtype=torch.int64, device=coll_dev)
        dist.all_gather_into_tensor(table, mine, group=self.group)
        table = table.view(self.world, 2).cpu()
        if bool((table[:, 0] < 0).any()):
            # an owner cannot answer: every rank looks its own shard up in the set, as scan_distinct() does
            self.scan_fallbacks += 1
            from kevlar_amd import bandmerge
            r, o, a = hk.novel_scan_set(batch, self.sketch_cls, self.ksize, S, all_hashes.data_ptr(), all_abund.data_ptr(), all_hashes.shape[0])
            t4 = time.perf_counter()
            self._merge_mask_np(np.asarray(r).astype(np.int64) + int(read_index_base), o)
            r, o, a = bandmerge.allgather_hits_device(np.asarray(r).astype(np.int64) + int(read_index_base), o, a, self.device, self.group, self.staged)
            self.timing['scan'] += (t1 - t0) + (t4 - t2)
            self.timing['gather'] += (t2 - t1) + (time.perf_counter() - t4)
            return r, o, a
        all_tags, total = gather_rows(tags, n_hits, -1, self.group, self.staged)
        all_rows, _ = gather_rows(rows, n_hits, 0, self.group, self.staged)
        skip = None
        if int(table[:, 1].sum()):
            mine_f = torch.from_numpy(flagged).to(self.device)
            skip, _ = gather_rows(mine_f, len(flagged), -1, self.group, self.staged)
            skip = np.unique(skip.cpu().numpy())
            skip = skip[skip >= 0]
        torch.cuda.synchronize()
        self._merge_mask_tags(tags, n_hits, skip)
        r, o, a = hk.hits_from_tagged(all_tags.data_ptr(), all_rows.data_ptr(), all_tags.shape[0], total, S)
        if skip is not None and len(skip):
            keep = ~np.isin(np.asarray(r), skip)
            r, o, a = np.asarray(r)[keep], np.asarray(o)[keep], np.asarray(a)[keep]
        self.timing['scan'] += (t1 - t0) + (t3 - t2)
        self.timing['gather'] += (t2 - t1) + (time.perf_counter() - t3)
        return r, o, a
