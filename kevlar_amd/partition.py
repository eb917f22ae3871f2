"""`kevlar partition`: group reads that share interesting k-mers into connected components and label them
`kvcc=N`, largest component first (kevlar/partition.py:15-80, kevlar/cli/partition.py).

The components come from the device union-find behind kevlar_amd.ReadGraph; this driver only sequences the
phases, reports their wall times with the reference's log lines, and renders each partition to text once."""
import kevlar_amd
from kevlar_amd.sequence import format_augmented_fastx


def _phase(timer, key, before, after_format):
    """context: log `before`, time the block, log after_format.format(seconds)"""
    class _Phase(object):
        def __enter__(self):
            timer.start(key)
            kevlar_amd.plog(*before)

        def __exit__(self, exc_type, exc, tb):
            if exc_type is None:
                kevlar_amd.plog('[kevlar::partition]', after_format.format(timer.stop(key)))
    return _Phase()


def write_gml(graph, outfilename):
    """The read graph as GML: one node per read name, one edge per pair of reads that share a retained interesting
    k-mer (what kevlar.to_gml / networkx.write_gml would hold, kevlar/__init__.py:115-120, minus the Record objects
    it would try to stringize)."""
    if not outfilename.endswith('.gml'):
        kevlar_amd.plog('[kevlar] WARNING: GML files usually need extension .gml')
    names = list(graph)
    index = {name: i for i, name in enumerate(names)}
    with open(outfilename, 'w') as out:
        out.write('graph [\n')
        for name in names:
            out.write('  node [\n    id {}\n    label "{}"\n  ]\n'.format(index[name], name.replace('"', '&#34;')))
        for a, b in graph.edge_list():
            out.write('  edge [\n    source {}\n    target {}\n  ]\n'.format(index[a], index[b]))
        out.write(']\n')
    kevlar_amd.plog('[kevlar] graph written to {}'.format(outfilename))


def partition(readstream, strict=False, minabund=None, maxabund=None, dedup=True, gmlfile=None):
    """Yield (N, reads) for N = 1, 2, ...; every read's name gets ' kvcc=N' appended."""
    timer = kevlar_amd.Timer()
    timer.start()
    graph = kevlar_amd.ReadGraph()
    with _phase(timer, 'loadreads', ('[kevlar::partition] Loading reads',), 'Reads loaded in {:.2f} sec'):
        graph.load(readstream, minabund=minabund, maxabund=maxabund)
    mode = ('[kevlar::partition]', 'Building read graph in {:s} mode'.format('strict' if strict else 'relaxed'))
    with _phase(timer, 'buildgraph', mode, 'Graph built in {:.2f} sec'):
        graph.populate_edges(strict=strict)
    if gmlfile:
        write_gml(graph, gmlfile)
    with _phase(timer, 'partition', ('[kevlar::partition] Partition readgraph',), 'Partitioning done in {:.2f} sec'):
        for number, members in enumerate(graph.partitions(dedup, minabund, maxabund, abundfilt=True), 1):
            reads = [graph.get_record(name) for name in members]
            for read in reads:
                read.name = '{} kvcc={:d}'.format(read.name, number)
            yield number, reads
    kevlar_amd.plog('[kevlar::partition]', 'Total time: {:.2f} seconds'.format(timer.stop()))


def _fixed_width(blob, offs):
    """the strings blob[offs[i]:offs[i + 1]] as one numpy bytes array (NUL padded): compares and sorts like the strings do.
    Rows are gathered whole from a sliding window over the blob (one index per string, not one per byte) and the bytes behind
    each string's end are zeroed; strings of one length back to back are just the blob reshaped."""
    import numpy as np
    offs = np.asarray(offs, dtype=np.int64)
    lens = np.diff(offs)
    width = max(1, int(lens.max())) if len(lens) else 1
    raw = np.frombuffer(blob, dtype=np.uint8)
    if len(raw) == 0 or len(lens) == 0:
        return np.zeros(len(lens), dtype='S1')
    kind = 'S{:d}'.format(width)
    if int(lens.min()) == width and offs[0] == 0 and int(offs[-1]) == len(lens) * width and len(raw) >= len(lens) * width:
        return np.ascontiguousarray(raw[:len(lens) * width]).view(kind).reshape(-1)
    padded = np.concatenate((raw, np.zeros(width, dtype=np.uint8)))
    windows = np.lib.stride_tricks.as_strided(padded, shape=(len(raw) + 1, width), strides=(1, 1), writeable=False)
    out = windows[np.minimum(offs[:-1], len(raw))]                  # a copy: one row of `width` bytes per string
    out[np.arange(width, dtype=np.int64)[None, :] >= lens[:, None]] = 0
    return np.ascontiguousarray(out).view(kind).reshape(-1)


_COMP_LUT = None


def _complement_lut():
    import numpy as np
    from kevlar_amd.sequence import _COMPLEMENT
    global _COMP_LUT
    if _COMP_LUT is None:
        lut = np.arange(256, dtype=np.uint8)
        for src, dst in _COMPLEMENT.items():
            if src < 256:
                lut[src] = dst
        _COMP_LUT = lut
    return _COMP_LUT


def _same_canonical(seqs, seq_offs, a, b):
    """bool per pair: reads a[j] and b[j] have the same kevlar_amd.revcommin() (kv_canonical_reads_equal builds both canonical
    forms and compares them)"""
    import ctypes
    import numpy as np
    from kevlar_amd import _lib
    a, b = np.ascontiguousarray(a, dtype=np.uint64), np.ascontiguousarray(b, dtype=np.uint64)
    same = np.zeros(len(a), dtype=np.uint8)
    if len(a):
        offs = np.ascontiguousarray(seq_offs, dtype=np.uint64)
        blob = bytes(seqs) if not isinstance(seqs, bytes) else seqs
        _lib.check(_lib.load().kv_canonical_reads_equal(
            ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p), offs.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p),
            b.ctypes.data_as(ctypes.c_void_p), len(a), _complement_lut().ctypes.data_as(ctypes.c_void_p), same.ctypes.data_as(ctypes.c_void_p)))
    return same.astype(bool)


def _canonical_hashes(seqs, seq_offs, reads):
    """two independent 64-bit hashes of min(sequence, reverse complement) -- kevlar_amd.revcommin(), the key partition()
    dedups by -- for the given reads: kv_canonical_read_hashes, host threads over the reads (the numpy form -- byte matrices
    padded to 64-bit words, strands chosen and hashed by word columns -- took 1.3 s for 2 M reads of 100 bases)."""
    import ctypes
    import numpy as np
    from kevlar_amd import _lib
    offs = np.ascontiguousarray(seq_offs, dtype=np.uint64)
    reads = np.ascontiguousarray(reads, dtype=np.uint64)
    h1 = np.zeros(len(reads), dtype=np.uint64)
    h2 = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        blob = bytes(seqs) if not isinstance(seqs, bytes) else seqs
        _lib.check(_lib.load().kv_canonical_read_hashes(
            ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p), offs.ctypes.data_as(ctypes.c_void_p), reads.ctypes.data_as(ctypes.c_void_p),
            len(reads), _complement_lut().ctypes.data_as(ctypes.c_void_p), h1.ctypes.data_as(ctypes.c_void_p), h2.ctypes.data_as(ctypes.c_void_p)))
    return h1, h2


_MIX = 0x9e3779b97f4a7c15


_DEVICE_SORT_MIN = 1 << 18          # below this numpy is done before the keys have crossed the link


def _argsort(keys):
    """numpy.argsort(keys, kind='stable') -- integer keys or an 'S<w>' array of names -- through the device's radix sort when
    there are enough of them and a GPU is there (kv_argsort_u64 / kv_argsort_rows: 6.7 M names in milliseconds; numpy's
    three sorts were 0.75 s of config 4's band).  Same order either way: both are stable."""
    import ctypes
    import os
    import numpy as np
    from kevlar_amd import _lib
    n = len(keys)
    if n < _DEVICE_SORT_MIN or n >= (1 << 32) or not _lib.device_visible() or _lib.knob('KV_HOST_SORT'):
        return np.argsort(keys, kind='stable')
    _lib.require_device()
    lib = _lib.load()
    order = np.empty(n, dtype=np.uint32)
    if keys.dtype.kind == 'S':
        rows = np.ascontiguousarray(keys)
        _lib.check(lib.kv_argsort_rows(ctypes.c_void_p(rows.ctypes.data), n, rows.dtype.itemsize, ctypes.c_void_p(order.ctypes.data)))
    else:
        if keys.dtype.kind == 'i' and n and int(keys.min()) < 0:
            return np.argsort(keys, kind='stable')
        k64 = np.ascontiguousarray(keys, dtype=np.uint64)
        _lib.check(lib.kv_argsort_u64(ctypes.c_void_p(k64.ctypes.data), n, ctypes.c_void_p(order.ctypes.data)))
    return order.astype(np.int64)


def _dedup_runs(part, h1, h2):
    """drop[i] = True for every member whose (partition, h1, h2) an EARLIER member has: ONE unstable sort by a 64-bit mix of
    partition and h1 (a three-key lexsort is three stable sorts: 4.7 s of 6.7 M reads against 0.5 s), member order restored inside
    the few runs of equal keys.  Returns (dup, head): the members that repeat an earlier one and, for each, that earlier member
    (the first of its run).  Two different (partition, h1) pairs that share a mixed key could interleave in a run; runs are
    therefore ordered by (partition, h1, h2, member) themselves, which is exact whatever the mix does."""
    import numpy as np
    n = len(part)
    if n == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    with np.errstate(over='ignore'):
        mixed = h1 ^ (part.astype(np.uint64) * np.uint64(_MIX))
    order = _argsort(mixed)
    ps, a1 = part[order], h1[order]
    with np.errstate(over='ignore'):
        ks = a1 ^ (ps.astype(np.uint64) * np.uint64(_MIX))           # (= mixed[order], without a third gather)
    tied = ks[1:] == ks[:-1]
    in_run = np.zeros(n, dtype=bool)
    in_run[1:] |= tied
    in_run[:-1] |= tied
    idx = np.flatnonzero(in_run)                                    # the members whose key somebody shares: a few per cent
    if not len(idx):
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    members = order[idx]
    by = np.lexsort((members, h2[members], h1[members], part[members]))     # exact order of those: by group, then member order
    members = members[by]
    gp, g1, g2 = part[members], h1[members], h2[members]
    again = np.zeros(len(members), dtype=bool)
    again[1:] = (gp[1:] == gp[:-1]) & (g1[1:] == g1[:-1]) & (g2[1:] == g2[:-1])
    head = members[np.maximum.accumulate(np.where(~again, np.arange(len(members)), 0))]
    return members[again], head[again]


def assemble_partitions(names, name_offs, seqs, seq_offs, component_of, minabund=None, dedup=True):
    """The host half of relaxed-mode partitioning on arrays.  names / seqs: the reads' names and sequences (blob + offsets);
    component_of(node_of_read, n_nodes) -> a component label per NODE (reads that share a name share a node, whose record is
    the last of them).  Returns (reads, number): the read indices in output order and the partition number of each --
    partitions largest first (ties: the one whose smallest name is larger first), members by name, with dedup only the
    first read of every sequence up to reverse complement, and a partition that dedup leaves below minabund dropped:
    kevlar/partition.py:15-55, kevlar/readgraph.py:123-161."""
    import numpy as np
    n = len(name_offs) - 1
    empty = (np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    if n <= 0:
        return empty
    fixed = _fixed_width(names, name_offs)
    by_node = _argsort(fixed)                                       # the reads by name, equal names in read order (ONE sort gives the
    in_order = fixed[by_node]                                       # node ids, the reads by node and each node's last read)
    fresh = np.ones(n, dtype=bool)
    fresh[1:] = in_order[1:] != in_order[:-1]
    node_of_read = np.empty(n, dtype=np.int64)
    node_of_read[by_node] = np.cumsum(fresh) - 1                    # node ids in name order: a node's id is its name's rank
    n_nodes = int(fresh.sum())
    last = np.ones(n, dtype=bool)
    last[:-1] = fresh[1:]
    holder = by_node[last]                                          # a node's record: the last read with its name
    labels = np.asarray(component_of(node_of_read.astype(np.uint32), n_nodes))
    by_comp = _argsort(labels)                                      # the nodes by component, within one by name
    in_order = labels[by_comp]
    head = np.ones(n_nodes, dtype=bool)
    head[1:] = in_order[1:] != in_order[:-1]
    comp = np.empty(n_nodes, dtype=np.int64)
    comp[by_comp] = np.cumsum(head) - 1
    size = np.bincount(comp)
    smallest = by_comp[head]                                        # a component's smallest name: its first node in name order
    keep = np.flatnonzero(size >= 2)                                # a read on its own is not a partition
    if not len(keep):
        return empty
    keep = keep[np.lexsort((-smallest[keep], -size[keep]))]          # largest first; ties: the larger smallest name first
    # by_comp already holds every component's nodes side by side, names ascending (node ids are name ranks): the output order is a
    # permutation of those stretches -- no sort over the nodes (an argsort by partition number was 1.7 s of 6.7 M reads)
    starts = np.zeros(len(size) + 1, dtype=np.int64)
    np.cumsum(size, out=starts[1:])
    lens = size[keep]
    out_start = np.cumsum(lens) - lens
    nodes = by_comp[np.repeat(starts[keep] - out_start, lens) + np.arange(int(lens.sum()), dtype=np.int64)]
    reads, part = holder[nodes], np.repeat(np.arange(len(keep), dtype=np.int64), lens)
    if dedup:
        h1, h2 = _canonical_hashes(seqs, seq_offs, reads)
        dup, head = _dedup_runs(part, h1, h2)
        if len(dup):
            # the reference compares the canonical sequences themselves (kevlar/partition.py:26-33): a read is dropped only if its
            # sequence really is the one its run of equal hashes started with (only would-be duplicates pay for the comparison)
            dup = dup[_same_canonical(seqs, seq_offs, reads[dup], reads[head])]
            stays = np.ones(len(reads), dtype=bool)
            stays[dup] = False
            reads, part = reads[stays], part[stays]
        if minabund:
            left = np.bincount(part, minlength=len(keep))
            ok = left >= minabund
            sel = ok[part]
            reads, part = reads[sel], part[sel]
            renumber = np.cumsum(ok) - 1
            part = renumber[part]
    return reads, part + 1


def partition_file(infile, minabund=None, maxabund=None, dedup=True):
    """partition() in relaxed mode for a file, on arrays: returns (annotated, read indices in output order, partition number of
    each).  Same components, numbering and log lines as partition(); the records are parsed natively
    (AnnotatedReads.from_file), the components come from the device union-find, the ordering is array arithmetic
    (assemble_partitions), and no Record object -- nor any per-read Python object -- is built."""
    import numpy as np
    from kevlar_amd import khmer
    from kevlar_amd.annotated import AnnotatedReads
    timer = kevlar_amd.Timer()
    timer.start()
    with _phase(timer, 'loadreads', ('[kevlar::partition] Loading reads',), 'Reads loaded in {:.2f} sec'):
        ann = AnnotatedReads.from_file(infile)
        if ann.n and ann.ksize is None and len(ann):
            raise ValueError('all interesting k-mers of one graph must share k')

    def component_of(node_of_read, n_nodes):
        mode = ('[kevlar::partition]', 'Building read graph in relaxed mode')
        with _phase(timer, 'buildgraph', mode, 'Graph built in {:.2f} sec'):
            labels = khmer.readgraph_components(ann.batch, ann.ksize or 1, ann.read, ann.offset, node_of_read, n_nodes,
                                                minabund or 0, maxabund or 0)
            ann.close()
        return labels
    if ann.n:
        reads, number = assemble_partitions(ann.names, ann.name_offs, ann.seqs, ann.seq_offs, component_of, minabund, dedup)
    else:
        with _phase(timer, 'buildgraph', ('[kevlar::partition]', 'Building read graph in relaxed mode'), 'Graph built in {:.2f} sec'):
            pass
        reads, number = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    with _phase(timer, 'partition', ('[kevlar::partition] Partition readgraph',), 'Partitioning done in {:.2f} sec'):
        pass
    kevlar_amd.plog('[kevlar::partition]', 'Total time: {:.2f} seconds'.format(timer.stop()))
    return ann, reads, number


class _Outputs(object):
    """where partitions go: one shared stream, or with --split PREFIX one gzipped file per partition"""

    def __init__(self, shared, prefix):
        self.prefix = prefix
        if prefix:
            kevlar_amd.mkdirp(prefix, trim=True)
        self.shared = None if prefix else kevlar_amd.open_sink(shared)

    def put(self, number, text):
        if self.shared is not None:
            self.shared.write(text)
            return
        with kevlar_amd.open_sink('{:s}.cc{:d}.augfastq.gz'.format(self.prefix, number)) as own:
            own.write(text)

    def close(self):
        if self.shared is not None:
            self.shared.close()


def _main_arrays(args, outputs):
    """relaxed mode, file input: everything on arrays, text rendered natively"""
    import numpy as np
    ann, reads, number = partition_file(args.infile, minabund=args.min_abund, maxabund=args.max_abund, dedup=args.dedup)
    count = int(number.max()) if len(number) else 0
    if len(reads):
        sizes = np.bincount(number, minlength=count + 1)[1:]
        labels = [' kvcc={:d}'.format(i) for i in range(1, count + 1)]
        if args.split:
            start = 0
            for i, size in enumerate(sizes.tolist(), 1):
                outputs.put(i, ann.format(reads[start:start + size], suffixes=[labels[i - 1]] * size))
                start += size
        else:
            blob = ''.join(label * size for label, size in zip(labels, sizes.tolist())).encode('latin-1')
            lens = np.repeat(np.array([len(label) for label in labels], dtype=np.uint64), sizes)
            offs = np.zeros(len(reads) + 1, dtype=np.uint64)
            np.cumsum(lens, out=offs[1:])
            ann.format_to(outputs.shared, reads, suffix_blob=(blob, offs))
    kevlar_amd.plog('[kevlar::partition]', 'grouped {:d} reads into {:d} connected components'.format(len(reads), count))


def main(args):
    outputs = _Outputs(args.out, args.split)
    if not args.strict and not args.gml and isinstance(args.infile, str) and args.infile != '-':
        _main_arrays(args, outputs)
        outputs.close()
        return
    labelled = partition(kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.infile, 'r')), strict=args.strict,
                         minabund=args.min_abund, maxabund=args.max_abund, dedup=args.dedup, gmlfile=args.gml)
    sizes = [0]                      # reads per component; sizes[0] pads the 1-based numbering
    for number, reads in labelled:
        sizes.append(len(reads))
        outputs.put(number, ''.join(map(format_augmented_fastx, reads)))
    outputs.close()
    kevlar_amd.plog('[kevlar::partition]', 'grouped {:d} reads into {:d} connected components'.format(sum(sizes), len(sizes) - 1))
