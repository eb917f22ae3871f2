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


def partition_file(infile, minabund=None, maxabund=None, dedup=True):
    """partition() in relaxed mode for a file, on arrays: yields (N, annotated, read indices) with the reads of
    component N in the order partition() lists them (sorted by name; with dedup the first read of every sequence, up
    to reverse complement).  Same components, numbering and log lines; the records are parsed natively
    (AnnotatedReads.from_file), the components come from the device union-find, and no Record object is built."""
    import numpy as np
    from kevlar_amd import khmer
    from kevlar_amd.annotated import AnnotatedReads
    timer = kevlar_amd.Timer()
    timer.start()
    with _phase(timer, 'loadreads', ('[kevlar::partition] Loading reads',), 'Reads loaded in {:.2f} sec'):
        ann = AnnotatedReads.from_file(infile)
        if ann.n and ann.ksize is None and len(ann):
            raise ValueError('all interesting k-mers of one graph must share k')
        blob, offs = ann.names.decode('latin-1'), ann.name_offs.tolist()
        names = [blob[offs[i]:offs[i + 1]] for i in range(ann.n)]
        node_id, node_names, holder = {}, [], []           # reads that share a name share a node; its record is the last of them
        node_of_read = np.empty(ann.n, dtype=np.uint32)
        for i, name in enumerate(names):
            node = node_id.get(name)
            if node is None:
                node = node_id[name] = len(node_names)
                node_names.append(name)
                holder.append(i)
            else:
                holder[node] = i
            node_of_read[i] = node
    mode = ('[kevlar::partition]', 'Building read graph in relaxed mode')
    with _phase(timer, 'buildgraph', mode, 'Graph built in {:.2f} sec'):
        if ann.n:
            labels = khmer.readgraph_components(ann.batch, ann.ksize or 1, ann.read, ann.offset, node_of_read, len(node_names),
                                                minabund or 0, maxabund or 0)
            ann.close()
        else:
            labels = np.zeros(0, dtype=np.uint32)
    with _phase(timer, 'partition', ('[kevlar::partition] Partition readgraph',), 'Partitioning done in {:.2f} sec'):
        # components, largest first; ties: by sorted read names, descending (names are unique per node, so the
        # smallest name of a component decides)
        order = np.argsort(labels, kind='stable')
        cuts = np.flatnonzero(np.diff(labels[order])) + 1 if len(order) else np.zeros(0, dtype=np.int64)
        groups = np.split(order, cuts) if len(order) else []
        keyed = []
        for nodes in groups:
            if len(nodes) < 2:
                continue                      # a read on its own is not a partition
            members = sorted(node_names[v] for v in nodes.tolist())
            keyed.append((len(members), members))
        keyed.sort(reverse=True)
        seqs, soffs = ann.seqs.decode('latin-1'), ann.seq_offs.tolist()
        number = 0
        for size, members in keyed:
            reads = [holder[node_id[name]] for name in members]
            if dedup:
                seen, kept = set(), []
                for r in reads:
                    canon = kevlar_amd.revcommin(seqs[soffs[r]:soffs[r + 1]])
                    if canon in seen:
                        continue
                    seen.add(canon)
                    kept.append(r)
                reads = kept
                if minabund and len(reads) < minabund:
                    continue
            number += 1
            yield number, ann, reads
    kevlar_amd.plog('[kevlar::partition]', 'Total time: {:.2f} seconds'.format(timer.stop()))


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
    sizes = [0]
    pending_reads, pending_suffix, ann = [], [], None
    for number, ann, reads in partition_file(args.infile, minabund=args.min_abund, maxabund=args.max_abund, dedup=args.dedup):
        sizes.append(len(reads))
        suffix = ' kvcc={:d}'.format(number)
        if args.split:
            outputs.put(number, ann.format(reads, suffixes=[suffix] * len(reads)))
        else:
            pending_reads.extend(reads)
            pending_suffix.extend([suffix] * len(reads))
    if pending_reads:
        outputs.put(0, ann.format(pending_reads, suffixes=pending_suffix))
    kevlar_amd.plog('[kevlar::partition]', 'grouped {:d} reads into {:d} connected components'.format(sum(sizes), len(sizes) - 1))


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
