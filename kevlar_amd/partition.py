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


class _Outputs(object):
    """where partitions go: one shared stream, or with --split PREFIX one gzipped file per partition"""

    def __init__(self, shared, prefix):
        self.prefix = prefix
        if prefix:
            kevlar_amd.mkdirp(prefix, trim=True)
        self.shared = None if prefix else kevlar_amd.open(shared, 'w')

    def put(self, number, text):
        if self.shared is not None:
            self.shared.write(text)
            return
        with kevlar_amd.open('{:s}.cc{:d}.augfastq.gz'.format(self.prefix, number), 'w') as own:
            own.write(text)


def main(args):
    outputs = _Outputs(args.out, args.split)
    labelled = partition(kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.infile, 'r')), strict=args.strict,
                         minabund=args.min_abund, maxabund=args.max_abund, dedup=args.dedup, gmlfile=args.gml)
    sizes = [0]                      # reads per component; sizes[0] pads the 1-based numbering
    for number, reads in labelled:
        sizes.append(len(reads))
        outputs.put(number, ''.join(map(format_augmented_fastx, reads)))
    kevlar_amd.plog('[kevlar::partition]', 'grouped {:d} reads into {:d} connected components'.format(sum(sizes), len(sizes) - 1))
