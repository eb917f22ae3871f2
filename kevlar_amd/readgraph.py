"""Read graph for `kevlar partition` (the reference's kevlar/readgraph.py:16-161).

The reference keeps a networkx graph, adds an edge for every pair of reads that share an
interesting k-mer and asks networkx for connected components.  Here the graph never
materialises: kv_readgraph_components groups the annotations by canonical k-mer in a device
hash table, applies the min/max abundance filter per k-mer and runs a lock-free union-find
over the reads; this class keeps the reference's method names on top of that.
"""
import numpy as np

import kevlar_amd
from kevlar_amd import khmer


class ReadGraph(object):
    def __init__(self):
        self.records = {}        # node name -> record (insertion ordered, last duplicate wins)
        self.readnames = set()
        self._reads = []         # every loaded read, duplicates included
        self._minabund = None
        self._maxabund = None
        self._labels = None
        self._nedges = None

    # ---- the networkx-shaped surface the drivers use ------------------------------------
    def __iter__(self):
        return iter(self.records)

    def __len__(self):
        return len(self.records)

    def __contains__(self, name):
        return name in self.records

    def number_of_nodes(self):
        return len(self.records)

    def number_of_edges(self):
        if self._nedges is None:
            self._solve(want_edges=True)
        return self._nedges

    def get_record(self, recordname):
        return self.records[recordname]

    # ---- kevlar/readgraph.py:43-84 ---------------------------------------------------------
    def load(self, readstream, minabund=None, maxabund=None, dedup=False):
        """Register reads as nodes.  With thresholds, only k-mers present in
        minabund <= #reads <= maxabund reads will link reads (0/None = unbounded).
        dedup=True keeps the first read of each sequence (up to reverse complement)."""
        seen = set()
        for record in readstream:
            if record is None:
                continue
            if dedup:
                minread = kevlar_amd.revcommin(record.sequence)
                if minread in seen:
                    continue
                seen.add(minread)
            self.records[record.name] = record
            self.readnames.add(record.name)
            self._reads.append(record)
        self._minabund = minabund
        self._maxabund = maxabund
        self._labels = None
        self._nedges = None

    # ---- kevlar/readgraph.py:104-125 ------------------------------------------------------
    def populate_edges(self, strict=False):
        if strict:
            self._solve_strict()
        else:
            self._solve(want_edges=False)

    def _solve_strict(self):
        """Strict mode (kevlar/readgraph.py:113-123): an edge only where the two reads overlap
        perfectly around a shared k-mer.  Host-side; see kevlar_amd/readpair.py."""
        from itertools import combinations
        from kevlar_amd.readpair import validate
        names = list(self.records)
        node_id = {name: i for i, name in enumerate(names)}
        ikmers = {}
        for record in self._reads:
            for ikmer in record.annotations:
                ikmers.setdefault(kevlar_amd.revcommin(record.ikmerseq(ikmer)), set()).add(record.name)
        parent = list(range(len(names)))

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x
        edges = set()
        for kmer, readset in ikmers.items():
            n = len(readset)
            if (self._minabund and n < self._minabund) or (self._maxabund and n > self._maxabund):
                continue
            for name1, name2 in combinations(sorted(readset), 2):
                if (name1, name2) in edges:
                    continue
                result = validate(self.records[name1], self.records[name2], kmer)
                if result is None:
                    continue
                tailname, headname = result[1], result[2]      # may be one and the same read (see readpair.validate)
                edges.add((min(tailname, headname), max(tailname, headname)))
                a, b = find(node_id[tailname]), find(node_id[headname])
                if a != b:
                    parent[max(a, b)] = min(a, b)
        self._labels = np.array([find(i) for i in range(len(names))], dtype=np.uint32)
        self._nedges = len(edges)

    def _solve(self, want_edges):
        names = list(self.records)
        node_id = {name: i for i, name in enumerate(names)}
        reads = self._reads
        node_of_read = np.fromiter((node_id[r.name] for r in reads), dtype=np.uint32, count=len(reads))
        ann_read, ann_off = [], []
        ksize = None
        for ridx, record in enumerate(reads):
            for ikmer in record.annotations:
                if ksize is None:
                    ksize = ikmer.ksize
                elif ikmer.ksize != ksize:
                    raise ValueError('all interesting k-mers of one graph must share k')
                ann_read.append(ridx)
                ann_off.append(ikmer.offset)
        if not reads:
            self._labels, self._nedges = np.zeros(0, dtype=np.uint32), 0
            return
        batch = khmer.ReadBatch([r.sequence for r in reads])
        result = khmer.readgraph_components(
            batch, ksize or 1, np.asarray(ann_read, dtype=np.uint32), np.asarray(ann_off, dtype=np.uint32),
            node_of_read, len(names), self._minabund or 0, self._maxabund or 0, want_edges=want_edges)
        batch.close()
        if want_edges:
            self._labels, self._nedges = result
        else:
            self._labels = result

    def edge_list(self):
        """Pairs of node names that share at least one retained interesting k-mer (relaxed mode), sorted; for
        `partition --gml`.  Enumerated on the host: the device path only ever needs their number."""
        from itertools import combinations
        holders = {}
        for record in self._reads:
            for ikmer in record.annotations:
                holders.setdefault(kevlar_amd.revcommin(record.ikmerseq(ikmer)), set()).add(record.name)
        pairs = set()
        for names in holders.values():
            n = len(names)
            if (self._minabund and n < self._minabund) or (self._maxabund and n > self._maxabund):
                continue
            pairs.update(combinations(sorted(names), 2))
        return sorted(pairs)

    def connected_components(self):
        """List of sets of node names."""
        if self._labels is None:
            self._solve(want_edges=False)
        names = list(self.records)
        groups = {}
        for name, label in zip(names, self._labels.tolist()):
            groups.setdefault(label, set()).add(name)
        return list(groups.values())

    # ---- kevlar/readgraph.py:127-161 ------------------------------------------------------
    def partitions(self, dedup=True, minabund=None, maxabund=None, abundfilt=False):
        """Connected components, largest first (ties: by sorted read names, descending)."""
        ccs = sorted(self.connected_components(), reverse=True, key=lambda c: (len(c), sorted(c)))
        for cc in ccs:
            if len(cc) == 1 and next(iter(cc)) in self.readnames:
                continue   # unassembled input read
            if not dedup:
                yield cc
                continue
            # The reference iterates a Python set here, so which duplicate survives depends on
            # PYTHONHASHSEED; sorted order makes it deterministic.
            partition = ReadGraph()
            partition.load([self.get_record(readid) for readid in sorted(cc)], minabund, maxabund, dedup=True)
            assert partition.number_of_nodes() > 0
            if abundfilt and minabund and partition.number_of_nodes() < minabund:
                continue
            yield partition
