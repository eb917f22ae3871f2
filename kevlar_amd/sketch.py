"""Sketch convenience functions (the reference's kevlar/sketch.py:14-170) over the HIP sketches."""
import kevlar_amd
from kevlar_amd import khmer

sketch_loader_by_filename_extension = {
    '.nt': khmer.Nodetable.load, '.ng': khmer.Nodegraph.load,
    '.ct': khmer.Counttable.load, '.cg': khmer.Countgraph.load,
    '.sct': khmer.SmallCounttable.load, '.scg': khmer.SmallCountgraph.load,
    '.nodetable': khmer.Nodetable.load, '.nodegraph': khmer.Nodegraph.load,
    '.counttable': khmer.Counttable.load, '.countgraph': khmer.Countgraph.load,
    '.smallcounttable': khmer.SmallCounttable.load, '.smallcountgraph': khmer.SmallCountgraph.load,
}

# (count, graph, smallcount) -> extensions
_EXTENSIONS = {
    (True, True, True): ('.scg', '.smallcountgraph'), (True, True, False): ('.cg', '.countgraph'),
    (True, False, True): ('.sct', '.smallcounttable'), (True, False, False): ('.ct', '.counttable'),
    (False, True, True): ('.ng', '.nodegraph'), (False, True, False): ('.ng', '.nodegraph'),
    (False, False, True): ('.nt', '.nodetable'), (False, False, False): ('.nt', '.nodetable'),
}


class KevlarSketchTypeError(ValueError):
    pass


class KevlarUnsuitableFPRError(SystemExit):
    pass


def estimate_fpr(sketch):
    """(occupied bins of table 0 / smallest table) ** number of tables."""
    sizes = sketch.hashsizes()
    return (float(sketch.n_occupied()) / min(sizes)) ** float(len(sizes))


def load(filename):
    """Pick the sketch class from the file extension and load it into HBM."""
    if not filename.endswith(tuple(sketch_loader_by_filename_extension)):
        raise KevlarSketchTypeError('unable to determine sketch type from filename ' + filename)
    ext = '.' + filename.split('.')[-1]
    return sketch_loader_by_filename_extension[ext](filename)


def get_extension(count=False, graph=False, smallcount=False):
    return _EXTENSIONS[(bool(count), bool(graph), bool(smallcount))]


def allocate(ksize, target_tablesize, num_tables=4, count=False, graph=False, smallcount=False):
    if count:
        if graph:
            cls = khmer.SmallCountgraph if smallcount else khmer.Countgraph
        else:
            cls = khmer.SmallCounttable if smallcount else khmer.Counttable
    else:
        cls = khmer.Nodegraph if graph else khmer.Nodetable
    return cls(ksize, target_tablesize, num_tables)


def autoload(infile, count=True, graph=False, ksize=31, table_size=1e4, num_tables=4,
             num_bands=None, band=None):
    """Load a saved sketch by extension, else count the file as FASTA/FASTQ."""
    try:
        return load(infile)
    except KevlarSketchTypeError:
        sketch = allocate(ksize, table_size, num_tables, count=count, graph=graph, smallcount=False)
        if num_bands:
            assert band >= 0 and band < num_bands
            sketch.consume_seqfile_banding(infile, num_bands, band)
        else:
            sketch.consume_seqfile(infile)
        return sketch


def load_sketchfiles(sketchfiles, maxfpr=0.2):
    sketches = []
    for sketchfile in sketchfiles:
        kevlar_amd.plog('[kevlar::sketch]    ', 'loading sketchfile "{}"...'.format(sketchfile), end='')
        sketch = autoload(sketchfile)
        fpr = estimate_fpr(sketch)
        message = 'done! estimated false positive rate is {:1.3f}'.format(fpr)
        if fpr > maxfpr:
            message += ' (FPR too high, bailing out!!!)'
            raise KevlarUnsuitableFPRError(message)
        kevlar_amd.plog(message)
        sketches.append(sketch)
    return sketches
