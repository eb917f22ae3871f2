"""`kevlar dist`: how deeply was a sample sequenced?  The abundance distribution of its k-mers over a mask of
single-copy k-mers (exons of a reference), and from it the mean and standard deviation of k-mer coverage that
`kevlar simlike --mu/--sigma` consumes (kevlar/dist.py:25-125).

Both passes over the reads run on the device: the masked count (only k-mers the mask holds are counted) and khmer's
abundance_distribution with a tracking Nodetable so that every distinct k-mer contributes once.  The histogram is a
numpy vector end to end; `--threads` is accepted for command-line compatibility (the device takes whole batches)."""
import json

import numpy as np

import kevlar_amd
from kevlar_amd import khmer


class KevlarZeroAbundanceDistError(ValueError):
    pass


def _each_file(infiles):
    for path in infiles:
        kevlar_amd.plog('    -', path)
        yield khmer.ReadParser(path)


def count_first_pass(infiles, counts, mask, nthreads=1):
    """count, into `counts`, the k-mers of the files that `mask` contains"""
    kevlar_amd.plog('[kevlar::dist]', 'Processing input with {:d} threads'.format(nthreads))
    for parser in _each_file(infiles):
        counts.consume_seqfile_with_mask(parser, mask, threshold=1, consume_masked=True)
    kevlar_amd.plog('[kevlar::dist] Done processing input!')


def _histogram(infiles, counts):
    """hist[c] = distinct k-mers of the files whose count in `counts` is c (0 <= c < 65536)"""
    seen = khmer.Nodetable(counts.ksize(), 1, 1, primes=counts.hashsizes())
    hist = np.zeros(65536, dtype=np.int64)
    for parser in _each_file(infiles):
        hist += np.asarray(counts.abundance_distribution(parser, seen), dtype=np.int64)
    return hist


def count_second_pass(infiles, counts, nthreads=1):
    """{abundance: number of distinct k-mers with that abundance}, abundance 0 left out"""
    kevlar_amd.plog('[kevlar::dist] Second pass over the data')
    hist = _histogram(infiles, counts)
    kevlar_amd.plog('[kevlar::dist] Done second pass over input!')
    present = np.flatnonzero(hist[1:]) + 1
    return dict(zip(present.tolist(), hist[present].tolist()))


def weighted_mean_std_dev(values, weights):
    v, w = np.asarray(values, dtype=np.float64), np.asarray(weights, dtype=np.float64)
    mu = float(np.average(v, weights=w))
    return mu, float(np.sqrt(np.average((v - mu) ** 2, weights=w)))


def calc_mu_sigma(abundance):
    if sum(abundance.values()) == 0:
        raise KevlarZeroAbundanceDistError('all k-mer abundances are 0, please check input files')
    return weighted_mean_std_dev(list(abundance), list(abundance.values()))


def compute_dist(abundance):
    """The distribution as a table: Abundance, Count, CumulativeCount, CumulativeFraction (floats, as the reference's
    row-by-row DataFrame has them)."""
    import pandas
    levels = np.array(sorted(abundance), dtype=np.float64)
    counts = np.array([abundance[int(a)] for a in levels], dtype=np.float64)
    assert (counts > 0).all(), abundance
    running = np.cumsum(counts)
    return pandas.DataFrame({'Abundance': levels, 'Count': counts, 'CumulativeCount': running,
                             'CumulativeFraction': running / counts.sum()})


def dist(infiles, mask, ksize=31, memory=1e6, threads=1):
    counts = khmer.Counttable(ksize, memory / 4, 4)
    count_first_pass(infiles, counts, mask, nthreads=threads)
    abundance = count_second_pass(infiles, counts, nthreads=threads)
    mu, sigma = calc_mu_sigma(abundance)
    return mu, sigma, compute_dist(abundance)


def _plot(data, mu, sigma, path, xlim):
    try:
        import matplotlib
        matplotlib.use('Agg')
        from matplotlib import pyplot
    except ImportError:
        raise RuntimeError('--plot needs matplotlib, which is not installed')
    matplotlib.rcParams['figure.figsize'] = [12, 6]
    pyplot.plot(data['Abundance'], data['Count'], color='blue')
    for x, colour, style in ((mu, 'blue', '--'), (mu - sigma, 'red', ':'), (mu + sigma, 'red', ':')):
        pyplot.axvline(x=x, color=colour, linestyle=style)
    pyplot.xlim(xlim)
    pyplot.xlabel('K-mer abundance')
    pyplot.ylabel('Frequency')
    pyplot.savefig(path, dpi=300)


def main(args):
    mu, sigma, data = dist(args.infiles, khmer.Nodetable.load(args.mask), ksize=args.ksize, memory=args.memory, threads=args.threads)
    summary = json.dumps({'mu': mu, 'sigma': sigma})
    if getattr(args, 'out', None):
        with open(args.out, 'w') as sink:
            print(summary, file=sink)
    else:
        print(summary)
    if args.tsv:
        data.to_csv(args.tsv, sep='\t', index=False)
    if args.plot:
        _plot(data, mu, sigma, args.plot, args.plot_xlim)
