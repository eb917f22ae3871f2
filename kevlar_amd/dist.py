"""`kevlar dist` driver (the reference's kevlar/dist.py:25-125): k-mer abundance distribution of a
sample over the k-mers of a mask (e.g. single-copy exonic k-mers) -> mean and standard deviation of
k-mer coverage, which `simlike --mu/--sigma` consumes.

Two passes over the reads, both on the GPU: count only masked k-mers into a Counttable
(consume_seqfile_with_mask, threshold 1, consume_masked), then `abundance_distribution` with a
tracking Nodetable so that every k-mer contributes once.  The reference fans each file out over T
Python threads sharing one parser; here one pass per file feeds the device and `--threads` is
accepted for command-line compatibility."""
import json
import math

import kevlar_amd
from kevlar_amd import khmer


class KevlarZeroAbundanceDistError(ValueError):
    pass


def count_first_pass(infiles, counts, mask, nthreads=1):
    kevlar_amd.plog('[kevlar::dist]', 'Processing input with {:d} threads'.format(nthreads))
    for filename in infiles:
        kevlar_amd.plog('    -', filename)
        counts.consume_seqfile_with_mask(khmer.ReadParser(filename), mask, threshold=1, consume_masked=True)
    kevlar_amd.plog('[kevlar::dist] Done processing input!')


def count_second_pass(infiles, counts, nthreads=1):
    kevlar_amd.plog('[kevlar::dist] Second pass over the data')
    tracking = khmer.Nodetable(counts.ksize(), 1, 1, primes=counts.hashsizes())
    abundance = {}
    for filename in infiles:
        kevlar_amd.plog('    -', filename)
        abund = counts.abundance_distribution(khmer.ReadParser(filename), tracking)
        for i, count in enumerate(abund):
            if i > 0 and count > 0:
                abundance[i] = abundance.get(i, 0) + count
    kevlar_amd.plog('[kevlar::dist] Done second pass over input!')
    return abundance


def weighted_mean_std_dev(values, weights):
    total = float(sum(weights))
    mu = sum(v * w for v, w in zip(values, weights)) / total
    sigma = math.sqrt(sum(w * (v - mu) ** 2 for v, w in zip(values, weights)) / total)
    return mu, sigma


def calc_mu_sigma(abundance):
    total = sum(abundance.values())
    if total == 0:
        raise KevlarZeroAbundanceDistError('all k-mer abundances are 0, please check input files')
    return weighted_mean_std_dev(list(abundance.keys()), list(abundance.values()))


def compute_dist(abundance):
    """Rows of the distribution table: Abundance, Count, CumulativeCount, CumulativeFraction (a pandas
    DataFrame of floats, as the reference builds it row by row)."""
    import pandas
    total = sum(abundance.values())
    rows = []
    cuml = 0
    for abund, count in sorted(abundance.items()):
        assert count > 0, (abund, count)
        cuml += count
        rows.append({'Abundance': float(abund), 'Count': float(count), 'CumulativeCount': float(cuml),
                     'CumulativeFraction': cuml / total})
    return pandas.DataFrame(rows, columns=['Abundance', 'Count', 'CumulativeCount', 'CumulativeFraction'])


def dist(infiles, mask, ksize=31, memory=1e6, threads=1):
    counts = khmer.Counttable(ksize, memory / 4, 4)
    count_first_pass(infiles, counts, mask, nthreads=threads)
    abundance = count_second_pass(infiles, counts, nthreads=threads)
    mu, sigma = calc_mu_sigma(abundance)
    data = compute_dist(abundance)
    return mu, sigma, data


def main(args):
    mask = khmer.Nodetable.load(args.mask)
    mu, sigma, data = dist(args.infiles, mask, ksize=args.ksize, memory=args.memory, threads=args.threads)
    out = {'mu': mu, 'sigma': sigma}
    if getattr(args, 'out', None):
        with open(args.out, 'w') as fh:
            print(json.dumps(out), file=fh)
    else:
        print(json.dumps(out))
    if args.tsv:
        data.to_csv(args.tsv, sep='\t', index=False)
    if args.plot:
        try:
            import matplotlib
            matplotlib.use('Agg')
            from matplotlib import pyplot as plt
        except ImportError:
            raise RuntimeError('--plot needs matplotlib, which is not installed')
        matplotlib.rcParams['figure.figsize'] = [12, 6]
        plt.plot(data['Abundance'], data['Count'], color='blue')
        plt.axvline(x=mu, color='blue', linestyle='--')
        plt.axvline(x=mu - sigma, color='red', linestyle=':')
        plt.axvline(x=mu + sigma, color='red', linestyle=':')
        plt.xlim(args.plot_xlim)
        plt.xlabel('K-mer abundance')
        plt.ylabel('Frequency')
        plt.savefig(args.plot, dpi=300)
