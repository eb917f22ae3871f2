"""Command line of the drop-in: `kevlar count | novel | filter | partition | unband | dist | split | augment`.

Flag names, defaults and dispatch follow the reference (kevlar/cli/__init__.py:31-108 and
kevlar/cli/{count,novel,filter,partition,unband,dist}.py); only the subcommands on the
novel-k-mer path exist here.  `novel` has one extra switch, --ref-band-quirk (see
kevlar_amd/novel.py).
"""
import argparse
import sys

import kevlar_amd
from kevlar_amd.khmer import khmer_args

memory = khmer_args.memory_setting


def _count(sub):
    p = sub.add_parser('count', description='Count the k-mers of one sample into a Count-Min sketch '
                       '(or Bloom filter) held in GPU memory and save it in khmer\'s file format. '
                       'Supports k-mer banding.')
    p.add_argument('-k', '--ksize', type=int, default=31, metavar='K', help='k-mer size (31)')
    p.add_argument('-c', '--counter-size', type=int, choices=(1, 4, 8), default=8, metavar='C',
                   help='bits per counter: 1 (presence), 4 (max 15) or 8 (max 255, default)')
    p.add_argument('-M', '--memory', type=memory, default=1e6, metavar='MEM',
                   help='bytes for the sketch, K/M/G/T suffixes allowed (1M)')
    p.add_argument('--max-fpr', type=float, default=0.2, metavar='FPR',
                   help='abort when the estimated false positive rate exceeds FPR (0.2)')
    p.add_argument('--mask', metavar='MSK', help='sketch of k-mers to leave out')
    p.add_argument('--count-masked', action='store_true',
                   help='invert the mask: count only k-mers it contains')
    p.add_argument('--num-bands', type=int, default=None, metavar='N', help='split the hash space into N bands')
    p.add_argument('--band', type=int, default=None, metavar='I', help='band to process, 1..N')
    p.add_argument('-t', '--threads', type=int, default=1, metavar='T', help='host threads feeding the GPU (1)')
    p.add_argument('counttable', help='output file; ".counttable" is appended unless it ends in .ct/.counttable')
    p.add_argument('seqfile', nargs='+', help='FASTA/FASTQ input')


def _novel(sub):
    p = sub.add_parser('novel', add_help=False, description='Report case reads that contain "interesting" '
                       'k-mers: abundant in every case sample, (nearly) absent from every control.')
    s = p.add_argument_group('Case/control config')
    s.add_argument('--case', metavar='F', nargs='+', required=True, action='append',
                   help='reads of one case sample; repeat the flag for more samples')
    s.add_argument('--case-counts', metavar='F', nargs='+', help='saved sketch per case sample')
    s.add_argument('--control', metavar='F', nargs='+', action='append',
                   help='reads of one control sample; repeat the flag for more samples')
    s.add_argument('--control-counts', metavar='F', nargs='+', help='saved sketch per control sample')
    s.add_argument('-x', '--ctrl-max', metavar='X', type=int, default=1, help='max abundance in a control (1)')
    s.add_argument('-y', '--case-min', metavar='Y', type=int, default=6, help='min abundance in a case (6)')
    s.add_argument('-M', '--memory', default='1e6', type=memory, metavar='MEM', help='bytes per sample sketch (1M)')
    s.add_argument('--max-fpr', type=float, default=0.2, metavar='FPR', help='abort above this sketch FPR (0.2)')
    b = p.add_argument_group('K-mer banding')
    b.add_argument('--num-bands', type=int, default=None, metavar='N', help='split the hash space into N bands')
    b.add_argument('--band', type=int, default=None, metavar='I', help='band to process, 1..N')
    b.add_argument('--ref-band-quirk', action='store_true',
                   help='apply the reference\'s literal low-bits band test during the scan')
    o = p.add_argument_group('Output settings')
    o.add_argument('-o', '--out', metavar='FILE', help='output augmented FASTQ (stdout)')
    o.add_argument('--save-case-counts', metavar='CT', nargs='+', help='save the case sketches')
    o.add_argument('--save-ctrl-counts', metavar='CT', nargs='+', help='save the control sketches')
    m = p.add_argument_group('Miscellaneous settings')
    m.add_argument('-h', '--help', action='help', help='show this help message and exit')
    m.add_argument('-k', '--ksize', type=int, default=31, metavar='K', help='k-mer size (31)')
    m.add_argument('--abund-screen', type=int, default=None, metavar='INT',
                   help='drop reads having any k-mer with case abundance < INT')
    m.add_argument('-t', '--threads', type=int, default=1, metavar='T', help='host threads for counting (1)')
    m.add_argument('--skip-until', type=str, metavar='ID', help='resume: skip case reads through read ID')


def _filter(sub):
    p = sub.add_parser('filter', description='Re-count the interesting k-mers of an augmented FASTQ '
                       '(ignoring masked k-mers) and drop k-mers/reads that no longer pass.')
    p.add_argument('-M', '--memory', type=memory, default=1e6, metavar='MEM', help='bytes for the recount sketch (1M)')
    p.add_argument('--max-fpr', type=float, default=0.01, metavar='FPR', help='abort above this recount FPR (0.01)')
    p.add_argument('--mask', metavar='MSK', help='sketch of k-mers to ignore')
    p.add_argument('-x', '--ctrl-max', metavar='X', type=int, default=1, help='max abundance in a control (1)')
    p.add_argument('-y', '--case-min', metavar='Y', type=int, default=6, help='min abundance in the case (6)')
    p.add_argument('-o', '--out', metavar='FILE', help='output file (stdout)')
    p.add_argument('augfastq', help='augmented FASTQ from `novel`')


def _partition(sub):
    p = sub.add_parser('partition', description='Group reads that share interesting k-mers: connected '
                       'components of the read graph, labelled kvcc=N.')
    p.add_argument('-s', '--strict', action='store_true', help='require a perfect overlap between linked reads')
    p.add_argument('--min-abund', metavar='X', type=int, default=2, help='ignore k-mers in fewer than X reads (2)')
    p.add_argument('--max-abund', metavar='Y', type=int, default=200, help='ignore k-mers in more than Y reads (200)')
    p.add_argument('--no-dedup', dest='dedup', action='store_false', default=True, help='keep duplicate reads')
    p.add_argument('--gml', metavar='FILE', help='write the read graph as GML')
    p.add_argument('--split', type=str, metavar='OUTPREFIX', help='one file per partition: OUTPREFIX.cc#.augfastq.gz')
    p.add_argument('-o', '--out', metavar='FILE', help='output file (stdout)')
    p.add_argument('infile', help='augmented FASTA/FASTQ')


def _unband(sub):
    p = sub.add_parser('unband', description='Merge the per-band outputs of a banded `novel` run into one '
                       'non-redundant set of reads carrying all of their annotations.')
    p.add_argument('-n', '--n-batches', metavar='N', type=int, default=16, help='temporary batches (16)')
    p.add_argument('-o', '--out', metavar='FILE', help='output file (stdout)')
    p.add_argument('infile', nargs='+', help='augmented FASTA/FASTQ files')


def _dist(sub):
    p = sub.add_parser('dist', description='Compute the k-mer abundance distribution for a data set.')
    p.add_argument('-o', '--out', metavar='FILE', help='output file; default is terminal (stdout)')
    p.add_argument('-k', '--ksize', metavar='K', type=int, default=31, help='k-mer size; default is 31')
    p.add_argument('-M', '--memory', type=memory, default=1e6, metavar='MEM', help='memory to allocate for k-mer counting')
    p.add_argument('-t', '--threads', type=int, metavar='T', default=1,
                   help='accepted for compatibility: the passes run on the GPU; default is 1')
    p.add_argument('-p', '--plot', metavar='PNG', help='plot k-mer abundance distribution to file `PNG`')
    p.add_argument('--tsv', metavar='TSV', help='write k-mer abundance distribution out to file formatted as '
                   'tab-separated values')
    p.add_argument('--plot-xlim', metavar=('MIN', 'MAX'), type=int, nargs=2, default=(0, 100),
                   help='define the minimum and maximum x values (k-mer abundance) for the plot; default is `0 100`')
    p.add_argument('mask', help='nodetable containing target k-mers to count (such as single-copy exonic k-mers)')
    p.add_argument('infiles', nargs='+', help='input files in FASTA/FASTQ format')


def _split(sub):
    p = sub.add_parser('split', description='Split partitioned reads into N output files.')
    p.add_argument('infile', help='input file; partitioned reads in augmented Fastq/Fasta format')
    p.add_argument('numfiles', type=int, help='number of output files to create')
    p.add_argument('base', help='prefix of all output files')


def _augment(sub):
    p = sub.add_parser('augment', description='Internally, kevlar annotates sequences with "interesting k-mers" '
                       'and uses "augmented" Fastq and Fasta formats. Processing sequences with third-party tools '
                       'usually requires discarding these annotations. This command is used to augment/reaugment '
                       'a set of sequences using annotations from an already augmented sequence file.')
    p.add_argument('-o', '--out', metavar='FILE', help='output file; default is terminal (stdout)')
    p.add_argument('augseqs', help='augmented sequence file')
    p.add_argument('seqs', help='sequences to annotate')


def _gentrio(sub):
    p = sub.add_parser('gentrio', description='Apply randomly generated mutations to the genome provided.')
    p.add_argument('-i', '--inherited', type=int, metavar='I', default=20, help='number of shared/inherited mutations to simulate')
    p.add_argument('-d', '--de-novo', type=int, metavar='D', default=10, help='number of unique/de novo mutations to simulate')
    p.add_argument('--vcf', metavar='FILE', help='write mutations to a VCF file')
    p.add_argument('--prefix', metavar='PFX', default='trio', help='prefix for output fasta files; default is "trio"')
    p.add_argument('--weights', metavar='WT', default='snv=0.8,ins=0.1,del=0.1',
                   help='comma-separated list of key/value pairs indicating the relative frequency of different variant types; '
                        'default is "snv=0.8,ins=0.1,del=0.1"')
    p.add_argument('-s', '--seed', metavar='S', default=None, type=int, help='seed for random number generator')
    p.add_argument('genome', help='genome to mutate')


mains = {
    'augment': kevlar_amd.augment.main,
    'gentrio': kevlar_amd.gentrio.main,
    'split': kevlar_amd.split.main,
    'count': kevlar_amd.count.main,
    'dist': kevlar_amd.dist.main,
    'novel': kevlar_amd.novel.main,
    'filter': kevlar_amd.filter.main,
    'partition': kevlar_amd.partition.main,
    'unband': kevlar_amd.unband.main,
}

subparser_funcs = {
    'augment': _augment,
    'gentrio': _gentrio,
    'split': _split,
    'count': _count,
    'dist': _dist,
    'novel': _novel,
    'filter': _filter,
    'partition': _partition,
    'unband': _unband,
}


def parser():
    top = argparse.ArgumentParser(
        prog='kevlar', formatter_class=argparse.RawDescriptionHelpFormatter,
        description='kevlar novel-k-mer discovery on AMD MI355X (count, novel, filter, partition, unband, dist, split, augment, gentrio)')
    top._positionals.title = 'Subcommands'
    top._optionals.title = 'Global arguments'
    top.add_argument('-v', '--version', action='version', version='kevlar v{}'.format(kevlar_amd.__version__))
    top.add_argument('-l', '--logfile', metavar='F', help='write diagnostics to F instead of stderr')
    top.add_argument('--tee', action='store_true', help='write diagnostics to the logfile and to stderr')
    sub = top.add_subparsers(dest='cmd', metavar='cmd', help='"' + '", "'.join(sorted(mains)) + '"')
    for func in subparser_funcs.values():
        func(sub)
    return top


def parse_args(arglist=None):
    args = parser().parse_args(arglist)
    kevlar_amd.logstream = sys.stderr
    if args.logfile and args.logfile != '-':
        kevlar_amd.logstream = kevlar_amd.open(args.logfile, 'w')
    kevlar_amd.teelog = args.tee
    return args


def run(arglist=None):
    """The command line: parse, announce the version ("[kevlar] running version ...", asserted by the reference's
    tests), dispatch to the subcommand's driver.  `python -m kevlar_amd` and the console entry point both come here;
    tests pass an argument list."""
    top = parser()
    args = parse_args(arglist)
    driver = mains.get(args.cmd)
    if driver is None:
        top.print_help()
        raise SystemExit(0 if args.cmd is None else 2)
    kevlar_amd.plog('[kevlar] running version {}'.format(kevlar_amd.__version__))
    return driver(args)

