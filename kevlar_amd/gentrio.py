"""`kevlar gentrio`: a simulated trio -- proband, mother, father, two haplotypes each -- from a reference genome, with
`--inherited` variants whose genotypes follow a valid inheritance pattern and `--de-novo` variants carried by the proband
only (kevlar/gentrio.py:116-257, kevlar/cli/gentrio.py:17-36).  Output: PREFIX-proband.fasta, PREFIX-mother.fasta,
PREFIX-father.fasta (records `>SEQID_haplo1`, `>SEQID_haplo2`) and, with --vcf, the variants with their genotypes.

Same flags, files and variant model (SNV, insertion of 5-350 bases partly copied from elsewhere, deletion of 5-350 bases;
weights snv=0.8,ins=0.1,del=0.1) as the reference.  The draw itself is this build's own: positions and kinds come from one
numpy generator seeded with --seed and every haplotype is assembled once from its sorted variant list, so the same seed does
not reproduce the reference's files (its output depends on the call order of Python's `random`)."""
import sys

import numpy as np

import kevlar_amd

# (child, mother, father) genotypes a variant can have: 0 = 0/0, 1 = heterozygous, 2 = 1/1; mendelian combinations only
INHERITANCE = [(0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 1), (1, 0, 2), (1, 1, 0), (1, 1, 1), (1, 1, 2), (1, 2, 0), (1, 2, 1),
               (2, 1, 1), (2, 1, 2), (2, 2, 1), (2, 2, 2)]
DEFAULT_WEIGHTS = {'snv': 0.8, 'ins': 0.1, 'del': 0.1}
_NUCL = 'ACGT'


def parse_weights(text):
    """'snv=0.8,ins=0.1,del=0.1' -> normalised dict"""
    weights = {}
    for pair in text.split(','):
        kind, value = pair.split('=')
        weights[kind.strip()] = float(value)
    unknown = set(weights) - set(DEFAULT_WEIGHTS)
    if unknown:
        raise ValueError('unknown mutation type {}'.format(sorted(unknown)[0]))
    total = sum(weights.values())
    return {kind: value / total for kind, value in weights.items()}


def _genotype(code, rng):
    return '0/0' if code == 0 else ('1/1' if code == 2 else ('0/1', '1/0')[int(rng.integers(2))])


def draw_variants(sequences, ninh, ndenovo, weights, rng):
    """[(seqid, position, ref, alt, (gt_child, gt_mother, gt_father))], non-overlapping, sorted by (seqid, position)"""
    seqids = sorted(sequences)
    kinds = sorted(weights)
    probs = np.array([weights[k] for k in kinds])
    taken = {sid: [] for sid in seqids}
    out = []
    for i in range(ninh + ndenovo):
        for _attempt in range(1000):
            sid = seqids[int(rng.integers(len(seqids)))]
            seq = sequences[sid]
            kind = kinds[int(rng.choice(len(kinds), p=probs))]
            pos = int(rng.integers(0, len(seq)))
            if kind == 'snv':
                ref = seq[pos]
                if ref not in _NUCL:
                    continue
                alt = _NUCL[(_NUCL.index(ref) + int(rng.integers(1, 4))) % 4]
                span = 1
            elif kind == 'ins':
                length = int(rng.integers(5, 351))
                src = int(rng.integers(0, max(1, len(seq) - length)))
                piece = list(seq[src:src + length])                 # a copy from elsewhere, lightly mutated
                for j in np.flatnonzero(rng.random(len(piece)) < 0.05):
                    piece[j] = _NUCL[int(rng.integers(4))]
                ref, alt, span = seq[pos], seq[pos] + ''.join(piece), 1
            else:
                length = int(rng.integers(5, 351))
                if pos + 1 + length > len(seq):
                    continue
                ref, alt, span = seq[pos:pos + 1 + length], seq[pos], 1 + length
            if any(pos < b and a < pos + span for a, b in taken[sid]):
                continue
            taken[sid].append((pos, pos + span))
            if i < ninh:
                gts = tuple(_genotype(code, rng) for code in INHERITANCE[int(rng.integers(len(INHERITANCE)))])
            else:
                gts = (('0/1', '1/0')[int(rng.integers(2))], '0/0', '0/0')
            out.append((sid, pos, ref, alt, gts))
            break
        else:
            raise ValueError('could not place {} non-overlapping variants'.format(ninh + ndenovo))
    return sorted(out, key=lambda v: (v[0], v[1]))


def haplotype(sequence, variants):
    """`sequence` with the (position, ref, alt) variants applied, positions ascending and non-overlapping"""
    pieces, cursor = [], 0
    for pos, ref, alt in variants:
        pieces.append(sequence[cursor:pos])
        pieces.append(alt)
        cursor = pos + len(ref)
    pieces.append(sequence[cursor:])
    return ''.join(pieces)


def gentrio(sequences, outstreams, ninh=20, ndenovo=10, weights=None, seed=None, logstream=sys.stderr):
    """Write the three individuals' haplotypes to `outstreams` (proband, mother, father); yield the variants."""
    assert len(outstreams) == 3
    if seed is None:
        seed = int(np.random.SeedSequence().entropy % (2 ** 63))
        print('[kevlar::gentrio] using random seed', seed, file=logstream)
    rng = np.random.default_rng(seed)
    variants = draw_variants(sequences, ninh, ndenovo, weights or DEFAULT_WEIGHTS, rng)
    for sid in sequences:
        mine = [v for v in variants if v[0] == sid]
        for ind in range(3):
            for hap in range(2):
                carried = [(pos, ref, alt) for _, pos, ref, alt, gts in mine if gts[ind][2 * hap] == '1']
                print('>', sid, '_haplo', hap + 1, '\n', haplotype(sequences[sid], carried), sep='', file=outstreams[ind])
    for v in variants:
        yield v


def main(args):
    timer = kevlar_amd.Timer()
    timer.start()
    print('[kevlar::gentrio] Loading genome...', end='', file=kevlar_amd.logstream)
    sequences = {}
    for record in kevlar_amd.parse_augmented_fastx(kevlar_amd.open(args.genome, 'r')):
        sequences[record.name.split()[0]] = record.sequence
    print('done!', file=kevlar_amd.logstream)
    samples = ('proband', 'mother', 'father')
    outstreams = [kevlar_amd.open('{:s}-{:s}.fasta'.format(args.prefix, s), 'w') for s in samples]
    variants = list(gentrio(sequences, outstreams, ninh=args.inherited, ndenovo=args.de_novo, weights=parse_weights(args.weights),
                            seed=args.seed, logstream=kevlar_amd.logstream))
    for stream in outstreams:
        stream.close()
    if args.vcf:
        with kevlar_amd.open(args.vcf, 'w') as vcf:
            print('##fileformat=VCFv4.2', file=vcf)
            print('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">', file=vcf)
            print('#CHROM', 'POS', 'ID', 'REF', 'ALT', 'QUAL', 'FILTER', 'INFO', 'FORMAT', *samples, sep='\t', file=vcf)
            for sid, pos, ref, alt, gts in variants:
                print(sid, pos + 1, '.', ref, alt, '.', '.', '.', 'GT', *gts, sep='\t', file=vcf)
    print('[kevlar::gentrio] wrote {} variants, {} individuals in {:.2f} seconds'.format(len(variants), 3, timer.stop()),
          file=kevlar_amd.logstream)
