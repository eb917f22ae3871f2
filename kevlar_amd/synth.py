"""Seeded synthetic trio generator for benchmarks and parity tests.

The reference's `kevlar gentrio` (kevlar/gentrio.py:185-257) emits haplotype FASTA only; the
reads in its tests came from an external `wgsim -e 0.005` (kevlar/tests/data/microtrios/README).
This module does both steps with numpy: an iid-uniform reference genome, inherited
heterozygous variants in the parents, de novo variants in the proband only, and 2 x coverage
reads sampled uniformly from an individual's two haplotypes, random strand, substitution errors.
Everything is a pure function of the seeds, so every rank / every run sees identical reads.

Reads are produced as 2-bit packed words (kv_reads_create_packed layout); `unpack_reads`
turns any subset back into ASCII for the CPU oracle.
"""
import numpy as np

ALPHABET = np.frombuffer(b'ACGT', dtype=np.uint8)


def make_genome(length, seed=42):
    return np.random.default_rng(seed).integers(0, 4, size=length, dtype=np.uint8)


def _apply_variants(genome, positions, kinds, rng):
    """kinds: 0 SNV, 1 insertion (1-8 random bases), 2 deletion (1-8 bases)."""
    hap = genome.copy()
    snv = positions[kinds == 0]
    hap[snv] = (hap[snv] + rng.integers(1, 4, size=len(snv), dtype=np.uint8)) & 3
    ins = np.sort(positions[kinds == 1])
    dele = np.sort(positions[kinds == 2])
    pieces, cursor = [], 0
    events = sorted([(int(p), 1) for p in ins] + [(int(p), 2) for p in dele])
    for pos, kind in events:
        if pos < cursor:
            continue
        pieces.append(hap[cursor:pos])
        n = int(rng.integers(1, 9))
        if kind == 1:
            pieces.append(rng.integers(0, 4, size=n, dtype=np.uint8))
            cursor = pos
        else:
            cursor = min(len(hap), pos + n)
    pieces.append(hap[cursor:])
    return np.concatenate(pieces)


def make_trio(genome_len, seed=42, inherited_per_mb=400, denovo_per_mb=200, weights=(0.8, 0.1, 0.1), extra_controls=0):
    """Returns {'proband': (hap1, hap2), 'mother': ..., 'father': ...} of uint8 code arrays.

    Defaults reproduce gentrio's -i 20 -d 10 on a 50 kb genome (cli/gentrio.py:17-36) and scale
    with genome size."""
    rng = np.random.default_rng(seed)
    genome = make_genome(genome_len, seed)
    n_inh = max(1, int(genome_len * inherited_per_mb / 1e6))
    n_dn = max(1, int(genome_len * denovo_per_mb / 1e6))
    pos = rng.choice(genome_len - 200, size=n_inh + n_dn, replace=False) + 100
    kinds = rng.choice(3, size=n_inh + n_dn, p=weights)
    owner = rng.integers(0, 4, size=n_inh)           # which parental haplotype carries it
    inh_pos, inh_kind = pos[:n_inh], kinds[:n_inh]
    dn_pos, dn_kind = pos[n_inh:], kinds[n_inh:]
    parents = [_apply_variants(genome, inh_pos[owner == h], inh_kind[owner == h], rng) for h in range(4)]
    # proband: father's haplotype 0 and mother's haplotype 0 (no recombination), plus the
    # de novo variants, each on one of its two haplotypes
    dn_hap = rng.integers(0, 2, size=n_dn)
    kid = []
    for h, parent_hap in enumerate((0, 2)):
        sel_inh = owner == parent_hap
        p = np.concatenate((inh_pos[sel_inh], dn_pos[dn_hap == h]))
        k = np.concatenate((inh_kind[sel_inh], dn_kind[dn_hap == h]))
        kid.append(_apply_variants(genome, p, k, rng))
    fam = {'father': (parents[0], parents[1]), 'mother': (parents[2], parents[3]), 'proband': tuple(kid)}
    # further controls (BASELINE.json config 5: proband + 3 controls): siblings that inherited other haplotype pairs
    # and carry no de novo variant; derived without touching the generator, so the trio itself is unchanged
    for i, (fh, mh) in enumerate(((1, 3), (0, 3), (1, 2))[:max(0, extra_controls)]):
        fam['sibling{}'.format(i + 1)] = (parents[fh], parents[mh])
    return fam


def sample_reads_packed(haps, n_reads, read_len=100, error_rate=0.005, seed=1001, chunk=1 << 20):
    """n_reads reads of read_len from the two haplotypes -> uint32 [n_reads, ceil(read_len/16)]."""
    rng = np.random.default_rng(seed)
    wpr = (read_len + 15) // 16
    out = np.empty((n_reads, wpr), dtype=np.uint32)
    out8 = out.view(np.uint8).reshape(n_reads, wpr * 4)
    windows = [np.lib.stride_tricks.sliding_window_view(h, read_len) for h in haps]
    nfull = read_len // 4
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        which = rng.integers(0, 2, size=n)
        codes = np.empty((n, read_len), dtype=np.uint8)
        for h in (0, 1):
            sel = np.flatnonzero(which == h)
            starts = rng.integers(0, len(haps[h]) - read_len + 1, size=len(sel))
            codes[sel] = windows[h][starts]
        flip = np.flatnonzero(rng.random(n) < 0.5)
        codes[flip] = 3 - codes[flip, ::-1]
        n_err = rng.binomial(n * read_len, error_rate)
        flat = codes.reshape(-1)
        epos = rng.integers(0, flat.size, size=n_err)
        flat[epos] = (flat[epos] + rng.integers(1, 4, size=n_err, dtype=np.uint8)) & 3
        # 4 bases per byte, little-endian within the 32-bit word: base j -> bits 2*(j%16) of word j//16
        dst = out8[lo:lo + n]
        dst[:] = 0
        quad = codes[:, :nfull * 4].reshape(n, nfull, 4)
        dst[:, :nfull] = quad[:, :, 0] | (quad[:, :, 1] << 2) | (quad[:, :, 2] << 4) | (quad[:, :, 3] << 6)
        for j in range(nfull * 4, read_len):
            dst[:, j // 4] |= codes[:, j] << (2 * (j % 4))
    return out


def unpack_reads(words, read_len):
    """Packed words -> list of ASCII strings (for the oracle / for writing FASTQ)."""
    n, wpr = words.shape
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    codes = ((words[:, :, None] >> shifts) & 3).reshape(n, wpr * 16)[:, :read_len].astype(np.uint8)
    ascii_ = ALPHABET[codes]
    return [row.tobytes().decode('ascii') for row in ascii_]


def trio_reads_packed(genome_len, coverage, read_len=100, seed=42, error_rate=0.005, extra_controls=0):
    """{'proband': words, 'mother': words, 'father': words[, 'sibling1': ...]}; seeds 1001/1002/1003(/1004...) per sample."""
    trio = make_trio(genome_len, seed, extra_controls=extra_controls)
    n_reads = int(genome_len * coverage / read_len)
    names = ['proband', 'mother', 'father'] + ['sibling{}'.format(i + 1) for i in range(extra_controls)]
    return {name: sample_reads_packed(trio[name], n_reads, read_len, error_rate, 1001 + i) for i, name in enumerate(names)}


# ---- the family kv_synth.hip generates on the device (kv_reads_generate), restated with numpy -------------------------
_M = np.uint64(0xffffffffffffffff)


def _mix(x):
    x = (x + np.uint64(0x9e3779b97f4a7c15)) & _M
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & _M
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & _M
    return x ^ (x >> np.uint64(31))


def device_family_bases(seed, pos, hap):
    """2-bit codes at positions `pos` (uint64 array) of haplotypes `hap` (same shape; 0..3 parental, 4 / 5 the proband's)"""
    with np.errstate(over='ignore'):
        seed = np.uint64(seed)
        pos = pos.astype(np.uint64)
        b = (_mix(seed ^ (pos * np.uint64(0x2545f4914f6cdd1d))) & np.uint64(3)).astype(np.uint32)
        parental = np.where(hap < 4, hap, np.where(hap == 4, 0, 2)).astype(np.uint64)
        hi = _mix((seed + np.uint64(1)) ^ (pos * np.uint64(0x9fb21c651e98df25)))
        inh = (hi % np.uint64(2500) == 0) & (((hi // np.uint64(2500)) & np.uint64(3)) == parental)
        b = np.where(inh, (b + 1 + ((hi >> np.uint64(40)) % np.uint64(3)).astype(np.uint32)) & 3, b)
        hd = _mix((seed + np.uint64(2)) ^ (pos * np.uint64(0xd6e8feb86659fd93)))
        dn = (hap >= 4) & (hd % np.uint64(5000) == 0) & (((hd // np.uint64(5000)) & np.uint64(1)) == (hap.astype(np.int64) - 4).astype(np.uint64))
        return np.where(dn, (b + 1 + ((hd >> np.uint64(40)) % np.uint64(3)).astype(np.uint32)) & 3, b).astype(np.uint8)


def device_family_reads(genome_len, seed, sample, read_index, read_len=100, error_rate=0.005):
    """codes [len(read_index), read_len] (uint8) of the given reads (global indices) of sample 0 proband / 1 mother / 2 father"""
    with np.errstate(over='ignore'):
        i = np.asarray(read_index, dtype=np.uint64)
        sseed = np.uint64(seed) + np.uint64(16 + 8 * sample)
        rr = _mix(sseed ^ (i * np.uint64(0xa0761d6478bd642f)))
        which = (rr & np.uint64(1)).astype(np.int64)
        flip = ((rr >> np.uint64(1)) & np.uint64(1)).astype(bool)
        start = (rr >> np.uint64(8)) % np.uint64(genome_len - read_len + 1)
        hap = 4 + which if sample == 0 else (which if sample == 2 else 2 + which)
        j = np.arange(read_len, dtype=np.uint64)[None, :]
        src = np.where(flip[:, None], start[:, None] + (np.uint64(read_len - 1) - j), start[:, None] + j)
        codes = device_family_bases(seed, src, np.broadcast_to(hap[:, None], src.shape))
        codes = np.where(flip[:, None], 3 - codes, codes).astype(np.uint32)
        e = _mix((sseed + np.uint64(1)) ^ ((i[:, None] * np.uint64(4096) + j) * np.uint64(0xe7037ed1a0b428db)))
        bad = (e & np.uint64(0xffffffff)) < np.uint64(int(error_rate * 4294967296.0))
        return np.where(bad, (codes + 1 + ((e >> np.uint64(32)) % np.uint64(3)).astype(np.uint32)) & 3, codes).astype(np.uint8)


def pack_codes(codes):
    """[n, L] codes -> uint32 [n, ceil(L / 16)] in the kv_reads layout"""
    n, L = codes.shape
    wpr = (L + 15) // 16
    padded = np.zeros((n, wpr * 16), dtype=np.uint32)
    padded[:, :L] = codes
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    return (padded.reshape(n, wpr, 16) << shifts).sum(axis=2, dtype=np.uint32)
