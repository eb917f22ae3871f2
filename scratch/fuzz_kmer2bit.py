"""Randomised parity of the per-k-mer kernels that hash from the 2-bit form (k_bin_hash_2bit, k_novel_mark_2bit) against the scalar oracle:
random k (16..64), read length, storage, table size, bands (range rule and the reference's quirk), masks, thresholds, reads with bases outside
ACGT, a first read.  The batches are device-parsed FASTQ files (the arithmetic layout the kernels need).  usage: fuzz_kmer2bit.py [trials] [seed]"""
import ctypes
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from kevlar_amd import _lib, khmer as hk, synth
from oracle import okhmer as ok


def launches(lib, name):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    lib.kv_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n))
    return n.value


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    lib = _lib.load()
    _lib.require_device()
    lib.kv_prof_enable(1)
    tmp = tempfile.mkdtemp(prefix='kv_fuzz2bit_')
    os.environ['KV_COUNT_PATH'] = 'binned'
    bad = 0
    for trial in range(trials):
        k = rng.choice([16, 17, 24, 25, 31, 32, 33, 40, 47, 48, 49, 51, 63, 64, rng.randrange(16, 65)])
        L = rng.choice([k, k + 1, k + 9, k + 10, k + 11, 100, 101, 150, rng.randrange(k, 260)])
        n = rng.randrange(300, 6000)
        kind = rng.choice(['Counttable', 'Counttable', 'SmallCounttable', 'Nodetable'])
        size = rng.choice([5e4, 3e5, 2e6])
        trio = synth.make_trio(rng.randrange(4000, 60000) + L, rng.randrange(1 << 30), inherited_per_mb=500, denovo_per_mb=3000)
        reads = {}
        for i, name in enumerate(('proband', 'mother', 'father')):
            reads[name] = synth.unpack_reads(synth.sample_reads_packed(trio[name], n, L, rng.choice([0.0, 0.005, 0.02]), rng.randrange(1 << 30)), L)
        sample = list(reads['proband'])
        for _ in range(rng.randrange(0, 4)):
            i = rng.randrange(n)
            j = rng.randrange(L)
            sample[i] = sample[i][:j] + rng.choice('NnRacgt') + sample[i][j + 1:]
        path = os.path.join(tmp, 't{}.fq'.format(trial))
        with open(path, 'w') as fh:
            for i, seq in enumerate(sample):
                fh.write('@r{}\n{}\n+\n{}\n'.format(i, seq, 'I' * L))
        parser = hk.ReadParser(path)
        text = parser.text_batch(8 * n + 64)
        assert text.n == n, (text.n, n)
        batch = text.batch
        bases, offs = ok.concat_reads(sample)
        nbands = rng.choice([0, 0, 2, 3, 8, 16])
        band = rng.randrange(nbands) if nbands else 0
        use_mask = rng.random() < 0.3
        masked = use_mask and rng.random() < 0.5
        mask_dev = mask_ref = None
        if use_mask:
            mask_dev, mask_ref = hk.Nodetable(k, 1e6, 2), ok.Nodetable(k, 1e6, 2)
            mask_dev.consume_batch(hk.ReadBatch(reads['mother'][:n // 2]))
            mb, mo = ok.concat_reads(reads['mother'][:n // 2])
            ok.consume_reads(mask_ref, mb, mo, n // 2)
        what = dict(trial=trial, k=k, L=L, n=n, kind=kind, size=size, nbands=nbands, band=band, mask=use_mask, masked=masked)
        lib.kv_prof_reset()
        dev, ref = getattr(hk, kind)(k, size, 4), getattr(ok, kind)(k, size, 4)
        n_dev = dev.consume_batch(batch, nbands, band, mask=mask_dev, threshold=1 if masked else 0, consume_masked=masked)
        n_ref = ok.consume_reads(ref, bases, offs, n, nbands, band, mask_ref, 1 if masked else 0, masked)
        ok_count = n_dev == n_ref and all(dev.table_bytes(t) == ref.table_bytes(t) for t in range(4)) and dev.n_occupied() == ref.n_occupied()
        ran = launches(lib, 'k_bin_hash_2bit')
        if not ok_count or ran != 1:
            bad += 1
            print('COUNT MISMATCH', what, n_dev, n_ref, 'launches', ran, flush=True)
        # scan: counts of the three samples (unbanded), then the case sample's batch under a band rule
        devs, refs = {}, {}
        for name, seqs in (('proband', sample), ('mother', reads['mother']), ('father', reads['father'])):
            devs[name], refs[name] = hk.Counttable(k, size, 4), ok.Counttable(k, size, 4)
            devs[name].consume_batch(hk.ReadBatch(seqs))
            b2, o2 = ok.concat_reads(seqs)
            ok.consume_reads(refs[name], b2, o2, len(seqs))
        band_mode = rng.choice([0, 0, 1, 2])
        nb = rng.choice([2, 4, 8]) if band_mode else 0
        bnd = rng.randrange(nb) if nb else 0
        first = rng.choice([0, 0, rng.randrange(n)])
        case_min, ctrl_max = rng.choice([2, 4, 6]), rng.choice([0, 1, 2])
        want, _ = ok.novel_scan([refs['proband']], [refs['mother'], refs['father']], bases, offs, n, k, case_min, ctrl_max, band_mode=band_mode, nbands=nb, band=bnd)
        want = [h for h in want if h[0] >= first]
        lib.kv_prof_reset()
        r, o, a, _ = hk.novel_scan([devs['proband']], [devs['mother'], devs['father']], batch, case_min, ctrl_max, band_mode=band_mode, nbands=nb, band=bnd, first_read=first)
        got = [(int(r[i]), int(o[i]), tuple(int(x) for x in a[i])) for i in range(len(r))]
        ran = launches(lib, 'k_novel_mark_2bit')
        if got != want or ran != 1:
            bad += 1
            print('SCAN MISMATCH', dict(what, band_mode=band_mode, nb=nb, bnd=bnd, first=first, case_min=case_min, ctrl_max=ctrl_max), len(got), len(want), 'launches', ran, flush=True)
        os.remove(path)
        if (trial + 1) % 25 == 0:
            print('{} trials, {} mismatches'.format(trial + 1, bad), flush=True)
    print('done: {} trials (seed {}), {} mismatches'.format(trials, seed, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
