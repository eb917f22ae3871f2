"""Randomised parity of the array paths of filter / partition / split / unband against the record-object paths, on
the novel output of random synthetic trios: python scratch/fuzz_host.py [trials] [seed]"""
import io, os, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import kevlar_amd
from kevlar_amd import _lib, khmer as hk, synth
from kevlar_amd.sequence import format_augmented_fastx, parse_augmented_fastx
_lib.load(); _lib.require_device()
kevlar_amd.logstream = io.StringIO()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
fails = 0


def records(path):
    with kevlar_amd.open(path, 'r') as fh:
        return list(parse_augmented_fastx(fh))


with tempfile.TemporaryDirectory() as tmp:
    for trial in range(trials):
        k = int(rng.choice([21, 25, 31, 45]))
        n = int(rng.choice([2000, 20000]))
        L = int(rng.choice([100, 151]))
        trio = synth.make_trio(int(rng.choice([20000, 150000])), int(rng.integers(0, 1 << 30)))
        files = {}
        for i, name in enumerate(('proband', 'mother', 'father')):
            seqs = synth.unpack_reads(synth.sample_reads_packed(trio[name], n, L, 0.004, int(rng.integers(0, 1 << 30))), L)
            if name == 'proband' and rng.random() < 0.5:            # duplicate sequences and a duplicated name downstream
                seqs[10:20] = seqs[:10]
            files[name] = os.path.join(tmp, '{}{}.fq'.format(name, trial))
            with open(files[name], 'w') as fh:
                fh.write(''.join('@{}_{}\n{}\n+\n{}\n'.format(name, j, s, 'I' * len(s)) for j, s in enumerate(seqs)))
        novel_out = os.path.join(tmp, 'novel{}.augfastq'.format(trial)) + ('.gz' if rng.random() < 0.3 else '')
        desc = 'trial {} k={} n={} L={}'.format(trial, k, n, L)
        try:
            a = kevlar_amd.cli.parser().parse_args(['novel', '--case', files['proband'], '--control', files['mother'], '--control', files['father'],
                                                    '--ksize', str(k), '--memory', '4M', '--case-min', '5', '--ctrl-max', '1', '-o', novel_out])
            kevlar_amd.novel.main(a)
            src = records(novel_out)
            if not src or src[0] is None:
                print('skip', desc, 'no novel reads'); continue
            # ---- filter: file (arrays) vs record stream (objects)
            casemin, ctrlmax = int(rng.integers(3, 8)), int(rng.integers(0, 3))
            mem = float(rng.choice([5e4, 1e6]))
            want = ''.join(format_augmented_fastx(r) for r in kevlar_amd.filter.filter(parse_augmented_fastx(kevlar_amd.open(novel_out, 'r')),
                                                                                      memory=mem, maxfpr=1.0, casemin=casemin, ctrlmax=ctrlmax))
            got = b''.join(kevlar_amd.filter._passes(novel_out, None, mem, 1.0, casemin, ctrlmax, as_text=True)).decode('latin-1')
            assert got == want, (desc, 'filter', len(got), len(want))
            # the same through the sink the CLI hands over (a plain file: the library renders and writes it, kv_format_records_fd)
            sunk = os.path.join(tmp, 'sunk{}.augfastq'.format(trial))
            with kevlar_amd.open_sink(sunk) as sink:
                for text in kevlar_amd.filter._passes(novel_out, None, mem, 1.0, casemin, ctrlmax, as_text=True, sink=sink):
                    if text:
                        sink.write(text)
            assert open(sunk).read() == want, (desc, 'filter to a sink')
            filtered = os.path.join(tmp, 'filtered{}.augfastq'.format(trial))
            with open(filtered, 'w') as fh:
                fh.write(got)
            # ---- partition
            dedup = bool(rng.random() < 0.7)
            minab, maxab = (None, None) if rng.random() < 0.5 else (int(rng.integers(1, 4)), int(rng.integers(50, 300)))
            parts_obj = [(num, ''.join(format_augmented_fastx(r) for r in (reads if dedup else sorted(reads, key=lambda r: r.name))))
                         for num, reads in kevlar_amd.partition.partition(parse_augmented_fastx(kevlar_amd.open(filtered, 'r')), strict=False,
                                                                         minabund=minab, maxabund=maxab, dedup=dedup)]
            # partition_file: (annotated reads, read indices in output order, partition number of each) -- cut into partitions here
            ann, out_reads, numbers = kevlar_amd.partition.partition_file(filtered, minabund=minab, maxabund=maxab, dedup=dedup)
            parts_arr = []
            for num in sorted(set(numbers.tolist())):
                mine = out_reads[numbers == num]
                parts_arr.append((int(num), ann.format(mine, suffixes=[' kvcc={:d}'.format(int(num))] * len(mine)).decode('latin-1')))
            # ... and the same text through the writer the CLI uses for a plain file (kv_format_records_fd)
            whole = os.path.join(tmp, 'whole{}.augfastq'.format(trial))
            with kevlar_amd.open_sink(whole) as sink:
                for num, text in parts_arr:
                    mine = out_reads[numbers == num]
                    ann.format_to(sink, mine, suffixes=[' kvcc={:d}'.format(num)] * len(mine))
            assert open(whole).read() == ''.join(text for _, text in parts_arr), (desc, 'format_to')
            assert parts_arr == parts_obj, (desc, 'partition', dedup, minab, maxab, len(parts_arr), len(parts_obj))
            parted = os.path.join(tmp, 'part{}.augfastq'.format(trial))
            with open(parted, 'w') as fh:
                fh.write(''.join(text for _, text in parts_arr))
            # ---- split
            nfiles = int(rng.integers(1, 5)); maxreads = int(rng.choice([5, 10000]))
            obj = [io.StringIO() for _ in range(nfiles)]; arr = [io.BytesIO() for _ in range(nfiles)]
            kevlar_amd.split.split(kevlar_amd.parse_partitioned_reads(parse_augmented_fastx(kevlar_amd.open(parted, 'r'))), obj, maxreads=maxreads)
            kevlar_amd.split.split_file(parted, arr, maxreads=maxreads)
            assert [s.getvalue().encode('latin-1') for s in obj] == [s.getvalue() for s in arr], (desc, 'split')
            # ---- unband: the novel output cut into band files
            nb = int(rng.integers(2, 5)); paths = []
            for b in range(nb):
                recs = []
                for i, rec in enumerate(src):
                    notes = [kk for j, kk in enumerate(rec.annotations) if (j + i) % nb == b]
                    if notes:
                        recs.append(kevlar_amd.sequence.Record(rec.name, rec.sequence, rec.quality, annotations=notes))
                paths.append(os.path.join(tmp, 'band{}_{}.augfastq'.format(trial, b)))
                with open(paths[-1], 'w') as fh:
                    fh.write(''.join(format_augmented_fastx(r) for r in recs))
            nbatch = int(rng.choice([1, 4, 16]))
            want = ''.join(format_augmented_fastx(r) for r in kevlar_amd.unband.unband(kevlar_amd.seqio.afxstream(paths), nbatch)).encode('latin-1')
            assert kevlar_amd.unband.unband_files(paths, nbatch) == want, (desc, 'unband')
            print('ok  ', desc, '{} novel reads, {} partitions'.format(len(src), len(parts_arr)), flush=True)
        except Exception as exc:
            fails += 1
            print('FAIL', desc, repr(exc)[:400], flush=True)
print('{} trials, {} failures'.format(trials, fails))
sys.exit(1 if fails else 0)
