"""Golden vectors for the bench's OWN generator ("a small seeded synthetic trio"): the 50 kb / 10x /
k=31 trio of BASELINE.json config 1, produced by kevlar_amd.synth with the seeds bench.py uses, written as FASTQ and
run through the REFERENCE's drivers (count, novel, filter, partition) over the CPU oracle standing in for khmer --
the same harness as make_golden.py.  Runs in the build container only (needs /root/reference); the FASTQ inputs and
the reference's outputs are committed under tests/golden/.

    PYTHONHASHSEED=0 python tests/golden/make_golden_synth.py
"""
import gzip
import io
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden  # noqa: E402

DATA = os.path.join(HERE, 'data', 'synth-cfg1')
EXPECTED = os.path.join(HERE, 'expected')
GENOME, COVERAGE, READLEN, KSIZE, MEMORY = 50000, 10, 100, 31, '1M'


def write_inputs():
    from kevlar_amd import synth
    os.makedirs(DATA, exist_ok=True)
    packed = synth.trio_reads_packed(GENOME, COVERAGE, READLEN)
    for name, words in packed.items():
        seqs = synth.unpack_reads(words, READLEN)
        with gzip.GzipFile(os.path.join(DATA, name + '.fq.gz'), 'wb', mtime=0) as out:
            out.write(''.join('@{}_{}\n{}\n+\n{}\n'.format(name, i, s, 'I' * READLEN) for i, s in enumerate(seqs)).encode())
    return {name: os.path.join(DATA, name + '.fq.gz') for name in packed}


def main():
    files = write_inputs()
    kevlar, scratch = make_golden.import_reference()
    manifest = {'PYTHONHASHSEED': os.environ.get('PYTHONHASHSEED'), 'genome': GENOME, 'coverage': COVERAGE, 'ksize': KSIZE,
                'memory': MEMORY, 'cases': {}}
    work = os.path.join(scratch, 'work')
    os.makedirs(work)
    for name, path in files.items():
        _, log = make_golden.run_cli(kevlar, ['count', '--ksize', str(KSIZE), '--memory', MEMORY, os.path.join(work, name + '.ct'), path])
        manifest['cases']['count-' + name] = make_golden.keep_lines(log, ['reads processed', 'estimated false'])
        with open(os.path.join(work, name + '.ct'), 'rb') as fh:
            import hashlib
            manifest['cases']['count-' + name].append('md5 ' + hashlib.md5(fh.read()).hexdigest())
    out, log = make_golden.run_cli(kevlar, ['novel', '--ksize', str(KSIZE), '--memory', MEMORY, '--case', files['proband'],
                                            '--control', files['mother'], '--control', files['father'],
                                            '--case-min', '6', '--ctrl-max', '1'])
    with open(os.path.join(EXPECTED, 'novel-synth-cfg1.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['novel-synth-cfg1.augfastq'] = make_golden.keep_lines(log, ['Found', 'reads processed'])
    novel_path = os.path.join(work, 'novel.augfastq')
    with open(novel_path, 'w') as fh:
        fh.write(out)
    # (the reference's `kevlar filter` command line cannot run without --mask, kevlar/filter.py:100: call the function)
    buf, logbuf = io.StringIO(), io.StringIO()
    kevlar.logstream = logbuf
    for rec in kevlar.filter.filter(novel_path, memory=5e5, casemin=6, ctrlmax=1):
        kevlar.print_augmented_fastx(rec, buf)
    kevlar.logstream = None
    out, log = buf.getvalue(), logbuf.getvalue()
    with open(os.path.join(EXPECTED, 'filter-synth-cfg1.augfastq'), 'w') as fh:
        fh.write(out)
    manifest['cases']['filter-synth-cfg1.augfastq'] = make_golden.keep_lines(log, ['Processed', 'Validated', 'FPR for'])
    filt_path = os.path.join(work, 'filtered.augfastq')
    with open(filt_path, 'w') as fh:
        fh.write(out)
    out, log = make_golden.run_cli(kevlar, ['partition', filt_path])
    parts = {}
    if out.strip():
        for rec in kevlar.parse_augmented_fastx(io.StringIO(out)):
            pid = kevlar.seqio.partition_id(rec.name)
            parts.setdefault(pid, []).append([rec.name.rsplit(' kvcc=', 1)[0], kevlar.revcommin(rec.sequence)])
    for pid in parts:
        parts[pid].sort()
    with open(os.path.join(EXPECTED, 'partition-synth-cfg1.json'), 'w') as fh:
        json.dump({'partitions': parts, 'log': make_golden.keep_lines(log, ['grouped'])}, fh, indent=0, sort_keys=True)
    with open(os.path.join(EXPECTED, 'manifest-synth.json'), 'w') as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    shutil.rmtree(scratch, ignore_errors=True)
    print(json.dumps(manifest['cases'], indent=1))


if __name__ == '__main__':
    main()
