"""End-to-end CLI timing on synthetic FASTQ files (host parse + pack + PCIe + kernels + text out)."""
import io, os, sys, time, contextlib
sys.path.insert(0, '.')
import numpy as np
import kevlar_amd
from kevlar_amd import synth
out = '/tmp/kv_e2e'; os.makedirs(out, exist_ok=True)
G, L = 3_400_000, 100
packed = synth.trio_reads_packed(G, 30, L)
n = packed['proband'].shape[0]
qual = 'I' * L
for name, words in packed.items():
    seqs = synth.unpack_reads(words, L)
    with open('%s/%s.fq' % (out, name), 'w') as fh:
        fh.write(''.join('@%s_%d\n%s\n+\n%s\n' % (name, i, s, qual) for i, s in enumerate(seqs)))
print('reads per sample', n, 'file MB', os.path.getsize(out + '/proband.fq') / 1e6)
kevlar_amd.logstream = io.StringIO()
def run(args):
    a = kevlar_amd.cli.parser().parse_args(args)
    t = time.time(); kevlar_amd.cli.mains[a.cmd](a); return time.time() - t
run(['count', '--memory', '300M', out + '/warm.ct', out + '/father.fq'])   # warm-up (library init)
tc = [run(['count', '--memory', '300M', '--threads', '2', '%s/%s.ct' % (out, s), '%s/%s.fq' % (out, s)]) for s in ('proband', 'mother', 'father')]
tc1 = run(['count', '--memory', '300M', out + '/p1.ct', out + '/proband.fq'])
tn = run(['novel', '--case', out + '/proband.fq', '--case-counts', out + '/proband.ct', '--control-counts', out + '/mother.ct', out + '/father.ct', '-o', out + '/novel.augfastq'])
tf = run(['filter', '--memory', '50M', '-o', out + '/filtered.augfastq', out + '/novel.augfastq'])
tp = run(['partition', '-o', out + '/part.augfastq', out + '/filtered.augfastq'])
print('count (threads=2, est. distinct) s per sample:', [round(x, 2) for x in tc], '-> %.2f M reads/s' % (n / np.mean(tc) / 1e6))
print('count (threads=1, exact distinct) %.2f s' % tc1)
print('novel %.2f s (%.2f M reads/s), filter %.2f s, partition %.2f s' % (tn, n / tn / 1e6, tf, tp))
log = kevlar_amd.logstream.getvalue()
print([l for l in log.split('\n') if 'Found' in l or 'grouped' in l or 'Validated' in l])
