"""count -> novel -> filter -> partition through the CLI drivers at config-2 scale, BGZF input: where the seconds go"""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kevlar_amd
from kevlar_amd import bgzf, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7500000
out = '/tmp/kv_pipe'; os.makedirs(out, exist_ok=True)
L = 100
packed = synth.trio_reads_packed(25_000_000, 30, L)
rng = np.random.default_rng(12)
t0 = time.time()
for name in packed:
    tag = '@{}_'.format(name).encode('ascii')
    rec = np.empty((n, len(tag) + 8 + 1 + L + 3 + L + 1), dtype=np.uint8)
    col = 0
    rec[:, :len(tag)] = np.frombuffer(tag, dtype=np.uint8); col += len(tag)
    digits = np.arange(n, dtype=np.int64)
    for d in range(8):
        rec[:, col + 7 - d] = 48 + digits % 10
        digits //= 10
    col += 8; rec[:, col] = 10; col += 1
    words = packed[name][:n]
    for j in range(L):
        rec[:, col + j] = np.frombuffer(b'ACGT', dtype=np.uint8)[(words[:, j >> 4] >> np.uint32(2 * (j & 15))) & np.uint32(3)]
    col += L
    rec[:, col:col + 3] = np.frombuffer(b'\n+\n', dtype=np.uint8); col += 3
    rec[:, col:col + L] = np.frombuffer(b'F:,#', dtype=np.uint8)[rng.choice(4, size=(n, L), p=[0.9, 0.06, 0.03, 0.01])]; col += L
    rec[:, col] = 10
    bgzf.write_file(out + '/' + name + '.fq.gz', rec.tobytes(), level=1, threads=16)
    del rec
print('files written in {:.1f} s'.format(time.time() - t0), flush=True)
kevlar_amd.logstream = io.StringIO()
_lib.load(); _lib.require_device()
def run(argv):
    a = kevlar_amd.cli.parser().parse_args(argv)
    t = time.perf_counter(); kevlar_amd.cli.mains[a.cmd](a); return time.perf_counter() - t
run(['count', '--memory', '2G', '--max-fpr', '0.99', out + '/warm.ct', out + '/father.fq.gz'])
tn = run(['novel', '--ksize', '31', '--memory', '2G', '--threads', '2', '--case', out + '/proband.fq.gz', '--control', out + '/mother.fq.gz', '--control', out + '/father.fq.gz',
          '--case-min', '6', '--ctrl-max', '1', '-o', out + '/novel.augfastq'])
print('novel (count 3 samples + scan + write): {:.2f} s = {:.1f} M reads/s; output {} MB'.format(tn, 3 * n / tn / 1e6, os.path.getsize(out + '/novel.augfastq') >> 20), flush=True)
tf = run(['filter', '--memory', '200M', '-o', out + '/filtered.augfastq', out + '/novel.augfastq'])
print('filter: {:.2f} s; output {} MB'.format(tf, os.path.getsize(out + '/filtered.augfastq') >> 20), flush=True)
if os.environ.get('PROFILE'):
    import cProfile, pstats
    prof = cProfile.Profile(); prof.enable()
tp = run(['partition', '-o', out + '/part.augfastq', out + '/filtered.augfastq'])
if os.environ.get('PROFILE'):
    prof.disable(); st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(14); print(st.getvalue()[:3500])
    prof = cProfile.Profile(); prof.enable()
    run(['filter', '--memory', '200M', '-o', out + '/filtered2.augfastq', out + '/novel.augfastq'])
    prof.disable(); st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(12); print(st.getvalue()[:3000])
print('partition: {:.2f} s'.format(tp))
log = kevlar_amd.logstream.getvalue()
print('\n'.join(l for l in log.split('\n') if 'Found' in l or 'grouped' in l or 'Validated' in l or 'reads' in l.lower())[-1500:])
