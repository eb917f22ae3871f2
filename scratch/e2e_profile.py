"""where the end-to-end `kevlar novel` time goes (cProfile over the CLI driver): python scratch/e2e_profile.py [reads] [bgzf|plain]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kevlar_amd
from kevlar_amd import bgzf, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
kind = sys.argv[2] if len(sys.argv) > 2 else 'bgzf'
out = '/tmp/kv_e2e_prof'; os.makedirs(out, exist_ok=True)
packed = synth.trio_reads_packed(25_000_000, 30, 100)
rng = np.random.default_rng(12)
suffix = '.fq.gz' if kind == 'bgzf' else '.fq'
for name, words in packed.items():
    seqs = synth.unpack_reads(words[:n], 100)
    quals = np.frombuffer(b'F:,#', dtype=np.uint8)[rng.choice(4, size=(n, 100), p=[0.9, 0.06, 0.03, 0.01])]
    text = ''.join('@{}_{}\n{}\n+\n{}\n'.format(name, i, s, q.tobytes().decode('ascii')) for i, (s, q) in enumerate(zip(seqs, quals)))
    sink = bgzf.BgzfWriter(out + '/' + name + suffix, level=4) if kind == 'bgzf' else open(out + '/' + name + suffix, 'w')
    sink.write(text); sink.close()
kevlar_amd.logstream = io.StringIO()
_lib.load(); _lib.require_device()
argv = ['novel', '--ksize', '31', '--memory', '2000000000', '--threads', '2', '--case', out + '/proband' + suffix, '--control', out + '/mother' + suffix,
        '--control', out + '/father' + suffix, '--case-min', '6', '--ctrl-max', '1', '-o', out + '/novel.augfastq']
a = kevlar_amd.cli.parser().parse_args(argv)
for rep in range(2):
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable(); kevlar_amd.cli.mains[a.cmd](a); prof.disable()
    print('run {}: {:.3f} s = {:.2f} M reads/s'.format(rep, time.perf_counter() - t0, 3 * n / (time.perf_counter() - t0) / 1e6))
st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('cumtime').print_stats(28); print(st.getvalue()[:6000])
