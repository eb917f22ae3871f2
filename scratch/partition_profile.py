"""cProfile of kevlar partition / filter on a config-2-size filtered file (needs /tmp/kv_pipe from scratch/pipeline_cfg2.py)"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kevlar_amd
from kevlar_amd import _lib
_lib.load(); _lib.require_device()
kevlar_amd.logstream = io.StringIO()
out = '/tmp/kv_pipe'
for cmd, argv in (('filter', ['filter', '--memory', '200M', '-o', out + '/f2.augfastq', out + '/novel.augfastq']),
                  ('partition', ['partition', '-o', out + '/p2.augfastq', out + '/filtered.augfastq'])):
    a = kevlar_amd.cli.parser().parse_args(argv)
    kevlar_amd.cli.mains[a.cmd](a)          # warm
    prof = cProfile.Profile(); t0 = time.perf_counter()
    prof.enable(); kevlar_amd.cli.mains[a.cmd](a); prof.disable()
    print(cmd, '{:.3f} s'.format(time.perf_counter() - t0))
    st = io.StringIO(); pstats.Stats(prof, stream=st).sort_stats('tottime').print_stats(14); print(st.getvalue()[:3500])
