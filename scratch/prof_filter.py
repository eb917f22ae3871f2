import io, os, sys, time, cProfile, pstats
sys.path.insert(0, '.')
import kevlar_amd
out = '/tmp/kv_e2e'
kevlar_amd.logstream = io.StringIO()
def run(args):
    a = kevlar_amd.cli.parser().parse_args(args)
    kevlar_amd.cli.mains[a.cmd](a)
for cmd in (['filter', '--memory', '50M', '-o', out + '/filtered.augfastq', out + '/novel.augfastq'],
            ['partition', '-o', out + '/part.augfastq', out + '/filtered.augfastq'],
            ['novel', '--case', out + '/proband.fq', '--case-counts', out + '/proband.ct', '--control-counts', out + '/mother.ct', out + '/father.ct', '-o', out + '/novel2.augfastq']):
    pr = cProfile.Profile(); pr.enable(); t = time.time(); run(cmd); dt = time.time() - t; pr.disable()
    print('====', cmd[0], '%.2f s' % dt)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(18); print('\n'.join(s.getvalue().split('\n')[6:34]))
