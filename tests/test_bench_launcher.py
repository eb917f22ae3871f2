"""bench.py --gpus N without a launcher around it starts its own ranks: the parent spawns
`python -m torch.distributed.run` as a child before anything touches a GPU and relays rank 0's line and the exit code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def run_bench(*args, timeout=600):
    env = dict(os.environ)
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(key, None)
    return subprocess.run([sys.executable, BENCH] + list(args), cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, universal_newlines=True)


def last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith('{')]
    assert lines, 'no JSON line in: ' + text[-2000:]
    return json.loads(lines[-1])


def test_self_launch_two_ranks_meet_over_gloo():
    res = run_bench('--gpus', '2', '--backend', 'gloo', '--launch-check')
    assert res.returncode == 0, res.stderr[-2000:]
    out = last_json(res.stdout)
    assert out['launch_check'] and out['n_gpus'] == 2 and out['ranks_seen'] == [0, 1]


def test_self_launch_relays_a_failing_rank():
    # a backend nobody knows passes the parent's argument parser and fails in every rank: the parent must come back
    # non-zero, not hang
    res = run_bench('--gpus', '2', '--backend', 'no-such-backend', '--launch-check', timeout=300)
    assert res.returncode != 0


@pytest.mark.gpu
def test_self_launch_runs_config_1_on_two_ranks_sharing_the_gpu():
    res = run_bench('--gpus', '2', '--backend', 'gloo', '--workload', 'cfg1', '--steps', '1', '--warmup', '1',
                    '--no-cpu-baseline', '--no-e2e', '--merge', 'mask')
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out['n_gpus'] == 2 and out['selfcheck']['ranks_seen'] == [0, 1]
    assert out['selfcheck']['replay_matches']
    assert out['selfcheck']['mask_allreduce_equals_gathered_hits']       # north_star's all-reduce of the per-band bit masks
    assert set(out['phases']['max_over_ranks']) >= {'count', 'scan', 'gather'}


@pytest.mark.gpu
@pytest.mark.parametrize('ranks,items', [(2, 'minimizer'), (3, 'minimizer'), (3, 'distinct')])
def test_exchange_layout_with_its_samples_interleaved_equals_the_banded_replay(ranks, items):
    """bench.py --multi exchange: a sample's records travel while the next sample's shard is cut, its pairs while the next sample's
    records are combined (ShardedTrio.cut_minimizer / combine_minimizer); the merged hits must equal rank 0's replay of the bands"""
    res = run_bench('--gpus', str(ranks), '--backend', 'gloo', '--workload', 'cfg1', '--steps', '2', '--warmup', '1',
                    '--no-cpu-baseline', '--no-e2e', '--multi', 'exchange', '--exchange-items', items)
    assert res.returncode == 0, res.stderr[-3000:]
    out = last_json(res.stdout)
    assert out['n_gpus'] == ranks and out['selfcheck']['ranks_seen'] == list(range(ranks))
    assert out['selfcheck']['replay_matches']
    assert 'exchanged by band' in out['config']['parallelism']
    assert out['selfcheck']['exchange']['layout_fallbacks'] == 0 and out['selfcheck']['exchange']['scan_fallbacks'] == 0
    assert out['selfcheck']['exchange']['unexpected_failures'] == 0 and out['selfcheck']['exchange']['own_failures'] == {}
    # north_star's collective on the default multi-GPU line: the per-owner bit masks all-reduced inside the step, equal to the gathered hits
    assert out['selfcheck']['mask_allreduce_equals_gathered_hits'] and 'bit masks' in out['config']['parallelism']
    assert out['selfcheck']['exchange']['scan'] == ('owner' if items == 'minimizer' else 'set')
