"""roofline.traffic of bench.py: the stage's HBM bytes are summed over the profiler's KERNEL names grouped by the same
prefixes as the HIP-event times -- never matched against the library's profile-scope names, which are a different name
space (round 3 lost k_skm_emit_wave and k_skm_split_sorted that way: 35.1 GB reported for 52.96 GB measured)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

R3 = os.path.join(ROOT, 'profiles', 'r3_final', 'pmc_hbm_bytes.json')


def test_count_stage_traffic_of_round_3_counts_all_five_kernels():
    pmc = json.load(open(R3))
    # the profile scopes of the run that produced the file (ms per step, profiles/r3_final/bench_one_stream.json)
    scopes = {'k_skm_emit': 4.05, 'k_skm_split': 2.75, 'k_skm_count': 10.7, 'k_bin_split_w': 3.25, 'k_bin_apply_w': 2.4,
              'k_skm_loose_count': 0.4, 'k_bin_spill': 0.05}
    total, by_kernel = bench.stage_traffic(pmc['kernels'], 'count', scopes)
    for name in ('k_skm_emit_wave', 'k_skm_split_sorted', 'k_skm_count', 'k_bin_split_w', 'k_bin_apply_w'):
        assert by_kernel[name] > 8e9, name
    assert abs(total - 52.96e9) < 0.3e9, total
    assert sum(by_kernel.values()) == total
    # the scan's kernels are the other stage
    novel, by_novel = bench.stage_traffic(pmc['kernels'], 'novel')
    assert 'k_skm_novel_list' in by_novel and 'k_skm_count' not in by_novel
    assert not set(by_novel) & set(by_kernel)


def test_a_timed_scope_without_a_counted_kernel_raises():
    pmc = json.load(open(R3))
    kernels = {k: v for k, v in pmc['kernels'].items() if k != 'k_skm_split_sorted'}
    scopes = {'k_skm_emit': 4.05, 'k_skm_split': 2.75, 'k_skm_count': 10.7}
    with pytest.raises(RuntimeError, match='k_skm_split'):
        bench.stage_traffic(kernels, 'count', scopes)
    # a scope below one per cent of the stage may be missing (a kernel that did not run in the one-step counter pass)
    bench.stage_traffic(kernels, 'count', {'k_skm_emit': 4.05, 'k_skm_split': 0.01, 'k_skm_count': 10.7})


def test_every_profile_scope_of_the_library_has_a_stage_or_is_known_to_have_none():
    import re
    csrc = os.path.join(ROOT, 'kevlar_amd', 'csrc')
    scopes = set()
    for f in os.listdir(csrc):
        if f.endswith('.hip'):
            for m in re.finditer(r'KvProfScope prof\(([^;]*)\);', open(os.path.join(csrc, f)).read()):
                scopes.update(re.findall(r'"([a-z0-9_]+)"', m.group(1)))
    staged = {s for s in scopes if bench.stage_of(s)}
    # what count and novel launch
    for s in ('k_skm_emit', 'k_skm_split', 'k_skm_count', 'k_skm_loose_count', 'k_bin_split_w', 'k_bin_apply_w', 'k_bin_spill', 'k_consume',
              'k_bin_hash_direct', 'k_skm_novel_list', 'k_skm_novel', 'k_skm_loose_novel', 'k_tile_hits', 'k_tile_scan', 'k_novel_mark', 'k_novel_emit',
              'k_case_bits', 'k_skm_set_hits', 'k_skm_route', 'k_mex_pack'):
        assert s in staged, s
    # ingest, point queries, partition, exchange bookkeeping: neither stage
    for s in ('k_inflate', 'k_pack_reads', 'k_get_hashes', 'k_readgraph', 'memset_tables'):
        assert s in scopes and s not in staged, s
