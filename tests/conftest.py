import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The tests pin kernel paths and shrink geometries through the library's TUNING switches (kevlar_amd/csrc/kv_knobs.h), which it
# honours only while KV_TUNING=1 is set; tests/test_host_logic.py holds the registry itself (and that without KV_TUNING a set switch
# is ignored).
os.environ.setdefault('KV_TUNING', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def data_file(rel):
    """A reference test data file copied under tests/golden/data by make_golden.py."""
    return os.path.join(GOLDEN, 'data', rel)


def expected_file(rel):
    return os.path.join(GOLDEN, 'expected', rel)


@pytest.fixture(scope='session')
def ok():
    """The CPU oracle (checker only)."""
    from oracle import okhmer
    return okhmer


@pytest.fixture(scope='session')
def hk():
    """The product's HIP sketch engine; GPU tests fail loudly if it is not built/loaded."""
    import __graft_entry__
    if not os.path.exists(__graft_entry__.LIB):
        __graft_entry__.build()
    # Tests that also use torch tensors (band masks) need torch's bundled HIP runtime to be the one
    # the process loads: initialise torch.cuda BEFORE libkvsketch_hip touches the device.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from kevlar_amd import khmer
    return khmer


@pytest.fixture
def kevlar_log():
    """Capture kevlar_amd.plog output."""
    import io
    import kevlar_amd
    buf = io.StringIO()
    old = kevlar_amd.logstream
    kevlar_amd.logstream = buf
    yield buf
    kevlar_amd.logstream = old

SIMLIKE_ALT = 'TGTCTCCCTCCCCTCCACCCCCAGAAATGGGTTTTTGATAGTCTTCCAAAGTTAGGGTAGT'
SIMLIKE_REF = 'TGTCTCCCTCCCCTCCACCCCCAGAAATGGCTTTTTGATAGTCTTCCAAAGTTAGGGTAGT'
SIMLIKE_INDEL = 'TGTCTCCCTCCCCTCCACCCCCAGAAATGGGAAATTTTTGATAGTCTTCCAAAGTTAGGGTAGT'
SIMLIKE_GOLD_ALT = [
    [7, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 9, 8, 8, 9, 9, 9, 7, 7, 8, 8, 8, 7, 7, 7, 7, 7, 7],
    [1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1],
    [0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
]
SIMLIKE_GOLD_REFR = [2, 2, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1]


def simlike_minitrio(khmer_module):
    """kevlar/tests/test_simlike.py:21-31"""
    kid = khmer_module.Counttable(31, 1e6, 4)
    mom = khmer_module.Counttable(31, 1e6, 4)
    dad = khmer_module.Counttable(31, 1e6, 4)
    ref = khmer_module.SmallCounttable(31, 125000, 4)
    kid.consume_seqfile(data_file('minitrio/trio-proband.fq.gz'))
    mom.consume_seqfile(data_file('minitrio/trio-mother.fq.gz'))
    dad.consume_seqfile(data_file('minitrio/trio-father.fq.gz'))
    ref.consume_seqfile(data_file('minitrio/refr.fa'))
    return kid, mom, dad, ref


def check_spanning_kmer_abundances(khmer_module):
    """kevlar/tests/test_simlike.py:82-106"""
    from kevlar_amd.simlike import spanning_kmer_abundances
    kid, mom, dad, ref = simlike_minitrio(khmer_module)
    altabund, refrabund, ndropped = spanning_kmer_abundances(SIMLIKE_ALT, SIMLIKE_REF, kid, (mom, dad), ref)
    assert ndropped == 3
    assert altabund == SIMLIKE_GOLD_ALT
    assert refrabund == SIMLIKE_GOLD_REFR
    altabund, refrabund, ndropped = spanning_kmer_abundances(SIMLIKE_ALT, SIMLIKE_INDEL, kid, (mom, dad), ref)
    assert ndropped == 3
    assert refrabund == [None] * len(altabund[0])
