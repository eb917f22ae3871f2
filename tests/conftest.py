import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


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
