import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'transtacos-retunegan_amd')
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # every weight image the lean pack drops is filled with NaN under test: a launch that still reads one without reporting
    # it fails loudly instead of computing with the weights of an earlier step (rtg/bank.py)
    from rtg import bank
    bank.LEAN_POISON = True


def pytest_collection_modifyitems(config, items):
    """multi-process / process-group tests run LAST whatever their file is called: under `pytest -x` an infrastructure
    failure (rendezvous, RCCL start-up, a worker crash) must not be able to hide a kernel-parity test (round 3 lost 163
    tests that way)"""
    late = [it for it in items if '_dp_' in os.path.basename(str(it.fspath))]
    if late:
        items[:] = [it for it in items if it not in late] + late


@pytest.fixture(scope='session')
def gold():
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', 'retunegan_b2_t8192.npz'), allow_pickle=False))


@pytest.fixture(scope='session')
def gold4():
    """fixtures of the SURVEY.md 8 f3/f4 rows (oracle/gen_golden_f4.py)"""
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', 'retunegan_f4_b2_t8192.npz'), allow_pickle=False))


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure; see oracle/rtg_oracle.py header)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('rtg_oracle', os.path.join(REPO, 'oracle', 'rtg_oracle.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules['rtg_oracle'] = mod
    spec.loader.exec_module(mod)
    return mod
