import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The HIP back-end.  No fallback: a missing library or device is a hard failure."""
    import rpg_monocular_pose_estimator_amd as mpe
    h = mpe.Handle()
    yield h
    h.close()


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.build()
    return oracle
