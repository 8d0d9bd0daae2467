import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ppca_oracle

    ppca_oracle.build()
    # The tests call the oracle thousands of times on a few hundred samples: on a 128-core host the OpenMP teams of such calls cost
    # more than the loops (a 300-component mixture over 400 samples: 20 minutes on all cores, half a second on two).
    if ppca_oracle.num_threads() > 16:
        ppca_oracle.set_threads(16)
    return ppca_oracle


@pytest.fixture(scope="session")
def hiplib():
    from ppca_rs_amd import build as b

    b.build()
    from ppca_rs_amd import _lib

    return _lib.lib()
