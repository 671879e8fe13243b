import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libofarn.so (built artefacts are git-ignored): cross-compile it once
    (hipcc needs no GPU).  An existing library is left alone -- on the GPU box the prebuilt one is used."""
    from hackathonopticalflow_amd import build as hb
    if not os.path.exists(hb.LIB):
        hb.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on first use with gcc."""
    from oracle import oracle as O
    O.build()
    return O
