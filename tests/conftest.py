import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as graft  # noqa: E402

graft.load_package()

GOLDEN = os.path.join(ROOT, "tests", "golden")
POT_REBOMOS = os.path.join(GOLDEN, "potentials", "MoS.REBO.set5b")
POT_AEAM = os.path.join(GOLDEN, "potentials", "AlSi.aeam")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_bindings
    return oracle_bindings.load()
