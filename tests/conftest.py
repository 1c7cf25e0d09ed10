import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Libraries are prebuilt in-tree (they travel to the GPU box); build only if missing."""
    pkg = graft.load_package()
    if not os.path.exists(pkg.lib_path()):
        graft.build()
    return pkg


@pytest.fixture(scope="session")
def host(built):
    return built.host


@pytest.fixture(scope="session")
def oracle(built):
    return graft.load_oracle()


@pytest.fixture(scope="session")
def ctx(host):
    c = host.BswContext(device=0)
    yield c
    c.close()
