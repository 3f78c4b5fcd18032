import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP library; built here if missing (hipcc cross-compiles without a GPU)."""
    import mizuroute_amd as m
    if not os.path.exists(m.lib_path()):
        m.build_library()
    return m.load_library()
