import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree libraries exist (built by __graft_entry__.build())."""
    import petsc_dev_amd as pda
    if not (os.path.exists(pda.kernels_lib_path()) and os.path.exists(pda.host_lib_path())
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))
            and os.path.exists(os.path.join(ROOT, "examples", "poisson2d"))
            and os.path.exists(os.path.join(ROOT, "examples", "mm2petsc"))
            and os.path.exists(os.path.join(ROOT, "examples", "loadsolve"))):
        pda.build_all()
    return pda
