import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """The in-tree native libraries (product .so and oracle .so); built on demand."""
    import __graft_entry__ as ge
    so = os.path.join(ROOT, "lpopc_amd", "csrc", "librpm_hip.so")
    orc = os.path.join(ROOT, "oracle", "liborpm.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        ge.build()
    return so
