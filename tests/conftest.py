import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libtm_oracle.so), built on demand.  Test infrastructure only."""
    so = os.path.join(ROOT, "oracle", "libtm_oracle.so")
    src = os.path.join(ROOT, "oracle", "tm_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libtm_oracle.so"])
    from tests import oracle_binding
    return oracle_binding.Oracle(so)
