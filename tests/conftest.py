import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A clean tree has no built library yet (it is git-ignored): build it in-tree once, as __graft_entry__.build()
    does (hipcc cross-compiles gfx950 without a GPU; ~3 min).  Nothing here falls back to a CPU path."""
    lib = os.path.join(ROOT, "fbus-ekf_amd", "lib", "libfbus_ekf.so")
    if not os.path.exists(lib) and not os.environ.get("FBUS_EKF_LIB"):
        import subprocess
        subprocess.run([sys.executable, os.path.join(ROOT, "fbus-ekf_amd", "build.py")], check=False)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
