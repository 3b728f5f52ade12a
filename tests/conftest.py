import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU suite only checks that the generated kernels COMPILE for gfx950 (hiprtc, no device); it builds a few hundred of them, and the
    # shipped -O3 costs 40 % more compile time than -O1 for nothing it looks at.  The -m gpu suite runs what ships.
    if "not gpu" in (config.getoption("-m") or ""):
        os.environ.setdefault("PTL_JIT_OPT", "-O1")


@pytest.fixture(scope="session")
def pa():
    """The product package; builds libportal_amd.so on first use if it is missing."""
    import portal_amd

    if not os.path.exists(portal_amd.LIB_PATH):
        import subprocess

        subprocess.run(["make", "-j8"], cwd=ROOT, check=True)
    portal_amd.lib()
    return portal_amd
