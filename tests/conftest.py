import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pa():
    """The product package; builds libportal_amd.so on first use if it is missing."""
    import portal_amd

    if not os.path.exists(portal_amd.LIB_PATH):
        import subprocess

        subprocess.run(["make", "-j8"], cwd=ROOT, check=True)
    portal_amd.lib()
    return portal_amd
