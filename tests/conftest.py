import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """`pytest -m gpu` (the driver's round-end command, single process, 1 200 s limit) spreads itself over four worker processes: the GPU
    suite is ~650 independent frames + kernel builds and most of its wall time is host work (hiprtc, the numpy oracle), 131 s with four
    workers against 498 s with one (profiles/r04).  `-n ...` on the command line or PTL_GPU_SUITE_WORKERS=0 keeps the caller's choice."""
    marker = (config.getoption("-m", default="") or "").strip()
    if marker not in ("gpu", "not gpu") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(config.option, "numprocesses", None) is not None or getattr(config.option, "tx", None) or os.environ.get("PYTEST_XDIST_WORKER"):
        return None
    # Round 6: the CPU suite (`-m "not gpu"`, 974 tests: hiprtc builds for gfx950, host builds of the generated sources through g++, the numpy oracle) does the
    # same with six workers: 7 min 52 s in one process, ~1 min 40 s spread (it has been run under `-n 8` since round 5; PTL_CPU_SUITE_WORKERS=0: one process)
    workers = int(os.environ.get("PTL_GPU_SUITE_WORKERS", "4")) if marker == "gpu" else int(os.environ.get("PTL_CPU_SUITE_WORKERS", "6"))
    if workers > 1:
        config.option.numprocesses = workers
    return None


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
