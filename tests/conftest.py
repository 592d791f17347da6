import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from se2lam_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a box without a GPU: they are skipped with a reason when
    # not selected with -m gpu, and FAIL (no fallback) if selected where no device is visible.
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run: pytest -m gpu on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def synth():
    from se2lam_amd import synth as s
    return s
