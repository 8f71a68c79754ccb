import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def _gpu_present():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """GPU tests are selected with ``-m gpu``; when no GPU device node exists
    they are skipped rather than failed (the CPU container has no /dev/kfd)."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
