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
        # a launch that never comes back fails its test with a stack dump (pytest-timeout, thread method: the main thread
        # sits in a ctypes call) instead of stalling the whole run until the box gives up on it
        if config.pluginmanager.hasplugin("timeout"):
            for item in items:
                if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                    item.add_marker(pytest.mark.timeout(300, method="thread"))
        return
    skip = pytest.mark.skip(reason="no GPU (/dev/kfd absent)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
