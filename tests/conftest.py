import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


@pytest.fixture(autouse=True)
def _reset_debug_options(request):
    """rtd_debug_option switches are process-wide: a GPU test that changes one must not leak it into the next (handles snapshot
    the plan-build switches at rtd_create, the conv dispatch switches are read at launch / capture)."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from telescope_cam_detection_amd import _capi
        _capi.debug_option("reset", 0)
