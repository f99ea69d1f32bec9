import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_gpu_objects(request):
    """GPU tests build engines (GBs of tables) and hipGraphs whose Python owners sit in reference cycles (closures of the
    capture): collect them when the test ends instead of whenever the cycle collector gets to it -- graph executables and their
    kernel-argument pools otherwise pile up across the ~170 tests of one process."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
