import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu tests are skipped (not failed) when no device is visible, e.g. `pytest tests` on the CPU box
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="module")
def _gpu_module_boundary():
    """Every GPU test module starts from an idle device and an empty allocator cache, and leaves them that way: what a module's graphs, streams and
    multi-GB scratch buffers did to the caching allocator (which blocks are cached, which were handed back to the driver under memory pressure)
    is not the next module's starting point."""
    yield
    try:
        import torch
        if torch.cuda.is_available():
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:                                                # noqa: BLE001 - never fail a test run in its teardown
        pass
