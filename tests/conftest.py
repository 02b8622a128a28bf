import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; without a device they would only error, so skip them.
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The built .so files normally travel with the snapshot; if one is missing on this box, build it (hipcc and gcc
    are in the image) rather than let every test fail on an ImportError."""
    import subprocess
    from approximatenn_amd import _lib
    if not all(os.path.exists(_lib.lib_path(p)) for p in ("f32", "f64")):
        _lib.build()
    harness = os.path.join(ROOT, "tests", "harness")
    if not os.path.exists(os.path.join(harness, "compare_results_f32")):
        from oracle import oracle_py
        oracle_py.build()
        subprocess.call(["make", "-s", "-C", harness])
    yield
