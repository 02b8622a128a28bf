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
    """The built .so files normally travel with the snapshot; anything missing or out of date on this box is (re)built
    (hipcc and gcc are in the image)."""
    import subprocess
    from approximatenn_amd import _lib
    # always `make` (a no-op when everything is up to date): a header edited after the last build must never be
    # tested through a stale binary
    _lib.build()
    harness = os.path.join(ROOT, "tests", "harness")
    from oracle import oracle_py
    oracle_py.build()
    subprocess.call(["make", "-s", "-C", harness])
    yield


@pytest.fixture(autouse=True)
def _fresh_env_switches():
    """The library caches its ANN_HIP_* switches; tests that set them call _lib.reload_env() themselves, and this
    makes sure no test inherits another one's switches."""
    yield
    from approximatenn_amd import _lib
    _lib.reload_env()
