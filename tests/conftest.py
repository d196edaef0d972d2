import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import lvdgs  # noqa: E402,F401  (registers the package alias)


def _ensure_built():
    """A fresh checkout has no lib/liblvdgs.so (built artefacts are not in history): build it once for the test
    session, the same way __graft_entry__.build() does.  The product itself never builds or falls back."""
    lib = os.path.join(ROOT, "lvd_gs-slam_amd", "lib", "liblvdgs.so")
    if os.path.exists(lib):
        return
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "lvd_gs-slam_amd", "csrc"), "-s", "-j4"])


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are only meaningful where a device exists; keep `-m "not gpu"` the CPU default.
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _name_the_test_for_the_parity_report(request):
    import parity_stats
    parity_stats.set_test(request.node.nodeid.replace("tests/", ""))
    yield


def pytest_sessionfinish(session, exitstatus):
    import parity_stats
    parity_stats.dump(ROOT)


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
