import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Parity first: under the driver's `pytest -x` nothing that measures (the bench.py contract runs, the A/B variant libraries)
# may stand in front of a parity test.  Files not named here keep their alphabetical place in front of these.
_LAST = ("test_gpu_determinism", "test_gpu_variant_libs", "test_gpu_bench_contract")


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        for i, key in enumerate(_LAST):
            if name.startswith(key):
                return i + 1
        return 0
    items.sort(key=rank)                 # stable: order inside each group is unchanged


@pytest.fixture(scope="session")
def hiplib():
    """The built C-ABI library; GPU tests fail (not skip) if it is missing."""
    import msnets_amd
    return msnets_amd._lib.load()


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return torch.device("cuda:0")
