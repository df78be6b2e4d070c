import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(autouse=True)
def _forget_search_memos():
    """The searches memoise per-tensor work (percentile grids, sorted copies) until a module's search ends; tests that
    call scoring functions directly never reach that point, so the memo is dropped after every test."""
    yield
    try:
        from adalog_amd import search
        search.forget_grids()
    except Exception:
        pass
