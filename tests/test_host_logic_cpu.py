"""Host-side logic of the product layers (FPCS driver, candidate flow, commits, reparam) on CPU.

The HIP kernels are replaced by the executable specs in tests/cpu_backend.py (injected through
adalog_amd.backend.set_backend, a test-only hook), so these tests pin everything *around* the kernels against the
golden fixtures captured from the reference.  The kernels themselves are pinned by the `-m gpu` tests, which run the
same cases (tests/layer_cases.py) on the HIP backend.
"""
import pytest

from adalog_amd import backend
from tests import cpu_backend, layer_cases as LC


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_search(golden, name):
    LC.case_linear_search(golden, name)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_channelwise_reparam(golden, bits):
    LC.case_channelwise_reparam(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu_search(golden, bits):
    LC.case_postgelu_search(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul_search(golden, bits):
    LC.case_matmul_search(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax_search(golden, bits):
    LC.case_postsoftmax_search(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv_search(golden, bits):
    LC.case_conv_search(golden, bits)


def test_modes_and_errors():
    LC.case_modes_and_errors()


def test_product_refuses_to_run_without_hip():
    """No silent fallback: with the test hook cleared and no GPU, resolving the backend raises."""
    import torch
    from adalog_amd._lib import AdalogHipError
    backend.set_backend(None)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(AdalogHipError):
        backend.get()
