"""CPU tier of tests/golden_forward_cases.py: the reference's final parameters loaded into the product layers, forward
through the stand-in kernel specs (tests/cpu_backend.py) -- pins the host logic of quant_forward / load_state_dict and the
case code itself; the `-m gpu` tier (tests/test_gpu_golden_forward.py) runs the same cases on the HIP kernels."""
import pytest

from adalog_amd import backend
from tests import cpu_backend, golden_forward_cases as GF


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_forward(golden, name):
    GF.case_linear_forward(golden, name)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_layer_forwards(golden, bits):
    GF.case_channelwise_forward(golden, bits)
    GF.case_postgelu_forward(golden, bits)
    GF.case_matmul_forward(golden, bits)
    GF.case_postsoftmax_forward(golden, bits)
    GF.case_conv_forward(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6, 8])
def test_uniform_quantizer_golden(golden, bits):
    GF.case_uniform_quantizer_golden(golden, bits)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_adalog_quantizer_golden(golden, bits):
    for q in (10, 23, 37, 53, 90, 137):
        GF.case_adalog_quantizer_golden(golden, bits, q)


@pytest.mark.parametrize("bits", [3, 4])
def test_adaround_quantizer_golden(golden, bits):
    GF.case_adaround_quantizer_golden(golden, bits)
