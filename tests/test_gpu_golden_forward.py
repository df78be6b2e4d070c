"""`-m gpu`: strict end-state parity on the HIP kernels (tests/golden_forward_cases.py) -- the reference's golden final
parameters loaded into the product layers, quant_forward vs the reference's golden output (<= 1e-3 relative, all six
layer classes x {3, 4, 6} bit, both bias_reparamed states, channel-wise); and the quantiser goldens
(quantizers_{uniform,adalog,adaround}.npz) fed straight to the HIP fake-quant kernels (values / bins exact)."""
import json
import os

import pytest

from tests import golden_forward_cases as GF

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "golden_forward_parity.jsonl")


@pytest.fixture(autouse=True)
def _hip_backend():
    from adalog_amd import backend
    backend.set_backend(None)
    backend.get()
    yield


def _log(case, err):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(json.dumps({"case": case, "max_rel_err": err}) + "\n")


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_forward(golden, name):
    _log(name, GF.case_linear_forward(golden, name, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_channelwise_forward(golden, bits):
    _log(f"linear_cw_w{bits}a{bits}", GF.case_channelwise_forward(golden, bits, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu_forward_both_bias_states(golden, bits):
    _log(f"postgelu_w{bits}a{bits}", GF.case_postgelu_forward(golden, bits, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul_forward(golden, bits):
    _log(f"matmul_a{bits}b{bits}", GF.case_matmul_forward(golden, bits, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax_forward(golden, bits):
    _log(f"postsoftmax_a{bits}b{bits}", GF.case_postsoftmax_forward(golden, bits, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv_forward(golden, bits):
    _log(f"conv_w{bits}", GF.case_conv_forward(golden, bits, DEV))


@pytest.mark.parametrize("bits", [3, 4, 6, 8])
def test_uniform_quantizer_golden(golden, bits):
    GF.case_uniform_quantizer_golden(golden, bits, DEV)


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("q", [10, 23, 37, 53, 90, 137])
def test_adalog_quantizer_golden(golden, bits, q):
    _log(f"adalog_b{bits}_q{q}", GF.case_adalog_quantizer_golden(golden, bits, q, DEV))


@pytest.mark.parametrize("bits", [3, 4])
def test_adaround_quantizer_golden(golden, bits):
    GF.case_adaround_quantizer_golden(golden, bits, DEV)
