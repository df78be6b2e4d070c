"""CPU tier of the per-call trace replay (tests/trace_replay.py): the same harness that runs on the HIP kernels under
`-m gpu`, here over the executable kernel specs of tests/cpu_backend.py -- pins the host-side candidate plumbing of every
scoring call against the reference's golden traces and keeps the harness itself tested without a GPU."""
import pytest

from adalog_amd import backend
from tests import cpu_backend, trace_replay as TR


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_traces(golden, name):
    r = TR.replay_linear(golden, name)
    assert r["calls"] == 48


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_channelwise_traces(golden, bits):
    assert TR.replay_channelwise(golden, bits)["calls"] == 54


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu_traces(golden, bits):
    assert TR.replay_postgelu(golden, bits)["calls"] == 45


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul_traces(golden, bits):
    assert TR.replay_matmul(golden, bits)["calls"] == 36


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax_traces(golden, bits):
    assert TR.replay_postsoftmax(golden, bits)["calls"] == 21


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv_traces(golden, bits):
    assert TR.replay_conv(golden, bits)["calls"] == 6
