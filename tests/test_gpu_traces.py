"""`-m gpu`: EVERY FPCS scoring call of all six layer classes at 3 / 4 / 6 bit, HIP kernels vs the reference's golden
traces (tests/trace_replay.py): score vectors <= 1e-4 relative, top-k sets equal up to reference ties (<1e-5)."""
import json
import os

import pytest

from tests import trace_replay as TR

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _hip_backend():
    from adalog_amd import backend
    backend.set_backend(None)
    backend.get()
    yield


def _log(name, r):
    """Keep the observed errors: gpurun_out/trace_parity.jsonl (copied to profiles/ by tools/final_run.sh)."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "trace_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=name, **r)) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear_traces(golden, name):
    r = TR.replay_linear(golden, name, DEV)
    assert r["calls"] == 48
    _log(name, r)


@pytest.mark.parametrize("bits", [3, 4])
def test_linear_traces_fp8_storage(golden, bits, monkeypatch):
    monkeypatch.setenv("ADALOG_INT_FP8", "1")
    r = TR.replay_linear(golden, f"linear_w{bits}a{bits}", DEV)
    _log(f"linear_w{bits}a{bits}_fp8", r)


@pytest.mark.parametrize("bits", [3, 4])
def test_linear_traces_int8_storage(golden, bits, monkeypatch):
    monkeypatch.setenv("ADALOG_INT_FP8", "0")
    r = TR.replay_linear(golden, f"linear_w{bits}a{bits}", DEV)
    _log(f"linear_w{bits}a{bits}_i8", r)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6"])     # (the ragged fixture has K = 24: token form)
def test_linear_traces_gram_form(golden, name, monkeypatch):
    """The reference's golden weight-search traces through the Gram-form kernels (csrc/gram.hip; ADALOG_GRAM_W=2 takes every
    supported shape, also these toy ones): same bars as the token form."""
    from adalog_amd import _lib
    monkeypatch.setenv("ADALOG_GRAM_W", "2")
    seen = []
    orig = _lib.load().adalog_last_kernel
    from adalog_amd import ops
    real = ops.GramState.score_w

    def spy(self, *a, **k):
        r = real(self, *a, **k)
        seen.append(orig().decode())
        return r
    monkeypatch.setattr(ops.GramState, "score_w", spy)
    r = TR.replay_linear(golden, name, DEV)
    assert r["calls"] == 48
    assert len(seen) >= 18 and set(seen) == {"k_gram_score<i8>"}, seen[:3]
    _log(name + "_gram", r)


@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6"])     # (the ragged fixture has K = 24: token form)
def test_linear_traces_gram_act_form(golden, name, monkeypatch):
    """The reference's golden activation-search traces through the Gram-form activation kernels (csrc/gram_act.hip; ADALOG_GRAM_A=2
    takes every supported shape, also these toy ones): same bars as the token form."""
    from adalog_amd import _lib, ops
    monkeypatch.setenv("ADALOG_GRAM_A", "2")
    seen = []
    real = ops.GramActState.score

    def spy(self, *a, **k):
        r = real(self, *a, **k)
        seen.append(_lib.load().adalog_last_kernel().decode())
        return r
    monkeypatch.setattr(ops.GramActState, "score", spy)
    r = TR.replay_linear(golden, name, DEV)
    assert r["calls"] == 48
    assert len(seen) >= 18 and set(seen) == {"k_gram_act<i8>"}, seen[:3]
    _log(name + "_gram_act", r)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_channelwise_traces(golden, bits):
    r = TR.replay_channelwise(golden, bits, DEV)
    assert r["calls"] == 54
    _log(f"linear_cw_w{bits}a{bits}", r)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu_traces(golden, bits):
    r = TR.replay_postgelu(golden, bits, DEV)
    assert r["calls"] == 45
    _log(f"postgelu_w{bits}a{bits}", r)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul_traces(golden, bits):
    r = TR.replay_matmul(golden, bits, DEV)
    assert r["calls"] == 36
    _log(f"matmul_a{bits}b{bits}", r)


@pytest.mark.parametrize("bits", [3, 4])
def test_matmul_traces_int8_storage(golden, bits, monkeypatch):
    monkeypatch.setenv("ADALOG_INT_FP8", "0")
    r = TR.replay_matmul(golden, bits, DEV)
    _log(f"matmul_a{bits}b{bits}_i8", r)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax_traces(golden, bits):
    r = TR.replay_postsoftmax(golden, bits, DEV)
    assert r["calls"] == 21
    _log(f"postsoftmax_a{bits}b{bits}", r)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv_traces(golden, bits):
    r = TR.replay_conv(golden, bits, DEV)
    assert r["calls"] == 6
    _log(f"conv_w{bits}", r)
