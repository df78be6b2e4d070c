"""`-m gpu`: the wrapped-model flow of tests/wrapper_cases.py on the HIP kernels -- the product's wrapper + calibrator +
un-wrap + reparam_bias on the tiny ViT against the reference's own run of that flow (tests/golden/wrapper_rules.npz),
both capture modes, and the attention forwards on the device."""
import json
import os

import pytest

from tests import wrapper_cases as WC

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _hip_backend():
    from adalog_amd import backend
    backend.set_backend(None)
    backend.get()
    yield


def test_attention_forwards_on_hip(golden):
    WC.case_attention_forwards(golden, DEV)


@pytest.mark.parametrize("capture", ["module", "block"])
def test_wrapped_vit_flow_on_hip(golden, capture):
    r = WC.case_wrapped_vit_flow(golden, DEV, capture)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "wrapper_flow_parity.jsonl"), "a") as f:
        f.write(json.dumps(dict(capture=capture, **{k: v for k, v in r.items() if k != "exact"},
                                inexact=[k for k, v in r["exact"].items() if not v])) + "\n")
    # near-tie flips aside (a 16x8 FPCS grid shares end points between neighbours), the reference's model
    assert r["scales_off"] <= 0.02 * r["scales"], r
    assert 0.9 <= r["mse_mine"] / r["mse_ref"] <= 1.1, r
    assert sum(not v for v in r["exact"].values()) <= 2, r
