"""CPU tier: the product's wrapper rules, attention forwards, block discovery and the wrap -> calibrate -> un-wrap flow
against what the REFERENCE's own utils/wrap_net.py / block_recon.py produce (tests/golden/wrapper_rules.npz)."""
import pytest

from adalog_amd import backend
from tests import cpu_backend, wrapper_cases as WC


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


@pytest.mark.parametrize("tag", ["deit_tiny", "swin_tiny"])
@pytest.mark.parametrize("bits,reparam", [(3, True), (4, True), (6, True), (4, False)])
def test_wrapper_rules(golden, tag, bits, reparam):
    WC.case_wrapper_rules(golden, tag, bits, reparam)


def test_attention_forwards(golden):
    WC.case_attention_forwards(golden)


def test_wrapped_vit_flow(golden):
    r = WC.case_wrapped_vit_flow(golden)
    assert r["scales_off"] == 0 and all(r["exact"].values()), r             # the reference's model, parameter for parameter
    assert r["max_out_diff"] <= 1e-3 * r["out_max"], r
