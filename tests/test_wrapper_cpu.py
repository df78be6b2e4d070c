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


def test_block_residual_routing_equals_the_plain_composition():
    """Block.forward hands the residual stream to its two branches (utils/models.py: the fused quant_forward route adds it inside the
    branch's last launch); on every other route the branch returns `residual + out` -- the timm composition
    x + attn(norm1(x)); x + mlp(norm2(x)) (reference models via timm.models.vision_transformer.Block) bit for bit."""
    import torch
    from adalog_amd.utils.models import Attention, Block, Mlp
    torch.manual_seed(3)
    blk = Block(64, num_heads=1, mlp_ratio=2.0).eval()
    x = torch.randn(2, 5, 64)
    with torch.no_grad():
        h = x + blk.attn._forward(blk.norm1(x))
        want = h + blk.mlp.fc2(blk.mlp.act(blk.mlp.fc1(blk.norm2(h))))
        assert torch.equal(blk(x), want)
        assert torch.equal(blk.attn(blk.norm1(x)), blk.attn._forward(blk.norm1(x)))          # no residual: unchanged API
        assert torch.equal(blk.mlp(x, residual=x), x + blk.mlp(x))
    # gradients flow through both routes the same way
    xg = x.clone().requires_grad_(True)
    blk(xg).sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()
