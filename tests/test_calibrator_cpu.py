"""QuantCalibrator orchestration on CPU (stand-in kernels): visiting order, captured shapes, mode switch, and the
end-to-end result against the reference's own calibrator run on the same toy module tree (golden calibrator_toy)."""
import numpy as np
import pytest
import torch

from adalog_amd import backend
from adalog_amd import quant_layers as Q
from adalog_amd.utils.calibrator import QuantCalibrator
from tests import cpu_backend


@pytest.fixture(autouse=True)
def _cpu_backend():
    backend.set_backend(cpu_backend)
    yield
    backend.set_backend(None)


from tests import calibrator_cases as CC


@pytest.mark.parametrize("capture", ["module", "block"])
def test_calibrator_matches_reference_run(golden, capture):
    CC.case_calibrator_matches_reference_run(golden, capture)


def test_calibrator_on_vit_block_with_reparam():
    """One DeiT-style block through wrap -> calibrate -> un-wrap: channel-wise layers fold into the LayerNorms and the
    FP function is preserved; head uses qhead_a_bit; modules come out in the reference's order."""
    import importlib.util
    import os
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cfg6", os.path.join(root, "configs", "6bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = 1, 2
    torch.manual_seed(5)
    model = VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=1, num_heads=2, num_classes=10).eval()
    for p in model.parameters():
        p.data.mul_(8.0)                                         # make the random-init activations non-degenerate
    x = torch.randn(4, 3, 32, 32)
    with torch.no_grad():
        y_fp = model(x)
    model = wrap_modules_in_net(model, cfg, reparam=True)
    names = [n for n, m in model.named_modules() if hasattr(m, "calibrated")]
    assert names == ["patch_embed.proj", "blocks.0.attn.qkv", "blocks.0.attn.proj", "blocks.0.attn.matmul1",
                     "blocks.0.attn.matmul2", "blocks.0.mlp.fc1", "blocks.0.mlp.fc2", "head"]
    assert isinstance(model.blocks[0].attn.qkv, Q.AsymmetricallyChannelWiseBatchingQuantLinear)
    assert model.blocks[0].attn.qkv.prev_layer is model.blocks[0].norm1 and model.blocks[0].attn.qkv.n_V == 3
    assert model.blocks[0].mlp.fc1.prev_layer is model.blocks[0].norm2
    QuantCalibrator(model, [(x[:2], None), (x[2:], None)]).batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    assert type(model.blocks[0].attn.qkv) is Q.AsymmetricallyBatchingQuantLinear
    assert model.blocks[0].attn.qkv.a_quantizer.scale.shape == (1,)
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "mode"):
                m.mode = "raw"
        torch.testing.assert_close(model(x), y_fp, rtol=1e-3, atol=1e-4)    # LayerNorm fold preserves the function
        for m in model.modules():
            if hasattr(m, "mode"):
                m.mode = "quant_forward"
        y_q = model(x)
    assert torch.isfinite(y_q).all()
    rel = ((y_q - y_fp).norm() / y_fp.norm()).item()
    assert rel < 0.3, rel                                        # W6A6 end to end on a random-init block


def test_block_capture_equals_module_capture():
    r = CC.case_capture_equivalence()
    assert r["scales_off"] == 0 and r["max_out_diff"] <= 1e-3 * r["out_max"], r


def test_converged_rounds_are_skipped_exactly():
    stats = CC.case_converged_rounds_are_skipped_exactly()
    assert stats[True]["checked"] > 0


@pytest.mark.parametrize("which", ["vit", "swin"])
def test_capture_cache_equals_recompute(which):
    passes = CC.case_capture_cache_equals_recompute("cpu", which)
    assert 0 < passes["1"] < passes["0"]       # block bodies run less often with the cache (2 per block instead of depth - i + 1)
