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


I, H = 16, 2


class Attn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        kw = dict(mode="raw", w_bit=4, a_bit=4, calib_batch_size=2, search_round=1, eq_n=128, fpcs=True, steps=2)
        self.qkv = Q.AsymmetricallyBatchingQuantLinear(I, 3 * I, True, n_V=3, **kw)
        self.proj = Q.AsymmetricallyBatchingQuantLinear(I, I, True, n_V=1, **kw)
        mk = dict(B_bit=4, mode="raw", calib_batch_size=2, search_round=1, eq_n=128, head_channel_wise=True, num_heads=H,
                  fpcs=True, steps=2)
        self.matmul1 = Q.AsymmetricallyBatchingQuantMatMul(A_bit=4, **mk)
        self.matmul2 = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=4, quantizer="adalog", **mk)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = self.matmul1(q, k.transpose(-2, -1)) * (C // H) ** -0.5
        attn = attn.softmax(dim=-1)
        x = self.matmul2(attn, v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.attn = Attn()
        kw = dict(mode="raw", w_bit=4, a_bit=4, calib_batch_size=2, search_round=1, eq_n=128, fpcs=True, steps=2)
        self.fc1 = Q.AsymmetricallyBatchingQuantLinear(I, 2 * I, True, n_V=1, **kw)
        self.fc2 = Q.PostGeluLogBasedBatchingQuantLinear(2 * I, I, True, n_V=1, quantizer="adalog", **kw)

    def forward(self, x):
        x = x + self.attn(x)
        return x + self.fc2(torch.nn.functional.gelu(self.fc1(x)))


@pytest.mark.parametrize("capture", ["module", "block"])
def test_calibrator_matches_reference_run(golden, capture):
    g = golden("calibrator_toy")
    model = Toy().eval()
    sd = {k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("in_")}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    xs = [torch.from_numpy(g["x0"]), torch.from_numpy(g["x1"])]
    order, shapes = [], {}
    for name, m in model.named_modules():
        if hasattr(m, "hyperparameter_searching"):
            orig = m.hyperparameter_searching

            def wrapped(orig=orig, name=name, m=m):
                order.append(name)
                ri = m.raw_input
                shapes[name] = (list(ri[0].shape) if isinstance(ri, list) else list(ri.shape), list(m.raw_out.shape))
                assert m.mode == "raw"                      # every capture is of the FP model
                return orig()

            m.hyperparameter_searching = wrapped
    cal = QuantCalibrator(model, [(x, None) for x in xs], capture=capture)
    cal.batching_quant_calib()
    assert order == [str(s) for s in g["order"]]            # qkv, proj, matmul1, matmul2, fc1, fc2
    for n in order:
        assert shapes[n][0] == list(g["shape_in_" + n.replace(".", "__")])
        assert shapes[n][1] == list(g["shape_out_" + n.replace(".", "__")])
    assert all(m.mode == "quant_forward" and m.calibrated for m in model.modules() if hasattr(m, "mode"))
    assert set(cal.timings) == set(order)
    with torch.no_grad():
        out = model(xs[0])
        fp = Toy().eval()
        fp.load_state_dict(sd, strict=False)
        ref_fp = fp(xs[0])
    ref_q = torch.from_numpy(g["qf_out"])
    e_mine, e_ref = ((out - ref_fp) ** 2).mean().item(), ((ref_q - ref_fp) ** 2).mean().item()
    assert 0.8 <= e_mine / e_ref <= 1.25, (e_mine, e_ref)
    # state_dict wire format equals the reference's (keys and shapes)
    ref_keys = {k[4:].replace("__", "."): v.shape for k, v in g.items() if k.startswith("out_")}
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(mine) == set(ref_keys)
    for k in mine:
        assert tuple(ref_keys[k]) == mine[k], k


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
