"""Calibrator cases shared by the CPU tier (stand-in kernels) and the `-m gpu` tier (HIP kernels): the reference's own
calibrator run on a toy module tree built from the reference's layers (golden calibrator_toy, tools/make_golden.py)."""
import numpy as np
import torch

from adalog_amd import quant_layers as Q
from adalog_amd.utils.calibrator import QuantCalibrator

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


def case_calibrator_matches_reference_run(golden, capture, device="cpu"):
    dev = torch.device(device)
    g = golden("calibrator_toy")
    model = Toy().eval().to(dev)
    sd = {k[3:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("in_")}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    xs = [torch.from_numpy(g["x0"]).to(dev), torch.from_numpy(g["x1"]).to(dev)]
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
        fp = Toy().eval().to(dev)
        fp.load_state_dict(sd, strict=False)
        ref_fp = fp(xs[0])
    ref_q = torch.from_numpy(g["qf_out"]).to(dev)
    e_mine, e_ref = ((out - ref_fp) ** 2).mean().item(), ((ref_q - ref_fp) ** 2).mean().item()
    assert 0.8 <= e_mine / e_ref <= 1.25, (e_mine, e_ref)
    # the calibrated model IS the reference's: every parameter / buffer of the golden state_dict (scales <= 1e-3 relative --
    # the fp32 fake-quant bar --, zero points, log bases and LUTs exactly) and the quantised forward pass
    ref_sd = {k[4:].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("out_")}
    for k, v in model.state_dict().items():
        r = ref_sd[k].to(v.device)
        if k.endswith(".scale") or k.endswith("weight") or k.endswith("bias"):
            torch.testing.assert_close(v, r.reshape(v.shape), rtol=1e-3, atol=1e-6, msg=lambda m: f"{k}: {m}")
        else:
            assert torch.equal(v.reshape(r.shape).to(r.dtype), r), k
    assert (out - ref_q).abs().max().item() <= 1e-3 * ref_q.abs().max().item()
    # state_dict wire format equals the reference's (keys and shapes)
    ref_keys = {k[4:].replace("__", "."): v.shape for k, v in g.items() if k.startswith("out_")}
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert set(mine) == set(ref_keys)
    for k in mine:
        assert tuple(ref_keys[k]) == mine[k], k




def case_capture_equivalence(device="cpu", bits=4):
    """capture='block' (one pass per transformer block, what bench.py times) against capture='module' (the reference's
    pass structure, calibrator.py:34-56) on a small ViT with the LayerNorm re-parameterisation in play: the two
    calibrated models must agree parameter by parameter (<= 1e-3 relative on scales, identical zero points / log bases
    wherever the search is not sitting on a near-tie) and in their quantised outputs."""
    import copy
    import importlib.util
    import os
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    dev = torch.device(device)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(f"cfg{bits}ce", os.path.join(root, "configs", f"{bits}bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = 1, 3
    torch.manual_seed(11)
    base = VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=2, num_heads=2, num_classes=10).eval()
    for p in base.parameters():
        p.data.mul_(8.0)
    x = torch.randn(8, 3, 32, 32).to(dev)
    loader = [(x[:4], None), (x[4:], None)]
    outs, sds = {}, {}
    for cap in ("module", "block"):
        model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
        QuantCalibrator(model, loader, capture=cap).batching_quant_calib()
        model = wrap_reparamed_modules_in_net(model)
        with torch.no_grad():
            outs[cap] = model(x)
        sds[cap] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    assert set(sds["module"]) == set(sds["block"])
    n_scale = n_bad = 0
    for k, a in sds["module"].items():
        b = sds["block"][k]
        if k.endswith(".scale"):
            n_scale += a.numel()
            n_bad += int(((a - b).abs() > 1e-3 * a.abs()).sum())
        elif k.endswith("weight") or k.endswith("bias"):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5, msg=lambda m: f"{k}: {m}")
    with torch.no_grad():
        y_fp = base.to(dev)(x)
    e_m = ((outs["module"] - y_fp) ** 2).mean().item()
    e_b = ((outs["block"] - y_fp) ** 2).mean().item()
    return {"scales": n_scale, "scales_off": n_bad, "mse_module": e_m, "mse_block": e_b,
            "max_out_diff": (outs["module"] - outs["block"]).abs().max().item(), "out_max": outs["module"].abs().max().item()}


def case_converged_rounds_are_skipped_exactly(device="cpu", bits=4, rounds=4):
    """adalog_amd.search.round_is_redundant: an output-MSE search whose inputs (the other operand's quantiser) are bit-identical
    to the ones it ran on in the previous round is not repeated.  The calibrated model must equal, tensor for tensor, the one
    obtained by re-running every search as the reference does (linear.py:538-541, matmul.py:275-277), and the case must
    actually exercise the shortcut."""
    import copy
    import importlib.util
    import os
    from adalog_amd import search
    from adalog_amd.quant_layers import linear as qlinear
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    dev = torch.device(device)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(f"cfg{bits}cr", os.path.join(root, "configs", f"{bits}bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = rounds, 3
    torch.manual_seed(11)
    base = VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=2, num_heads=2, num_classes=10).eval()
    for p in base.parameters():
        p.data.mul_(8.0)
    x = torch.randn(8, 3, 32, 32).to(dev)
    sds, stats = {}, {}
    keep = (search.SKIP_CONVERGED, qlinear.RUN_DEAD_W_SELF)
    keep_collect, search.COLLECT_ROUND_STATS = search.COLLECT_ROUND_STATS, True
    try:
        for skip in (False, True):                                       # False: the reference's schedule, True: the product's
            search.SKIP_CONVERGED = skip
            qlinear.RUN_DEAD_W_SELF = not skip                            # (the weights' self-MSE search the first round overwrites)
            search.reset_round_stats()
            model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
            QuantCalibrator(model, [(x, None)], capture="block").batching_quant_calib()
            model = wrap_reparamed_modules_in_net(model)
            sds[skip] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            stats[skip] = search.round_stats()
    finally:
        search.SKIP_CONVERGED, qlinear.RUN_DEAD_W_SELF = keep
        search.COLLECT_ROUND_STATS = keep_collect
    assert set(sds[False]) == set(sds[True])
    for k, a in sds[False].items():
        assert torch.equal(a, sds[True][k]), k
    assert stats[True]["unchanged"] > 0, stats                       # the shortcut was taken ...
    assert stats[False]["unchanged"] >= stats[True]["unchanged"], stats   # ... and a full run meets at least as many fixed points
    return stats


def case_capture_cache_equals_recompute(device="cpu", which="vit"):
    """QuantCalibrator's cache of finished blocks (the blocks upstream of the one being captured return their recorded outputs
    instead of recomputing them; reference utils/calibrator.py:44-47 re-runs the whole network) must not change one captured
    value: the calibrated model is equal tensor for tensor with the cache on and off."""
    import copy
    import os
    import numpy as np
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    from tests import wrapper_cases as WC
    dev = torch.device(device)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, "tests", "golden", "wrapper_rules.npz"))
    cfg = WC.cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 1, 2, 4
    base = WC._load(WC.tiny_vit() if which == "vit" else WC.tiny_swin(), g, f"{which}_in_", torch.device("cpu"))
    side = 32 if which == "vit" else 56
    x = torch.randn(8, 3, side, side, generator=torch.Generator().manual_seed(3)).to(dev)
    loader = [(x[:4], None), (x[4:], None)]                              # two calibration batches
    sds, passes = {}, {}
    keep = os.environ.get("ADALOG_CAPTURE_CACHE")
    try:
        for cache in ("0", "1"):
            os.environ["ADALOG_CAPTURE_CACHE"] = cache
            model = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(dev)
            calls = [0]
            inner = [m for n, m in model.named_modules() if n.endswith(".norm1")]          # one per block, inside it
            hooks = [b.register_forward_pre_hook(lambda m, i, _c=calls: _c.__setitem__(0, _c[0] + 1)) for b in inner]
            QuantCalibrator(model, loader, capture="block").batching_quant_calib()
            for h in hooks:
                h.remove()
            model = wrap_reparamed_modules_in_net(model)
            sds[cache] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            passes[cache] = calls[0]
    finally:
        if keep is None:
            os.environ.pop("ADALOG_CAPTURE_CACHE", None)
        else:
            os.environ["ADALOG_CAPTURE_CACHE"] = keep
    assert set(sds["0"]) == set(sds["1"])
    for k, a in sds["0"].items():
        assert torch.equal(a, sds["1"][k]), k
    return passes
