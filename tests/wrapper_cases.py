"""The product's model surgery pinned against the REFERENCE's own wrapper code.

tests/golden/wrapper_rules.npz (tools/make_golden.py::gen_wrapper) holds what the reference's utils/wrap_net.py:55-210,
its attention forwards (:19-52) and utils/block_recon.py:23-36 produce when they run over the product's module trees
(a container-only `timm` stub whose Attention / WindowAttention / Block ... are adalog_amd.utils.models' classes):
  (1) per model x bit width x reparam flag: module name -> quant class, (w|A, a|B) bits, n_V, prev_layer name, bias flag,
      num_heads, activation quantiser class, mode -- in calibration order; the full named_modules() order; the blocks BRECQ
      reconstructs; classes / `calibrated` flags / state_dict keys and shapes after wrap_reparamed_modules_in_net;
  (2) raw-mode outputs of a tiny wrapped ViT / Swin and of one attention module each (Swin: with and without a mask);
  (3) the whole flow on the tiny ViT: the reference's calibrator (LayerNorm fold included), un-wrap, reparam_bias ->
      final state_dict and quantised output.
These cases run the PRODUCT wrapper / models / calibrator and compare.  Shared by the CPU tier and the `-m gpu` tier.
"""
import importlib.util
import os

import numpy as np
import torch

from adalog_amd.utils import models as PM
from adalog_amd.utils.calibrator import QuantCalibrator
from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg_of(bits):
    spec = importlib.util.spec_from_file_location(f"cfg{bits}w", os.path.join(ROOT, "configs", f"{bits}bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Config()


def wrap_table(model):
    mods = dict(model.named_modules())
    by_id = {id(m): n for n, m in mods.items()}
    rows = []
    for name, m in mods.items():
        if not hasattr(m, "calibrated"):
            continue
        if hasattr(m, "A_quantizer"):
            rows.append((name, type(m).__name__, m.A_quantizer.n_bits, m.B_quantizer.n_bits, 0, "", 0, int(m.num_heads),
                         type(m.A_quantizer).__name__, m.mode))
        else:
            pl = getattr(m, "prev_layer", None)
            rows.append((name, type(m).__name__, m.w_quantizer.n_bits, m.a_quantizer.n_bits, int(getattr(m, "n_V", 0)),
                         by_id[id(pl)] if pl is not None else "", int(m.bias is not None), 0,
                         type(m.a_quantizer).__name__, m.mode))
    return rows


def _strs(a):
    return [str(v) for v in a]


def case_wrapper_rules(golden, tag, bits, reparam):
    g = golden("wrapper_rules")
    key = f"{tag}_w{bits}_{'reparam' if reparam else 'plain'}"
    model = PM.create_model(tag).eval()
    model = wrap_modules_in_net(model, cfg_of(bits), reparam=reparam)
    rows = wrap_table(model)
    assert [r[0] for r in rows] == _strs(g[key + "_name"])                   # same modules, same calibration order
    assert [r[1] for r in rows] == _strs(g[key + "_class"])
    assert [[r[2], r[3]] for r in rows] == g[key + "_bits"].tolist()
    assert [r[4] for r in rows] == g[key + "_nV"].tolist()
    assert [r[5] for r in rows] == _strs(g[key + "_prev"])
    assert [r[6] for r in rows] == g[key + "_bias"].tolist()
    assert [r[7] for r in rows] == g[key + "_heads"].tolist()
    assert [r[8] for r in rows] == _strs(g[key + "_aq"])
    assert [r[9] for r in rows] == _strs(g[key + "_mode"])
    assert [n for n, _ in model.named_modules()] == _strs(g[key + "_all_modules"])
    # sanity on the rules themselves, so a fixture regenerated from a broken generator cannot pass silently
    cls = dict(zip([r[0] for r in rows], [r[1] for r in rows]))
    some_qkv = next(n for n in cls if n.endswith("attn.qkv"))
    assert cls[some_qkv] == ("AsymmetricallyChannelWiseBatchingQuantLinear" if reparam else "AsymmetricallyBatchingQuantLinear")
    assert all(c == "PostGeluLogBasedBatchingQuantLinear" for n, c in cls.items() if n.endswith("mlp.fc2"))
    assert all(c == "PostSoftmaxAsymmetricallyBatchingQuantMatMul" for n, c in cls.items() if n.endswith("matmul2"))
    if not (reparam and bits == 4):
        return
    from adalog_amd.utils.block_recon import BlockReconstructor
    rec = BlockReconstructor(model, PM.create_model(tag).eval(), None)
    assert list(rec.blocks.keys()) == _strs(g[f"{tag}_blocks"])
    assert list(rec.full_blocks.keys()) == _strs(g[f"{tag}_full_blocks"])
    for m in model.modules():
        if type(m).__name__ == "AsymmetricallyChannelWiseBatchingQuantLinear":
            del m.a_quantizer.scale, m.a_quantizer.zero_point
            m.a_quantizer.channel_wise = False
            m.a_quantizer.scale = torch.nn.Parameter(torch.ones(1))
            m.a_quantizer.zero_point = torch.nn.Parameter(torch.zeros(1))
    model = wrap_reparamed_modules_in_net(model)
    rows2 = wrap_table(model)
    assert [r[1] for r in rows2] == _strs(g[key + "_unwrapped_class"])
    mods = dict(model.named_modules())
    assert [int(mods[r[0]].calibrated) for r in rows2] == g[key + "_unwrapped_calibrated"].tolist()
    sd = model.state_dict()
    assert list(sd.keys()) == _strs(g[key + "_sd_keys"])                     # the checkpoint wire format, key order included
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == _strs(g[key + "_sd_shapes"])


def _load(model, g, prefix, dev):
    sd = {k[len(prefix):].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix)}
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    return model.to(dev)


def tiny_vit():
    return PM.VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=2, num_heads=4, num_classes=10).eval()


def tiny_swin():
    return PM.SwinTransformer(img_size=56, patch_size=4, embed_dim=16, depths=(2, 2), num_heads=(2, 4), window_size=7,
                              num_classes=10).eval()


def _close(a, ref, rtol=1e-5, atol=1e-5):
    torch.testing.assert_close(a.detach().cpu(), torch.from_numpy(np.asarray(ref)).reshape(a.shape), rtol=rtol, atol=atol)


def case_attention_forwards(golden, device="cpu"):
    """The product's Attention.forward / WindowAttention.forward against the reference's vit_attn_forward /
    swin_attn_forward (wrap_net.py:19-52) on the same seeded weights, before and after the product's wrapping (raw mode)."""
    dev = torch.device(device)
    g = golden("wrapper_rules")
    cfg = cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 1, 2, 2
    x, tok = torch.from_numpy(g["vit_x"]).to(dev), torch.from_numpy(g["vit_tok"]).to(dev)
    vit = _load(tiny_vit(), g, "vit_in_", dev)
    with torch.no_grad():
        _close(vit(x), g["vit_out_raw"])
        _close(vit.blocks[1].attn(tok), g["vit_attn_out_raw"])
        vit = wrap_modules_in_net(vit, cfg, reparam=True)
        _close(vit(x), g["vit_out_raw"])
        _close(vit.blocks[1].attn(tok), g["vit_attn_out_raw"])
    xs, wtok, mask = (torch.from_numpy(g[k]).to(dev) for k in ("swin_x", "swin_wtok", "swin_mask"))
    swin = _load(tiny_swin(), g, "swin_in_", dev)
    with torch.no_grad():
        _close(swin(xs), g["swin_out_raw"], 1e-4, 1e-4)
        swin = wrap_modules_in_net(swin, cfg, reparam=True)
        _close(swin(xs), g["swin_out_raw"], 1e-4, 1e-4)
        att = swin.layers[0].blocks[1].attn
        _close(att(wtok), g["swin_attn_out_nomask"])
        _close(att(wtok, mask), g["swin_attn_out_mask"])


def case_wrapped_vit_flow(golden, device="cpu", capture="module"):
    """wrap -> calibrate (channel-wise search + LayerNorm fold) -> un-wrap -> reparam_bias on the tiny ViT, against the
    reference's own run of the same flow: calibration order, every parameter of the final state_dict (scales / weights /
    biases <= 1e-3 relative, zero points / log bases / LUTs / flags exact), the quantised output <= 1e-3."""
    dev = torch.device(device)
    g = golden("wrapper_rules")
    cfg = cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 1, 2, 2
    x = torch.from_numpy(g["vit_x"]).to(dev)
    vit = wrap_modules_in_net(_load(tiny_vit(), g, "vit_in_", dev), cfg, reparam=True)
    order = []
    for name, m in vit.named_modules():
        if hasattr(m, "hyperparameter_searching"):
            orig = m.hyperparameter_searching
            m.hyperparameter_searching = (lambda orig=orig, name=name: (order.append(name), orig())[1])
    QuantCalibrator(vit, [(x[:2], None), (x[2:], None)], capture=capture).batching_quant_calib()
    assert order == _strs(g["vit_calib_order"])
    vit = wrap_reparamed_modules_in_net(vit)
    with torch.no_grad():
        for m in vit.modules():
            if hasattr(m, "mode") and hasattr(m, "reparam_bias"):
                m.reparam_bias()
        out = vit(x)
    ref_sd = {k[len("vit_out_"):].replace("__", "."): torch.from_numpy(np.asarray(v)) for k, v in g.items()
              if k.startswith("vit_out_") and k != "vit_out_raw" and k != "vit_out_quant"}
    mine = vit.state_dict()
    assert list(mine.keys()) == list(ref_sd.keys())
    n_scale = n_off = 0
    for k, v in mine.items():
        r = ref_sd[k]
        assert tuple(v.shape) == tuple(r.shape), k
        v = v.detach().cpu()
        if k.endswith(".scale"):
            n_scale += v.numel()
            n_off += int(((v - r).abs() > 1e-3 * r.abs()).sum())
        elif k.endswith("weight") or k.endswith("bias") or "norm" in k or k in ("cls_token", "pos_embed"):
            torch.testing.assert_close(v, r, rtol=2e-3, atol=2e-4, msg=lambda m: f"{k}: {m}")
    ref_q, raw = torch.from_numpy(g["vit_out_quant"]), torch.from_numpy(g["vit_out_raw"])
    e_mine = ((out.cpu() - raw) ** 2).mean().item()
    e_ref = ((ref_q - raw) ** 2).mean().item()
    return {"scales": n_scale, "scales_off": n_off, "mse_mine": e_mine, "mse_ref": e_ref,
            "max_out_diff": (out.cpu() - ref_q).abs().max().item(), "out_max": ref_q.abs().max().item(),
            "exact": {k: bool(torch.equal(mine[k].detach().cpu().to(ref_sd[k].dtype), ref_sd[k])) for k in mine
                      if not (k.endswith(".scale") or k.endswith("weight") or k.endswith("bias") or "norm" in k
                              or k in ("cls_token", "pos_embed"))}}
