"""`-m gpu` end-to-end: the test_quant.py entry point (calibrate -> checkpoint -> reload), a Swin stage with PatchMerging
(`reduction` + LayerNorm fold), and fused quant_forward vs the composed fake-quant forward."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
# floor of the logit SQNR (quantised vs FP model, held-out synthetic images, seeded random-init weights) a CALIBRATED model of the
# zoo reaches at each bit width; observed on MI355X: see gpurun_out/e2e_outcomes.jsonl / profiles/r05_e2e_outcomes.jsonl
MIN_SQNR_DB = {3: 1.0, 4: 5.0, 6: 10.0}
# per run: within 1 dB below what the calibrated model of that run reached on MI355X (profiles/r05_e2e_outcomes.jsonl: deit_tiny W6A6
# 14.3 dB, vit_base W4A4 7.85, swin_base W3A3 on two ranks 9.87, deit_base W3A3 3.35) -- the searches are bit-reproducible, so a drop
# of a dB is a changed search, not noise
MIN_SQNR_RUN = {"deit_tiny_w6": 13.3, "vit_base_w4": 6.85, "swin_base_w3": 8.87, "deit_base_w3": 2.35}


def _rec_loss_lines(text):
    """[(block, before, after)] from the reconstruction's per-block report (utils/block_recon.py: the block's reconstruction loss on a
    fixed set of its optimisation images, soft rounding targets, before the first and after the last iteration)"""
    import re
    return [(n, float(a), float(b)) for n, a, b in
            re.findall(r"(\S+): reconstruction loss on its first \d+ images ([-0-9.eE+naninf]+) -> ([-0-9.eE+naninf]+) after", text)]


def _assert_reconstruction_lowers_the_loss(text, n_blocks):
    """BRECQ is gradient descent on each block's reconstruction loss (block_recon.py:114-127): over the run the loss on a FIXED set of
    the block's images must not rise -- summed over the blocks it must fall, and no single block may end more than 2 % above where it
    started (24 Adam iterations of lr 1e-3 / 4e-5 on mini-batches of 32: a block's fixed-set loss moves by a few per cent)."""
    rec = _rec_loss_lines(text)
    assert len(rec) == n_blocks, (len(rec), n_blocks, rec[:3])
    assert all(a == a and b == b and a >= 0 and b >= 0 for _, a, b in rec), rec
    try:
        with open(os.path.join(ROOT, "gpurun_out", "e2e_outcomes.jsonl"), "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "rec_loss_before_after": rec}) + "\n")
    except OSError:
        pass
    assert sum(b for _, _, b in rec) <= sum(a for _, a, _ in rec), rec
    assert all(b <= 1.02 * a + 1e-12 for _, a, b in rec), [r for r in rec if r[2] > 1.02 * r[1]]


def _cfg(bits=4, rounds=1, steps=2):
    import importlib.util
    spec = importlib.util.spec_from_file_location(f"cfg{bits}e", os.path.join(ROOT, "configs", f"{bits}bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = rounds, steps
    return cfg


def _fidelity_lines(text):
    """(top-1 agreement %, logit SQNR dB) of every `validate_fidelity` report of a test_quant.py run on synthetic data, in order:
    the OUTCOME of the run -- the quantised model's logits against the FP model's on held-out images (a calibration that wrote
    garbage scales gives a non-positive or NaN SQNR, and passes every file / shape check)."""
    import re
    out = [(float(a), float(b)) for a, b in re.findall(r"top-1 agreement with FP model ([-0-9.naninf]+)%\s+logit SQNR ([-0-9.naninf]+) dB", text)]
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "e2e_outcomes.jsonl"), "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "fidelity": out}) + "\n")
    except OSError:
        pass
    return out


def test_cli_calibrate_save_and_reload(tmp_path):
    out = str(tmp_path / "run")
    cmd = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "deit_tiny", "--config",
           os.path.join(ROOT, "configs", "6bit.py"), "--calibrate", "--calib-size", "8", "--calib-batch-size", "8",
           "--val-size", "16", "--val-batch-size", "16", "--output-dir", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ckpt = os.path.join(out, "deit_tiny_w6_a6_s6_calibsize_8.pth")            # reference naming, test_quant.py:98-99
    assert os.path.exists(ckpt)
    sd = torch.load(ckpt, map_location="cpu")
    # reference checkpoint wire format: plain layers after un-wrapping, AdaLog buffers, re-parameterised bias flag
    assert sd["blocks.0.attn.qkv.w_quantizer.scale"].shape == (3, 192, 1)
    assert sd["blocks.0.attn.qkv.a_quantizer.scale"].shape == (1,)
    assert sd["blocks.0.attn.matmul2.A_quantizer.q"].dtype == torch.int64
    assert sd["blocks.0.mlp.fc2.a_quantizer.table2"].shape == (64,)
    assert bool(sd["blocks.0.mlp.fc2.a_quantizer.bias_reparamed"])
    fid = _fidelity_lines(r.stdout + r.stderr)
    assert len(fid) == 1 and fid[0][1] == fid[0][1] and fid[0][1] > MIN_SQNR_RUN["deit_tiny_w6"], fid
    assert sd["patch_embed.proj.w_quantizer.zero_point"].shape == (192, 1)
    assert "agreement" in r.stdout + r.stderr
    cmd2 = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "deit_tiny", "--config",
            os.path.join(ROOT, "configs", "6bit.py"), "--load-calibrate-checkpoint", ckpt, "--test-calibrate-checkpoint",
            "--val-size", "16", "--val-batch-size", "16", "--output-dir", out]
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]

    def agreement(text):
        line = [ln for ln in text.splitlines() if "agreement" in ln][-1]
        return float(line.split("model")[1].split("%")[0])
    assert abs(agreement(r.stdout + r.stderr) - agreement(r2.stdout + r2.stderr)) < 1e-6     # reload reproduces the model


def test_swin_stage_calibration():
    from adalog_amd import quant_layers as Q
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import SwinTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    torch.manual_seed(3)
    model = SwinTransformer(img_size=56, patch_size=4, embed_dim=32, depths=(2, 2), num_heads=(2, 4), window_size=7,
                            num_classes=10).eval()
    for p in model.parameters():
        p.data.mul_(6.0)
    model.to(DEV)
    x = torch.randn(8, 3, 56, 56, device=DEV)
    with torch.no_grad():
        y_fp = model(x)
    model = wrap_modules_in_net(model, _cfg(6, 1, 3), reparam=True)
    names = [n for n, m in model.named_modules() if hasattr(m, "calibrated")]
    assert "layers.1.downsample.reduction" in names and "layers.0.blocks.1.attn.matmul2" in names and "head.fc" in names
    red = model.layers[1].downsample.reduction
    assert isinstance(red, Q.AsymmetricallyChannelWiseBatchingQuantLinear) and red.prev_layer is model.layers[1].downsample.norm
    assert red.bias is None                                       # timm's reduction is bias-free (test_quant.py:116-117)
    QuantCalibrator(model, [(x[:4], None), (x[4:], None)]).batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    assert model.layers[1].downsample.reduction.bias is not None  # the LayerNorm fold creates it (linear.py:609-611)
    with torch.no_grad():
        y_q = model(x)
    rel = ((y_q - y_fp).norm() / y_fp.norm()).item()
    assert torch.isfinite(y_q).all() and rel < 0.35, rel


def test_fused_quant_forward_matches_composed():
    """The fused int8/bf16 MFMA forward (pack + GEMM epilogue) equals fake-quant(x) @ fake-quant(W)^T to fp32 rounding."""
    from adalog_amd import quant_layers as Q
    from tests import layer_cases as LC
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "linear_w4a4.npz"))
    LC.DEV[0] = torch.device(DEV)
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "quant_forward", wb, ab, n_V=n_V, fpcs=True).to(DEV)
    sd = {k[4:].replace("__", "."): LC.t(v) for k, v in g.items() if k.startswith("out_")}
    lay.load_state_dict(sd)
    lay.calibrated = lay.a_quantizer.inited = lay.w_quantizer.inited = True
    x = LC.t(g["x"])
    with torch.no_grad():
        fused = lay(x)
        composed = Q.MinMaxQuantLinear.quant_forward(lay, x)
    torch.testing.assert_close(fused, composed, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fused.cpu(), torch.from_numpy(g["qf_out"]), rtol=1e-4, atol=1e-4)
    big = torch.randn(3, 300, I, device=DEV)                        # rows not a multiple of any tile
    with torch.no_grad():
        torch.testing.assert_close(lay(big), Q.MinMaxQuantLinear.quant_forward(lay, big), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("capture", ["module", "block"])
def test_calibrator_matches_reference_run_on_hip(golden, capture):
    """The reference's own calibrator run on the toy tree (golden calibrator_toy): visiting order, captured shapes,
    every calibrated parameter and the quantised output, with the HIP kernels doing the searches."""
    from adalog_amd import backend
    from tests import calibrator_cases as CC
    backend.set_backend(None)
    CC.case_calibrator_matches_reference_run(golden, capture, DEV)


@pytest.mark.parametrize("bits", [4, 6])
def test_block_capture_equals_module_capture_on_hip(bits):
    from adalog_amd import backend
    from tests import calibrator_cases as CC
    backend.set_backend(None)
    r = CC.case_capture_equivalence(DEV, bits)
    assert r["scales_off"] <= 0.01 * r["scales"], r                # near-tie flips only
    assert abs(r["mse_block"] / r["mse_module"] - 1.0) <= 0.02, r


def test_converged_rounds_are_skipped_exactly_on_hip():
    from adalog_amd import backend
    from tests import calibrator_cases as CC
    backend.set_backend(None)
    stats = CC.case_converged_rounds_are_skipped_exactly(DEV)
    assert stats[True]["checked"] > 0


@pytest.mark.parametrize("which", ["vit", "swin"])
def test_capture_cache_equals_recompute_on_hip(which):
    from adalog_amd import backend
    from tests import calibrator_cases as CC
    backend.set_backend(None)
    passes = CC.case_capture_cache_equals_recompute(DEV, which)
    assert 0 < passes["1"] < passes["0"]


def test_two_lanes_on_one_gpu_give_the_sequential_parameters(monkeypatch):
    """ADALOG_LANES=2 (one process: two modules' searches side by side on two streams, calibrator._search_interleaved) must
    calibrate every parameter exactly as the sequential schedule does -- the modules' searches are independent
    (reference utils/calibrator.py:34-67)."""
    import copy
    import numpy as np
    from adalog_amd import backend
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.wrap_net import wrap_modules_in_net
    from tests import wrapper_cases as WC
    backend.set_backend(None)
    g = np.load(os.path.join(ROOT, "tests", "golden", "wrapper_rules.npz"))
    cfg = WC.cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 2, 3, 8
    base = WC._load(WC.tiny_vit(), g, "vit_in_", torch.device("cpu"))
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    states = {}
    for lanes in ("1", "2"):
        monkeypatch.setenv("ADALOG_LANES", lanes)
        vit = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(DEV)
        QuantCalibrator(vit, [(x, None)], capture="block").batching_quant_calib()
        torch.cuda.synchronize()
        states[lanes] = {k: v.detach().cpu().clone() for k, v in vit.state_dict().items()}
    assert states["1"].keys() == states["2"].keys()
    for k, v in states["1"].items():
        assert torch.equal(v, states["2"][k]), k


def _score_w_f64(x, a_s, a_z, a_bits, w2, b, raw_out, w_s, w_z, w_bits):
    """The reference's weight-search objective (quant_layers/linear.py:355-384) of ONE (scale, zero point) per output row, with the
    fake quantisations done by the oracle exactly as the reference does them (fp32: uniform.py:29-36) and everything after them --
    the products, the squared error, mean over tokens, sum over images -- in fp64: -> [O] scores (higher is better)."""
    from oracle import adalog_oracle as O
    xq = O.uniform_fake_quant(x, a_s, a_z, a_bits)[0].double()
    wq = O.uniform_fake_quant(w2, w_s.view(-1, 1), w_z.view(-1, 1), w_bits)[0].double()
    out = torch.nn.functional.linear(xq, wq, None if b is None else b.double())
    err = (raw_out.double() - out) ** 2
    return -err.reshape(err.shape[0], -1, err.shape[-1]).mean(1).sum(0)


def test_gram_forms_calibrate_like_the_token_forms(monkeypatch):
    """The Gram forms of the Linear searches (csrc/gram.hip, gram_act.hip; forced wherever supported: ADALOG_GRAM_W = ADALOG_GRAM_A = 2)
    against the token-form kernels (= 0), whole calibrations of the same model.  Scores agree to ~1e-6, so the committed parameters may
    only differ where neighbouring candidates of a late FPCS step tie.  What is asserted (round 6) is the CLAIM behind that, per search:
    inside the Gram calibration every output-MSE weight search is run a second time on the token-form kernels from the same state, and
    on every output row where the two commit different parameters the reference's own objective (linear.py:355-384, fake quantisation
    by the oracle in fp32, accumulation in fp64) of the Gram pick is at least the token pick's (to 1e-6 relative) -- the Gram form is
    the more faithful of the two, not merely close.  Activation parameters of the two calibrations agree to 1e-4."""
    import copy
    import numpy as np
    from adalog_amd import backend, ops, search
    from adalog_amd import quant_layers as Q
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.wrap_net import wrap_modules_in_net
    from tests import wrapper_cases as WC
    backend.set_backend(None)
    g = np.load(os.path.join(ROOT, "tests", "golden", "wrapper_rules.npz"))
    cfg = WC.cfg_of(4)
    cfg.search_round, cfg.steps, cfg.calib_batch_size = 2, 4, 32
    base = WC._load(WC.tiny_vit(), g, "vit_in_", torch.device("cpu"))
    x = torch.randn(32, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    xh = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(4)).to(DEV)
    calls = {"w": 0, "a": 0}
    sw, sa = ops.GramState.score_w, ops.GramActState.score
    monkeypatch.setattr(ops.GramState, "score_w", lambda self, *a, **k: (calls.__setitem__("w", calls["w"] + 1), sw(self, *a, **k))[1])
    monkeypatch.setattr(ops.GramActState, "score", lambda self, *a, **k: (calls.__setitem__("a", calls["a"] + 1), sa(self, *a, **k))[1])
    per_search = []                               # one record per output-MSE weight search of the Gram calibration
    orig_wfpcs = Q.AsymmetricallyBatchingQuantLinear.weight_fpcs

    def weight_fpcs_both(self, fpcs_width=16, steps=6, search_strategy="output"):
        if search_strategy != "output" or os.environ.get("ADALOG_GRAM_W") != "2":
            return orig_wfpcs(self, fpcs_width, steps, search_strategy)
        before = calls["w"]
        seen = copy.copy(self.__dict__.get("_round_inputs", {}))
        orig_wfpcs(self, fpcs_width, steps, search_strategy)
        if calls["w"] == before:
            return                                # this search is not one the Gram form takes (or was skipped as converged)
        wq, aq = self.w_quantizer, self.a_quantizer
        g_s, g_z = wq.scale.detach().clone(), wq.zero_point.detach().clone()
        # the same search from the same state on the token-form kernels
        self.__dict__["_round_inputs"] = copy.copy(seen)
        os.environ["ADALOG_GRAM_W"] = "0"
        try:
            orig_wfpcs(self, fpcs_width, steps, search_strategy)
        finally:
            os.environ["ADALOG_GRAM_W"] = "2"
        assert calls["w"] == before + steps, "the second run must have taken the token-form kernels"
        t_s, t_z = wq.scale.detach().clone(), wq.zero_point.detach().clone()
        differ = ((g_s != t_s) | (g_z != t_z)).view(-1)
        rec = {"rows": int(differ.numel()), "differ": int(differ.sum()), "worst_deficit_rel": 0.0}
        if rec["differ"]:
            args = (self.raw_input.cpu(), aq.scale.detach().cpu(), aq.zero_point.detach().cpu(), aq.n_bits,
                    self.weight.detach().cpu(), None if self.bias is None else self.bias.detach().cpu(), self.raw_out.cpu())
            sg = _score_w_f64(*args, g_s.cpu().view(-1), g_z.cpu().view(-1), wq.n_bits)
            st = _score_w_f64(*args, t_s.cpu().view(-1), t_z.cpu().view(-1), wq.n_bits)
            d = differ.cpu()
            deficit = ((st - sg) / st.abs().clamp_min(1e-300))[d]        # > 0: the token pick scores better by the reference's objective
            rec["worst_deficit_rel"] = float(deficit.max())
            rec["gram_better_rows"] = int((sg[d] > st[d]).sum())
        per_search.append(rec)
        wq.scale.data.copy_(g_s)                  # the calibration goes on with the Gram pick
        wq.zero_point.data.copy_(g_z)
        self.invalidate_packed_weight()

    monkeypatch.setattr(Q.AsymmetricallyBatchingQuantLinear, "weight_fpcs", weight_fpcs_both)
    states, outs, used = {}, {}, {}
    for mode in ("0", "2"):
        monkeypatch.setenv("ADALOG_GRAM_W", mode)
        monkeypatch.setenv("ADALOG_GRAM_A", mode)
        calls["w"] = calls["a"] = 0
        vit = wrap_modules_in_net(copy.deepcopy(base), cfg, reparam=True).to(DEV)
        QuantCalibrator(vit, [(x, None)], capture="block").batching_quant_calib()
        with torch.no_grad():
            outs[mode] = vit(xh).float().cpu()
        states[mode] = {k: v.detach().float().cpu() for k, v in vit.state_dict().items() if "quantizer" in k}
        used[mode] = dict(calls)
    assert used["0"] == {"w": 0, "a": 0} and used["2"]["w"] > 0 and used["2"]["a"] > 0, used
    assert per_search, "no output-MSE weight search of the Gram calibration was compared"
    worst_a, rows, rows_off, worst_w = 0.0, 0, 0, 0.0
    for k, v in states["0"].items():
        u = states["2"][k]
        rel = ((u - v).abs() / v.abs().clamp_min(1e-12))
        if ".a_quantizer." in k or ".A_quantizer." in k or ".B_quantizer." in k:
            worst_a = max(worst_a, rel.max().item())
        elif k.endswith("w_quantizer.scale"):
            rows += rel.numel()
            rows_off += int((rel > 0).sum())
            worst_w = max(worst_w, rel.max().item())
    sqnr = 10 * torch.log10(outs["0"].pow(2).sum() / (outs["0"] - outs["2"]).pow(2).sum().clamp_min(1e-30)).item()
    worst_deficit = max(r["worst_deficit_rel"] for r in per_search)
    with open(os.path.join(ROOT, "gpurun_out", "e2e_outcomes.jsonl"), "a") as f:
        f.write(json.dumps({"case": "gram_vs_token_forms", "gram_calls": used["2"], "worst_activation_param_rel": worst_a,
                            "weight_rows": rows, "weight_rows_differing": rows_off, "worst_weight_scale_rel": worst_w,
                            "logit_sqnr_db_between_the_two": sqnr, "searches_compared": len(per_search),
                            "rows_differing_per_search": [r["differ"] for r in per_search],
                            "gram_better_rows": sum(r.get("gram_better_rows", 0) for r in per_search),
                            "worst_token_over_gram_deficit_rel": worst_deficit}) + "\n")
    assert worst_a <= 1e-4, worst_a
    # wherever the two forms commit different parameters, the Gram pick is at least as good by the reference's own (fp64) objective
    assert worst_deficit <= 1e-6, per_search


def test_cli_vit_base_calibrate_and_optimize(tmp_path):
    """BASELINE config 3 in reduced form: vit_base W4A4 `--calibrate --optimize` through the CLI -- calibration of the
    768-wide model (two-row-tile fused search, K = 768 int8 searches), then BRECQ over every block (HIP-graph replay,
    fused kernels) over the config's 1024 optimisation images for a few iterations, checkpoints in the reference's naming, reload of the optimised checkpoint."""
    out = str(tmp_path / "run3")
    cmd = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "vit_base", "--config",
           os.path.join(ROOT, "configs", "4bit.py"), "--calibrate", "--optimize", "--calib-size", "32", "--calib-batch-size", "32",
           "--optim-iters", "24", "--val-size", "32", "--val-batch-size", "32", "--output-dir", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)     # optim_size = 1024 (configs/4bit.py)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    files = os.listdir(out)
    assert any(f.startswith("vit_base_w4_a4_s4_calibsize_32") for f in files), files
    opt = [f for f in files if f.startswith("vit_base_w4_a4_s4_optimsize_")]
    assert opt, files
    sd = torch.load(os.path.join(out, opt[0]), map_location="cpu")
    assert sd["blocks.11.mlp.fc2.w_quantizer.scale"].shape == (1, 768, 1)
    assert not any(k.endswith("alpha") for k in sd)                           # hard rounding committed (block_recon.py:151-157)
    # outcome: logits against the FP model on held-out images, after calibration and after the (24-iteration) reconstruction --
    # a 4-bit calibration keeps the logits well above the noise floor, and a reconstruction this short must leave the model about
    # where the calibration put it (hard rounding of barely trained alphas = nearest rounding), not wreck it
    fid = _fidelity_lines(r.stdout + r.stderr)
    assert len(fid) == 2, fid
    assert all(q == q and q > MIN_SQNR_DB[4] for _, q in fid), fid
    assert fid[0][1] >= MIN_SQNR_RUN["vit_base_w4"], fid                      # the calibrated model
    # the reconstruction lowers every block's reconstruction loss (patch embedding + 12 blocks + head); the logit SQNR after so short
    # a run is reported, not asserted against the calibrated one (hard rounding of barely trained alphas moves it either way)
    _assert_reconstruction_lowers_the_loss(r.stdout + r.stderr, 14)
    cmd2 = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "vit_base", "--config",
            os.path.join(ROOT, "configs", "4bit.py"), "--load-optimize-checkpoint", os.path.join(out, opt[0]),
            "--test-optimize-checkpoint", "--val-size", "32", "--val-batch-size", "32", "--output-dir", out]
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    # the reloaded checkpoint scores what the run scored (not to the digit: the run's model still carries the AdaRound quantisers and
    # takes the composed fp32 forward, the reloaded one the integer-MFMA forward -- a 4-bit random-init model amplifies that rounding-level
    # difference over its 12 blocks; observed 7.40 vs 7.46 dB)
    fid2 = _fidelity_lines(r2.stdout + r2.stderr)
    assert len(fid2) == 1 and abs(fid2[0][1] - fid[1][1]) < 0.5, (fid, fid2)


def test_cli_dataset_folder_calibrate_and_validate(tmp_path):
    """`--dataset <ImageNet-style folder>`: the reference's loader / validate path (utils/datasets.py, utils/test_utils.py)
    on a generated toy folder -- calibration subset drawn from train/, Prec@1 / Prec@5 on val/."""
    import numpy as np
    from PIL import Image
    rng = np.random.RandomState(0)
    root = tmp_path / "toy_imagenet"
    for split, per in (("train", 6), ("val", 3)):
        for ci in range(4):
            d = root / split / f"n{ci:04d}"
            d.mkdir(parents=True)
            for j in range(per):
                Image.fromarray((rng.rand(70 + 8 * j, 96 - 5 * ci, 3) * 255).astype(np.uint8)).save(d / f"img{j}.jpg")
    out = str(tmp_path / "run4")
    cmd = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "deit_tiny", "--config",
           os.path.join(ROOT, "configs", "4bit.py"), "--dataset", str(root), "--calibrate", "--calib-size", "16",
           "--calib-batch-size", "8", "--val-batch-size", "4", "--num-workers", "0", "--output-dir", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    text = r.stdout + r.stderr
    assert "Prec@1" in text and "FP model Prec@1" in text
    assert os.path.exists(os.path.join(out, "deit_tiny_w4_a4_s4_calibsize_16.pth"))


def test_cli_swin_base_w3a3_sharded_over_two_ranks(tmp_path):
    """BASELINE config 4 in reduced form: swin_base W3A3 `--calibrate` with the calibration images SHARDED over the ranks
    of a torch.distributed.run launch (score all-reduce per search step, quantile statistics all-gathered) -- two ranks
    sharing this box's one GPU over gloo (RCCL needs one GPU per rank); rank 0 writes the checkpoint."""
    out = str(tmp_path / "run5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "test_quant.py"), "--model", "swin_base", "--config",
           os.path.join(ROOT, "configs", "3bit.py"), "--calibrate", "--calib-size", "32", "--calib-batch-size", "16",
           "--val-size", "16", "--val-batch-size", "16", "--output-dir", out]
    env = dict(os.environ, ADALOG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ckpt = os.path.join(out, "swin_base_w3_a3_s3_calibsize_32.pth")
    assert os.path.exists(ckpt)
    sd = torch.load(ckpt, map_location="cpu")
    assert sd["layers.0.blocks.0.attn.qkv.w_quantizer.scale"].shape == (3, 128, 1)
    assert "on 2 GPU(s)" in r.stdout + r.stderr
    fid = _fidelity_lines(r.stdout + r.stderr)                                # (both ranks report: the same calibrated model)
    assert fid and all(q == q and q > MIN_SQNR_RUN["swin_base_w3"] for _, q in fid), fid


def test_cli_deit_base_w3a3_calibrate_and_optimize(tmp_path):
    """BASELINE config 5 on one GPU (its eight-GPU part is covered by the sharded test above and the gloo tests): deit_base W3A3
    `--calibrate --optimize` through the CLI -- calibration at 3 bit (fp8 storage of the q.k^T / linear operands), then BRECQ over
    every block for a few iterations on the config's 1024 optimisation images, checkpoints in the reference's naming."""
    out = str(tmp_path / "run6")
    cmd = [sys.executable, os.path.join(ROOT, "test_quant.py"), "--model", "deit_base", "--config",
           os.path.join(ROOT, "configs", "3bit.py"), "--calibrate", "--optimize", "--optim-iters", "24", "--calib-size", "32",
           "--calib-batch-size", "32", "--val-size", "32", "--val-batch-size", "32", "--output-dir", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    files = os.listdir(out)
    assert os.path.exists(os.path.join(out, "deit_base_w3_a3_s3_calibsize_32.pth"))
    opt = [f for f in files if f.startswith("deit_base_w3_a3_s3_optimsize_")]
    assert opt, files
    sd = torch.load(os.path.join(out, opt[0]), map_location="cpu")
    assert sd["blocks.11.attn.qkv.w_quantizer.scale"].shape == (3, 768, 1)
    assert not any(k.endswith("alpha") for k in sd)                           # hard rounding committed (block_recon.py:151-157)
    fid = _fidelity_lines(r.stdout + r.stderr)
    assert len(fid) == 2 and all(q == q and q > MIN_SQNR_DB[3] for _, q in fid), fid
    assert fid[0][1] >= MIN_SQNR_RUN["deit_base_w3"], fid                     # the calibrated model
    _assert_reconstruction_lowers_the_loss(r.stdout + r.stderr, 14)


@pytest.mark.parametrize("bits", [4, 6])
def test_fused_block_quant_forward_matches_the_module_route(bits):
    """quant_forward of a calibrated deit_tiny on the fused block route (utils/models.py: Attention._fused_quant_forward, Mlp.forward
    -- q / k / v split-quantise-pack in one launch, softmax + AdaLog quantiser + pack in one launch, softmax.v written heads-last, GELU
    inside fc2's packer, residuals in the projections' epilogues) against the module-by-module route (every quantised module's own
    quant_forward, torch softmax / GELU / adds between them; ADALOG_QF_FUSED=0).  Every fused piece is bit-identical to what it
    replaces (tests/test_gpu_kernels.py), so the logits must agree to fp32 rounding of the few re-associated adds -- and the fused
    route must launch far fewer kernels."""
    from torch.profiler import ProfilerActivity, profile
    from adalog_amd.utils import models as M
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    torch.manual_seed(5)
    model = wrap_modules_in_net(create_model("deit_tiny").eval(), _cfg(bits), reparam=True).to(DEV)
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(DEV)
    QuantCalibrator(model, [(x, None)], capture="block").batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model).to(DEV).eval()
    for m in model.modules():
        if hasattr(m, "reparam_bias"):
            m.reparam_bias()
        if hasattr(m, "mode"):
            m.mode = "quant_forward"

    def run(fused):
        M.QF_FUSED = fused
        with torch.no_grad():
            model(x)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                y = model(x)
                torch.cuda.synchronize()
        n = sum(e.count for e in prof.key_averages() if "DeviceType.CUDA" in str(getattr(e, "device_type", "")))
        return y, n
    try:
        y_mod, n_mod = run(False)
        y_fused, n_fused = run(True)
    finally:
        M.QF_FUSED = True
    rel = ((y_fused - y_mod).norm() / y_mod.norm()).item()
    assert torch.isfinite(y_fused).all() and rel <= 1e-5, rel
    assert n_fused <= 0.72 * n_mod, (n_fused, n_mod)
    y2 = None
    with torch.no_grad():
        y2 = model(x)
    assert torch.equal(y2, y_fused)                                    # deterministic


def _minmax_params(t, bits, per=None):
    """(scale, zero_point) of an asymmetric uniform quantiser from the tensor's range; per: dims to KEEP (None = per tensor)."""
    if per is None:
        mn, mx = t.min(), t.max()
    else:
        red = [d for d in range(t.dim()) if d not in per]
        mn, mx = t.amin(dim=red, keepdim=True), t.amax(dim=red, keepdim=True)
    s = (mx - mn).clamp_min(1e-6) / (2 ** bits - 1)
    return s, torch.round(-mn / s).clamp(0, 2 ** bits - 1)


def _arm(q, s, z=None):
    q.scale.data.copy_(s.reshape(q.scale.shape))
    if z is not None:
        q.zero_point.data.copy_(z.reshape(q.zero_point.shape))
    q.inited = True
    q._zp_on_grid = True


@pytest.mark.parametrize("bits", [4, 6])
@pytest.mark.parametrize("cls", ["linear", "linear_channelwise", "postgelu", "postgelu_bias_reparamed", "matmul", "postsoftmax", "conv"])
def test_quant_forward_matches_composed_at_full_shape(cls, bits):
    """quant_forward of each of the six concrete layer classes at deit_small's shapes (32 images) on the product route (operand
    generation / packs + integer or bf16 MFMA product with the dequantising epilogue) against the composed route of the base class
    (fake-quantise both operands, then the fp32 product: reference linear.py:46-51, matmul.py:43-45, conv.py:60-65) -- the same
    quantised operands, so the results agree to fp32 accumulation order (1e-5 of the output's range)."""
    from adalog_amd import quant_layers as Q
    g = torch.Generator().manual_seed(77 + bits)
    N, T, D, H = 32, 197, 384, 6

    def close(a, b):
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert ((a - b).abs().max() / b.abs().max()).item() <= 1e-5, ((a - b).abs().max() / b.abs().max()).item()

    with torch.no_grad():
        if cls in ("linear", "linear_channelwise"):
            C = Q.AsymmetricallyBatchingQuantLinear if cls == "linear" else Q.AsymmetricallyChannelWiseBatchingQuantLinear
            lay = C(D, 3 * D, True, "quant_forward", bits, bits, n_V=3 if cls != "linear" else 1, fpcs=True).to(DEV)
            x = torch.randn(N, T, D, generator=g).to(DEV) * (0.5 + torch.rand(D, generator=g).to(DEV))
            ws, wz = _minmax_params(lay.weight.data.view(lay.n_V, lay.crb_rows, D), bits, per=(0, 1))
            _arm(lay.w_quantizer, ws, wz)
            a_s, a_z = _minmax_params(x, bits, per=(2,) if cls != "linear" else None)
            _arm(lay.a_quantizer, a_s, a_z)
            lay.calibrated = True
            close(lay(x), Q.MinMaxQuantLinear.quant_forward(lay, x))
        elif cls.startswith("postgelu"):
            lay = Q.PostGeluLogBasedBatchingQuantLinear(4 * D, D, True, "quant_forward", bits, bits, n_V=1, quantizer="adalog", fpcs=True).to(DEV)
            x = torch.nn.functional.gelu(2.0 * torch.randn(N, T, 4 * D, generator=g)).to(DEV)
            ws, wz = _minmax_params(lay.weight.data.view(1, D, 4 * D), bits, per=(0, 1))
            _arm(lay.w_quantizer, ws, wz)
            aq = lay.a_quantizer
            aq.shift.data.fill_(0.16997124254703522)
            aq.scale.data.fill_((x.max().item() + 0.17) * 0.9)
            aq.q.fill_(41)
            aq.update_table(41)
            aq.inited = True
            lay._q_host = None                                   # (the layer's host mirror of q, as after load_state_dict)
            lay.calibrated = True
            if cls.endswith("reparamed"):
                lay.reparam_bias()
            close(lay(x), Q.MinMaxQuantLinear.quant_forward(lay, x))
            h = (2.0 * torch.randn(N, T, 4 * D, generator=g)).to(DEV)              # fc1's output: GELU inside the packer
            close(lay.quant_forward(h, pre_gelu=True), Q.MinMaxQuantLinear.quant_forward(lay, torch.nn.functional.gelu(h)))
        elif cls in ("matmul", "postsoftmax"):
            if cls == "matmul":
                lay = Q.AsymmetricallyBatchingQuantMatMul(bits, bits, "quant_forward", head_channel_wise=True, num_heads=H, fpcs=True).to(DEV)
                A = (torch.randn(N, H, T, 64, generator=g) * (0.5 + torch.rand(1, H, 1, 1, generator=g))).to(DEV)
                B = (torch.randn(N, H, T, 64, generator=g) * (0.5 + torch.rand(1, H, 1, 1, generator=g))).to(DEV).transpose(-2, -1)
                _arm(lay.A_quantizer, *_minmax_params(A, bits, per=(1,)))
            else:
                lay = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(bits, bits, "quant_forward", head_channel_wise=True, num_heads=H,
                                                                     fpcs=True, quantizer="adalog").to(DEV)
                A = (torch.randn(N, H, T, T, generator=g) * 3).softmax(-1).to(DEV)
                B = (torch.randn(N, H, T, 64, generator=g) * (0.5 + torch.rand(1, H, 1, 1, generator=g))).to(DEV)
                lay.A_quantizer.q.fill_(29)
                lay.A_quantizer.update_table(29)
                lay._q_host = None
            _arm(lay.B_quantizer, *_minmax_params(B, bits, per=(1,)))
            lay.calibrated = True
            close(lay(A, B), Q.MinMaxQuantMatMul.quant_forward(lay, A, B))
        else:
            lay = Q.AsymmetricallyBatchingQuantConv2d(3, D, 16, 16, mode="quant_forward", w_bit=bits, a_bit=8, fpcs=True).to(DEV)
            x = torch.randn(N, 3, 224, 224, generator=g).to(DEV)
            _arm(lay.w_quantizer, *_minmax_params(lay.weight.data.view(D, -1), bits, per=(0,)))
            lay.a_quantizer.scale.data.fill_(x.abs().max().item() / 127)
            lay.a_quantizer.inited = True
            lay.calibrated = True
            close(lay(x), Q.MinMaxQuantConv2d.quant_forward(lay, x))
