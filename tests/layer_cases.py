"""Layer-level search cases shared by the CPU host-logic tests (stand-in kernels) and the `-m gpu` tests (HIP kernels).

Each case loads a golden fixture captured from the reference (tools/make_golden.py), runs the PRODUCT layer's
hyperparameter_searching() on ``device`` and checks it against the reference's result.
"""
import numpy as np
import pytest
import torch

from adalog_amd import quant_layers as Q

DEV = [torch.device('cpu')]


def t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV[0])


def close(a, b, rtol=2e-4, atol=1e-7):
    torch.testing.assert_close(a.detach().cpu().reshape(b.shape), b.detach().cpu(), rtol=rtol, atol=atol)


def equivalent_uniform(x, s1, z1, s2, z2, bits, tol=1e-3, max_zp_flips=0.1):
    """Two (scale, zero_point) sets are equivalent when they fake-quantise ``x`` to the same tensor within
    tol * max|x|.  Exact score ties are structural in FPCS (neighbouring survivors' grids share their end points, and
    zero points that cause no clamping give identical tensors), and torch.topk's order among ties is unspecified
    (SURVEY A.7), so raw parameters may differ where the quantised tensors do not."""
    from oracle import adalog_oracle as O
    x, s1, z1, s2, z2 = [v.detach().cpu() for v in (x, s1, z1, s2, z2)]
    y1 = O.uniform_fake_quant(x, s1.reshape(s2.shape), z1.reshape(z2.shape), bits)[0]
    y2 = O.uniform_fake_quant(x, s2, z2, bits)[0]
    assert (y1 - y2).abs().max().item() <= tol * x.abs().max().item(), (y1 - y2).abs().max().item()
    assert (z1.reshape(z2.shape) != z2).float().mean().item() <= max_zp_flips


def objective_ok(mine, ref, raw, lo=0.9, hi=1.1):
    """Full-search parity criterion: the reached output MSE equals the reference's within 10 %.

    Individual scoring calls agree with the reference to ~5e-6 on identical candidates (pinned separately), but the last
    FPCS steps compare candidates whose scores differ by less than that, so the *path* (and with it the final
    parameters, most visibly at 6 bit on tiny tensors) is only reproducible up to such near-ties."""
    mine, ref, raw = mine.detach().cpu(), ref.detach().cpu(), raw.detach().cpu()
    m1 = ((mine.reshape(raw.shape) - raw) ** 2).mean().item()
    m0 = ((ref.reshape(raw.shape) - raw) ** 2).mean().item()
    assert lo <= m1 / m0 <= hi, (m1, m0)


def case_linear_search(golden, name, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(name)
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                              eq_n=128, n_V=n_V, fpcs=True, steps=6)
    lay.to(DEV[0])
    lay.weight.data.copy_(t(g["weight"]))
    lay.bias.data.copy_(t(g["bias"]))
    x = t(g["x"])
    with torch.no_grad():
        lay.raw_input, lay.raw_out = x, lay(x)
        close(lay.raw_out, t(g["raw_out"]), 1e-5, 1e-6)
        lay.hyperparameter_searching()
    assert lay.calibrated and not hasattr(lay, "raw_input")
    if wb <= 4:
        W3 = t(g["weight"]).view(n_V, Oc // n_V, I)
        equivalent_uniform(W3, lay.w_quantizer.scale.data, lay.w_quantizer.zero_point.data,
                           t(g["out_w_quantizer__scale"]), t(g["out_w_quantizer__zero_point"]), wb)
        close(lay.a_quantizer.zero_point.data, t(g["out_a_quantizer__zero_point"]), 0, 0)
        close(lay.a_quantizer.scale.data, t(g["out_a_quantizer__scale"]), 1e-3)
    lay.mode = "quant_forward"
    with torch.no_grad():
        objective_ok(lay(x), t(g["qf_out"]), t(g["raw_out"]))


def case_channelwise_reparam(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"linear_cw_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyChannelWiseBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs,
                                                         search_round=3, eq_n=128, n_V=n_V, fpcs=True, steps=6)
    lay.to(DEV[0])
    lay.weight.data.copy_(t(g["weight"]))
    lay.bias.data.copy_(t(g["bias"]))
    ln = torch.nn.LayerNorm(I).to(DEV[0])
    ln.weight.data.copy_(t(g["ln_weight"]))
    ln.bias.data.copy_(t(g["ln_bias"]))
    lay.prev_layer = ln
    assert "prev_layer" not in dict(lay.named_modules()) and lay.prev_layer is ln
    h = t(g["h"])
    with torch.no_grad():
        x = ln(h)
        lay.raw_input, lay.raw_out = x, lay(x)
        lay.hyperparameter_searching()
        equivalent_uniform(x, lay.a_quantizer.scale.data, lay.a_quantizer.zero_point.data, t(g["cw_a_scale"]),
                           t(g["cw_a_zp"]), ab, tol=2e-3, max_zp_flips=1.0)
        # the fold itself is deterministic arithmetic: check it from the reference's channel-wise parameters
        lay.a_quantizer.scale.data.copy_(t(g["cw_a_scale"]))
        lay.a_quantizer.zero_point.data.copy_(t(g["cw_a_zp"]))
        lay.reparam()
    close(ln.weight.data, t(g["reparam_ln_weight"]), 1e-6, 0)
    close(ln.bias.data, t(g["reparam_ln_bias"]), 1e-5, 1e-6)
    close(lay.weight.data, t(g["out_weight"]), 1e-6, 0)
    close(lay.bias.data, t(g["out_bias"]), 1e-4, 1e-5)
    assert lay.a_quantizer.scale.shape == (1,) and not lay.a_quantizer.channel_wise
    lay.mode = "quant_forward"
    with torch.no_grad():
        objective_ok(lay(ln(h)), t(g["qf_out"]), t(g["raw_out"]))


def case_postgelu_search(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"postgelu_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.PostGeluLogBasedBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                                eq_n=128, n_V=1, quantizer="adalog", fpcs=True, steps=6)
    lay.to(DEV[0])
    lay.weight.data.copy_(t(g["weight"]))
    lay.bias.data.copy_(t(g["bias"]))
    assert torch.equal(lay.table, t(g["search_table"]).cpu())
    x = t(g["x"])
    with torch.no_grad():
        lay.raw_input, lay.raw_out = x, lay(x)
        ud, sc = lay.calculate_percentile_activation_candidates()
        close(ud, t(g["cand_ud"]), 0, 0)
        close(sc, t(g["cand_a_scale"]), 1e-6, 0)
        lay.hyperparameter_searching()
    if wb <= 4:
        assert int(lay.a_quantizer.q.item()) == int(g["out_a_quantizer__q"][0])
        close(lay.a_quantizer.scale.data, t(g["out_a_quantizer__scale"]), 1e-3)
        close(lay.a_quantizer.table1, t(g["out_a_quantizer__table1"]), 0, 0)
        close(lay.a_quantizer.table2, t(g["out_a_quantizer__table2"]), 0, 0)
        equivalent_uniform(t(g["weight"]).view(1, Oc, I), lay.w_quantizer.scale.data, lay.w_quantizer.zero_point.data,
                           t(g["out_w_quantizer__scale"]), t(g["out_w_quantizer__zero_point"]), wb)
    lay.mode = "quant_forward"
    with torch.no_grad():
        out0 = lay(x)
        objective_ok(out0, t(g["qf_out"]), t(g["raw_out"]))
        lay.reparam_bias()
        assert bool(lay.a_quantizer.bias_reparamed)
        close(lay(x), out0, 1e-4, 1e-5)                          # re-parameterisation preserves the function
        objective_ok(lay(x), t(g["qf_out_reparamed"]), t(g["raw_out"]))
        b0 = lay.bias.data.clone()
        lay.reparam_bias()                                      # idempotent (linear.py:1000-1001)
        assert torch.equal(b0, lay.bias.data)


def case_matmul_search(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"matmul_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=cbs, search_round=3,
                                              eq_n=128, head_channel_wise=True, num_heads=H, fpcs=True, steps=6)
    A, B = t(g["A"]), t(g["B"])
    lay.to(DEV[0])
    with torch.no_grad():
        lay.raw_input, lay.raw_out = [A, B], lay(A, B)
        lay.hyperparameter_searching()
    if bits <= 4:
        for k in ("A_quantizer", "B_quantizer"):
            sd = dict(lay.state_dict())
            equivalent_uniform(A if k[0] == "A" else B, sd[k + ".scale"], sd[k + ".zero_point"],
                               t(g[f"out_{k}__scale"]), t(g[f"out_{k}__zero_point"]), bits)
    lay.mode = "quant_forward"
    with torch.no_grad():
        out0 = lay(A, B)
        objective_ok(out0, t(g["qf_out"]), t(g["raw_out"]))
        # q@k^T hands over a transposed *view* (wrap_net.py:25): same result without a copy
        Bv = B.transpose(-2, -1).contiguous().transpose(-2, -1)
        close(lay(A, Bv), out0, 0, 0)


def case_matmul_gen_route(device="cpu", bits=4, dims=(8, 2, 12, 16), gen="all"):
    """The attention searches with the candidate operand generated inside the scoring kernel (adalog_gemm_score_gen, layer switch
    quant_layers.matmul.GEN_MM) commit exactly the parameters the packed-operand route commits (K = 16: a shape both take)."""
    import adalog_amd.quant_layers.matmul as MM
    from adalog_amd import backend
    DEV[0] = torch.device(device)
    N, H, S, K = dims
    gen_ = torch.Generator().manual_seed(11)
    A = (torch.randn(N, H, S, K, generator=gen_) * 1.7).to(DEV[0])
    B = (torch.randn(N, H, K, S, generator=gen_) * 0.9 + 0.2).to(DEV[0])
    res, calls = {}, {}
    be = backend.get()
    orig = be.gemm_score_gen
    for mode in ("0", gen):
        old, MM.GEN_MM = MM.GEN_MM, mode
        n = [0]

        def spy(*a, **k):
            n[0] += 1
            return orig(*a, **k)
        be.gemm_score_gen = spy
        try:
            lay = Q.AsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=N, search_round=2,
                                                      eq_n=128, head_channel_wise=True, num_heads=H, fpcs=True, steps=4)
            lay.to(DEV[0])
            with torch.no_grad():
                lay.raw_input, lay.raw_out = [A, B], lay(A, B)
                lay.hyperparameter_searching()
            res[mode] = {k: v.detach().cpu().clone() for k, v in lay.state_dict().items()}
            calls[mode] = n[0]
        finally:
            MM.GEN_MM = old
            be.gemm_score_gen = orig
    assert calls["0"] == 0
    for k in res["0"]:
        assert torch.equal(res["0"][k], res[gen][k]), k
    return calls[gen]


def case_postsoftmax_search(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"postsoftmax_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    lay = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=cbs,
                                                         search_round=3, eq_n=128, head_channel_wise=True, num_heads=H,
                                                         fpcs=True, steps=6, quantizer="adalog")
    assert torch.equal(lay.table, t(g["search_table"]).cpu())
    A, B = t(g["A"]), t(g["B"])
    lay.to(DEV[0])
    with torch.no_grad():
        lay.raw_input, lay.raw_out = [A, B], lay(A, B)
        lay.hyperparameter_searching()
    if bits <= 4:
        assert int(lay.A_quantizer.q.item()) == int(g["out_A_quantizer__q"][0])
        close(lay.A_quantizer.table2, t(g["out_A_quantizer__table2"]), 0, 0)
        equivalent_uniform(B, lay.B_quantizer.scale.data, lay.B_quantizer.zero_point.data,
                           t(g["out_B_quantizer__scale"]), t(g["out_B_quantizer__zero_point"]), bits)
    lay.mode = "quant_forward"
    with torch.no_grad():
        objective_ok(lay(A, B), t(g["qf_out"]), t(g["raw_out"]))


def case_conv_search(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"conv_w{bits}")
    wb, _, N, ic, oc, k, hw, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantConv2d(in_channels=ic, out_channels=oc, kernel_size=(k, k), stride=(k, k),
                                              mode="raw", w_bit=wb, a_bit=8, calib_batch_size=cbs, search_round=3,
                                              eq_n=128, fpcs=True, steps=6)
    lay.to(DEV[0])
    lay.weight.data.copy_(t(g["weight"]))
    lay.bias.data.copy_(t(g["bias"]))
    x = t(g["x"])
    with torch.no_grad():
        lay.raw_input, lay.raw_out = x, lay(x)
        lay.hyperparameter_searching()
    if wb <= 4:
        equivalent_uniform(t(g["weight"]).view(oc, -1), lay.w_quantizer.scale.data, lay.w_quantizer.zero_point.data,
                           t(g["out_w_quantizer__scale"]), t(g["out_w_quantizer__zero_point"]), wb)
    lay.mode = "quant_forward"
    with torch.no_grad():
        objective_ok(lay(x), t(g["qf_out"]), t(g["raw_out"]))


def case_modes_and_errors(device="cpu"):
    DEV[0] = torch.device(device)
    lay = Q.AsymmetricallyBatchingQuantLinear(8, 8, True, "raw", 4, 4, calib_batch_size=2, search_round=1, eq_n=128,
                                              n_V=1, fpcs=True, steps=2)
    lay.to(DEV[0])
    x = torch.randn(2, 3, 8).to(DEV[0])
    lay.mode = "bogus"
    with pytest.raises(NotImplementedError):
        lay(x)
    lay.mode = "quant_forward"
    with pytest.raises(AssertionError):
        lay(x)                                                   # not calibrated (linear.py:47)
    with pytest.raises(AssertionError):
        lay.a_quantizer(x)                                       # quantiser not inited (uniform.py:28)
    with pytest.raises(NotImplementedError):
        Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(quantizer="log2")
    conv = Q.AsymmetricallyBatchingQuantConv2d(3, 4, (3, 3), (1, 1), w_bit=4, a_bit=8, fpcs=True, eq_n=128, steps=2)
    conv.to(DEV[0])
    conv.raw_input, conv.raw_out = torch.randn(2, 3, 8, 8).to(DEV[0]), torch.randn(2, 4, 6, 6).to(DEV[0])
    with pytest.raises(NotImplementedError):
        conv.hyperparameter_searching()                          # overlapping conv is off the accelerated path


# ------------------------------------------------------------------------------------------------ BRECQ (K17)
def case_brecq_toy(golden, device="cpu"):
    """One training-mode forward/backward of a toy block against the reference's own autograd (golden brecq_toy):
    reconstruction + rounding loss, d/d alpha (AdaRound), d/d scale through the uniform and AdaLog STE quantisers."""
    DEV[0] = torch.device(device)
    from adalog_amd.utils.block_recon import BlockReconstructor, LinearTempDecay, LossFunction
    g = golden("brecq_toy")
    assert abs(LossFunction.lp_loss(torch.ones(2, 3, 4), torch.zeros(2, 3, 4)).item() - float(g["lp_ones"])) < 1e-6
    td = LinearTempDecay(1000, rel_start_decay=0.2, start_b=20, end_b=2)
    for tt, bb in zip(g["temp_decay_t"], g["temp_decay_b"]):
        assert abs(td(int(tt)) - float(bb)) < 1e-9
    I, Hd, H = 16, 32, 2

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            kw = dict(mode="raw", w_bit=4, a_bit=4, calib_batch_size=4, search_round=1, eq_n=128, fpcs=True, steps=2)
            self.fc1 = Q.AsymmetricallyBatchingQuantLinear(I, Hd, True, n_V=1, **kw)
            self.fc2 = Q.PostGeluLogBasedBatchingQuantLinear(Hd, I, True, n_V=1, quantizer="adalog", **kw)
            mk = dict(B_bit=4, mode="raw", calib_batch_size=4, search_round=1, eq_n=128, head_channel_wise=True,
                      num_heads=H, fpcs=True, steps=2)
            self.matmul1 = Q.AsymmetricallyBatchingQuantMatMul(A_bit=4, **mk)

        def forward(self, x):
            B, N, C = x.shape
            h = x.reshape(B, N, H, C // H).permute(0, 2, 1, 3)
            a = self.matmul1(h, h.transpose(-2, -1)).softmax(-1) @ h
            x = x + a.permute(0, 2, 1, 3).reshape(B, N, C)
            return x + self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    blk = Blk().eval().to(DEV[0])
    sd = {k[4:].replace("__", "."): t(v) for k, v in g.items() if k.startswith("cal_")}
    res = blk.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, res
    for m in blk.modules():
        if hasattr(m, "mode"):
            m.calibrated = True
            for a in ("a_quantizer", "w_quantizer", "A_quantizer", "B_quantizer"):
                if hasattr(m, a):
                    getattr(m, a).inited = True
    rec = object.__new__(BlockReconstructor)
    rec.wrap_quantizers_in_net(blk, "blk")
    for m in blk.modules():
        if hasattr(m, "training_mode"):
            m.init_training()
        if hasattr(m, "mode"):
            m.mode = "quant_forward"
    close(blk.fc2.w_quantizer.alpha.data, t(g["alpha_fc2"]), 1e-4, 1e-5)        # init_alpha (adaround.py:62-67)
    blk.fc1.w_quantizer.alpha.data.copy_(t(g["alpha_fc1"]))
    blk.fc2.w_quantizer.alpha.data.copy_(t(g["alpha_fc2"]))
    lf = LossFunction(blk, round_loss="relaxation", weight=0.01, max_count=10, rec_loss="mse", b_range=(20, 2),
                      decay_start=0, warmup=0.2, p=2.0)
    lf.count = 4
    x, tgt = t(g["x"]), t(g["tgt"])
    out = blk(x)
    loss = lf(out, tgt)
    loss.backward()
    close(out, t(g["train_out"]), 1e-4, 1e-5)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    close(blk.fc1.w_quantizer.alpha.grad, t(g["g_alpha_fc1"]), 2e-3, 1e-6)
    close(blk.fc2.w_quantizer.alpha.grad, t(g["g_alpha_fc2"]), 2e-3, 1e-6)
    close(blk.fc1.a_quantizer.scale.grad, t(g["g_a_scale_fc1"]), 2e-3, 1e-5)
    close(blk.fc2.a_quantizer.scale.grad, t(g["g_a_scale_fc2"]), 2e-3, 1e-5)
    close(blk.matmul1.A_quantizer.scale.grad, t(g["g_A_scale_mm"]), 2e-3, 1e-5)
    close(blk.matmul1.B_quantizer.scale.grad, t(g["g_B_scale_mm"]), 2e-3, 1e-5)
    close(blk.fc1.w_quantizer.get_hard_value(blk.fc1.weight.data), t(g["hard_fc1"]), 1e-6, 1e-7)


def case_brecq_traj(golden, device="cpu", graph=None, tol=1e-3):
    """The LOOP of utils/block_recon.py:84-137 against the reference's own 20-iteration run (golden brecq_traj, made by
    tools/make_golden.py: gen_brecq_traj): the reference's mini-batch index sequence is injected, and alpha, every trained
    activation scale, the per-iteration loss and b are compared after iterations 1, 5 and 20.  ``graph``: None = the
    loop's own choice, True / False = force the HIP-graph replay on / off (ADALOG_BRECQ_GRAPH)."""
    import os
    DEV[0] = torch.device(device)
    from adalog_amd.utils.block_recon import BlockReconstructor
    g = golden("brecq_traj")
    iters, bs = [int(v) for v in g["cfg"]]
    I, Hd, H = 16, 32, 2

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            kw = dict(mode="raw", w_bit=4, a_bit=4, calib_batch_size=4, search_round=1, eq_n=128, fpcs=True, steps=2)
            self.fc1 = Q.AsymmetricallyBatchingQuantLinear(I, Hd, True, n_V=1, **kw)
            self.fc2 = Q.PostGeluLogBasedBatchingQuantLinear(Hd, I, True, n_V=1, quantizer="adalog", **kw)
            mk = dict(B_bit=4, mode="raw", calib_batch_size=4, search_round=1, eq_n=128, head_channel_wise=True,
                      num_heads=H, fpcs=True, steps=2)
            self.matmul1 = Q.AsymmetricallyBatchingQuantMatMul(A_bit=4, **mk)
            self.matmul2 = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=4, quantizer="adalog", **mk)

        def forward(self, x):
            B, N, C = x.shape
            h = x.reshape(B, N, H, C // H).permute(0, 2, 1, 3)
            a = self.matmul2(self.matmul1(h, h.transpose(-2, -1)).softmax(-1), h)
            x = x + a.permute(0, 2, 1, 3).reshape(B, N, C)
            return x + self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    blk = Blk().eval().to(DEV[0])
    sd = {k[4:].replace("__", "."): t(v) for k, v in g.items() if k.startswith("cal_")}
    res = blk.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not res.missing_keys, res
    for m in blk.modules():
        if hasattr(m, "mode"):
            m.calibrated = True
            for a in ("a_quantizer", "w_quantizer", "A_quantizer", "B_quantizer"):
                if hasattr(m, a):
                    getattr(m, a).inited = True
    blk.raw_input, blk.raw_out = t(g["x"]).clone(), t(g["tgt"]).clone()
    perms = torch.from_numpy(g["perms"])
    rec = object.__new__(BlockReconstructor)
    rec.index_source = lambda it, n, b: perms[it][:b]
    seen = {}

    def names():
        return [("alpha_fc1", blk.fc1.w_quantizer.alpha), ("alpha_fc2", blk.fc2.w_quantizer.alpha),
                ("a_scale_fc1", blk.fc1.a_quantizer.scale), ("a_scale_fc2", blk.fc2.a_quantizer.scale),
                ("A_scale_mm1", blk.matmul1.A_quantizer.scale), ("B_scale_mm1", blk.matmul1.B_quantizer.scale),
                ("A_scale_mm2", blk.matmul2.A_quantizer.scale), ("B_scale_mm2", blk.matmul2.B_quantizer.scale)]
    losses, bvals = [], []

    def hook(it, lf):
        rec_t, rnd_t = lf.cur
        losses.append(float(torch.as_tensor(rec_t).detach()) + float(torch.as_tensor(rnd_t).detach()))
        bvals.append(float(lf.b))
        if it in (1, 5, 20):
            for n_, p_ in names():
                seen[f"it{it:02d}_{n_}"] = p_.detach().clone()
    rec.iter_hook = hook
    prev = os.environ.get("ADALOG_BRECQ_GRAPH")
    if graph is not None:
        os.environ["ADALOG_BRECQ_GRAPH"] = "1" if graph else "0"
    try:
        rec.reconstruct_single_block("blk", blk, DEV[0], batch_size=bs, iters=iters, quant_act=True)
    finally:
        if graph is not None:
            if prev is None:
                os.environ.pop("ADALOG_BRECQ_GRAPH", None)
            else:
                os.environ["ADALOG_BRECQ_GRAPH"] = prev
    assert len(losses) == iters
    worst = {}
    for k, v in seen.items():
        ref = t(g[k])
        err = ((v.reshape(ref.shape) - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
        worst[k] = err
        assert err <= tol, (k, err)
    lref = torch.from_numpy(g["losses"])
    lerr = ((torch.tensor(losses, dtype=torch.float64) - lref).abs() / lref.abs()).max().item()
    assert lerr <= tol, ("loss", lerr, losses, lref)
    assert max(abs(a - b) for a, b in zip(bvals, g["b"].tolist())) < 1e-9
    # hard rounding committed from the trained alpha equals the reference's
    close(blk.fc1.w_quantizer.get_hard_value(blk.fc1.weight.data), t(g["hard_fc1"]), 1e-6, 1e-7)
    close(blk.fc2.w_quantizer.get_hard_value(blk.fc2.weight.data), t(g["hard_fc2"]), 1e-6, 1e-7)
    worst["loss"] = lerr
    return worst


def case_brecq_reconstruct(device="cpu", iters=60):
    """reconstruct_model end to end on a tiny ViT: the training loop runs, the reconstruction loss goes down, hard
    rounding is committed (weights land on the quantisation grid) and the model stays in quant_forward."""
    DEV[0] = torch.device(device)
    import copy
    import importlib.util
    import os
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cfg4b", os.path.join(root, "configs", "4bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = 1, 2
    torch.manual_seed(7)
    model = VisionTransformer(img_size=32, patch_size=8, embed_dim=32, depth=1, num_heads=2, num_classes=10).eval()
    for p_ in model.parameters():
        p_.data.mul_(8.0)
    model.to(DEV[0])
    full = copy.deepcopy(model)
    x = torch.randn(16, 3, 32, 32).to(DEV[0])
    loader = [(x[:8], None), (x[8:], None)]
    model = wrap_modules_in_net(model, cfg, reparam=True)
    QuantCalibrator(model, loader).batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    with torch.no_grad():
        y_fp = full(x)
        e0 = ((model(x) - y_fp) ** 2).mean().item()
    br = BlockReconstructor(model, full, loader)
    assert list(br.blocks) == ["patch_embed", "blocks.0", "head"]
    br.reconstruct_model(quant_act=True, keep_gpu=True, iters=iters)
    with torch.no_grad():
        e1 = ((model(x) - y_fp) ** 2).mean().item()
    # a few dozen iterations cannot beat nearest rounding yet (b decays 20 -> 2 within them); this is a sanity bound
    assert e1 == e1 and e1 < 3.0 * e0, (e0, e1)
    lin = model.blocks[0].mlp.fc1
    assert lin.mode == "quant_forward" and lin.w_quantizer.round_mode == "nearest" and not hasattr(lin.w_quantizer, "alpha")
    w3 = lin.weight.data.view(lin.n_V, lin.crb_rows, lin.in_features)
    grid = w3 / lin.w_quantizer.scale.data
    assert (grid - grid.round()).abs().max().item() < 1e-3            # committed to the integer grid


def case_brecq_converges(device="cuda", iters=20000, images=256):
    """BRECQ as the reference runs it (block_recon.py:84-137: 20 000 Adam iterations of batch 32 per block, rounding
    regulariser after the 20 % warm-up, b: 20 -> 2, hard rounding committed at the end) on the transformer block of a
    depth-1 deit_small at W4A4: the block's reconstruction error against the FP block, over the optimisation images,
    must be LOWER after the reconstruction (learned rounding + tuned activation scales) than before it (nearest rounding,
    calibrated scales).  Returns the numbers for the log."""
    DEV[0] = torch.device(device)
    import copy
    import importlib.util
    import os
    import time
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import create_model
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cfg4c", os.path.join(root, "configs", "4bit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.Config()
    torch.manual_seed(5)
    base = create_model("deit_small", depth=1).eval()
    for p_ in base.parameters():                                   # random-init weights: give the activations some spread
        p_.data.mul_(4.0)
    full = copy.deepcopy(base).to(DEV[0]).eval()
    model = wrap_modules_in_net(base, cfg, reparam=True).to(DEV[0])
    g = torch.Generator().manual_seed(5)
    calib = torch.randn(32, 3, 224, 224, generator=g).to(DEV[0])
    QuantCalibrator(model, [(calib, None)], capture="block").batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    for m in model.modules():
        if hasattr(m, "reparam_bias"):
            m.reparam_bias()
    opt = torch.randn(images, 3, 224, 224, generator=g).to(DEV[0])
    loader = [(opt[i:i + 32], None) for i in range(0, images, 32)]
    rec = BlockReconstructor(model, full, loader)
    name = "blocks.0"
    block, fblock = rec.blocks[name], rec.full_blocks[name]
    rec.init_block_raw_data(block, fblock, name, DEV[0])
    xin, tgt = block.raw_input.clone(), block.raw_out.clone()

    def block_err():
        rec.set_block_mode(block, "quant_forward")
        with torch.no_grad():
            e = sum(((block(xin[i:i + 32]) - tgt[i:i + 32]) ** 2).sum().item() for i in range(0, images, 32)) / tgt.numel()
        rec.set_block_mode(block, "raw")
        return e
    e0 = block_err()
    torch.cuda.synchronize() if DEV[0].type == "cuda" else None
    t0 = time.perf_counter()
    rec.reconstruct_single_block(name, block, DEV[0], quant_act=True, iters=iters)
    torch.cuda.synchronize() if DEV[0].type == "cuda" else None
    dt = time.perf_counter() - t0
    # commit the hard rounding exactly as reconstruct_model does (block_recon.py:151-157)
    from adalog_amd.quantizers.adaround import AdaRoundQuantizer
    for m in block.modules():
        if hasattr(m, "w_quantizer") and isinstance(m.w_quantizer, AdaRoundQuantizer):
            m.weight.data.copy_(m.w_quantizer.get_hard_value(m.weight.data))
            del m.w_quantizer.alpha
            m.w_quantizer.round_mode = "nearest"
    e1 = block_err()
    return {"mse_before": e0, "mse_after": e1, "iters": iters, "seconds": dt, "iters_per_s": iters / dt}
