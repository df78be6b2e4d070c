"""Pin the CPU oracle against fixtures captured from the reference itself (tools/make_golden.py).

CPU-only: these run in the `-m "not gpu"` tier.
"""
import numpy as np
import pytest
import torch

from oracle import adalog_oracle as O

T = torch.from_numpy


def t(a):
    return torch.from_numpy(np.asarray(a))


def check_trace(g, tr, prefix="trace", rtol=2e-5):
    n = int(g[f"{prefix}_n"])
    assert len(tr.scores) == n, (len(tr.scores), n)
    for i in range(n):
        ref = t(g[f"{prefix}_{i:03d}_scores"])
        got = tr.scores[i].reshape(ref.shape)
        assert int(g[f"{prefix}_{i:03d}_k"]) == tr.ks[i]
        torch.testing.assert_close(got, ref, rtol=rtol, atol=1e-9, msg=lambda m: f"call {i}: {m}")
        ridx = t(g[f"{prefix}_{i:03d}_idx"])
        assert torch.equal(tr.idx[i].reshape(ridx.shape), ridx), f"call {i}: top-k indices differ"


# ------------------------------------------------------------------ quantisers
@pytest.mark.parametrize("bits", [3, 4, 6, 8])
def test_uniform(golden, bits):
    g = golden("quantizers_uniform")
    x = t(g[f"u{bits}_pt_x"])
    y, q = O.uniform_fake_quant(x, t(g[f"u{bits}_pt_scale"]), t(g[f"u{bits}_pt_zp"]), bits)
    assert torch.equal(y, t(g[f"u{bits}_pt_y"]))
    assert q.min() >= 0 and q.max() <= 2 ** bits - 1 and torch.equal(q, q.round())
    y, _ = O.uniform_fake_quant(x, t(g[f"u{bits}_pc_scale"]), t(g[f"u{bits}_pc_zp"]), bits)
    assert torch.equal(y, t(g[f"u{bits}_pc_y"]))
    y, _ = O.uniform_fake_quant(t(g[f"u{bits}_row_w"]), t(g[f"u{bits}_row_scale"]), t(g[f"u{bits}_row_zp"]), bits)
    assert torch.equal(y, t(g[f"u{bits}_row_y"]))
    y, _ = O.uniform_fake_quant(t(g[f"u{bits}_head_a"]), t(g[f"u{bits}_head_scale"]), t(g[f"u{bits}_head_zp"]), bits)
    assert torch.equal(y, t(g[f"u{bits}_head_y"]))
    y, _ = O.uniform_fake_quant(x, t(g[f"u{bits}_sym_scale"]), None, bits, sym=True)
    assert torch.equal(y, t(g[f"u{bits}_sym_y"]))
    assert torch.equal(y, t(g[f"u{bits}_sym_train_y"]))          # STE form has the same forward value
    L = O.n_levels(bits)                                          # STE: d/dx = 1 inside the clamp range, 0 outside
    xi = torch.round(x / t(g[f"u{bits}_sym_scale"]))
    assert torch.equal(t(g[f"u{bits}_sym_train_gx"]), ((xi >= -L) & (xi <= L - 1)).float())


def test_uniform_passthrough(golden):
    g = golden("quantizers_uniform")
    y, q = O.uniform_fake_quant(t(g["u32_x"]), None, None, 32)
    assert torch.equal(y, t(g["u32_y"])) and q is None


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("q", [10, 23, 37, 53, 90, 137])
def test_adalog(golden, bits, q):
    g = golden("quantizers_adalog")
    t1, t2 = O.adalog_tables(q, bits)
    assert torch.equal(t1, t(g[f"a{bits}_q{q}_t1"])) and torch.equal(t2, t(g[f"a{bits}_q{q}_t2"]))
    num = t2 * (4 * O.n_levels(bits) - 2)                       # integer numerators
    assert torch.allclose(num, num.round(), atol=1e-4)
    sm, ge = t(g[f"a{bits}_sm_x"]), t(g[f"a{bits}_ge_x"])
    y, k, m = O.adalog_fake_quant(sm, torch.ones(1, 1, 1, 1), q, bits)
    assert torch.equal(y, t(g[f"a{bits}_q{q}_sm_y"]))
    y, _, _ = O.adalog_fake_quant(sm, torch.tensor([0.83]), q, bits)
    assert torch.equal(y, t(g[f"a{bits}_q{q}_sm_y_s083"]))
    y, _, _ = O.adalog_fake_quant_train(sm, torch.tensor([0.83]), q, bits)
    assert torch.equal(y, t(g[f"a{bits}_q{q}_sm_ytrain_s083"]))
    sc = t(g[f"a{bits}_q{q}_ge_scale"])
    sh = torch.tensor(O.GELU_SHIFT)
    y, _, _ = O.shift_adalog_fake_quant(ge, sc, q, bits, sh, False)
    assert torch.equal(y, t(g[f"a{bits}_q{q}_ge_y"]))
    y, _, _ = O.shift_adalog_fake_quant(ge, sc, q, bits, sh, True)
    assert torch.equal(y, t(g[f"a{bits}_q{q}_ge_y_reparamed"]))


@pytest.mark.parametrize("bits", [3, 4])
def test_adaround(golden, bits):
    g = golden("quantizers_adaround")
    w, sc, zp = t(g[f"r{bits}_w"]), t(g[f"r{bits}_scale"]), t(g[f"r{bits}_zp"])
    a0 = O.adaround_init_alpha(w, sc)
    assert torch.equal(a0, t(g[f"r{bits}_alpha0"]))
    assert torch.equal(O.adaround_fake_quant(w, sc, zp, a0, bits, soft=False), t(g[f"r{bits}_hard_y"]))
    assert torch.equal(O.adaround_fake_quant(w, sc, zp, a0, bits, soft=True), t(g[f"r{bits}_soft_y"]))
    assert torch.equal(O.adaround_soft_targets(a0), t(g[f"r{bits}_soft_targets"]))
    # at init the hard value equals plain uniform rounding (SURVEY 8c probe)
    assert torch.equal(t(g[f"r{bits}_hard_y"]), t(g[f"r{bits}_uq_y"]))
    a1 = t(g[f"r{bits}_alpha1"]).clone().requires_grad_(True)
    y = O.adaround_fake_quant(w, sc, zp, a1, bits, soft=True)
    assert torch.equal(y.detach(), t(g[f"r{bits}_soft_y1"]))
    (y * y).sum().backward()
    torch.testing.assert_close(a1.grad, t(g[f"r{bits}_galpha1"]), rtol=1e-6, atol=1e-9)
    hv = O.adaround_hard_value(w.view(48, 32), sc, a1.detach())
    assert torch.equal(hv, t(g[f"r{bits}_hardval1"]))


# ------------------------------------------------------------------ layers
@pytest.mark.parametrize("name", ["linear_w3a3", "linear_w4a4", "linear_w6a6", "linear_w4a4_ragged"])
def test_linear(golden, name):
    g = golden(name)
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    ws, wz = O.weight_candidates(W.view(n_V, Oc // n_V, I), wb)
    assert torch.equal(ws, t(g["cand_w_scale"])) and torch.equal(wz, t(g["cand_w_zp"]))
    as_, az = O.activation_candidates(x, ab, False)
    assert torch.equal(as_, t(g["cand_a_scale"])) and torch.equal(az, t(g["cand_a_zp"]))
    tr = O.Trace()
    p = O.search_linear(W, b, x, ro, wb, ab, n_V=n_V, batch=cbs, trace=tr)
    check_trace(g, tr)
    assert torch.equal(p.w_scale, t(g["out_w_quantizer__scale"]))
    assert torch.equal(p.w_zp, t(g["out_w_quantizer__zero_point"]))
    assert torch.equal(p.a_scale, t(g["out_a_quantizer__scale"]))
    assert torch.equal(p.a_zp, t(g["out_a_quantizer__zero_point"]))
    torch.testing.assert_close(O.linear_quant_forward(x, p, wb, ab, n_V), t(g["qf_out"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_linear_channelwise_reparam(golden, bits):
    g = golden(f"linear_cw_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro, h = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"]), t(g["h"])
    as_, az = O.activation_candidates(x, ab, True)
    assert torch.equal(as_, t(g["cand_a_scale"])) and torch.equal(az, t(g["cand_a_zp"]))
    tr = O.Trace()
    s, z = O.search_linear_channelwise(x, ab, batch=cbs, trace=tr)
    check_trace(g, tr, "cwtrace")
    assert torch.equal(s, t(g["cw_a_scale"])) and torch.equal(z, t(g["cw_a_zp"]))
    r, bb, ts, tz, lw, lb, W2, b2 = O.reparam_step1(s, z, t(g["ln_weight"]), t(g["ln_bias"]), W, b)
    assert torch.equal(lw, t(g["reparam_ln_weight"])) and torch.equal(lb, t(g["reparam_ln_bias"]))
    assert torch.equal(W2, t(g["out_weight"])) and torch.equal(b2, t(g["out_bias"]))
    x2 = x / r - bb                                               # linear.py:616
    tr = O.Trace()
    p = O.search_linear(W2, b2, x2, ro, wb, ab, n_V=n_V, batch=cbs, trace=tr)
    check_trace(g, tr)
    assert torch.equal(p.w_scale, t(g["out_w_quantizer__scale"]))
    assert torch.equal(p.a_scale, t(g["out_a_quantizer__scale"]))
    assert torch.equal(p.a_zp, t(g["out_a_quantizer__zero_point"]))
    # the LayerNorm fold preserves the FP function
    ln_x2 = torch.nn.functional.layer_norm(h, (I,), lw, lb)
    torch.testing.assert_close(torch.nn.functional.linear(ln_x2, W2, b2), ro, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postgelu(golden, bits):
    g = golden(f"postgelu_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    assert torch.equal(O.search_table(ab), t(g["search_table"]))
    pp = O.positive_percentile(x.view(-1), torch.tensor([0.9, 1.0, 0.5, 0.013]))
    assert torch.equal(pp, t(g["pospct"]))
    ud, sc = O.postgelu_candidates(x, torch.tensor(O.GELU_SHIFT).item())
    assert torch.equal(ud, t(g["cand_ud"])) and torch.equal(sc, t(g["cand_a_scale"]))
    tr = O.Trace()
    p = O.search_postgelu(W, b, x, ro, wb, ab, batch=cbs, trace=tr)
    check_trace(g, tr)
    assert torch.equal(p.a_scale, t(g["out_a_quantizer__scale"]))
    assert p.a_q == int(g["out_a_quantizer__q"][0])
    assert torch.equal(p.w_scale, t(g["out_w_quantizer__scale"]))
    assert torch.equal(p.w_zp, t(g["out_w_quantizer__zero_point"]))
    t1, t2 = O.adalog_tables(p.a_q, ab)
    assert torch.equal(t1, t(g["out_a_quantizer__table1"])) and torch.equal(t2, t(g["out_a_quantizer__table2"]))
    torch.testing.assert_close(O.postgelu_quant_forward(x, p, wb, ab), t(g["qf_out"]), rtol=1e-5, atol=1e-6)
    wq = O._w_fq(W.view(1, Oc, I), p, wb).view(Oc, I)
    nb = O.reparam_bias(wq, b)
    torch.testing.assert_close(nb, t(g["reparamed_bias"]), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(O.postgelu_quant_forward(x, p, wb, ab, True, nb), t(g["qf_out_reparamed"]),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_matmul(golden, bits):
    g = golden(f"matmul_a{bits}b{bits}")
    A, B, ro = t(g["A"]), t(g["B"]), t(g["raw_out"])
    sA, zA = O.matmul_candidates(A, bits)
    assert torch.equal(sA, t(g["cand_A_scale"])) and torch.equal(zA, t(g["cand_A_zp"]))
    sB, zB = O.matmul_candidates(B, bits)
    assert torch.equal(sB, t(g["cand_B_scale"])) and torch.equal(zB, t(g["cand_B_zp"]))
    cbs = int(g["cfg"][-1])
    tr = O.Trace()
    p = O.search_matmul(A, B, ro, bits, bits, batch=cbs, trace=tr)
    check_trace(g, tr)
    for k, v in (("A_quantizer__scale", p.A_scale), ("A_quantizer__zero_point", p.A_zp),
                 ("B_quantizer__scale", p.B_scale), ("B_quantizer__zero_point", p.B_zp)):
        assert torch.equal(v, t(g["out_" + k])), k
    torch.testing.assert_close(O.matmul_quant_forward(A, B, p, bits, bits), t(g["qf_out"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_postsoftmax(golden, bits):
    g = golden(f"postsoftmax_a{bits}b{bits}")
    A, B, ro = t(g["A"]), t(g["B"]), t(g["raw_out"])
    cbs = int(g["cfg"][-1])
    tr = O.Trace()
    p = O.search_postsoftmax(A, B, ro, bits, bits, batch=cbs, trace=tr)
    check_trace(g, tr)
    assert p.A_q == int(g["out_A_quantizer__q"][0])
    assert torch.equal(p.B_scale, t(g["out_B_quantizer__scale"]))
    assert torch.equal(p.B_zp, t(g["out_B_quantizer__zero_point"]))
    torch.testing.assert_close(O.matmul_quant_forward(A, B, p, bits, bits, True), t(g["qf_out"]),
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_conv(golden, bits):
    g = golden(f"conv_w{bits}")
    wb, _, N, ic, oc, k, hw, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    ws, wz = O.weight_candidates(W.view(oc, -1), wb, conv=True)
    assert torch.equal(ws, t(g["cand_w_scale"])) and torch.equal(wz, t(g["cand_w_zp"]))
    tr = O.Trace()
    s, z = O.search_conv(W, b, x, ro, wb, (k, k), batch=cbs, trace=tr)
    check_trace(g, tr)
    assert torch.equal(s, t(g["out_w_quantizer__scale"])) and torch.equal(z, t(g["out_w_quantizer__zero_point"]))
    torch.testing.assert_close(O.conv_quant_forward(x, W, b, s, z, wb, (k, k)), t(g["qf_out"]), rtol=1e-5, atol=1e-6)


def test_quantile_above_2pow24(golden):
    """Per-tensor candidates when numel > 2**24: rows double until torch.quantile accepts (linear.py:465-471)."""
    g = golden("quantile_large")
    N, Tn, I = [int(v) for v in g["shape"]]
    x = torch.randn(N, Tn, I, generator=torch.Generator().manual_seed(int(g["seed"])))
    assert torch.equal(x.view(-1)[:64], t(g["x_head"]))
    as_, az = O.activation_candidates(x, 4, False)
    assert torch.equal(as_, t(g["cand_a_scale"])) and torch.equal(az, t(g["cand_a_zp"]))
