"""Strict end-state parity cases: the REFERENCE's final parameters (golden ``out_*`` state captured by
tools/make_golden.py after its own hyperparameter_searching) are loaded into the product layer, and the product's
``quant_forward`` output is compared with the reference's golden ``qf_out`` -- for all six layer classes, at 3 / 4 / 6 bit,
both ``bias_reparamed`` states of the post-GELU layer and the channel-wise (LayerNorm-folded) layer.  No search runs
here, so there are no ties and no path divergence: the criterion is the north star's 1e-3 relative on the fp32 fake-quant
output (observed: ~1e-6).  Also: the quantiser goldens fed straight to the product quantiser modules (bins exact).

Shared by the CPU tier (stand-in backend: pins the host logic around the kernels) and the `-m gpu` tier (HIP kernels).
"""
import numpy as np
import torch

from adalog_amd import quant_layers as Q
from adalog_amd import quantizers as QZ

DEV = [torch.device("cpu")]
REL = 1e-3            # north star: fp32 fake-quant tensors within 1e-3 relative


def t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV[0])


def _state(g, prefix="out_"):
    return {k[len(prefix):].replace("__", "."): t(v) for k, v in g.items() if k.startswith(prefix)}


def _mark_inited(lay):
    lay.calibrated = True
    for m in lay.modules():
        if hasattr(m, "inited"):
            m.inited = True


def _rel_ok(got, ref, rel=REL):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double().reshape(got.shape)
    err = (got - ref).abs().max().item()
    bound = rel * ref.abs().max().item()
    assert err <= bound, f"max |diff| {err:.3e} > {rel} * max|ref| = {bound:.3e}"
    return err / max(ref.abs().max().item(), 1e-30)


def case_linear_forward(golden, name, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(name)
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "quant_forward", wb, ab, n_V=n_V, fpcs=True).to(DEV[0])
    lay.load_state_dict(_state(g))
    _mark_inited(lay)
    with torch.no_grad():
        return _rel_ok(lay(t(g["x"])), t(g["qf_out"]))


def case_channelwise_forward(golden, bits, device="cpu"):
    """After reparam the reference's channel-wise layer is a plain per-tensor layer behind the folded LayerNorm
    (linear.py:596-621): golden LayerNorm parameters + out_* state -> qf_out."""
    DEV[0] = torch.device(device)
    g = golden(f"linear_cw_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyChannelWiseBatchingQuantLinear(I, Oc, True, "quant_forward", wb, ab, n_V=n_V, fpcs=True)
    # the reference re-creates the activation parameters as per-tensor in reparam() (linear.py:617-619)
    del lay.a_quantizer.scale, lay.a_quantizer.zero_point
    lay.a_quantizer.channel_wise = False
    lay.a_quantizer.scale = torch.nn.Parameter(torch.zeros(1))
    lay.a_quantizer.zero_point = torch.nn.Parameter(torch.zeros(1))
    lay.to(DEV[0])
    lay.load_state_dict(_state(g))
    _mark_inited(lay)
    ln = torch.nn.LayerNorm(I).to(DEV[0])
    ln.weight.data.copy_(t(g["reparam_ln_weight"]))
    ln.bias.data.copy_(t(g["reparam_ln_bias"]))
    with torch.no_grad():
        return _rel_ok(lay(ln(t(g["h"]))), t(g["qf_out"]))


def case_postgelu_forward(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"postgelu_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    lay = Q.PostGeluLogBasedBatchingQuantLinear(I, Oc, True, "quant_forward", wb, ab, n_V=1, quantizer="adalog",
                                                fpcs=True).to(DEV[0])
    # The fixture's out_* arrays were taken from live state_dict tensors and written after reparam_bias() ran, so `out_bias`
    # and `out_a_quantizer__bias_reparamed` hold the RE-PARAMETERISED state; the searched state is the same dict with
    # the original bias (the post-GELU search never touches it, linear.py:969-997) and the flag cleared.
    sd = _state(g)
    assert bool(sd["a_quantizer.bias_reparamed"]) and torch.equal(sd["bias"].cpu(), t(g["reparamed_bias"]).cpu())
    x = t(g["x"])
    with torch.no_grad():
        lay.load_state_dict(sd)                              # the reference's re-parameterised state (linear.py:999-1006)
        _mark_inited(lay)
        assert bool(lay.a_quantizer.bias_reparamed) and lay.a_quantizer._shift_args()[1] is False
        e1 = _rel_ok(lay(x), t(g["qf_out_reparamed"]))
        sd0 = dict(sd, bias=t(g["bias"]))
        sd0["a_quantizer.bias_reparamed"] = torch.tensor(False)
        lay.load_state_dict(sd0)                             # the searched, not yet re-parameterised state
        assert not bool(lay.a_quantizer.bias_reparamed) and lay.a_quantizer._shift_args()[1] is True
        e0 = _rel_ok(lay(x), t(g["qf_out"]))
        lay.reparam_bias()                                   # the product's own fold lands on the reference's bias
        _rel_ok(lay.bias.data, t(g["reparamed_bias"]), 1e-4)
        e1 = max(e1, _rel_ok(lay(x), t(g["qf_out_reparamed"])))
    return max(e0, e1)


def case_matmul_forward(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"matmul_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="quant_forward", head_channel_wise=True,
                                              num_heads=H, fpcs=True).to(DEV[0])
    lay.load_state_dict(_state(g))
    _mark_inited(lay)
    with torch.no_grad():
        return _rel_ok(lay(t(g["A"]), t(g["B"])), t(g["qf_out"]))


def case_postsoftmax_forward(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"postsoftmax_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    lay = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="quant_forward",
                                                         head_channel_wise=True, num_heads=H, fpcs=True,
                                                         quantizer="adalog").to(DEV[0])
    lay.load_state_dict(_state(g))
    _mark_inited(lay)
    with torch.no_grad():
        return _rel_ok(lay(t(g["A"]), t(g["B"])), t(g["qf_out"]))


def case_conv_forward(golden, bits, device="cpu"):
    DEV[0] = torch.device(device)
    g = golden(f"conv_w{bits}")
    wb, _, N, ic, oc, k, hw, cbs = [int(v) for v in g["cfg"]]
    lay = Q.AsymmetricallyBatchingQuantConv2d(in_channels=ic, out_channels=oc, kernel_size=(k, k), stride=(k, k),
                                              mode="quant_forward", w_bit=wb, a_bit=8, fpcs=True).to(DEV[0])
    lay.load_state_dict(_state(g))
    _mark_inited(lay)
    with torch.no_grad():
        return _rel_ok(lay(t(g["x"])), t(g["qf_out"]))


# ---------------------------------------------------------------------------------------------- quantiser goldens, directly
def _uq(bits, scale, zp, sym=False, cw=True):
    uq = QZ.UniformQuantizer(n_bits=bits, symmetric=sym, channel_wise=cw)
    uq.scale = torch.nn.Parameter(scale.clone())
    if zp is not None:
        uq.zero_point = torch.nn.Parameter(zp.clone())
    uq.inited = True
    return uq.to(DEV[0])


def case_uniform_quantizer_golden(golden, bits, device="cpu"):
    """quantizers_uniform.npz -> the product UniformQuantizer (one kernel): values bit-equal to the reference's, bins equal
    to round(y/s)+round(zp) of the reference's output (uniform.py:29-36)."""
    DEV[0] = torch.device(device)
    g = golden("quantizers_uniform")
    x = t(g[f"u{bits}_pt_x"])
    for tag, xin in (("pt", x), ("pc", x), ("row", t(g[f"u{bits}_row_w"])), ("head", t(g[f"u{bits}_head_a"]))):
        s, z = t(g[f"u{bits}_{tag}_scale"]), t(g[f"u{bits}_{tag}_zp"])
        uq = _uq(bits, s, z)
        y_ref = t(g[f"u{bits}_{tag}_y"])
        with torch.no_grad():
            y = uq(xin)
            assert torch.equal(y.cpu(), y_ref.cpu()), f"u{bits}_{tag}: fake-quant values differ"
            if bits <= 7:
                bins = uq.bins(xin).cpu().to(torch.float32)
                want = torch.round(y_ref.cpu() / s.cpu()) + torch.round(z.cpu())
                assert torch.equal(bins.reshape(want.shape), want), f"u{bits}_{tag}: bins differ"
    uq = _uq(bits, t(g[f"u{bits}_sym_scale"]), None, sym=True, cw=False)
    with torch.no_grad():
        assert torch.equal(uq(x).cpu(), t(g[f"u{bits}_sym_y"]).cpu())


def case_adalog_quantizer_golden(golden, bits, q, device="cpu"):
    """quantizers_adalog.npz -> the product AdaLog / ShiftAdaLog quantisers: LUTs exact, bins exact (k recomputed from the
    reference's output by its own formula), values within 1e-6 relative."""
    DEV[0] = torch.device(device)
    g = golden("quantizers_adalog")
    sm, ge = t(g[f"a{bits}_sm_x"]), t(g[f"a{bits}_ge_x"])
    L2 = 2 ** bits

    def ref_bins(xs, s):
        """logarithm.py:87-96 on CPU ATen: the reference's bin index, 255 where masked"""
        u = (xs.cpu() / s).clamp(1e-15, 1.0)
        k = torch.round(-1 * u.log2() * 37.0 / q)
        return torch.where(k >= L2, torch.full_like(k, 255.0), k.clamp(0, L2 - 1))

    aq = QZ.AdaLogQuantizer(n_bits=bits).to(DEV[0])
    aq.q.data.copy_(torch.tensor([q]))
    aq.update_table(q)
    aq.inited = True
    assert torch.equal(aq.table1.cpu(), t(g[f"a{bits}_q{q}_t1"]).cpu()) and torch.equal(aq.table2.cpu(), t(g[f"a{bits}_q{q}_t2"]).cpu())
    worst = 0.0
    with torch.no_grad():
        for s, key in ((1.0, "sm_y"), (0.83, "sm_y_s083")):
            aq.scale = torch.nn.Parameter(torch.full((1,), s, device=DEV[0]))
            y, y_ref = aq(sm).cpu(), t(g[f"a{bits}_q{q}_{key}"]).cpu()
            torch.testing.assert_close(y, y_ref, rtol=1e-6, atol=1e-30)
            assert torch.equal(aq.bins(sm).cpu().float(), ref_bins(sm, s)), f"a{bits}_q{q}_{key}: bins differ"
            worst = max(worst, ((y - y_ref).abs() / y_ref.abs().clamp_min(1e-30)).max().item())
        aq.init_training()                                  # training form (logarithm.py:88-92): no LUT rounding
        torch.testing.assert_close(aq(sm).cpu(), t(g[f"a{bits}_q{q}_sm_ytrain_s083"]).cpu(), rtol=2e-6, atol=1e-30)
        aq.end_training()
        sq = QZ.ShiftAdaLogQuantizer(n_bits=bits).to(DEV[0])
        sq.scale = torch.nn.Parameter(t(g[f"a{bits}_q{q}_ge_scale"]).clone())
        sq.shift.data.copy_(torch.tensor(0.16997124254703522))
        sq.q.data.copy_(torch.tensor([q]))
        sq.update_table(q)
        sq.inited = True
        # y - shift cancels: the absolute error is that of the un-shifted value (<= 1e-6 * max|y + shift|)
        torch.testing.assert_close(sq(ge).cpu(), t(g[f"a{bits}_q{q}_ge_y"]).cpu(), rtol=1e-6, atol=2e-6)
        s_ge = float(g[f"a{bits}_q{q}_ge_scale"][0])
        assert torch.equal(sq.bins(ge).cpu().float(), ref_bins(ge.cpu() + sq.shift.data.cpu(), s_ge))
        sq.mark_bias_reparamed()
        torch.testing.assert_close(sq(ge).cpu(), t(g[f"a{bits}_q{q}_ge_y_reparamed"]).cpu(), rtol=1e-6, atol=1e-30)
    return worst


def case_adaround_quantizer_golden(golden, bits, device="cpu"):
    """quantizers_adaround.npz -> the product AdaRoundQuantizer: alpha init, hard / soft forward, soft targets, d/d alpha
    (the reference's autograd), get_hard_value."""
    DEV[0] = torch.device(device)
    g = golden("quantizers_adaround")
    w, sc, zp = t(g[f"r{bits}_w"]), t(g[f"r{bits}_scale"]), t(g[f"r{bits}_zp"])
    uq = _uq(bits, sc, zp)
    with torch.no_grad():
        assert torch.equal(uq(w).cpu(), t(g[f"r{bits}_uq_y"]).cpu())
    from adalog_amd.quantizers.adaround import AdaRoundQuantizer
    ar = AdaRoundQuantizer(uq=uq, weight_tensor=w, round_mode="learned_hard_sigmoid").to(DEV[0])
    torch.testing.assert_close(ar.alpha.detach().cpu(), t(g[f"r{bits}_alpha0"]).cpu(), rtol=1e-5, atol=1e-6)
    ar.alpha.data.copy_(t(g[f"r{bits}_alpha0"]))
    with torch.no_grad():
        assert torch.equal(ar(w).cpu(), t(g[f"r{bits}_hard_y"]).cpu())
        ar.soft_targets = True
        torch.testing.assert_close(ar(w).cpu(), t(g[f"r{bits}_soft_y"]).cpu(), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(ar.get_soft_targets().cpu(), t(g[f"r{bits}_soft_targets"]).cpu(), rtol=1e-6, atol=1e-7)
    ar.alpha.data.copy_(t(g[f"r{bits}_alpha1"]))
    y = ar(w)
    (y * y).sum().backward()
    torch.testing.assert_close(y.detach().cpu(), t(g[f"r{bits}_soft_y1"]).cpu(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ar.alpha.grad.cpu(), t(g[f"r{bits}_galpha1"]).cpu(), rtol=1e-5, atol=1e-7)
    with torch.no_grad():
        hv = ar.get_hard_value(w.view(48, 32))
    assert torch.equal(hv.cpu().reshape(48, 32), t(g[f"r{bits}_hardval1"]).cpu().reshape(48, 32))
