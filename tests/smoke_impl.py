"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle."""
import torch


def run_smoke():
    from adalog_amd import backend, quant_layers as Q
    from oracle import adalog_oracle as O
    backend.set_backend(None)
    ops = backend.get()                                   # fails loudly without the HIP library / device
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    N, T, I, Oc, bits = 4, 19, 64, 96, 4
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", bits, bits, calib_batch_size=4, search_round=1,
                                              eq_n=128, n_V=3, fpcs=True, steps=6)
    lay.weight.data.normal_(0, 0.1)
    lay.bias.data.normal_(0, 0.1)
    x = torch.randn(N, T, I) * 1.2 + 0.1
    W, b = lay.weight.data.clone(), lay.bias.data.clone()
    with torch.no_grad():
        ro = lay(x)
    # 1. elementwise kernel: bit-exact bins and values against the oracle
    s, z = torch.tensor([0.21]), torch.tensor([7.0])
    y, bins = ops.uniform_fake_quant(x.to(dev), s.to(dev), z.to(dev), bits, want_bins=True)
    y_ref, q_ref = O.uniform_fake_quant(x, s, z, bits)
    assert torch.equal(y.cpu(), y_ref) and torch.equal(bins.cpu(), q_ref.to(torch.uint8)), "uniform fake-quant mismatch"
    # 2. one full layer search on the GPU vs the oracle's search: same reached objective
    lay.to(dev)
    with torch.no_grad():
        lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
        lay.hyperparameter_searching()
        lay.mode = "quant_forward"
        out = lay(x.to(dev)).cpu()
    p = O.search_linear(W, b, x, ro, bits, bits, n_V=3, rounds=1, batch=4)
    ref = O.linear_quant_forward(x, p, bits, bits, 3)
    m1, m0 = ((out - ro) ** 2).mean().item(), ((ref - ro) ** 2).mean().item()
    # W4A4: the searches land on equivalent parameters (ties apart): the reached objective agrees to 1e-3
    assert abs(m1 / m0 - 1.0) <= 1e-3, f"search objective differs from the oracle: {m1} vs {m0}"
    # 3. the fused post-GELU activation search (quantise-in-loader kernel) against the oracle, small shape
    I2, O2, T2, N2 = 256, 64, 9, 4
    xg = torch.nn.functional.gelu(2.0 * torch.randn(N2, T2, I2))
    W2, b2 = torch.randn(O2, I2) * 0.05, torch.randn(O2) * 0.1
    ro2 = torch.nn.functional.linear(xg, W2, b2)
    pg = Q.PostGeluLogBasedBatchingQuantLinear(I2, O2, True, "raw", bits, bits, calib_batch_size=N2, search_round=1, eq_n=128,
                                               n_V=1, quantizer="adalog", fpcs=True, steps=6).to(dev)
    pg.weight.data.copy_(W2)
    pg.bias.data.copy_(b2)
    pg.raw_input, pg.raw_out = xg.to(dev), ro2.to(dev)
    w3 = W2.view(1, O2, I2)
    scw, zpw = O.weight_candidates(w3, bits)
    pg.w_quantizer.scale.data.copy_(scw[60])
    pg.w_quantizer.zero_point.data.copy_(zpw[60].float())
    pg.w_quantizer.inited = True
    wq = O.uniform_fake_quant(w3, scw[60], zpw[60].float(), bits)[0].view(O2, I2)
    shift = torch.tensor(O.GELU_SHIFT)
    ud, _ = O.postgelu_candidates(xg, shift.item())
    scs = (ud[:, 0:1] + (ud[:, 1:] - ud[:, 0:1]) * torch.tensor([i / 15 for i in range(16)]).view(1, -1)).repeat(1, 8)
    qs = torch.tensor([17, 23, 31, 37, 45, 60, 90, 137]).view(1, -1).repeat_interleave(16, dim=-1)
    ref_j = O.score_postgelu(xg, wq, b2, ro2, scs, qs, shift, bits, O.search_table(bits), N2).reshape(-1, 128).t()
    from adalog_amd.ops import BF16
    with torch.no_grad():
        wp, rowsum = pg._pack_w_fixed(BF16, want_rowsum=True)
        fold = ops.shift_fold(rowsum.view(1, -1), pg.w_quantizer.scale.data.view(1, -1), pg.a_quantizer.shift.data, pg.bias.data).view(-1)
        assert ops.score_act_fused_ok(O2, N2 * T2, I2, wp.shape[-1], 128, bits), "fused activation search not taken"
        got_j = pg._score_scale_logbase(wp, fold, scs.t().contiguous().to(dev), qs.t().float().contiguous().to(dev)).cpu()
    err = ((got_j - ref_j).abs() / ref_j.abs()).max().item()
    assert err <= 1e-4, f"fused activation search differs from the oracle: {err}"
    torch.cuda.synchronize()
    print(f"smoke ok: output MSE hip {m1:.6e} / oracle {m0:.6e}; fused search max rel err {err:.2e}")


if __name__ == "__main__":
    run_smoke()
