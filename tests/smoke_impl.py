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
    assert 0.9 <= m1 / m0 <= 1.1, f"search objective differs from the oracle: {m1} vs {m0}"
    torch.cuda.synchronize()
    print(f"smoke ok: output MSE hip {m1:.6e} / oracle {m0:.6e}")


if __name__ == "__main__":
    run_smoke()
