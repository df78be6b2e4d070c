"""`-m gpu` parity tests: every HIP kernel, called through the C ABI, against its CPU specification / the oracle.

Bars (north star): integer bin indices and packed integer operands bit-exact; fp32 fake-quant tensors and scores within
1e-3 relative (observed: bit-exact for the elementwise kernels, ~1e-6 for scores).
"""
import numpy as np
import pytest
import torch

from oracle import adalog_oracle as O
from tests import cpu_backend as CB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from adalog_amd import backend
    backend.set_backend(None)
    return backend.get()          # raises loudly if the HIP library or the device is missing


DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_err(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("bits", [3, 4, 6, 8])
@pytest.mark.parametrize("layout", ["tensor", "channel", "rows", "heads", "ragged"])
def test_uniform_fake_quant(ops, bits, layout):
    L = 2 ** (bits - 1)
    gen = g(100 + bits)
    if layout == "tensor":
        x = torch.randn(7, 197, 96, generator=gen) * 2
        s = torch.tensor([0.11]); z = torch.tensor([L - 0.3])
    elif layout == "channel":
        x = torch.randn(5, 33, 96, generator=gen) * torch.linspace(0.2, 3, 96)
        s = torch.rand(96, generator=gen) * 0.3 + 0.02; z = torch.randint(0, 2 * L, (96,), generator=gen).float()
    elif layout == "rows":
        x = torch.randn(3, 64, 384, generator=gen) * 0.2
        s = torch.rand(3, 64, 1, generator=gen) * 0.05 + 0.005; z = torch.randint(0, 2 * L, (3, 64, 1), generator=gen).float()
    elif layout == "heads":
        x = torch.randn(4, 6, 50, 64, generator=gen)
        s = torch.rand(1, 6, 1, 1, generator=gen) * 0.3 + 0.05; z = torch.randint(0, 2 * L, (1, 6, 1, 1), generator=gen).float()
    else:   # odd sizes: scalar (non-vectorised) path, empty tail handling
        x = torch.randn(3, 7, 13, generator=gen)
        s = torch.rand(3, 7, 1, generator=gen) * 0.3 + 0.05; z = torch.randint(0, 2 * L, (3, 7, 1), generator=gen).float()
    y_ref, q_ref = O.uniform_fake_quant(x, s, z, bits)
    y, bins = ops.uniform_fake_quant(x.to(DEV), s.to(DEV), z.to(DEV), bits, want_bins=True)
    assert torch.equal(bins.cpu(), q_ref.to(torch.uint8)), "integer bins must be exact"
    assert torch.equal(y.cpu(), y_ref), "fp32 fake-quant is the same IEEE op sequence: expected bit-exact"
    ys = ops.uniform_fake_quant(x.to(DEV), s.abs().to(DEV) * 4, None, bits, sym=True)
    assert torch.equal(ys.cpu(), O.uniform_fake_quant(x, s.abs() * 4, None, bits, sym=True)[0])


def test_uniform_empty_and_errors(ops):
    from adalog_amd._lib import AdalogHipError
    x = torch.empty(0, 8, device=DEV)
    assert ops.uniform_fake_quant(x, torch.ones(1, device=DEV), torch.zeros(1, device=DEV), 4).numel() == 0
    with pytest.raises(AdalogHipError):
        ops.uniform_fake_quant(torch.randn(4, 4), torch.ones(1), torch.zeros(1), 4)      # CPU tensors: no fallback
    with pytest.raises(AdalogHipError):
        ops.uniform_fake_quant(torch.randn(4, 4, device=DEV), torch.ones(1, device=DEV), torch.zeros(1, device=DEV), 9)


# ------------------------------------------------------------------------------------------------ K2/K3
@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("q", [10, 37, 53, 137])
def test_adalog_fake_quant(ops, bits, q):
    gen = g(200 + bits + q)
    sm = torch.softmax(4 * torch.randn(8, 6, 197, 197, generator=gen), dim=-1)          # 1.86 M elements
    t1, t2 = O.adalog_tables(q, bits)
    qd = torch.tensor([q]).to(DEV)
    for scale in (torch.ones(1, 1, 1, 1), torch.tensor([0.83])):
        y_ref, k_ref, m_ref = O.adalog_fake_quant(sm, scale, q, bits, (t1, t2))
        y, bins = ops.log_fake_quant(sm.to(DEV), scale.to(DEV), qd, t1.to(DEV), t2.to(DEV), bits, want_bins=True)
        b_ref = torch.where(m_ref, k_ref, torch.full_like(k_ref, 255.0)).to(torch.uint8)
        flips = (bins.cpu() != b_ref).sum().item()
        assert flips == 0, f"{flips} bin flips of {sm.numel()} (expected ~1e-9 per element, see DESIGN.md)"
        assert rel_err(y.cpu(), y_ref) <= 1e-6
    # shifted post-GELU form, both re-parameterisation states, and the training (no-LUT) form
    ge = torch.nn.functional.gelu(2 * torch.randn(16, 197, 384, generator=gen))
    sc = torch.tensor([ge.max().item() * 0.9 + 0.17]); sh = torch.tensor([O.GELU_SHIFT])
    for reparamed in (False, True):
        y_ref, k_ref, m_ref = O.shift_adalog_fake_quant(ge, sc, q, bits, sh, reparamed, (t1, t2))
        y, bins = ops.log_fake_quant(ge.to(DEV), sc.to(DEV), qd, t1.to(DEV), t2.to(DEV), bits, shift=sh.to(DEV),
                                     sub_shift=not reparamed, want_bins=True)
        b_ref = torch.where(m_ref, k_ref, torch.full_like(k_ref, 255.0)).to(torch.uint8)
        assert (bins.cpu() != b_ref).sum().item() == 0
        assert (y.cpu() - y_ref).abs().max().item() <= 1e-6 * sc.item()
    y_ref, _, _ = O.adalog_fake_quant_train(sm, torch.tensor([0.83]), q, bits)
    y = ops.log_fake_quant(sm.to(DEV), torch.tensor([0.83]).to(DEV), qd, None, None, bits, train_form=True)
    assert rel_err(y.cpu(), y_ref) <= 1e-5


def test_adalog_zero_and_tiny_inputs(ops):
    """x = 0 / denormal / above the scale: clamp to [1e-15, 1] then mask, exactly as logarithm.py:87-98."""
    x = torch.tensor([0.0, 1e-45, 1e-30, 1e-16, 1e-15, 0.5, 1.0, 2.0, -1.0, 0.999999, 2 ** -15.5])
    for bits in (3, 4, 6):
        t1, t2 = O.adalog_tables(37, bits)
        y_ref, k_ref, m_ref = O.adalog_fake_quant(x, torch.ones(1), 37, bits, (t1, t2))
        y, bins = ops.log_fake_quant(x.to(DEV), torch.ones(1, device=DEV), torch.tensor([37]).to(DEV), t1.to(DEV),
                                     t2.to(DEV), bits, want_bins=True)
        assert torch.equal(y.cpu(), y_ref)
        assert torch.equal(bins.cpu(), torch.where(m_ref, k_ref, torch.full_like(k_ref, 255.0)).to(torch.uint8))


# ------------------------------------------------------------------------------------------------ operand packing
@pytest.mark.parametrize("bits", [3, 4, 6])
def test_pack_uniform(ops, bits):
    gen = g(300 + bits)
    L = 2 ** (bits - 1)
    # activation candidates (per-tensor, C = 16), K not a multiple of 64
    x3 = torch.randn(1, 500, 197, generator=gen)
    sc = torch.rand(16, 1, generator=gen) * 0.2 + 0.05; zp = torch.randint(L - 4, L + 4, (16, 1), generator=gen).float()
    ref = CB.pack_uniform(x3, sc, zp, 16, 1, 1, 0, 0, bits, CB.I8)
    out = ops.pack_uniform(x3.to(DEV), sc.to(DEV), zp.to(DEV), 16, 1, 1, 0, 0, bits, ops.I8)
    assert torch.equal(out.cpu(), ref)
    if bits <= 4:                                  # fp8 (e4m3) storage of q - z: exact for <= 4-bit operands
        ref8 = CB.pack_uniform(x3, sc, zp, 16, 1, 1, 0, 0, bits, CB.FP8)
        out8 = ops.pack_uniform(x3.to(DEV), sc.to(DEV), zp.to(DEV), 16, 1, 1, 0, 0, bits, ops.FP8)
        assert torch.equal(out8.cpu().float(), ref8.float()) and torch.equal(out8.cpu().float(), ref.float())
        w8 = torch.randn(1, 96, 384, generator=gen) * 0.1
        s8 = torch.rand(8, 96, generator=gen) * 0.02 + 0.005; z8 = torch.randint(0, 2 * L, (8, 96), generator=gen).float()
        r8, rs8 = CB.pack_uniform(w8, s8, z8, 8, 96, 1, 0, 1, bits, CB.FP8, want_rowsum=True, c_inner=True)
        o8, ro8 = ops.pack_uniform(w8.to(DEV), s8.to(DEV), z8.to(DEV), 8, 96, 1, 0, 1, bits, ops.FP8, want_rowsum=True,
                                   c_inner=True)
        assert torch.equal(o8.cpu().float(), r8.float()) and torch.equal(ro8.cpu(), rs8)
    # per-row weight candidates with rowsum, bf16 and fp32 outputs
    w3 = torch.randn(1, 96, 384, generator=gen) * 0.1
    sc = torch.rand(8, 96, generator=gen) * 0.02 + 0.005; zp = torch.randint(0, 2 * L, (8, 96), generator=gen).float()
    for dt_c, dt_o in ((CB.I8, ops.I8), (CB.BF16, ops.BF16), (CB.F32, ops.F32)):
        ref, rs = CB.pack_uniform(w3, sc, zp, 8, 96, 1, 0, 1, bits, dt_c, want_rowsum=True)
        out, rso = ops.pack_uniform(w3.to(DEV), sc.to(DEV), zp.to(DEV), 8, 96, 1, 0, 1, bits, dt_o, want_rowsum=True)
        assert torch.equal(out.cpu().float(), ref.float()) and torch.equal(rso.cpu(), rs)
    # per-head candidates on a transposed view (B^T of softmax@v: row stride 1)
    B = torch.randn(3, 4, 49, 32, generator=gen)                               # [N,H,S,C]
    bt3 = B.transpose(-2, -1).reshape(-1, 32, 49)
    sc = torch.rand(8, 4, generator=gen) * 0.2 + 0.05; zp = torch.randint(L - 3, L + 3, (8, 4), generator=gen).float()
    ref = CB.pack_uniform(bt3, sc, zp, 8, 4, 4, 1, 0, bits, CB.I8)
    Bd = B.to(DEV)
    out = ops.pack_uniform(Bd.transpose(-2, -1).reshape(-1, 32, 49), sc.to(DEV), zp.to(DEV), 8, 4, 4, 1, 0, bits, ops.I8)
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_pack_adalog(ops, bits):
    gen = g(400 + bits)
    L = 2 ** (bits - 1)
    table = O.search_table(bits)
    mant = torch.round(table[:37] * (4 * L - 2))
    # post-softmax: no clamp, scale 1, 16 candidate bases
    A3 = torch.softmax(4 * torch.randn(6, 50, 50, generator=gen), -1)
    A3[0, 0, :5] = torch.tensor([0.0, 1e-40, 1e-20, 1.0, 0.5])
    qv = torch.tensor([10., 11, 23, 36, 37, 38, 53, 64, 77, 90, 100, 111, 120, 130, 136, 137])
    ones = torch.ones(16)
    ref = CB.pack_adalog(A3, ones, qv, 16, 1, 1, 0, bits, mant, None, False)
    out = ops.pack_adalog(A3.to(DEV), ones.to(DEV), qv.to(DEV), 16, 1, 1, 0, bits, mant.to(DEV), None, False)
    bad = (out.cpu().float() != ref.float()).sum().item()
    assert bad == 0, f"{bad} of {ref.numel()} AdaLog operands differ"
    # post-GELU: shift + clamp, per-candidate (scale, base)
    x3 = torch.nn.functional.gelu(2 * torch.randn(1, 300, 200, generator=gen))
    sc = torch.rand(16, generator=gen) * 2 + 1.0
    sh = torch.tensor([O.GELU_SHIFT])
    ref = CB.pack_adalog(x3, sc, qv, 16, 1, 1, 0, bits, mant, sh, True)
    out = ops.pack_adalog(x3.to(DEV), sc.to(DEV), qv.to(DEV), 16, 1, 1, 0, bits, mant.to(DEV), sh.to(DEV), True)
    assert (out.cpu().float() != ref.float()).sum().item() == 0
    # the packed operand times s/(4L-2) reproduces the reference's search-time value (linear.py:830-837)
    v_ref = O.adalog_search_value(-((x3.unsqueeze(-1) + sh) / sc).clamp(1e-15, 1.0).log2(), qv, bits, table)
    got = out.cpu().float()[..., :200].permute(1, 2, 3, 0) / (4 * L - 2)        # [G,R,K,C]
    torch.testing.assert_close(got, v_ref, rtol=1e-6, atol=0)


def test_pack_adalog_fast_vs_generic(ops, monkeypatch):
    """The LDS-parameter AdaLog packer (narrow tie zone, clamped LUT index) against the generic kernel on 1.3e8
    element-candidates, and against the CPU specification on the no-clamp K % 4 == 0 form."""
    gen = g(431)
    bits, P = 4, 128
    table = O.search_table(bits)
    mant = torch.round(table[:37] * (4 * 2 ** (bits - 1) - 2)).to(DEV)
    x3 = torch.nn.functional.gelu(2 * torch.randn(1, 2000, 512, generator=gen)).to(DEV)
    x3[0, 0, :4] = torch.tensor([0.0, -0.1699, 1e-30, 5.0])
    sc = (torch.rand(P, generator=gen) * 2 + 0.5).to(DEV)
    qv = torch.randint(10, 138, (P,), generator=gen).float().to(DEV)
    sh = torch.tensor([O.GELU_SHIFT]).to(DEV)
    for clamp, shift in ((True, sh), (False, None)):
        src = x3 if clamp else torch.softmax(x3 * 3, -1)
        for c_inner in (True, False):
            monkeypatch.delenv("ADALOG_PACK_GENERIC", raising=False)
            fast = ops.pack_adalog(src, sc if clamp else torch.ones_like(sc), qv, P, 1, 1, 0, bits, mant, shift, clamp,
                                   c_inner=c_inner)
            monkeypatch.setenv("ADALOG_PACK_GENERIC", "1")
            slow = ops.pack_adalog(src, sc if clamp else torch.ones_like(sc), qv, P, 1, 1, 0, bits, mant, shift, clamp,
                                   c_inner=c_inner)
            monkeypatch.delenv("ADALOG_PACK_GENERIC", raising=False)
            assert torch.equal(fast.view(torch.int16), slow.view(torch.int16)), (clamp, c_inner)
    # uniform int8, per-tensor candidates: LDS-parameter kernel vs the generic one (6.6e7 element-candidates, ragged K)
    xa = torch.randn(2, 1000, 257, generator=gen).to(DEV) * 2
    su = (torch.rand(P, 1, generator=gen) * 0.3 + 0.02).to(DEV)
    zu = torch.randint(0, 16, (P, 1), generator=gen).float().to(DEV)
    for c_inner in (True, False):
        fast = ops.pack_uniform(xa, su, zu, P, 1, 1, 0, 0, 4, ops.I8, c_inner=c_inner)
        monkeypatch.setenv("ADALOG_PACK_GENERIC", "1")
        slow = ops.pack_uniform(xa, su, zu, P, 1, 1, 0, 0, 4, ops.I8, c_inner=c_inner)
        monkeypatch.delenv("ADALOG_PACK_GENERIC", raising=False)
        assert torch.equal(fast, slow), c_inner
    # uniform, per-ROW candidates (weight searches): LDS-table kernel (k_pack_uniform_tab) vs the generic one, every output type,
    # ragged K (rows of 1530 and 257 elements), with and without the integer row sums
    for K, rows in ((1530, 96), (257, 130)):
        wa = (torch.randn(1, rows, K, generator=gen) * 0.1).to(DEV)
        sw = (torch.rand(P, rows, generator=gen) * 0.02 + 0.004).to(DEV)
        zw = torch.randint(0, 16, (P, rows), generator=gen).float().to(DEV)
        for dt in (ops.I8, ops.FP8, ops.BF16, ops.F32):
            for c_inner in (True, False):
                fast, rs_f = ops.pack_uniform(wa, sw, zw, P, rows, 1, 0, 1, 4, dt, want_rowsum=True, c_inner=c_inner)
                monkeypatch.setenv("ADALOG_PACK_GENERIC", "1")
                slow, rs_s = ops.pack_uniform(wa, sw, zw, P, rows, 1, 0, 1, 4, dt, want_rowsum=True, c_inner=c_inner)
                monkeypatch.delenv("ADALOG_PACK_GENERIC", raising=False)
                assert torch.equal(fast.view(torch.uint8), slow.view(torch.uint8)) and torch.equal(rs_f, rs_s), (K, dt, c_inner)
    # per-head candidates (attention q / k operands): grid.z walks the groups; int8 and fp8, 64-byte rows, ragged K
    xh = torch.randn(24, 197, 61, generator=gen).to(DEV) * 2
    sh_ = (torch.rand(P, 6, generator=gen) * 0.3 + 0.05).to(DEV)
    zh = torch.randint(0, 16, (P, 6), generator=gen).float().to(DEV)
    for dt in (ops.I8, ops.FP8):
        for c_inner in (True, False):
            fast = ops.pack_uniform(xh, sh_, zh, P, 6, 6, 1, 0, 4, dt, c_inner=c_inner, k_align=64)
            monkeypatch.setenv("ADALOG_PACK_GENERIC", "1")
            slow = ops.pack_uniform(xh, sh_, zh, P, 6, 6, 1, 0, 4, dt, c_inner=c_inner, k_align=64)
            monkeypatch.delenv("ADALOG_PACK_GENERIC", raising=False)
            assert torch.equal(fast.float(), slow.float()), (dt, c_inner)       # (values: fp8 has a -0 encoding)
    A3 = torch.softmax(4 * torch.randn(6, 50, 52, generator=gen), -1)
    A3[0, 0, :5] = torch.tensor([0.0, 1e-40, 1e-20, 1.0, 0.5])
    q16 = torch.tensor([10., 11, 23, 36, 37, 38, 53, 64, 77, 90, 100, 111, 120, 130, 136, 137])
    ref = CB.pack_adalog(A3, torch.ones(16), q16, 16, 1, 1, 0, bits, mant.cpu(), None, False)
    out = ops.pack_adalog(A3.to(DEV), torch.ones(16).to(DEV), q16.to(DEV), 16, 1, 1, 0, bits, mant, None, False)
    assert (out.cpu().float() != ref.float()).sum().item() == 0


def test_pack_raw(ops):
    x3 = torch.randn(1, 77, 48 * 3, generator=g(5))
    assert torch.equal(ops.pack_raw(x3.to(DEV)).cpu(), CB.pack_raw(x3))


@pytest.mark.parametrize("heads,bits,S,K,Sp,imgs", [(6, 4, 197, 197, 64, 4), (3, 3, 197, 197, 64, 4), (4, 4, 49, 49, 32, 128),
                                                      (32, 3, 49, 49, 32, 16), (8, 4, 49, 49, 32, 40)])
def test_gemm_mixed_bf16_rows_fp8_columns(ops, heads, bits, S, K, Sp, imgs):
    """matmul.py:173-201 with the AdaLog-quantised probabilities as the fixed operand: candidates packed as fp8 and converted to
    bf16 in the kernel (k_gemm_grpk8) must score like the all-bf16 launch (same exact operand values, other summation order)
    and like the CPU spec."""
    gen = g(21)
    P = 128
    G = imgs * heads
    KP = 64 if K <= 64 else 256
    assert ops.gemm_mixed_ok(S, Sp, G, heads, P, K)
    a = torch.softmax(torch.randn(G, S, K, generator=gen) * 2, -1)
    a = (a * 64).round().clamp(0, 255) / 64                                     # bf16-exact stand-in for the AdaLog values
    a[0, 5] = 0
    v = torch.randn(G, Sp, K, generator=gen)                                   # B^T: [G, dims, keys]
    sc = torch.rand(P, heads, generator=gen) * 0.2 + 0.05
    zp = torch.randint(2 ** (bits - 1) - 2, 2 ** (bits - 1) + 2, (P, heads), generator=gen).float()
    ref = torch.randn(G, Sp, S, generator=gen)                                 # transposed reference [G, S', S]
    one = torch.ones(1)
    res = {}
    for name in ("mixed", "bf16", "cpu"):
        mod = CB if name == "cpu" else ops
        dev = "cpu" if name == "cpu" else DEV
        dt = mod.BF16_FP8 if name == "mixed" else mod.BF16
        Kp = KP if name == "mixed" else mod.pad_k(K, mod.BF16, 64)
        ap = torch.zeros(1, G, S, Kp, dtype=torch.bfloat16, device=dev)
        ap[0, :, :, :K] = a.to(torch.bfloat16).to(dev)
        ap.k_valid = K
        cand = mod.pack_uniform(v.to(dev), sc.to(dev), zp.to(dev), P, heads, heads, 1, 0, bits, mod.FP8 if name == "mixed" else mod.BF16,
                                c_inner=True, k_align=KP if name == "mixed" else 64)
        res[name] = mod.gemm_score(dt, ap, cand, S, Sp, P, G, heads, ref.to(dev), mod.Strided(one.to(dev)),
                                   mod.Strided(sc.to(dev), c=heads, g=1), None, True, False, 1.0 / (S * Sp), ref_div=P, order=2,
                                   ref_transposed=True).cpu()
    assert res["mixed"].shape == (P, heads)
    assert rel_err(res["mixed"], res["bf16"]) <= 2e-6 and rel_err(res["mixed"], res["cpu"]) <= 2e-6


@pytest.mark.parametrize("M,K,O_,bits,with_bias", [(6304, 1536, 384, 4, True), (1000, 512, 96, 3, False), (777, 320, 40, 4, True),
                                                   (500, 1000, 37, 4, True), (193, 260, 3, 2, False)])
def test_gemm_mixed_streaming_weight_search(ops, M, K, O_, bits, with_bias):
    """linear.py:355-392 for the post-GELU layer: bf16 activation operand (AdaLog values) against fp8 weight candidates on the
    wide streaming kernel (k_gemm_stream<..., MX>) must score like the all-bf16 launch and like the CPU spec; ragged M, K not
    a multiple of 128, column bias per (candidate, channel)."""
    gen = g(23)
    P = 128
    assert ops.gemm_mixed_ok(M, O_, 1, 1, P, K)
    x = (torch.rand(1, M, K, generator=gen) * 32).round() / 32 * torch.pow(2.0, -torch.randint(0, 6, (1, M, K), generator=gen).float())
    W = torch.randn(1, O_, K, generator=gen) * 0.1
    sc = torch.rand(P, O_, generator=gen) * 0.02 + 0.005
    zp = torch.randint(2 ** (bits - 1) - 2, 2 ** (bits - 1) + 2, (P, O_), generator=gen).float()
    ref = torch.randn(1, O_, M, generator=gen)
    bias = torch.randn(P, O_, generator=gen) if with_bias else None
    one = torch.ones(1)
    res = {}
    for name in ("mixed", "bf16", "cpu"):
        mod = CB if name == "cpu" else ops
        dev = "cpu" if name == "cpu" else DEV
        Kp = mod.pad_k(K, mod.BF16, 128)
        xp = torch.zeros(1, 1, M, Kp, dtype=torch.bfloat16, device=dev)
        xp[0, 0, :, :K] = x[0].to(torch.bfloat16).to(dev)
        assert torch.equal(xp[0, 0, :, :K].float().cpu(), x[0])
        xp.k_valid = K
        wp = mod.pack_uniform(W.to(dev), sc.to(dev), zp.to(dev), P, O_, 1, 0, 1, bits, mod.FP8 if name == "mixed" else mod.BF16,
                              c_inner=True, k_align=64 if name == "mixed" else 128)
        b = None if bias is None else mod.Strided(bias.to(dev), c=O_, n=1)
        res[name] = mod.gemm_score(mod.BF16_FP8 if name == "mixed" else mod.BF16, xp, wp, M, O_, P, 1, 1, ref.to(dev),
                                   mod.Strided(one.to(dev)), mod.Strided(sc.to(dev), c=O_, n=1), b, False, True, 1.0 / 197,
                                   sa_mul=0.5, ref_div=P, order=2, ref_transposed=True).cpu()
    assert res["mixed"].shape == (P, O_)
    assert rel_err(res["mixed"], res["bf16"]) <= 2e-6 and rel_err(res["mixed"], res["cpu"]) <= 2e-6


def test_pack_split3_is_exact_and_scores_like_fp32(ops):
    """conv.py:226-255 with the unquantised input as three bf16 terms: hi + mid + lo == x bit for bit, and the scores of the
    bf16 GEMM over [hi | mid | lo] x [W | W | W] equal the fp32-operand GEMM's (same fp32 products, other summation order)."""
    gen = g(15)
    M, K, O_, P, bits = 300, 48 * 3, 40, 128, 4
    x3 = torch.randn(1, M, K, generator=gen) * torch.logspace(-3, 2, K).view(1, 1, K)
    x3[0, 0, :4] = torch.tensor([0.0, 1e-30, -3.0e38, 1.0 + 2.0 ** -23])
    got = ops.pack_split3(x3.to(DEV)).cpu()
    assert torch.equal(got.view(torch.int16), CB.pack_split3(x3).view(torch.int16))
    Kt = got.shape[-1] // 3
    s3 = got[..., :Kt].float() + got[..., Kt:2 * Kt].float() + got[..., 2 * Kt:].float()
    assert torch.equal(s3[..., :K], x3[None]) and (s3[..., K:] == 0).all()
    x3[0, 0, :4] = torch.tensor([0.0, 1e-30, -3.0, 1.0 + 2.0 ** -23])        # (3e38 squared overflows any score)
    W = torch.randn(1, O_, K, generator=gen) * 0.1
    sc = torch.rand(P, O_, generator=gen) * 0.02 + 0.005
    zp = torch.randint(4, 12, (P, O_), generator=gen).float()
    ref = torch.randn(1, O_, M, generator=gen)
    bias = torch.randn(O_, generator=gen)
    one = torch.ones(1)
    res = {}
    for name, dt in (("f32", ops.F32), ("split3", ops.BF16)):
        xa = ops.pack_raw(x3.to(DEV)) if name == "f32" else ops.pack_split3(x3.to(DEV))
        wb = ops.pack_uniform(W.to(DEV), sc.to(DEV), zp.to(DEV), P, O_, 1, 0, 1, bits, dt, c_inner=True)
        if name == "split3":
            wb = wb.repeat(1, 1, 1, 3)
        res[name] = ops.gemm_score(dt, xa, wb, M, O_, P, 1, 1, ref.to(DEV), ops.Strided(one.to(DEV)),
                                   ops.Strided(sc.to(DEV), c=O_, n=1), ops.Strided(bias.to(DEV), n=1), False, True, 1.0 / 7,
                                   ref_div=P, order=2, ref_transposed=True).cpu()
    assert res["split3"].shape == (P, O_) and rel_err(res["split3"], res["f32"]) <= 2e-6


# ------------------------------------------------------------------------------------------------ scoring GEMM
def _strided(mod, t, **kw):
    return mod.Strided(t, **kw)


@pytest.mark.parametrize("dtype", ["i8", "bf16", "f32"])
@pytest.mark.parametrize("shape", [(300, 200, 197, 1, 1, 5), (197, 197, 64, 12, 6, 4), (130, 129, 65, 2, 2, 3)])
def test_gemm_score_vs_spec(ops, dtype, shape):
    """M, N, K, G, gmod, C with ragged edges in every dimension; all three epilogue reductions."""
    M, N, K, G, gmod, C = shape
    gen = g(500 + M)
    dt_c = {"i8": CB.I8, "bf16": CB.BF16, "f32": CB.F32}[dtype]
    dt_o = {"i8": ops.I8, "bf16": ops.BF16, "f32": ops.F32}[dtype]
    Kp = CB.pad_k(K, dt_c)
    tdt = {"i8": torch.int8, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    A = torch.zeros(C, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, N, Kp, dtype=tdt)
    A[..., :K] = torch.randint(-15, 16, (C, G, M, K), generator=gen).to(tdt)
    B[..., :K] = torch.randint(-15, 16, (1, G, N, K), generator=gen).to(tdt)
    if dtype != "i8":
        A[..., :K] = (A[..., :K].float() * 0.25).to(tdt)
    ref = torch.randn(G, M, N, generator=gen) * 3
    sa = torch.rand(C, gmod, generator=gen) * 0.02 + 0.01
    sb = torch.rand(gmod, N, generator=gen) * 0.5 + 0.5
    bias = torch.randn(N, generator=gen)
    for keep_h, keep_n in ((False, True), (True, False), (False, False)):
        want = CB.gemm_score(dt_c, A, B, M, N, C, G, gmod, ref, CB.Strided(sa, c=gmod, g=1), CB.Strided(sb, g=N, n=1),
                             CB.Strided(bias, n=1), keep_h, keep_n, 1.0 / M, sa_mul=0.5)
        got = ops.gemm_score(dt_o, A.to(DEV), B.to(DEV), M, N, C, G, gmod, ref.to(DEV),
                             ops.Strided(sa.to(DEV), c=gmod, g=1), ops.Strided(sb.to(DEV), g=N, n=1),
                             ops.Strided(bias.to(DEV), n=1), keep_h, keep_n, 1.0 / M, sa_mul=0.5)
        assert got.shape == want.shape
        assert rel_err(got.cpu(), want) <= 2e-6, (keep_h, keep_n, rel_err(got.cpu(), want))
    out = ops.gemm_out(dt_o, A[:1].to(DEV), B.to(DEV), M, N, G, gmod, ops.Strided(sa[:1].to(DEV), g=1),
                       ops.Strided(sb.to(DEV), g=N, n=1), ops.Strided(bias.to(DEV), n=1), sa_mul=0.5)
    want = CB.gemm_out(dt_c, A[:1], B, M, N, G, gmod, CB.Strided(sa[:1], g=1), CB.Strided(sb, g=N, n=1),
                       CB.Strided(bias, n=1), sa_mul=0.5)
    assert rel_err(out.cpu(), want) <= 2e-6


def test_gemm_score_transpose_detecting(ops):
    """A = I with an ASYMMETRIC B: catches a swapped C/D fragment layout (cdna guide, G9)."""
    M = N = 128; K = 128
    A = torch.zeros(1, 1, M, K, dtype=torch.int8); A[0, 0, torch.arange(M), torch.arange(M)] = 1
    B = torch.zeros(1, 1, N, K, dtype=torch.int8)
    B[0, 0] = (torch.arange(N).view(-1, 1) * 3 + torch.arange(K).view(1, -1) * 5) % 23 - 11
    one = torch.ones(1)
    out = ops.gemm_out(ops.I8, A.to(DEV), B.to(DEV), M, N, 1, 1, ops.Strided(one.to(DEV)), ops.Strided(one.to(DEV)), None)
    assert torch.equal(out.cpu()[0], B[0, 0].float().t())


def test_gemm_score_deterministic(ops):
    gen = g(7)
    A = torch.randint(-8, 8, (8, 1, 1000, 256), generator=gen).to(torch.int8).to(DEV)
    B = torch.randint(-8, 8, (1, 1, 300, 256), generator=gen).to(torch.int8).to(DEV)
    ref = torch.randn(1, 1000, 300, generator=gen).to(DEV)
    s = torch.rand(8, generator=gen).to(DEV) * 0.01; sb = torch.rand(300, generator=gen).to(DEV)
    run = lambda: ops.gemm_score(ops.I8, A, B, 1000, 300, 8, 1, 1, ref, ops.Strided(s, c=1), ops.Strided(sb, n=1), None,
                                 False, False, 1e-3)
    a = run()
    for _ in range(3):
        assert torch.equal(a, run()), "scores must be bit-reproducible (fixed-order fp64 finish)"


# ------------------------------------------------------------------------------------------------ FPCS pieces
def test_topk_ties_and_nan(ops):
    gen = g(11)
    s = torch.randn(128, 300, generator=gen)
    s[5, :] = s[9, :]                          # exact ties -> lower index first
    s[100, 7] = float("nan")
    for k in (1, 8, 16, 32):
        assert torch.equal(ops.topk(s.to(DEV), k).cpu(), CB.topk(s, k))
    vals, _ = torch.topk(torch.nan_to_num(s, nan=float("inf")), 16, dim=0)
    got = torch.gather(torch.nan_to_num(s, nan=float("inf")), 0, ops.topk(s.to(DEV), 16).cpu().long())
    assert torch.equal(got, vals)


def test_candidate_grid_and_fpcs_next(ops):
    gen = g(12)
    cols = 70
    q4 = torch.stack([torch.rand(cols, generator=gen) + 1, torch.rand(cols, generator=gen) + 2,
                      -torch.rand(cols, generator=gen) - 1, -torch.rand(cols, generator=gen) - 2])
    for bits, num_zp, clamp in ((4, 8, None), (6, 16, 1e-4), (3, 4, None)):
        L = 2 ** (bits - 1)
        ns = 128 // num_zp
        lin = torch.linspace(0, 1, ns)
        want = CB.candidate_grid(q4, ns, num_zp, int(L - num_zp / 2), bits, lin, clamp)
        got = ops.candidate_grid(q4.to(DEV), ns, num_zp, int(L - num_zp / 2), bits, lin.to(DEV), clamp)
        for a, b in zip(got, want):
            torch.testing.assert_close(a.cpu(), b, rtol=2e-7, atol=0)
    scale, zp, delta = want
    scores = torch.randn(128, cols, generator=gen)
    idx = CB.topk(scores, 16)
    lin8 = torch.linspace(0, 1, 8)
    d_ref = delta.clone()
    w = CB.fpcs_next(scale, zp, zp * 2, idx, 16, 8, lin8, d_ref, 1e-4)
    d_dev = delta.clone().to(DEV)
    o = ops.fpcs_next(scale.to(DEV), zp.to(DEV), (zp * 2).to(DEV), idx.to(DEV), 16, 8, lin8.to(DEV), d_dev, 1e-4)
    for a, b in zip(o, w):
        assert torch.equal(a.cpu(), b)
    assert torch.equal(d_dev.cpu(), d_ref)
    idx1 = CB.topk(scores, 1)
    w = CB.fpcs_next(scale, zp, None, idx1, 1, 0, None, None, None)
    o = ops.fpcs_next(scale.to(DEV), zp.to(DEV), None, idx1.to(DEV), 1, 0, None, None, None)
    assert torch.equal(o[0].cpu(), w[0]) and torch.equal(o[1].cpu(), w[1]) and o[2] is None


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_self_mse_scores(ops, bits):
    gen = g(600 + bits)
    L = 2 ** (bits - 1)
    W = torch.randn(3, 64, 192, generator=gen) * 0.1
    sc, zp = O.weight_candidates(W, bits)
    want = O.score_w_self(W, sc, zp, bits).reshape(128, -1)
    got = ops.score_w_self(W.view(-1, 192).to(DEV), sc.reshape(128, -1).to(DEV), zp.reshape(128, -1).float().to(DEV), bits)
    assert rel_err(got.cpu(), want) <= 1e-5
    x = torch.randn(6, 197, 96, generator=gen) * torch.linspace(0.3, 2, 96)
    for cw in (True, False):
        s2, z2 = O.activation_candidates(x, bits, cw)
        want = O.score_a_self(x, s2, z2, bits, cw).t()                       # [P, C]
        T = 197
        norm = 1.0 / T if cw else 1.0 / (T * 96)
        got = ops.score_a_self(x.view(-1, 96).to(DEV), s2.t().contiguous().to(DEV), z2.t().contiguous().float().to(DEV),
                               cw, bits, norm)
        assert rel_err(got.cpu(), want) <= 1e-5


@pytest.mark.parametrize("bits", [3, 4, 6, 8])
def test_self_mse_scores_sorted_prefix(ops, bits):
    """csrc/sorted_score.hip: the self-MSE searches scored from the sorted tensor (segments sorted by a library radix sort,
    fp64 prefix sums, 2^bits bisections per candidate with the reference's exact rounding predicate) against the oracle's
    element-by-element evaluation: weights (a segment per row), per-channel and per-tensor activations, ragged segment
    lengths, duplicated values, +-0, candidates with a NON-integral zero point and with clamping at both ends."""
    gen = g(650 + bits)
    W = torch.randn(3, 64, 192, generator=gen) * 0.1
    sc, zp = O.weight_candidates(W, min(bits, 6))
    if bits <= 6:
        want = O.score_w_self(W, sc, zp, bits).reshape(128, -1)
        sp = ops.sorted_prefix(W.view(-1, 192).to(DEV))
        assert torch.equal(sp.sorted.cpu(), W.view(-1, 192).sort(dim=-1).values)
        pf = sp.prefix.cpu()
        ref_pf = torch.cat([torch.zeros(192, 1, dtype=torch.float64), sp.sorted.cpu().double().cumsum(-1)], -1)
        torch.testing.assert_close(pf[..., 0], ref_pf, rtol=1e-13, atol=1e-13)
        got = ops.score_self_sorted(sp, sc.reshape(128, -1).to(DEV), zp.reshape(128, -1).float().to(DEV), bits, 1.0 / 192)
        assert rel_err(got.cpu(), want) <= 1e-5
    for rows, C in ((6 * 197, 96), (1027, 5), (4096, 3)):                      # n % 4 != 0, n % 1024 == 0, tiny segment counts
        x = torch.randn(rows, C, generator=gen) * torch.linspace(0.3, 2, C)
        x[:7, 0] = x[7:14, 0]
        x[20, :] = 0.0
        x[21, :] = -0.0
        for cw in (True, False):
            cols = C if cw else 1
            P = 128
            s = (x.abs().max() / (2 ** bits - 1)) * (0.5 + 1.5 * torch.rand(P, cols, generator=gen))
            z = torch.randint(0, 2 ** bits, (P, cols), generator=gen).float()
            z[::7] += 0.37                                                      # non-integral zero points (never produced by FPCS)
            z[5] = 0.0
            z[6] = 2 ** bits - 1.0
            want = CB.score_a_self(x, s, z, cw, bits, 0.01)
            x2 = x.t().contiguous() if cw else x.reshape(1, -1)
            sp = ops.sorted_prefix(x2.to(DEV))
            got = ops.score_self_sorted(sp, s.to(DEV), z.to(DEV), bits, 0.01)
            assert rel_err(got.cpu(), want) <= 2e-5, (rows, C, cw)
    # a large single segment (deit_small: 6304 x 384 = 2.4 M elements), per-tensor candidates from the reference's grid
    x = torch.randn(32 * 197 * 384, generator=gen) * 1.3 + 0.2
    if bits <= 6:
        s2, z2 = O.activation_candidates(x.view(32, 197, 384), bits, False)                  # [1, 128]
        sub = [0, 31, 64, 127]
        want = O.score_a_self(x.view(32, 197, 384), s2[:, sub], z2[:, sub], bits, False, 32).reshape(-1)
        sp = ops.sorted_prefix(x.view(1, -1).to(DEV))
        got = ops.score_self_sorted(sp, s2.t().contiguous().to(DEV), z2.t().contiguous().float().to(DEV), bits, 1.0 / (197 * 384))
        assert rel_err(got.cpu().reshape(-1)[sub], want) <= 1e-5


@pytest.mark.parametrize("M,T,K,P,bits,dt", [
    (1152, 4 * 197, 384, 128, 4, "fp8"), (1536, 3 * 197, 384, 128, 4, "i8"), (384, 5 * 197, 384, 128, 4, "fp8"),
    (384, 591, 384, 128, 3, "fp8"),                    # odd token count: the last slab holds one token
    (768, 1001, 192, 128, 6, "i8"), (512, 777, 96, 128, 4, "fp8"),      # K = 96: the second K-step is half padding
    (1152, 403, 384, 64, 4, "fp8"), (1152, 402, 384, 256, 4, "i8"),     # 4 tokens / 1 token per slab
    (2304, 600, 768, 128, 4, "fp8"), (256, 640, 768, 128, 6, "i8"),     # K = 768: 128-column slabs
])
def test_score_act_gen_matches_packed_path(ops, M, T, K, P, bits, dt):
    """gemm_k_slab.inc GEN form (candidate operand generated in the kernel) against pack_uniform + gemm_score on the same
    candidates (the path it replaces, itself pinned to the reference's traces), incl. values planted ON rounding ties, and
    against the oracle on four candidates."""
    gen = g(4000 + M + T + K + P)
    DT = ops.FP8 if dt == "fp8" else ops.I8
    x = torch.randn(T, K, generator=gen) * 1.3 + 0.2
    W = torch.randn(M, K, generator=gen) * 0.05
    b = torch.randn(M, generator=gen) * 0.1
    ref = torch.nn.functional.linear(x, W, b)
    L = 2 ** (bits - 1)
    w_s = (W.amax(1) - W.amin(1)) / (2 * L - 1)
    w_z = torch.round(-W.amin(1) / w_s).clamp(0, 2 * L - 1)
    s = (x.abs().max() * 2 / (2 * L - 1)) * (0.4 + 1.2 * torch.rand(P, 1, generator=gen))
    z = torch.randint(0, 2 * L, (P, 1), generator=gen).float()
    # plant exact ties: x = (k + 0.5) * s_p for some candidates (the IEEE quotient decides those bins)
    for j in range(0, P, 9):
        kk = torch.randint(-L, L, (K,), generator=gen).float() + 0.5
        x[(7 * j) % T] = kk * s[j, 0]
    xd, Wd = x.to(DEV), W.to(DEV)
    wp = ops.pack_uniform(Wd.unsqueeze(0), w_s.to(DEV), w_z.to(DEV), 1, 0, 1, 0, 1, bits, DT)
    norm = 1.0 / (197 * M)
    assert ops.score_act_gen_ok(DT, M, T, K, wp.shape[-1], P)
    for bias in (b.to(DEV), None):
        got = ops.score_act_gen(DT, wp, xd, s.to(DEV), z.to(DEV), bits, (ref if bias is not None else ref - b).to(DEV), w_s.to(DEV), bias, norm)
        xp = ops.pack_uniform(xd.unsqueeze(0), s.to(DEV), z.to(DEV), P, 1, 1, 0, 0, bits, DT, c_inner=True)
        one = torch.ones(1, device=DEV)
        want = ops.gemm_score(DT, wp, xp, M, T, P, 1, 1, (ref if bias is not None else ref - b).to(DEV).reshape(1, T, M),
                              ops.Strided(one), ops.Strided(s.to(DEV).contiguous(), c=1), None, False, False, norm, ref_div=P,
                              order=2, ref_transposed=True, row_scale=w_s.to(DEV),
                              row_bias=bias if bias is not None else torch.zeros(M, device=DEV))
        assert rel_err(got.cpu(), want.cpu()) <= 2e-6, (bias is None)
    sub = [0, P // 3, P // 2, P - 1]
    wq = O.uniform_fake_quant(W, w_s.view(-1, 1), w_z.view(-1, 1), bits)[0]
    refo = O.score_a(x.view(1, T, K), wq, b, ref.view(1, T, M), s[sub].t().contiguous(), z[sub].t().contiguous(), bits, 1)
    got = ops.score_act_gen(DT, wp, xd, s.to(DEV), z.to(DEV), bits, ref.to(DEV), w_s.to(DEV), b.to(DEV), 1.0 / (T * M))
    assert rel_err(got.cpu().reshape(-1)[sub], refo.reshape(-1)) <= 1e-4


def test_finish_topk_next_fused_equals_two_launches(ops):
    """csrc/gemm_finish.inc: finish + top-k + next grid in ONE launch (per-column partials: a block ranks its own columns;
    per-workgroup accumulators: the last block to arrive ranks) against adalog_finish_scores + adalog_topk_next, bit for bit:
    weight search (a winner per output channel), per-tensor activation search through the packed path and through the
    in-kernel-generated operand, per-head attention search; expansion (16 x 8), width-32 form with a third plane, and the commit."""
    from adalog_amd import quant_layers as Q, search
    gen = g(77)
    I, Oc, T, N, bits = 384, 768, 197, 8, 4
    x = torch.randn(N, T, I, generator=gen) * 1.3
    W = torch.randn(Oc, I, generator=gen) * 0.05
    b = torch.randn(Oc, generator=gen) * 0.1
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", bits, bits, calib_batch_size=8, search_round=1, eq_n=128, n_V=1,
                                              fpcs=True, steps=6).to(DEV)
    lay.weight.data.copy_(W); lay.bias.data.copy_(b)
    lay.raw_input = x.to(DEV); lay.raw_out = torch.nn.functional.linear(lay.raw_input, lay.weight.data, lay.bias.data)
    scw, zpw = O.weight_candidates(W.view(1, Oc, I), bits)
    sca, zpa = O.activation_candidates(x, bits, False)
    for q, sc, zp in ((lay.a_quantizer, sca[:, 60], zpa[:, 60].float()), (lay.w_quantizer, scw[60], zpw[60].float())):
        q.scale.data.copy_(sc.reshape(q.scale.shape)); q.zero_point.data.copy_(zp.reshape(q.zero_point.shape))
        q.inited = True; q._zp_on_grid = True
    lin8, lin4 = search.linspace01(8, torch.device(DEV)), search.linspace01(4, torch.device(DEV))

    def both(make_pending, scale, zp, third, cases):
        for k, new_cnt, lin, clamp in cases:
            cols = scale.shape[1]
            d1 = (torch.rand(cols, generator=gen) * 0.01 + 0.001).to(DEV)
            d2 = d1.clone()
            pend = make_pending()
            assert isinstance(pend, ops.PendingScores)
            want_scores = pend.finish()
            want = ops.topk_next(want_scores, scale, zp, third, k, new_cnt, lin, d1 if new_cnt else None, clamp)
            got = ops.finish_topk_next(make_pending(), scale, zp, third, k, new_cnt, lin, d2 if new_cnt else None, clamp)
            for a, bb in zip(got, want):
                assert (a is None) == (bb is None)
                if a is not None:
                    assert torch.equal(a, bb), (k, new_cnt)
            assert torch.equal(d1, d2)

    cases = [(16, 8, lin8, None), (16, 8, lin8, 1e-4), (1, 0, None, None)]
    with torch.no_grad():
        cw_s, cw_z = scw.reshape(128, -1).to(DEV), zpw.reshape(128, -1).float().to(DEV)
        fixed = lay._pack_x_fixed()
        both(lambda: lay._score_w(fixed, cw_s, cw_z, defer=True), cw_s, cw_z, None, cases)                       # [128, 768]
        ca_s, ca_z = sca.t().contiguous().to(DEV), zpa.t().contiguous().float().to(DEV)
        dt = lay._int_dt(N * T, prefer_fp8=True)
        wp = lay._pack_w_fixed(dt); wp.int_dt = dt
        third = (torch.arange(128, dtype=torch.float32).view(128, 1) + 10).to(DEV)
        both(lambda: lay._score_a(wp, ca_s, ca_z, defer=True), ca_s, ca_z, third, cases + [(32, 4, lin4, None)])   # generated operand
        from adalog_amd.quant_layers import linear as LM
        LM.GEN_ACT_SEARCH = False
        try:
            both(lambda: lay._score_a(wp, ca_s, ca_z, defer=True), ca_s, ca_z, None, cases)                      # packed operand
        finally:
            LM.GEN_ACT_SEARCH = True
    H, S, hd = 6, 197, 64
    A = torch.randn(N, H, S, hd, generator=gen) * (0.5 + torch.rand(1, H, 1, 1, generator=gen))
    Bt = torch.randn(N, H, S, hd, generator=gen)
    mm = Q.AsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=8, search_round=1, eq_n=128,
                                             head_channel_wise=True, num_heads=H, fpcs=True, steps=6).to(DEV)
    Bd = Bt.to(DEV).transpose(-2, -1)
    mm.raw_input, mm.raw_out = [A.to(DEV), Bd], A.to(DEV) @ Bd
    mm._initialize_calib_parameters()
    sA, zA = O.matmul_candidates(A, bits)
    sB, zB = O.matmul_candidates(Bt.transpose(-2, -1), bits)
    for q, sc, zp in ((mm.A_quantizer, sA[60], zA[60].float()), (mm.B_quantizer, sB[60], zB[60].float())):
        q.scale.data.copy_(sc.reshape(q.scale.shape).to(DEV)); q.zero_point.data.copy_(zp.reshape(q.zero_point.shape).to(DEV))
        q.inited = True; q._zp_on_grid = True
    with torch.no_grad():
        from tests.trace_replay import _mm_dt
        dtm = _mm_dt(mm)
        fixedB = mm._pack_fixed("B", dtm)
        cs, cz = sA.reshape(128, H).to(DEV), zA.reshape(128, H).float().to(DEV)
        both(lambda: mm._score("A", fixedB, cs, cz, dtm, defer=True), cs, cz, None, cases)                        # [128, 6 heads]


# ------------------------------------------------------------------------------------------------ order statistics
@pytest.mark.parametrize("S,n", [(1, 1000003), (96, 384), (7, 6304), (4, 65536)])
def test_quantile_rows(ops, S, n):
    gen = g(700 + S)
    x = torch.randn(S, n, generator=gen)
    x[0, :10] = x[0, 10:20]                                                 # duplicates
    qs = torch.tensor([0.9, 1.0]).tolist() + (1 - torch.tensor([0.9, 1.0])).tolist()
    want = torch.quantile(x, torch.tensor(qs), dim=-1)
    got = ops.quantile_rows(x.to(DEV), qs, 1)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-6, atol=1e-7)
    if S % 2 == 0:
        got = ops.quantile_rows(x.to(DEV), qs, 2)
        torch.testing.assert_close(got.cpu(), want.view(4, S // 2, 2).mean(-1), rtol=1e-6, atol=1e-7)


def test_quantile_negative_zero_and_constant(ops):
    x = torch.tensor([[0.0, -0.0, 1.0, -1.0, 0.0, 2.0, -2.0, 0.0], [3.0] * 8])
    qs = [0.5, 0.0, 1.0, 0.25]
    torch.testing.assert_close(ops.quantile_rows(x.to(DEV), qs, 1).cpu(), torch.quantile(x, torch.tensor(qs), dim=-1),
                               rtol=0, atol=0)


def test_positive_percentile(ops):
    gen = g(800)
    x = torch.nn.functional.gelu(2 * torch.randn(1, 500000, generator=gen))
    qs = [0.9, 1.0, 0.5, 0.013]
    want = O.positive_percentile(x.view(-1), torch.tensor(qs)).view(-1, 1)
    assert torch.equal(ops.positive_percentile_rows(x.to(DEV), qs).cpu(), want)
    neg = -torch.rand(1, 1000, generator=gen)                               # no positive entry -> 0 (linear.py:796-797)
    assert torch.equal(ops.positive_percentile_rows(neg.to(DEV), [0.9, 1.0]).cpu(), torch.zeros(2, 1))


def test_misc_kernels(ops):
    gen = g(900)
    W = torch.randn(100, 333, generator=gen)
    mn, mx = ops.minmax_rows(W.to(DEV))
    assert torch.equal(mn.cpu(), W.amin(1)) and torch.equal(mx.cpu(), W.amax(1))
    x = torch.randn(5000, 96, generator=gen)
    for pc in (True, False):
        a, b = ops.absminmax(x.to(DEV), pc)
        wa, wb = CB.absminmax(x, pc)
        assert torch.equal(a.cpu(), wa) and torch.equal(b.cpu(), wb)
    rs = torch.randint(-500, 500, (4, 50), generator=gen).to(torch.int32)
    ws = torch.rand(4, 50, generator=gen); sh = torch.tensor([0.17]); bias = torch.randn(50, generator=gen)
    torch.testing.assert_close(ops.shift_fold(rs.to(DEV), ws.to(DEV), sh.to(DEV), bias.to(DEV)).cpu(),
                               CB.shift_fold(rs, ws, sh, bias), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", ["i8", "f32"])
def test_gemm_score_candidates_in_columns(ops, dtype):
    """Weight-search layout: B packed candidates-innermost (c_inner), ref_div = P, every tile order -> same scores as
    the candidate-batched layout and as the CPU specification."""
    gen = g(31)
    bits, P, O_, I_, M = 4, 128, 40, 100, 300
    dt_c, dt_o = (CB.I8, ops.I8) if dtype == "i8" else (CB.F32, ops.F32)
    W = torch.randn(1, O_, I_, generator=gen) * 0.1
    sc = torch.rand(P, O_, generator=gen) * 0.02 + 0.005
    zp = torch.randint(4, 12, (P, O_), generator=gen).float()
    x3 = torch.randn(1, M, I_, generator=gen)
    xs, xz = torch.tensor([0.2]), torch.tensor([8.0])
    ref = torch.randn(1, M, O_, generator=gen)
    bias = torch.randn(O_, generator=gen)
    if dtype == "i8":
        xa_c = CB.pack_uniform(x3, xs, xz, 1, 0, 1, 0, 0, bits, dt_c)
        xa = ops.pack_uniform(x3.to(DEV), xs.to(DEV), xz.to(DEV), 1, 0, 1, 0, 0, bits, dt_o)
    else:
        xa_c, xa = CB.pack_raw(x3), ops.pack_raw(x3.to(DEV))
    wb_c = CB.pack_uniform(W, sc, zp, P, O_, 1, 0, 1, bits, dt_c, c_inner=True)
    wb = ops.pack_uniform(W.to(DEV), sc.to(DEV), zp.to(DEV), P, O_, 1, 0, 1, bits, dt_o, c_inner=True)
    assert torch.equal(wb.cpu().float(), wb_c.float())
    want = CB.gemm_score(dt_c, xa_c, wb_c, M, O_, P, 1, 1, ref, CB.Strided(xs), CB.Strided(sc, c=O_, n=1),
                         CB.Strided(bias, n=1), False, True, 1.0 / 7, ref_div=P)
    for order in (0, 1, 2):
        got = ops.gemm_score(dt_o, xa, wb, M, O_, P, 1, 1, ref.to(DEV), ops.Strided(xs.to(DEV)),
                             ops.Strided(sc.to(DEV), c=O_, n=1), ops.Strided(bias.to(DEV), n=1), False, True, 1.0 / 7,
                             ref_div=P, order=order)
        assert got.shape == (P, O_) and rel_err(got.cpu(), want) <= 2e-6, order
    # candidate-batched layout gives the same numbers
    wb2 = ops.pack_uniform(W.to(DEV), sc.to(DEV), zp.to(DEV), P, O_, 1, 0, 1, bits, dt_o)
    for order in (0, 1, 2):
        got2 = ops.gemm_score(dt_o, xa, wb2, M, O_, P, 1, 1, ref.to(DEV), ops.Strided(xs.to(DEV)),
                              ops.Strided(sc.to(DEV), c=O_, n=1), ops.Strided(bias.to(DEV), n=1), False, True, 1.0 / 7,
                              order=order)
        assert rel_err(got2.cpu(), want) <= 2e-6


# ------------------------------------------------------------------------------------------------ K17 (BRECQ)
@pytest.mark.parametrize("layout", ["tensor", "heads", "rows"])
def test_uniform_backward(ops, layout):
    gen = g(41)
    bits = 4
    if layout == "tensor":
        x = torch.randn(6, 50, 96, generator=gen) * 2; s = torch.tensor([0.21]); z = torch.tensor([7.0])
    elif layout == "heads":
        x = torch.randn(4, 6, 20, 16, generator=gen); s = torch.rand(1, 6, 1, 1, generator=gen) * 0.3 + 0.1
        z = torch.randint(4, 12, (1, 6, 1, 1), generator=gen).float()
    else:
        x = torch.randn(3, 16, 64, generator=gen) * 0.2; s = torch.rand(3, 16, 1, generator=gen) * 0.03 + 0.01
        z = torch.randint(4, 12, (3, 16, 1), generator=gen).float()
    gy = torch.randn(x.shape, generator=gen)
    want = CB.uniform_fake_quant_backward(gy, x, s, z, bits, False, True, True)
    got = ops.uniform_fake_quant_backward(gy.to(DEV), x.to(DEV), s.to(DEV), z.to(DEV), bits, False, True, True)
    torch.testing.assert_close(got[0].cpu(), want[0], rtol=1e-6, atol=0)      # autograd's gy*s/s vs gy: 1 ulp
    torch.testing.assert_close(got[1].cpu(), want[1], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[2].cpu(), want[2], rtol=1e-4, atol=1e-5)
    want = CB.uniform_fake_quant_backward(gy, x, s.abs() * 3, None, bits, True, True, False)
    got = ops.uniform_fake_quant_backward(gy.to(DEV), x.to(DEV), (s.abs() * 3).to(DEV), None, bits, True, True, False)
    torch.testing.assert_close(got[0].cpu(), want[0], rtol=1e-6, atol=0)
    torch.testing.assert_close(got[1].cpu(), want[1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("q", [23, 37, 90])
def test_adalog_backward(ops, q):
    gen = g(42 + q)
    bits = 4
    x = torch.nn.functional.gelu(2 * torch.randn(8, 50, 64, generator=gen))
    s = torch.tensor([x.max().item() * 0.8]); sh = torch.tensor([O.GELU_SHIFT]); qd = torch.tensor([q])
    gy = torch.randn(x.shape, generator=gen)
    for sub in (True, False):
        y = ops.log_fake_quant(x.to(DEV), s.to(DEV), qd.to(DEV), None, None, bits, shift=sh.to(DEV), sub_shift=sub,
                               train_form=True)
        want = CB.log_fake_quant_backward(gy, x, y.cpu(), s, qd, bits, sh, sub)
        got = ops.log_fake_quant_backward(gy.to(DEV), x.to(DEV), y, s.to(DEV), qd.to(DEV), bits, sh.to(DEV), sub)
        torch.testing.assert_close(got[0].cpu(), want[0], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(got[1].cpu(), want[1], rtol=1e-3, atol=1e-3)


def test_adaround_and_round_loss(ops):
    gen = g(43)
    bits = 4
    w = torch.randn(48, 96, generator=gen) * 0.2
    s = (w.amax(1) - w.amin(1)) / 15; z = torch.round(-w.amin(1) / s)
    alpha = torch.randn(48, 96, generator=gen) * 2
    gy = torch.randn(48, 96, generator=gen)
    for soft in (True, False):
        want = CB.adaround(w, alpha, s, z, bits, soft)
        got = ops.adaround(w.to(DEV), alpha.to(DEV), s.to(DEV), z.to(DEV), bits, soft)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)
        wantg = CB.adaround(w, alpha, s, z, bits, soft, gy=gy)
        gotg = ops.adaround(w.to(DEV), alpha.to(DEV), s.to(DEV), z.to(DEV), bits, soft, gy=gy.to(DEV))
        torch.testing.assert_close(gotg.cpu(), wantg, rtol=1e-4, atol=1e-6)
    for b in (20.0, 11.0, 2.0):
        ga_ref = torch.zeros_like(alpha)
        l_ref = CB.round_loss(alpha, b, galpha=ga_ref, gscale=0.5)
        ga = torch.zeros_like(alpha).to(DEV)
        l = ops.round_loss(alpha.to(DEV), b, galpha=ga, gscale=0.5)
        torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(ga.cpu(), ga_ref, rtol=1e-3, atol=1e-5)
        # the form the autograd node uses: upstream gradient read on the device, gradient written (not accumulated)
        ga2 = torch.full_like(alpha, 7.0).to(DEV)
        assert ops.round_loss(alpha.to(DEV), b, galpha=ga2, gscale=1.0, want_loss=False, gmul=torch.tensor([0.5]).to(DEV),
                              overwrite=True) is None
        torch.testing.assert_close(ga2.cpu(), ga_ref, rtol=1e-3, atol=1e-5)
    # many launches back to back reuse the ticket counters: every value must stay the same
    big = torch.randn(300_000, generator=gen).to(DEV)
    vals = torch.stack([ops.round_loss(big, 4.0) for _ in range(2100)])
    assert torch.equal(vals, vals[0].expand_as(vals))


@pytest.mark.parametrize("dtype", ["i8", "bf16"])
@pytest.mark.parametrize("M", [64, 100, 197, 384, 600])
def test_gemm_cand_kernel_row_scale_and_tiles(ops, dtype, M):
    """Large-tile kernel (C = 1, candidates in columns): every row-tile size, K tails of 64 and 128 bytes, transposed
    reference, per-row scale/bias -- against the CPU specification."""
    gen = g(900 + M)
    dt_c, dt_o = (CB.I8, ops.I8) if dtype == "i8" else (CB.BF16, ops.BF16)
    tdt = torch.int8 if dtype == "i8" else torch.bfloat16
    P, Ncols, G, gmod = 128, 37, 4, 2
    for K in (64, 200):
        Kp = CB.pad_k(K, dt_c)
        A = torch.zeros(1, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, Ncols * P, Kp, dtype=tdt)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).to(tdt)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).to(tdt)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3                      # stored [G, N, M] (transposed)
        sa = torch.rand(gmod, generator=gen) * 0.02 + 0.01
        sb = torch.rand(P, gmod, generator=gen) * 0.5 + 0.5
        rs = torch.rand(M, generator=gen) + 0.5; rb = torch.randn(M, generator=gen)
        for keep_h in (True, False):
            want = CB.gemm_score(dt_c, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa, g=1), CB.Strided(sb, c=gmod, g=1), None,
                                 keep_h, False, 0.01, sa_mul=0.5, ref_div=P, ref_transposed=True, row_scale=rs, row_bias=rb)
            for order in (1, 2):
                got = ops.gemm_score(dt_o, A.to(DEV), B.to(DEV), M, Ncols, P, G, gmod, ref.to(DEV),
                                     ops.Strided(sa.to(DEV), g=1), ops.Strided(sb.to(DEV), c=gmod, g=1), None, keep_h, False,
                                     0.01, sa_mul=0.5, ref_div=P, order=order, ref_transposed=True, row_scale=rs.to(DEV),
                                     row_bias=rb.to(DEV))
                assert got.shape == want.shape and rel_err(got.cpu(), want) <= 3e-6, (K, keep_h, order, rel_err(got.cpu(), want))


@pytest.mark.parametrize("dtype", ["i8", "bf16", "fp8"])
@pytest.mark.parametrize("P", [64, 128, 256])
def test_gemm_stream_kernel_variants(ops, dtype, P):
    """Persistent streaming kernel (ref_div in {64, 128, 256}, transposed reference): column bias shared by the candidates
    (folded into the staged reference), per-(candidate, column) bias, no bias; with and without a row scale; K padding
    skipped through k_valid; ragged M and N edges; several groups -- against the CPU specification."""
    gen = g(1200 + P)
    dt_c, dt_o = {"i8": (CB.I8, ops.I8), "bf16": (CB.BF16, ops.BF16), "fp8": (CB.FP8, ops.FP8)}[dtype]
    tdt = {"i8": torch.int8, "bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn}[dtype]
    G, gmod = 3, 1
    # the last three shapes have >= 1024 bytes of K: the one-workgroup-per-CU 192/256-row form (ragged and exact M)
    for M, Ncols, K in ((197, 5, 64), (300, 11, 136), (64, 3, 40), (384, 3, 1100), (200, 5, 1030), (700, 2, 1100)):
        Kp = CB.pad_k(K, dt_c)
        A = torch.zeros(1, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, Ncols * P, Kp, dtype=tdt)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).float().to(tdt)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).float().to(tdt)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3
        sa = torch.rand(1, generator=gen) * 0.002 + 0.001
        sb = torch.rand(P, Ncols, generator=gen) * 0.5 + 0.5
        rs = torch.rand(M, generator=gen) + 0.5; rb = torch.randn(M, generator=gen)
        b_n = torch.randn(Ncols, generator=gen); b_cn = torch.randn(P, Ncols, generator=gen)
        for bias_kind in ("none", "n", "cn"):
            for rows in (False, True):
                cb = {"none": None, "n": CB.Strided(b_n, n=1), "cn": CB.Strided(b_cn, c=Ncols, n=1)}[bias_kind]
                want = CB.gemm_score(dt_c, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa), CB.Strided(sb, c=Ncols, n=1), cb,
                                     False, True, 0.01, ref_div=P, ref_transposed=True,
                                     row_scale=rs if rows else None, row_bias=rb if rows else None)
                ob = {"none": None, "n": ops.Strided(b_n.to(DEV), n=1), "cn": ops.Strided(b_cn.to(DEV), c=Ncols, n=1)}[bias_kind]
                for kv in (None, K):
                    Ad, Bd = A.to(DEV), B.to(DEV)
                    if kv is not None:
                        Ad.k_valid = kv; Bd.k_valid = kv
                    got = ops.gemm_score(dt_o, Ad, Bd, M, Ncols, P, G, gmod, ref.to(DEV), ops.Strided(sa.to(DEV)),
                                         ops.Strided(sb.to(DEV), c=Ncols, n=1), ob, False, True, 0.01, ref_div=P, order=2,
                                         ref_transposed=True, row_scale=rs.to(DEV) if rows else None,
                                         row_bias=rb.to(DEV) if rows else None)
                    assert got.shape == want.shape
                    assert rel_err(got.cpu(), want) <= 3e-6, (M, Ncols, K, bias_kind, rows, kv, rel_err(got.cpu(), want))


@pytest.mark.parametrize("dtype", ["i8", "fp8"])
@pytest.mark.parametrize("P", [64, 128, 256])
def test_gemm_slab_kernel(ops, dtype, P):
    """Slab kernel (int8 or fp8 storage, one group, K <= 384 bytes, M a multiple of 32 and >= 768): slabs cut into several pieces by the
    workgroup ranges (small N) and ranges spanning several slabs (large N), a ragged last slab, 2..6 K-steps with and
    without K padding, every bias kind, with and without the row scale, per-column scores and scores summed over the
    columns (per-workgroup accumulators) -- against the CPU specification."""
    gen = g(4100 + P)
    G, gmod = 1, 1
    dt_c, dt_o = {"i8": (CB.I8, ops.I8), "fp8": (CB.FP8, ops.FP8)}[dtype]
    tdt = {"i8": torch.int8, "fp8": torch.float8_e4m3fn}[dtype]
    # (K = 512 / 768 / 700: the 128-column form of the kernel, for 64 or 128 candidates; with 256 they stay on the streaming kernel)
    # (K = 320 / 256: five and four K-steps -- an odd step count ends a unit with the streamed fragment in the other buffer)
    for M, Ncols, K in ((768, 21 * 128 // P + 1, 384), (1152, 9, 136), (800, 70000 // P + 3, 100), (768, 11, 512), (1024, 5, 768),
                        (800, 7, 700), (832, 13, 320), (768, 6, 256)):
        Kp = CB.pad_k(K, dt_c)
        A = torch.zeros(1, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, Ncols * P, Kp, dtype=tdt)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).float().to(tdt)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).float().to(tdt)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3
        sa = torch.rand(1, generator=gen) * 0.002 + 0.001
        sb = torch.rand(P, Ncols, generator=gen) * 0.5 + 0.5
        rs = torch.rand(M, generator=gen) + 0.5; rb = torch.randn(M, generator=gen)
        b_n = torch.randn(Ncols, generator=gen); b_cn = torch.randn(P, Ncols, generator=gen)
        Ad, Bd = A.to(DEV), B.to(DEV)
        Ad.k_valid = K; Bd.k_valid = K
        for bias_kind in (("none", "n", "cn") if Ncols * P < 50000 else ("n",)):      # (the large case is slow on the CPU side)
            cb = {"none": None, "n": CB.Strided(b_n, n=1), "cn": CB.Strided(b_cn, c=Ncols, n=1)}[bias_kind]
            ob = {"none": None, "n": ops.Strided(b_n.to(DEV), n=1), "cn": ops.Strided(b_cn.to(DEV), c=Ncols, n=1)}[bias_kind]
            for rows in (False, True):
                for keep_n in (True, False):
                    want = CB.gemm_score(dt_c, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa), CB.Strided(sb, c=Ncols, n=1), cb,
                                         False, keep_n, 0.01, ref_div=P, ref_transposed=True,
                                         row_scale=rs if rows else None, row_bias=rb if rows else None)
                    ops.GEMM_EVENTS = []                                        # (the ctypes route reports the kernel behind a launch)
                    try:
                        got = ops.gemm_score(dt_o, Ad, Bd, M, Ncols, P, G, gmod, ref.to(DEV), ops.Strided(sa.to(DEV)),
                                             ops.Strided(sb.to(DEV), c=Ncols, n=1), ob, False, keep_n, 0.01, ref_div=P, order=2,
                                             ref_transposed=True, row_scale=rs.to(DEV) if rows else None,
                                             row_bias=rb.to(DEV) if rows else None)
                        kernel = ops.GEMM_EVENTS[-1][-1]
                    finally:
                        ops.GEMM_EVENTS = None
                    if K <= 384:
                        assert kernel.startswith("k_gemm_slab<"), kernel
                    elif P in (64, 128):
                        assert kernel.startswith("k_gemm_slab128<"), kernel
                    assert got.shape == want.shape
                    assert rel_err(got.cpu(), want) <= 3e-6, (M, Ncols, K, bias_kind, rows, keep_n, rel_err(got.cpu(), want))


@pytest.mark.parametrize("dtype", ["i8", "fp8"])
@pytest.mark.parametrize("P", [64, 128, 256])
def test_gemm_group_kernel(ops, dtype, P):
    """Group kernel (attention q.k^T searches: int8 or fp8 storage, one 64-byte K-step, 129..224 rows, many image x head groups, scores
    summed over the columns): ragged and exact last row block, a column count that leaves the last stage and the last
    chunk partly empty, head-wise and tensor-wise scores -- against the CPU specification."""
    gen = g(5200 + P)
    gmod = 6
    dt_c, dt_o = {"i8": (CB.I8, ops.I8), "fp8": (CB.FP8, ops.FP8)}[dtype]
    tdt = {"i8": torch.int8, "fp8": torch.float8_e4m3fn}[dtype]
    for M, Ncols, G, K in ((197, 197, 12, 64), (160, 37, 24, 48), (224, 9, 18, 64)):
        Kp = CB.pad_k(K, dt_c, 64)
        A = torch.zeros(1, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, Ncols * P, Kp, dtype=tdt)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).float().to(tdt)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).float().to(tdt)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3                      # stored [G, N, M] (transposed)
        sa = torch.rand(gmod, generator=gen) * 0.02 + 0.01
        sb = torch.rand(P, gmod, generator=gen) * 0.5 + 0.5
        Ad, Bd = A.to(DEV), B.to(DEV)
        Ad.k_valid = K; Bd.k_valid = K
        for keep_h in (True, False):
            want = CB.gemm_score(dt_c, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa, g=1), CB.Strided(sb, c=gmod, g=1), None,
                                 keep_h, False, 0.01, sa_mul=0.5, ref_div=P, ref_transposed=True)
            got = ops.gemm_score(dt_o, Ad, Bd, M, Ncols, P, G, gmod, ref.to(DEV), ops.Strided(sa.to(DEV), g=1),
                                 ops.Strided(sb.to(DEV), c=gmod, g=1), None, keep_h, False, 0.01, sa_mul=0.5, ref_div=P,
                                 order=2, ref_transposed=True)
            assert got.shape == want.shape and rel_err(got.cpu(), want) <= 3e-6, (M, Ncols, G, K, keep_h, rel_err(got.cpu(), want))


@pytest.mark.parametrize("dtype", ["i8", "fp8"])
@pytest.mark.parametrize("P", [64, 128])
def test_gemm_window_kernel(ops, dtype, P):
    """Window kernel (attention searches of windowed models: <= 64 rows, one K-step, hundreds of (window, head) groups; a wave
    per group, 32-byte rows when K <= 32): two / one row blocks, ragged rows and K, head counts that do and do not divide the
    wave count, head-wise and tensor-wise scores -- against the CPU specification."""
    gen = g(7400 + P)
    dt_c, dt_o = {"i8": (CB.I8, ops.I8), "fp8": (CB.FP8, ops.FP8)}[dtype]
    tdt = {"i8": torch.int8, "fp8": torch.float8_e4m3fn}[dtype]
    for M, Ncols, G, K, gmod in ((49, 49, 512, 32, 4), (49, 49, 384, 32, 3), (33, 17, 300, 20, 6), (64, 64, 256, 32, 32), (20, 9, 320, 64, 8)):
        use32 = K <= 32 and ops.gemm_win_ok(dt_o, M, Ncols, G, gmod, P, K)
        assert ops.gemm_win_ok(dt_o, M, Ncols, G, gmod, P, K) == (K <= 32)          # (the query is about 32-byte rows)
        Kp = 32 if use32 else CB.pad_k(K, dt_c, 64)
        A = torch.zeros(1, G, M, Kp, dtype=tdt); B = torch.zeros(1, G, Ncols * P, Kp, dtype=tdt)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).float().to(tdt)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).float().to(tdt)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3                      # stored [G, N, M] (transposed)
        sa = torch.rand(gmod, generator=gen) * 0.02 + 0.01
        sb = torch.rand(P, gmod, generator=gen) * 0.5 + 0.5
        Ad, Bd = A.to(DEV), B.to(DEV)
        Ad.k_valid = K; Bd.k_valid = K
        for keep_h in (True, False):
            want = CB.gemm_score(dt_c, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa, g=1), CB.Strided(sb, c=gmod, g=1), None,
                                 keep_h, False, 0.01, sa_mul=0.5, ref_div=P, ref_transposed=True)
            ops.GEMM_EVENTS = []                                                # (the ctypes route reports the kernel behind a launch)
            try:
                got = ops.gemm_score(dt_o, Ad, Bd, M, Ncols, P, G, gmod, ref.to(DEV), ops.Strided(sa.to(DEV), g=1),
                                     ops.Strided(sb.to(DEV), c=gmod, g=1), None, keep_h, False, 0.01, sa_mul=0.5, ref_div=P,
                                     order=2, ref_transposed=True)
                kernel = ops.GEMM_EVENTS[-1][-1]
            finally:
                ops.GEMM_EVENTS = None
            assert kernel.startswith("k_gemm_win"), kernel
            assert got.shape == want.shape and rel_err(got.cpu(), want) <= 3e-6, (M, Ncols, G, K, keep_h, rel_err(got.cpu(), want))


@pytest.mark.parametrize("P", [64, 128, 256])
def test_gemm_group_kernel_k_steps(ops, P):
    """Group kernel with seven K-steps (softmax.v weight search: bf16, K = 197, 129..224 rows, 64 x P columns per
    group): ragged and exact row blocks, ragged K, chunks that end inside the column range, head-wise and tensor-wise
    scores -- against the CPU specification."""
    gen = g(6300 + P)
    gmod = 6
    for M, Ncols, G, K in ((197, 64, 12, 197), (160, 11, 24, 200), (224, 5, 18, 224)):
        Kp = CB.pad_k(K, CB.BF16, 64)
        A = torch.zeros(1, G, M, Kp, dtype=torch.bfloat16); B = torch.zeros(1, G, Ncols * P, Kp, dtype=torch.bfloat16)
        A[..., :K] = torch.randint(-15, 16, (1, G, M, K), generator=gen).to(torch.bfloat16)
        B[..., :K] = torch.randint(-15, 16, (1, G, Ncols * P, K), generator=gen).to(torch.bfloat16)
        ref = torch.randn(G, Ncols, M, generator=gen) * 3                      # stored [G, N, M] (transposed)
        sa = torch.rand(gmod, generator=gen) * 0.02 + 0.01
        sb = torch.rand(P, gmod, generator=gen) * 0.5 + 0.5
        Ad, Bd = A.to(DEV), B.to(DEV)
        Ad.k_valid = K; Bd.k_valid = K
        for keep_h in (True, False):
            want = CB.gemm_score(CB.BF16, A, B, M, Ncols, P, G, gmod, ref, CB.Strided(sa, g=1), CB.Strided(sb, c=gmod, g=1), None,
                                 keep_h, False, 0.01, sa_mul=0.5, ref_div=P, ref_transposed=True)
            got = ops.gemm_score(ops.BF16, Ad, Bd, M, Ncols, P, G, gmod, ref.to(DEV), ops.Strided(sa.to(DEV), g=1),
                                 ops.Strided(sb.to(DEV), c=gmod, g=1), None, keep_h, False, 0.01, sa_mul=0.5, ref_div=P,
                                 order=2, ref_transposed=True)
            assert got.shape == want.shape and rel_err(got.cpu(), want) <= 3e-6, (M, Ncols, G, K, keep_h, rel_err(got.cpu(), want))


def test_sharded_select_matches_fused(ops):
    """adalog_select_* (histograms summed over emulated ranks between counting and pick) == the fused single-process
    kernels == the CPU specification, for every shard/chunk layout the grids use, and for the positive percentile."""
    gen = g(77)
    qs = [0.9, 1.0, 1 - 0.9, 0.0]

    def run(mod, shards, S, layouts, n_total, mbs, dev):
        lohi, w = mod.quantile_ranks(qs, n_total)
        sels = [mod.ShardedSelect(x2.to(dev), S, 8, *lay, ranks=lohi) for x2, lay in zip(shards, layouts)]
        for p in range(4):
            for s in sels:
                s.hist_pass(p)
            total = sum(s.hist.clone() for s in sels)
            for s in sels:
                s.hist.copy_(total)
                s.pick(p)
        return sels[0].quantiles(w, mbs)

    x = torch.randn(4, 50000, generator=gen)
    cases = [
        ([x[0:1].reshape(1, -1), x[1:2].reshape(1, -1), x[2:3].reshape(1, -1), x[3:4].reshape(1, -1)], 1, [(0, 1, 0)] * 4, 200000, 1,
         x.reshape(1, -1), 1),                                                       # one segment over 4 ranks
        ([x[0:2], x[2:4]], 4, [(0, 2, 4), (2, 2, 4)], 50000, 4, x, 4),                # 2 chunks per rank, mean of 4 chunks
        ([x[:, :20000].contiguous(), x[:, 20000:].contiguous()], 4, [(0, 4, 0), (0, 4, 0)], 50000, 1, x, 1),   # channel-wise
    ]
    for shards, S, layouts, n_total, mbs, full, fmbs in cases:
        got = run(ops, shards, S, layouts, n_total, mbs, DEV).cpu()
        fused = ops.quantile_rows(full.to(DEV), qs, fmbs).cpu()
        spec = run(CB, shards, S, layouts, n_total, mbs, "cpu")
        assert torch.equal(got, fused), (S, mbs)
        torch.testing.assert_close(got, spec, rtol=1e-6, atol=0)
    # positive percentile: rank from the GLOBAL count of positive entries
    y = torch.nn.functional.gelu(2 * torch.randn(3, 40000, generator=gen))
    pq = torch.tensor([0.9, 1.0]).tolist()
    sels = [ops.ShardedSelect(y[i:i + 1].to(DEV), 1, 2, 0, 1, 0, qfrac=pq) for i in range(3)]
    for p in range(4):
        for s in sels:
            s.hist_pass(p)
        total = sum(s.hist.clone() for s in sels)
        for s in sels:
            s.hist.copy_(total)
            s.pick(p)
    got = sels[0].values().cpu()
    want = ops.positive_percentile_rows(y.reshape(1, -1).to(DEV), pq).cpu()
    assert torch.equal(got, want)


def test_topk_next_matches_two_kernels(ops):
    """adalog_topk_next (one launch) == adalog_topk + adalog_fpcs_next, for expansion and commit, with ties and NaNs."""
    gen = g(91)
    for cols in (1, 7, 1152):
        P, k, new_cnt = 128, 16, 8
        scores = torch.randn(P, cols, generator=gen)
        scores[5] = scores[9]                                    # exact ties
        if cols > 1:
            scores[3, 1] = float("nan")
        scale = torch.rand(P, cols, generator=gen) + 0.1
        zp = torch.randint(0, 16, (P, cols), generator=gen).float()
        third = torch.rand(P, cols, generator=gen)
        lin = torch.linspace(0, 1, new_cnt)
        delta = torch.rand(cols, generator=gen) * 0.01 + 0.001
        for clamp in (None, 0.3):
            d1, d2 = delta.clone().to(DEV), delta.clone().to(DEV)
            idx = ops.topk(scores.to(DEV), k)
            a = ops.fpcs_next(scale.to(DEV), zp.to(DEV), third.to(DEV), idx, k, new_cnt, lin.to(DEV), d1, clamp)
            b = ops.topk_next(scores.to(DEV), scale.to(DEV), zp.to(DEV), third.to(DEV), k, new_cnt, lin.to(DEV), d2, clamp)
            for x, y in zip(a, b):
                assert torch.equal(x, y)
            assert torch.equal(d1, d2)
        idx1 = ops.topk(scores.to(DEV), 1)
        a = ops.fpcs_next(scale.to(DEV), zp.to(DEV), None, idx1, 1, 0, None, None, None)
        b = ops.topk_next(scores.to(DEV), scale.to(DEV), zp.to(DEV), None, 1, 0, None, None, None)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] is None and b[2] is None


# ------------------------------------------------------------------------------------------------ fused activation search
def _postgelu_layer(I, Oc, T, N, bits, seed, tie_frac=0.0, const_x=False):
    from adalog_amd import quant_layers as Q
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.gelu(2.0 * torch.randn(N, T, I, generator=g))
    if const_x:
        x = torch.full((N, T, I), 0.731)
    W = torch.randn(Oc, I, generator=g) * 0.03
    b = torch.randn(Oc, generator=g) * 0.1
    lay = Q.PostGeluLogBasedBatchingQuantLinear(I, Oc, True, "raw", bits, bits, calib_batch_size=N, search_round=1,
                                                eq_n=128, n_V=1, quantizer="adalog", fpcs=True, steps=6).to(DEV)
    lay.weight.data.copy_(W)
    lay.bias.data.copy_(b)
    shift = float(torch.tensor(0.16997124254703522, dtype=torch.float32))
    # candidates as activation_fpcs builds them: 16 scales x 8 bases
    hi = float(x.max()) + shift
    scs = (torch.linspace(0.55 * hi, hi, 16)).repeat(8)
    qs = torch.tensor([13., 23., 31., 37., 45., 60., 90., 137.]).repeat_interleave(16)
    if tie_frac > 0:                                   # plant values ON rounding ties of random (candidate, bin) pairs
        n = int(tie_frac * x.numel())
        idx = torch.randint(0, x.numel(), (n,), generator=g)
        pc = torch.randint(0, 128, (n,), generator=g)
        kb = torch.randint(0, 2 ** bits - 1, (n,), generator=g).float() + 0.5
        xs = scs[pc].double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), -(kb.double() * qs[pc].double() / 37.0))
        x.view(-1)[idx] = (xs - shift).float()
    lay.raw_input = x.to(DEV)
    lay.raw_out = torch.nn.functional.linear(x, W, b).to(DEV)
    lay.w_quantizer.scale.data.fill_(float(W.abs().max()) / 7.0)
    lay.w_quantizer.zero_point.data.fill_(float(2 ** (bits - 1)))
    lay.w_quantizer.inited = True
    return lay, scs.view(-1, 1).contiguous().to(DEV), qs.view(-1, 1).contiguous().to(DEV)


@pytest.mark.parametrize("I,Oc,T,N,bits,tie,const", [
    (1536, 384, 197, 4, 4, 0.0, False),      # deit_small fc2 (12 row blocks, one row tile)
    (1536, 384, 197, 3, 4, 0.05, False),     # ties planted: queue, exact path and the rank-1 fix-ups all run; odd token count
    (768, 192, 50, 2, 3, 0.05, False),       # deit_tiny (6 row blocks)
    (512, 128, 49, 8, 3, 0.02, False),       # swin stage 0 (4 row blocks)
    (1024, 256, 33, 3, 6, 0.02, False),      # 6 bit: 66-row LUT, 8 row blocks
    (3072, 768, 41, 2, 4, 0.02, False),      # vit_base: two row tiles of 384
    (200, 200, 37, 3, 4, 0.05, False),       # ragged: K and M not multiples of 32
    (256, 96, 31, 2, 4, 0.0, True),          # constant input: every lane ties with the same candidates (queue flood)
])
def test_score_act_fused_matches_packed_path(I, Oc, T, N, bits, tie, const):
    """gemm_fused.hip against pack_adalog + gemm_score (itself pinned to the reference's traces and to the oracle)."""
    from adalog_amd.quant_layers import linear as LM
    from adalog_amd.ops import BF16
    from adalog_amd import backend
    be = backend.get()
    lay, scs, qs = _postgelu_layer(I, Oc, T, N, bits, 1234 + I + T, tie, const)
    aq = lay.a_quantizer
    with torch.no_grad():
        wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
        fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data, lay.bias.data).view(-1)
        assert be.score_act_fused_ok(Oc, N * T, I, wp.shape[-1], 128, bits)
        LM.FUSED_ACT_SEARCH = False
        try:
            ref = lay._score_scale_logbase(wp, fold, scs, qs)
        finally:
            LM.FUSED_ACT_SEARCH = True
        got = lay._score_scale_logbase(wp, fold, scs, qs)
        got2 = lay._score_scale_logbase(wp, fold, scs, qs)
    assert torch.equal(got, got2)                                     # bit-reproducible
    err = ((got - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
    assert err <= 2e-6, err


def test_torch_ops_route_matches_c_abi():
    """torch.ops.adalog.* (TORCH_LIBRARY shims) give bit-identical results to the ctypes route of the same C ABI, and
    adalog_amd.ops goes through them when the extension is built."""
    from adalog_amd import _torch_ops, ops as OPS
    assert _torch_ops.available(), "libadalog_torch.so not built (python adalog_amd/csrc/build_torch_ops.py)"
    g = torch.Generator().manual_seed(4)
    x = torch.randn(64, 197, 96, generator=g).to(DEV)
    s = torch.rand(96, generator=g).to(DEV) * 0.1 + 0.05
    z = torch.randint(3, 12, (96,), generator=g).float().to(DEV)
    y_t = torch.ops.adalog.uniform_fake_quant(x, s, z, 96, 1, 4, False)
    y_c, _ = OPS.uniform_fake_quant(x, s, z, 4, want_bins=True)            # want_bins forces the ctypes route
    assert torch.equal(y_t, y_c) and torch.equal(OPS.uniform_fake_quant(x, s, z, 4), y_c)
    sc = torch.rand(128, 1, generator=g).to(DEV)
    assert torch.equal(torch.ops.adalog.topk(sc, 16), OPS.topk(sc, 16))
    xg = torch.nn.functional.gelu(x)
    assert torch.equal(torch.ops.adalog.log2_shift(xg, 0.17), OPS.log2_shift(xg, 0.17))
    # one scoring call of each kind through BOTH routes (bench.py times the ctypes route, the product runs the torch.ops one):
    # gemm_score (+ deferred finish and the fused finish / top-k), the in-kernel-generated activation search, the sorted self-MSE search
    M, T, K, P = 768, 4 * 197, 384, 128
    xa = (torch.randn(T, K, generator=g) * 1.3).to(DEV)
    W = (torch.randn(M, K, generator=g) * 0.05).to(DEV)
    ref = torch.nn.functional.linear(xa, W)
    ws, wz = torch.full((M,), 0.01, device=DEV), torch.full((M,), 8.0, device=DEV)
    cs = (torch.rand(P, 1, generator=g) * 0.2 + 0.2).to(DEV)
    cz = torch.randint(4, 12, (P, 1), generator=g).float().to(DEV)
    lin8 = torch.linspace(0, 1, 8).to(DEV)

    def run_all():
        wp = OPS.pack_uniform(W.unsqueeze(0), ws, wz, 1, 0, 1, 0, 1, 4, OPS.FP8)
        xp = OPS.pack_uniform(xa.unsqueeze(0), cs, cz, P, 1, 1, 0, 0, 4, OPS.FP8, c_inner=True)
        one = torch.ones(1, device=DEV)
        kw = dict(ref_div=P, order=2, ref_transposed=True, row_scale=ws, row_bias=torch.zeros(M, device=DEV))
        a = OPS.gemm_score(OPS.FP8, wp, xp, M, T, P, 1, 1, ref.reshape(1, T, M), OPS.Strided(one), OPS.Strided(cs, c=1), None, False, False,
                           1e-3, **kw)
        pend = OPS.gemm_score(OPS.FP8, wp, xp, M, T, P, 1, 1, ref.reshape(1, T, M), OPS.Strided(one), OPS.Strided(cs, c=1), None, False,
                              False, 1e-3, defer=True, **kw)
        b = pend.finish()
        d = torch.full((1,), 0.01, device=DEV)
        nxt = OPS.finish_topk_next(pend, cs, cz, None, 16, 8, lin8, d, 1e-4)
        gsc = OPS.score_act_gen(OPS.FP8, wp, xa, cs, cz, 4, ref, ws, None, 1e-3)
        sp = OPS.sorted_prefix(xa.reshape(1, -1))
        ss = OPS.score_self_sorted(sp, cs, cz, 4, 1e-3)
        return [a, b, nxt[0], nxt[1], d, gsc, sp.sorted, sp.prefix, ss]

    via_torch = run_all()
    _torch_ops._state = False                                     # force the ctypes route of adalog_amd.ops
    try:
        via_ctypes = run_all()
    finally:
        _torch_ops._state = True
    for i, (u, v) in enumerate(zip(via_torch, via_ctypes)):
        assert torch.equal(u, v), f"torch.ops and ctypes routes disagree on result {i}"


@pytest.mark.parametrize("shape", [(4, 7, 33), (32, 197, 384), (3, 5)])
def test_rec_loss_forward_backward(ops, shape):
    """Fused BRECQ reconstruction loss (lp_loss, p = 2) against the ATen composition the reference runs."""
    gen = g(77)
    pred = torch.randn(*shape, generator=gen)
    tgt = pred + 0.1 * torch.randn(*shape, generator=gen)
    scale = 0.1 * shape[1] / pred.numel()
    p_ref = pred.clone().requires_grad_(True)
    l_ref = (p_ref - tgt).abs().pow(2.0).sum(1).mean() / 10
    l_ref.backward()
    l = ops.rec_loss(pred.to(DEV), tgt.to(DEV), scale)
    torch.testing.assert_close(l.cpu().view(()), l_ref.detach(), rtol=2e-6, atol=0)
    gp = ops.rec_loss_backward(pred.to(DEV), tgt.to(DEV), scale, torch.tensor([1.0]).to(DEV))
    torch.testing.assert_close(gp.cpu(), p_ref.grad, rtol=1e-5, atol=1e-9)


def test_round_loss_multi_matches_per_tensor(ops):
    """One launch for all of a block's alpha tensors == the per-tensor kernel, value and gradients."""
    gen = g(78)
    alphas = [torch.randn(n, generator=gen) * 2 for n in (147456, 1152 * 384, 5, 384 * 1536 + 3)]
    for b in (20.0, 7.3, 2.0):
        l_ref, g_ref = CB.round_loss_multi(alphas, b, 0.01)
        loss, grads = ops.round_loss_multi([a.to(DEV) for a in alphas], b, 0.01)
        torch.testing.assert_close(loss.cpu(), l_ref, rtol=1e-5, atol=1e-4)
        for gg, gr in zip(grads, g_ref):
            torch.testing.assert_close(gg.cpu(), gr, rtol=1e-3, atol=1e-7)
    loss_b, _ = ops.round_loss_multi([a.to(DEV) for a in alphas], torch.tensor([2.0]).to(DEV), 0.01)    # exponent on the device
    torch.testing.assert_close(loss_b, loss)


# ------------------------------------------------------------------------------------------------ radix sort (csrc/radix_sort.hip)
@pytest.mark.parametrize("S,n", [(1, 1), (1, 100), (3, 8192), (5, 8191), (384, 6304), (1152, 384), (1, 8193), (2, 100352), (1, 2420736),
                                 (7, 20000)])
def test_sort_f32_matches_a_stable_sort(ops, S, n):
    """adalog_sort_f32 (the hand-written LSD radix sort that replaced hipCUB under the sorted self-MSE searches and the Gram
    activation form): every segment ascending and equal to torch.sort's values bit for bit, the permutation the STABLE one (equal
    values in input order).  Data with many duplicates, both signs, denormals, zeros and huge values; both the one-launch LDS form
    (n <= 8192) and the tiled form."""
    gen = g(9100 + S + n)
    x = torch.randn(S, n, generator=gen)
    x[:, ::3] = torch.round(x[:, ::3] * 4) / 4 + 0.0                # duplicates (+ 0.0: no -0.0, which the bit order puts before +0.0
                                                                    # while torch.sort calls the two equal and keeps their input order)
    if n > 16:
        x[:, 5] = 0.0; x[:, 6] = 1e-42; x[:, 7] = -1e-42; x[:, 8] = 3e38; x[:, 9] = -3e38; x[:, 10] = 0.0
    xd = x.to(DEV)
    out, perm = ops.sort_f32(xd, want_perm=True)
    ref, ridx = torch.sort(x, dim=1, stable=True)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(perm.cpu().long(), ridx)
    out2, none = ops.sort_f32(xd, want_perm=False)
    assert none is None and torch.equal(out2, out)


def test_sorted_prefix_and_gram_act_prepare_use_no_library_sort(ops):
    """The two call sites that sorted with hipCUB (rocPRIM kernels) until round 5 launch none now."""
    from torch.profiler import ProfilerActivity, profile
    gen = g(9200)
    x = torch.randn(6304, 384, generator=gen).to(DEV)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        sp1 = ops.sorted_prefix(x.view(1, -1))
        sp2 = ops.sorted_prefix(x.t().contiguous())
        prep = ops.GramActPrepared(x)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    assert not any("rocprim" in n_ or "hipcub" in n_ for n_ in names), [n_ for n_ in names if "rocprim" in n_ or "hipcub" in n_]
    assert any("k_rs_" in n_ for n_ in names), names
    ref = torch.sort(x.view(-1).cpu())[0]
    assert torch.equal(sp1.sorted.view(-1).cpu(), ref) and torch.equal(prep.sorted.cpu(), ref)
    assert torch.equal(x.view(-1)[prep.perm.long()].cpu(), ref)
    assert torch.equal(sp2.sorted.cpu(), torch.sort(x.t().cpu(), dim=1)[0])


# ------------------------------------------------------------------------------------------------ K17b: BRECQ contractions
def _last_kernel():
    from adalog_amd import _lib
    return _lib.load().adalog_last_kernel().decode()


@pytest.mark.parametrize("M,N,K", [(64, 128, 16), (100, 36, 40), (300, 260, 52), (1000, 16, 1000), (33, 1000, 384), (129, 1028, 36),
                                   (197, 197, 64), (6304, 384, 384)])
def test_gemm_f32x3_every_orientation(ops, M, N, K):
    """adalog_gemm_f32x3 against its fp64 specification: both operands in both memory orientations, with and without the
    fused bias, ragged M / N / K (K tails, N off a multiple of 4), at fp32-class accuracy (rocBLAS fp32 is at 5e-7 .. 3e-6)."""
    gen = g(7000 + M + N + K)
    for ta in (0, 1):
        for tb in (0, 1):
            for with_bias in (False, True):
                if with_bias and N % 16:
                    continue
                a = (torch.randn(K, M, generator=gen).to(DEV).t() if ta else torch.randn(M, K, generator=gen).to(DEV))
                b = (torch.randn(K, N, generator=gen).to(DEV).t() if tb else torch.randn(N, K, generator=gen).to(DEV))
                bias = torch.randn(N, generator=gen).to(DEV) if with_bias else None
                assert ops.gemm_f32x3_ok(a, b, bias)
                out = ops.gemm_f32x3(a, b, bias)
                assert _last_kernel().startswith("bq_gemm<")
                ref = CB.gemm_f32x3(a.cpu(), b.cpu(), None if bias is None else bias.cpu()).double()
                assert rel_err(out.cpu().double(), ref) <= 2e-6, (M, N, K, ta, tb, with_bias)


def test_gemm_f32x3_integer_operand_and_planes(ops):
    """The integer-activation forms (forward: A = x_int K-contiguous; dL/dw: B = x_int K-major; scale as a device scalar) and the
    pre-split weight planes give the same product as the general form."""
    gen = g(7100)
    for (M, N, K) in [(300, 64, 48), (6304, 384, 384), (200, 144, 40)]:
        xi = torch.randint(-15, 16, (M, K), generator=gen).float().to(DEV)
        w = torch.randn(N, K, generator=gen).to(DEV)
        sc = torch.tensor([0.37], device=DEV)
        bias = torch.randn(N, generator=gen).to(DEV) if N % 16 == 0 else None
        ref = CB.gemm_f32x3(xi.cpu(), w.cpu(), None if bias is None else bias.cpu(), alpha_dev=sc.cpu()).double()
        out = ops.gemm_f32x3(xi, w, bias, alpha_dev=sc, exact_a=True)
        assert rel_err(out.cpu().double(), ref) <= 2e-6
        wp = ops.pack_split3(w.view(1, N, K), 64)
        for ex in (False, True):
            outp = ops.gemm_f32x3_planes(xi, wp, K, bias, alpha_dev=sc, exact_a=ex)
            assert rel_err(outp.cpu().double(), ref) <= 2e-6
        gy = torch.randn(M, N, generator=gen).to(DEV)                      # dL/dw = s * gy^T . x_int
        gw = ops.gemm_f32x3(gy.t(), xi.t(), alpha_dev=sc, exact_b=True)
        refw = CB.gemm_f32x3(gy.cpu().t(), xi.cpu().t(), alpha_dev=sc.cpu()).double()
        assert rel_err(gw.cpu().double(), refw) <= 2e-6


@pytest.mark.parametrize("M,N,K", [(100, 36, 40), (300, 260, 52), (197, 197, 64), (6304, 384, 384), (384, 1536, 6304)])
def test_gemm_f32x3_two_term_operands(ops, M, N, K):
    """exact = 2 (round 6: the gradient contractions of a BRECQ iteration): two bf16 terms per general operand -- three MFMA products
    (hi.hi, hi.mid, mid.hi), two against an integer operand -- in every orientation, against the fp64 product.  The dropped terms
    are 2^-16 relative per factor: the result stays within 6e-5 of the fp64 product (rms-normalised, like the three-term test's
    2e-6) and is far more accurate than a plain bf16 product (4e-3)."""
    gen = g(7150 + M + N + K)
    for ta in (0, 1):
        for tb in (0, 1):
            a = (torch.randn(K, M, generator=gen).to(DEV).t() if ta else torch.randn(M, K, generator=gen).to(DEV))
            b = (torch.randn(K, N, generator=gen).to(DEV).t() if tb else torch.randn(N, K, generator=gen).to(DEV))
            assert ops.gemm_f32x3_ok(a, b, None)
            out = ops.gemm_f32x3(a, b, exact_a=2, exact_b=2)
            assert _last_kernel().startswith("bq_gemm<")
            ref = CB.gemm_f32x3(a.cpu(), b.cpu()).double()
            err = rel_err(out.cpu().double(), ref)
            assert err <= 6e-5, (M, N, K, ta, tb, err)
    # dL/dw against the integer activation: gy two-term, x_int exact (two products)
    xi = torch.randint(-15, 16, (M, K), generator=gen).float().to(DEV)
    gy = torch.randn(M, N, generator=gen).to(DEV)
    sc = torch.tensor([0.37], device=DEV)
    gw = ops.gemm_f32x3(gy.t(), xi.t(), alpha_dev=sc, exact_a=2, exact_b=True)
    refw = CB.gemm_f32x3(gy.cpu().t(), xi.cpu().t(), alpha_dev=sc.cpu()).double()
    assert rel_err(gw.cpu().double(), refw) <= 6e-5


def test_gemm_f32x3_split_k_is_bit_reproducible(ops):
    """dL/dw-shaped products are split along K and reduced in a fixed order: two launches give identical bits."""
    gen = g(7200)
    gy = torch.randn(6304, 1152, generator=gen).to(DEV)
    x = torch.randn(6304, 384, generator=gen).to(DEV)
    o1 = ops.gemm_f32x3(gy.t(), x.t())
    o2 = ops.gemm_f32x3(gy.t(), x.t())
    assert torch.equal(o1, o2)
    ref = (gy.double().t() @ x.double()).cpu()
    assert rel_err(o1.cpu().double(), ref) <= 2e-6


def test_gemm_f32x3_attention_products(ops):
    """The four attention products of a BRECQ iteration at 197 tokens (rows not 16-byte aligned, N off a multiple of 4)."""
    gen = g(7300)
    for (G, S, C) in [((2, 6), 197, 64), ((3, 4), 49, 32)]:
        q = torch.randn(*G, S, C, generator=gen).to(DEV)
        kt = torch.randn(*G, C, S, generator=gen).to(DEV)
        v = torch.randn(*G, S, C, generator=gen).to(DEV)
        s_ = ops.gemm_f32x3(q, kt.transpose(-1, -2))
        assert rel_err(s_.cpu().double(), (q.double() @ kt.double()).cpu()) <= 2e-6
        pr = torch.softmax(s_, -1)
        o = ops.gemm_f32x3(pr, v.transpose(-1, -2))
        assert rel_err(o.cpu().double(), (pr.double() @ v.double()).cpu()) <= 2e-6
        gy = torch.randn(*G, S, C, generator=gen).to(DEV)
        gp = ops.gemm_f32x3(gy, v)
        assert rel_err(gp.cpu().double(), (gy.double() @ v.double().transpose(-1, -2)).cpu()) <= 2e-6
        gv = ops.gemm_f32x3(pr.transpose(-1, -2), gy.transpose(-1, -2))
        assert rel_err(gv.cpu().double(), (pr.double().transpose(-1, -2) @ gy.double()).cpu()) <= 2e-6


@pytest.mark.parametrize("B,H,N,D", [(32, 6, 197, 64), (3, 4, 49, 32), (2, 3, 70, 20)])
def test_gemm_f32x3_two_level_groups(ops, B, H, N, D):
    """adalog_gemm_f32x3_g2: operands whose [B][H] strides do not collapse -- q / k / v read in place from a [B, N, 3, H, D]
    tensor, and a result written as [B, N, H, D] storage through its [B, H, N, D] view -- against fp64; then the two backward
    products of softmax.v with the gradient arriving as such a view (what train_mm._MatmulFn does with heads_last)."""
    gen = torch.Generator().manual_seed(B * H + N)
    qkv = torch.randn(B, N, 3, H, D, generator=gen).to(DEV)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)                    # [B, H, N, D] views: strides (N*3HD, D, 3HD, 1)
    assert not q.is_contiguous()
    if ops.gemm_f32x3_ok(q, k):
        s = ops.gemm_f32x3(q, k)                                        # q . k^T, both read in place
        assert rel_err(s.cpu().double(), (q.double() @ k.double().transpose(-1, -2)).cpu()) <= 2e-6
    else:
        assert (3 * H * D) % 4 or qkv.data_ptr() % 16 or (D * 4) % 16
    pr = torch.softmax(torch.randn(B, H, N, N, generator=gen), -1).to(DEV)
    vc = v.contiguous()
    out = torch.empty(B, N, H, D, device=DEV).permute(0, 2, 1, 3)
    assert ops.gemm_f32x3_ok(pr, vc.transpose(-1, -2), None, out)
    r = ops.gemm_f32x3(pr, vc.transpose(-1, -2), out=out)
    assert r.data_ptr() == out.data_ptr()
    want = (pr.double() @ vc.double()).cpu()
    assert rel_err(out.cpu().double(), want) <= 2e-6
    merged = out.transpose(1, 2).reshape(B, N, H * D)                   # the view the projection layer reads
    assert merged.data_ptr() == out.data_ptr()
    gy = torch.randn(B, N, H, D, generator=gen).to(DEV).permute(0, 2, 1, 3)     # gradient arriving as the [B, H, N, D] view
    assert ops.gemm_f32x3_ok(gy, vc) and ops.gemm_f32x3_ok(pr.transpose(-1, -2), gy.transpose(-1, -2))
    gp = ops.gemm_f32x3(gy, vc)
    assert rel_err(gp.cpu().double(), (gy.double() @ vc.double().transpose(-1, -2)).cpu()) <= 2e-6
    gv = ops.gemm_f32x3(pr.transpose(-1, -2), gy.transpose(-1, -2))
    assert rel_err(gv.cpu().double(), (pr.double().transpose(-1, -2) @ gy.double()).cpu()) <= 2e-6


@pytest.mark.parametrize("shape", [(32, 197, 3, 6, 64), (5, 49, 3, 4, 32), (2, 7, 1, 3, 4)])
def test_permute_heads_matches_the_view_route(ops, shape):
    """adalog_permute_heads ([B, N, P*H*D] -> [P, B, H, N, D] and back) moves exactly what reshape / permute / unbind move
    (reference utils/wrap_net.py:21-22), and train_mm.split_heads' gradient is the view route's gradient bit for bit."""
    from adalog_amd import train_mm
    B, N, P, H, D = shape
    gen = torch.Generator().manual_seed(B * N + D)
    x = torch.randn(B, N, P * H * D, generator=gen).to(DEV)
    y = ops.permute_heads(x, P, H)
    assert torch.equal(y.cpu(), CB.permute_heads(x.cpu(), P, H))
    assert torch.equal(ops.permute_heads(y, P, H, inverse=True), x)
    assert torch.equal(ops.merge_heads([y[i] for i in range(P)], B, N, H, D), x)
    if P > 1:                                                            # a missing part counts as zeros
        got = ops.merge_heads([None] + [y[i] for i in range(1, P)], B, N, H, D).cpu()
        assert torch.equal(got, CB.merge_heads([None] + [y[i].cpu() for i in range(1, P)], B, N, H, D))
    ws = [torch.randn(B, H, N, D, generator=gen).to(DEV) for _ in range(P)]
    grads = []
    for on in (True, False):
        old, train_mm.SPLIT_HEADS = train_mm.SPLIT_HEADS, on
        try:
            xr = x.clone().requires_grad_(True)
            parts = train_mm.split_heads(xr, P, H)
            assert all(pt.shape == (B, H, N, D) for pt in parts)
            sum((pt * w).sum() for pt, w in zip(parts, ws)).backward()
            grads.append(xr.grad.clone())
        finally:
            train_mm.SPLIT_HEADS = old
    assert torch.equal(grads[0], grads[1])


def test_brecq_iteration_runs_no_library_gemm():
    """Every contraction of a BRECQ iteration of a transformer block runs on csrc/brecq_gemm.hip: the kernel trace of one
    forward/backward holds k_bq_gemm launches and no rocBLAS / hipBLASLt (`Cijk_*`) kernel."""
    import copy
    from torch.profiler import ProfilerActivity, profile
    from adalog_amd import backend
    from adalog_amd.utils.block_recon import BlockReconstructor
    from adalog_amd.utils.calibrator import QuantCalibrator
    from adalog_amd.utils.models import VisionTransformer
    from adalog_amd.utils.wrap_net import wrap_modules_in_net, wrap_reparamed_modules_in_net
    import importlib.util, os
    backend.set_backend(None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cfg4k", os.path.join(root, "configs", "4bit.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    cfg = mod.Config()
    cfg.search_round, cfg.steps = 1, 2
    torch.manual_seed(3)
    model = VisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=1, num_heads=2, num_classes=16).eval()
    for p_ in model.parameters():
        p_.data.mul_(6.0)
    model.to(DEV)
    full = copy.deepcopy(model)
    x = torch.randn(16, 3, 64, 64).to(DEV)
    loader = [(x[:8], None), (x[8:], None)]
    model = wrap_modules_in_net(model, cfg, reparam=True)
    QuantCalibrator(model, loader).batching_quant_calib()
    model = wrap_reparamed_modules_in_net(model)
    rec = BlockReconstructor(model, full, loader)
    name = "blocks.0"
    block, fblock = rec.blocks[name], rec.full_blocks[name]
    rec.init_block_raw_data(block, fblock, name, torch.device(DEV))
    os.environ["ADALOG_BRECQ_GRAPH"] = "0"
    from adalog_amd import train_mm
    taken = {"offers": 0, "hits": 0}
    real_offer, real_kmajor = train_mm.offer_kmajor, train_mm._kmajor

    def offer(w2, w2_t):
        taken["offers"] += 1
        return real_offer(w2, w2_t)

    def kmajor(w2):
        had = w2.data_ptr() in train_mm._KMAJOR_OFFER
        out = real_kmajor(w2)
        taken["hits"] += int(had and w2.data_ptr() not in train_mm._KMAJOR_OFFER and out.data_ptr() != w2.data_ptr())
        return out

    train_mm.offer_kmajor, train_mm._kmajor = offer, kmajor
    try:
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            rec.reconstruct_single_block(name, block, torch.device(DEV), batch_size=8, iters=3, quant_act=True)
            torch.cuda.synchronize()
    finally:
        os.environ.pop("ADALOG_BRECQ_GRAPH", None)
        train_mm.offer_kmajor, train_mm._kmajor = real_offer, real_kmajor
    names = [e.key for e in prof.key_averages()]
    assert any("k_bq_gemm" in n for n in names), names
    assert not any(n.startswith("Cijk_") for n in names), [n for n in names if n.startswith("Cijk_")]
    # the AdaRound forward of a training iteration leaves the K-major image of w_sim (k_adaround_t) and the layer's forward product
    # takes it: round 5 read the grad mode INSIDE the autograd Function (always off there) and every layer paid a transposing copy
    if train_mm.W_KMAJOR and train_mm.ENABLED:
        assert any("k_adaround_t" in n for n in names), [n for n in names if "adaround" in n]
        assert taken["offers"] >= 4 and taken["hits"] >= taken["offers"] - 1, taken


@pytest.mark.parametrize("bits,dt_name", [(4, "fp8"), (3, "fp8"), (6, "i8")])
@pytest.mark.parametrize("T,O,K", [(6304, 1152, 384), (6304, 384, 384), (1568, 768, 768)])
def test_score_w_gen_matches_packed_path(ops, bits, dt_name, T, O, K):
    """Weight-candidate scores with the candidate operand generated in the slab kernel (adalog_score_w_gen) against the packed
    path (pack_uniform + gemm_score) on the same candidates, with weight values planted on rounding ties."""
    from adalog_amd.ops import FP8, I8, Strided
    dt = FP8 if dt_name == "fp8" else I8
    gen = g(8000 + bits + T + O)
    P = 128
    x = torch.randn(T, K, generator=gen)
    W = torch.randn(O, K, generator=gen) * 0.05
    b = torch.randn(O, generator=gen) * 0.1
    qmax = 2 ** bits - 1
    w_lo, w_hi = W.min(1).values, W.max(1).values
    sc = ((w_hi - w_lo) / qmax).view(1, O) * torch.linspace(0.6, 1.2, P).view(P, 1)
    zp = torch.round(-w_lo.view(1, O) / sc).clamp(0, qmax)
    W[:, 5] = sc[17] * 2.5                                              # exact ties of candidate 17 (and near-ties of its neighbours)
    W[:, 9] = sc[90] * -1.5
    a_s = torch.tensor([x.abs().max().item() * 2 / qmax])
    a_z = torch.tensor([float(2 ** (bits - 1))])
    ref = torch.nn.functional.linear(x, W, b)
    xd, Wd = x.to(DEV), W.to(DEV)
    xp = ops.pack_uniform(xd.unsqueeze(0), a_s.to(DEV), a_z.to(DEV), 1, 0, 1, 0, 0, bits, dt)
    assert ops.score_w_gen_ok(dt, T, O, K, xp.shape[-1], P)
    ref_t = ref.t().contiguous().to(DEV)
    got = ops.score_w_gen(dt, xp, Wd, sc.to(DEV).contiguous(), zp.to(DEV).contiguous(), bits, ref_t, a_s.to(DEV), b.to(DEV), 1.0 / 197)
    assert _last_kernel().startswith(("k_gemm_slab_wgen<", "k_gemm_slab128_wgen<"))
    wp = ops.pack_uniform(Wd.unsqueeze(0), sc.to(DEV).contiguous(), zp.to(DEV).contiguous(), P, O, 1, 0, 1, bits, dt, c_inner=True)
    want = ops.gemm_score(dt, xp, wp, T, O, P, 1, 1, ref_t, Strided(a_s.to(DEV)), Strided(sc.to(DEV).contiguous(), c=O, n=1),
                          Strided(b.to(DEV), n=1), False, True, 1.0 / 197, ref_div=P, order=2, ref_transposed=True)
    assert got.shape == want.shape == (P, O)
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6


@pytest.mark.parametrize("bits,dt_name", [(4, "fp8"), (3, "fp8"), (6, "i8")])
@pytest.mark.parametrize("shape", [("win", 2048, 4, 49, 32, 32), ("win", 512, 8, 49, 32, 32), ("grpw", 192, 6, 197, 64, 64)])
def test_gemm_score_gen_matches_packed_path(ops, bits, dt_name, shape):
    """Attention-search scores with the candidate operand generated in the kernel (adalog_gemm_score_gen: window kernel for swin's
    49 x 49 x 32 windows, wave-private group kernel for 197 x 197 x 64) against the packed route (pack_uniform + gemm_score) on the
    same candidates, with source values planted on rounding ties.  Both searched operands: rows = queries and rows = keys."""
    from adalog_amd.ops import FP8, I8, Strided
    kind, G, H, S, K, al = shape
    dt = FP8 if dt_name == "fp8" else I8
    gen = g(9100 + bits + G + S)
    P = 128
    qmax = 2 ** bits - 1
    src = torch.randn(G, S, K, generator=gen) * 1.3 + 0.4                 # the searched operand
    fix = torch.randn(G, S, K, generator=gen)                            # the other operand
    lo, hi = src.amin(), src.amax()
    sc = ((hi - lo) / qmax) * torch.linspace(0.5, 1.1, P).view(P, 1) * torch.linspace(0.9, 1.1, H).view(1, H)
    zp = torch.round(-lo / sc).clamp(0, qmax)
    src[:, 3, 5] = (sc[17, 0] * 2.5).item()                              # exact ties of candidate 17 / head 0, near-ties of neighbours
    src[:, 7, 20] = (sc[90, H - 1] * -1.5).item()
    f_s = torch.full((H,), fix.abs().max().item() * 2 / qmax)
    f_z = torch.full((H,), float(2 ** (bits - 1)))
    pg = 1 if H > 1 else 0
    ref = torch.einsum("gsk,gtk->gst", src, fix)                          # [G, S(src rows), S(fixed rows)] = the transposed reference
    d = lambda t_: t_.to(DEV).contiguous()
    fp = ops.pack_uniform(d(fix), d(f_s), d(f_z), 1, 0, H, pg, 0, bits, dt, k_align=al)
    assert ops.gemm_score_gen_ok(dt, S, S, G, H, P, K, fp.shape[-1])
    sb = Strided(d(sc), c=H, g=pg)
    sa = Strided(d(f_s), g=pg)
    got = ops.gemm_score_gen(dt, fp, d(src), d(zp), bits, S, S, P, G, H, d(ref), sa, sb, True, 1.0 / (S * S)).finish()
    label = _last_kernel()
    assert label.startswith("k_gemm_win_gen<" if kind == "win" else "k_gemm_grpw_gen<"), label
    cand = ops.pack_uniform(d(src), d(sc), d(zp), P, H, H, pg, 0, bits, dt, c_inner=True, k_align=al)
    want = ops.gemm_score(dt, fp, cand, S, S, P, G, H, d(ref), sa, sb, None, True, False, 1.0 / (S * S), ref_div=P, order=2,
                          ref_transposed=True)
    assert not _last_kernel().endswith("_gen<fp8>") and "_gen" not in _last_kernel()
    assert got.shape == want.shape == (P, H)
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6


@pytest.mark.parametrize("bits", [4, 3, 6])
@pytest.mark.parametrize("shape", [("vit", 192, 6, 197, 64), ("win", 2048, 4, 49, 32), ("win16", 512, 16, 49, 32)])
def test_gemm_score_avq_matches_packed_path(ops, bits, shape):
    """Log-base scores of softmax.v with the 128 AdaLog quantisations of the probabilities generated in the kernel
    (adalog_gemm_score_avq) against the packed route (pack_adalog + bf16 gemm_score) on the same bases; probabilities planted on
    bin ties of some bases, exact zeros (masked) and values below the last bin included."""
    from adalog_amd.ops import BF16, Strided
    kind, G, H, S, D = shape
    gen = g(9300 + bits + G + S)
    P = 128
    A = torch.softmax(4.0 * torch.randn(G, S, S, generator=gen), dim=-1)
    A[:, 3, 5] = 0.0                                                     # masked code
    A[:, 4, 6] = 1e-30                                                   # below the last bin of every base
    for qq, kk in ((17, 2.5), (90, 0.5), (137, 1.5)):                    # t = k + 0.5 exactly representable -> the tie zone
        A[:, 7, qq % S] = float(2.0 ** (-(kk * qq / 37.0)))
    v = torch.randn(G, S, D, generator=gen) * (0.5 + torch.rand(1, 1, 1, generator=gen))
    b_s = torch.full((H,), v.abs().max().item() * 2 / (2 ** bits - 1))
    b_z = torch.full((H,), float(2 ** (bits - 1)))
    pg = 1 if H > 1 else 0
    ref = torch.einsum("gsk,gkd->gsd", A, v)                             # [G, S, D] = the transposed reference of D^T = v^T . A^T
    d = lambda t_: t_.to(DEV).contiguous()
    q_all = torch.arange(10, 10 + P).float()
    mant = torch.round(torch.tensor([2 ** (-j / 37.0) for j in range(37)]) * (4 * 2 ** (bits - 1) - 2))
    ones = torch.ones(P)
    vt = d(v.transpose(1, 2))                                            # [G, D, S]: rows = head-dim columns of v
    bp = ops.pack_uniform(vt, d(b_s), d(b_z), 1, 0, H, pg, 0, bits, BF16, k_align=64)
    assert ops.gemm_score_avq_ok(D, S, G, H, P, S, bp.shape[-1], bits)
    lut = ops.adalog_value_lut(d(q_all), bits, d(mant))
    sa, sb = Strided(d(b_s), g=pg), Strided(d(ones), c=1)
    ts = 1.0 / (4 * 2 ** (bits - 1) - 2)
    got = ops.gemm_score_avq(bp, d(A), d(q_all), lut, bits, D, S, P, G, H, d(ref), sa, sb, 1.0 / (S * D), sa_mul=ts)
    assert _last_kernel() == ("k_gemm_avq<13,bf16>" if S > 64 else "k_gemm_avq<4,bf16>")
    ap = ops.pack_adalog(d(A), d(ones), d(q_all), P, 1, 1, 0, bits, d(mant), shift=None, clamp_u=False, c_inner=True, k_align=64)
    want = ops.gemm_score(BF16, bp, ap, D, S, P, G, H, d(ref), sa, sb, None, False, False, 1.0 / (S * D), sa_mul=ts, ref_div=P,
                          order=2, ref_transposed=True)
    assert "avq" not in _last_kernel()
    assert got.shape == want.shape == (P, 1)
    assert rel_err(got.cpu(), want.cpu()) <= 2e-6


@pytest.mark.parametrize("rows,inner", [(1152, 384), (384, 1536), (100, 70), (33, 5)])
def test_adaround_t_writes_both_orientations(ops, rows, inner):
    """adalog_adaround_t: the AdaRound forward and its [inner][rows] image in one launch, both bit-equal to adalog_adaround."""
    gen = g(9500 + rows)
    w = torch.randn(rows, inner, generator=gen) * 0.05
    al = torch.randn(rows, inner, generator=gen) * 3.0
    s_ = (torch.rand(rows, generator=gen) * 0.01 + 0.005)
    z_ = torch.randint(4, 12, (rows,), generator=gen).float()
    d = lambda t_: t_.to(DEV)
    for soft in (True, False):
        want = ops.adaround(d(w), d(al), d(s_), d(z_), 4, soft)
        y, yt = ops.adaround_t(d(w), d(al), d(s_), d(z_), 4, soft)
        assert torch.equal(y, want) and torch.equal(yt, want.t().contiguous())


@pytest.mark.parametrize("shape", [(32, 6, 197, 197), (7, 3, 49, 49), (5, 1000), (3, 2, 1)])
def test_scaled_softmax_forward_and_backward(ops, shape):
    """adalog_scaled_softmax / _backward against ATen's (x * scale).softmax(-1) and its autograd gradient (bar 2e-6 of the
    largest value: the two evaluate exp with different routines)."""
    from adalog_amd import train_mm
    gen = g(9700 + shape[-1])
    x = (torch.randn(*shape, generator=gen) * 3.0).to(DEV)
    gy = torch.randn(*shape, generator=gen).to(DEV)
    scale = 0.125
    xr = x.clone().requires_grad_(True)
    want = (xr * scale).softmax(dim=-1)
    want.backward(gy)
    y = ops.scaled_softmax(x, scale)
    assert rel_err(y.cpu(), want.detach().cpu()) <= 2e-6
    gx = ops.scaled_softmax_backward(gy, y, scale)
    assert rel_err(gx.cpu(), xr.grad.cpu()) <= 4e-6
    x2 = x.clone().requires_grad_(True)
    y2 = train_mm.scaled_softmax(x2, scale)
    y2.backward(gy)
    assert torch.equal(y2.detach(), y) and torch.equal(x2.grad, gx)


@pytest.mark.parametrize("shape", [(32, 197, 6, 64, 4), (8, 49, 4, 32, 3), (2, 7, 1, 32, 6)])
def test_qkv_split_quant_matches_the_separate_route(ops, shape):
    """adalog_qkv_split_quant / adalog_qkv_merge_quant_backward against the head split followed by three uniform straight-through
    quantisers (per head for q and k, per tensor for v): values and dL/dx bit-equal (same IEEE operations per element), the scale
    gradients to 1e-5 (different partial-sum order)."""
    B, N, H, D, bits = shape
    gen = g(9900 + B + N)
    x = (torch.randn(B, N, 3 * H * D, generator=gen) * 1.5).to(DEV)
    qmax = 2 ** bits - 1
    scales = [(torch.rand(1, H, 1, 1, generator=gen) * 0.2 + 0.2).to(DEV), (torch.rand(1, H, 1, 1, generator=gen) * 0.2 + 0.2).to(DEV),
              (torch.rand(1, 1, 1, 1, generator=gen) * 0.2 + 0.2).to(DEV)]
    zps = [torch.randint(1, qmax, (1, H, 1, 1), generator=gen).float().to(DEV), torch.randint(1, qmax, (1, H, 1, 1), generator=gen).float().to(DEV),
           torch.randint(1, qmax, (1, 1, 1, 1), generator=gen).float().to(DEV)]
    nb = (bits, bits, bits)
    ys = ops.qkv_split_quant(x, H, scales, zps, nb)
    parts = x.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    gys = [torch.randn(B, H, N, D, generator=gen).to(DEV) for _ in range(3)]
    gx, gs = ops.qkv_merge_quant_backward(gys, x, H, scales, zps, nb)
    gx_want = []
    for p_ in range(3):
        xp = parts[p_].contiguous()
        want = ops.uniform_fake_quant(xp, scales[p_], zps[p_], bits)
        assert torch.equal(ys[p_], want), p_
        wgx, wgs, _ = ops.uniform_fake_quant_backward(gys[p_], xp, scales[p_], zps[p_], bits, False, True, False)
        gx_want.append(wgx)
        assert rel_err(gs[p_].cpu().double(), wgs.cpu().double()) <= 1e-5, p_
    gx_ref = torch.stack(gx_want, 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * D)
    assert torch.equal(gx, gx_ref)
    cb_y = CB.qkv_split_quant(x.cpu(), H, [t.cpu() for t in scales], [t.cpu() for t in zps], nb)
    for p_ in range(3):
        assert torch.equal(ys[p_].cpu(), cb_y[p_])
    gx2, gs2 = ops.qkv_merge_quant_backward([gys[0], None, gys[2]], x, H, scales, zps, nb)      # a part without gradient
    assert torch.equal(gx2.reshape(B, N, 3, H, D)[:, :, 1], torch.zeros(B, N, H, D, device=DEV)) and float(gs2[1].abs().max()) == 0.0



# ------------------------------------------------------------------------------------------------ Gram-form weight search (round 5)
GRAM_SHAPES = [  # T, O, K, tokens per image
    (6304, 1152, 384, 197),      # deit_small qkv
    (6304, 384, 384, 197),       # deit_small proj: 3 blocks per workgroup (thin passes)
    (6304, 576, 192, 197),       # deit_tiny qkv (NJ = 6, four blocks per wave)
    (1000, 200, 96, 125),        # NJ = 3, tokens and rows that are no multiple of any tile, no bias
    (25088, 768, 256, 784),      # swin_base stage 1 (NJ = 8)
    (6272, 1536, 512, 196),      # swin_base stage 2 (NJ = 16)
    (3136, 96, 32, 3136),        # NJ = 1
    (1568, 100, 64, 49),         # NJ = 2
]


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("T,O,K,tok", GRAM_SHAPES)
def test_gram_score_matches_spec_and_token_form(ops, bits, T, O, K, tok):
    """adalog_gram_build + adalog_gram_score_w (G = X^T X, c = X^T r, S0 once; K^2 multiply-adds per candidate row and limb) against
    (1) the integer-arithmetic specification tests/cpu_backend.GramState -- the two must agree to fp64 rounding, i.e. the kernels'
    integer pipeline is exact -- and (2) the token-form slab kernel where it takes the shape; weight values planted on rounding ties."""
    from adalog_amd.ops import FP8, I8
    gen = g(12000 + bits + T + O + K)
    P = 128
    x = torch.randn(T, K, generator=gen)
    x[:, : max(1, K // 8)] *= 3.0
    W = torch.randn(O, K, generator=gen) * 0.05
    b = torch.randn(O, generator=gen) * 0.1 if K != 96 else None
    qmax = 2 ** bits - 1
    w_lo, w_hi = W.min(1).values, W.max(1).values
    sc = ((w_hi - w_lo) / qmax).view(1, O) * torch.linspace(0.6, 1.2, P).view(P, 1)
    zp = torch.round(-w_lo.view(1, O) / sc).clamp(0, qmax)
    W[:, 5] = sc[17] * 2.5                                              # exact ties of candidate 17 (and near-ties of its neighbours)
    W[:, 9] = sc[90] * -1.5
    a_s = torch.tensor([x.abs().max().item() * 1.2 / qmax])
    a_z = torch.tensor([float(2 ** (bits - 1) - 1)])
    ref = torch.nn.functional.linear(x, W, b)
    ref[:, 3] *= 40.0                                                   # one loud output column: the per-column fixed-point exponent
    d = lambda t_: t_.to(DEV).contiguous()
    assert ops._lib.load().adalog_gram_supported(T, O, K, bits, bits, P)
    ref_t = d(ref.t())
    st = ops.GramState(d(x), d(a_s), d(a_z), bits, ref_t, None if b is None else d(b))
    got = st.score_w(d(W), d(sc), d(zp), bits, 1.0 / tok)
    assert _last_kernel() == "k_gram_score<i8>"
    spec = CB.GramState(x, a_s, a_z, bits, ref.t().contiguous(), b).score_w(W, sc, zp, bits, 1.0 / tok)
    assert got.shape == spec.shape == (P, O)
    assert rel_err(got.cpu(), spec) <= 2e-7                             # fp32 output rounding only
    dt = FP8 if bits <= 4 else I8
    xp = ops.pack_uniform(d(x).unsqueeze(0), d(a_s), d(a_z), 1, 0, 1, 0, 0, bits, dt)
    if ops.score_w_gen_ok(dt, T, O, K, xp.shape[-1], P):
        want = ops.score_w_gen(dt, xp, d(W), d(sc), d(zp), bits, ref_t, d(a_s), None if b is None else d(b), 1.0 / tok)
        assert rel_err(got.cpu(), want.cpu()) <= 2e-6
    # a second step from the same state (the FPCS loop re-uses it) with other candidates, and determinism of the first
    got2 = st.score_w(d(W), d(sc * 1.01), d(zp), bits, 1.0 / tok)
    spec2 = CB.GramState(x, a_s, a_z, bits, ref.t().contiguous(), b).score_w(W, sc * 1.01, zp, bits, 1.0 / tok)
    assert rel_err(got2.cpu(), spec2) <= 2e-7
    assert torch.equal(got, st.score_w(d(W), d(sc), d(zp), bits, 1.0 / tok))


def test_gram_form_rejects_what_it_cannot_score(ops):
    lib = ops._lib.load()
    assert not lib.adalog_gram_supported(6304, 1152, 400, 4, 4, 128)    # K % 32
    assert not lib.adalog_gram_supported(6304, 1152, 384, 8, 4, 128)    # q - z of an 8-bit activation does not fit int8
    assert not lib.adalog_gram_supported(6304, 1152, 384, 4, 4, 100)    # candidates per row: multiples of 32
    assert lib.adalog_gram_supported(197, 1152, 384, 4, 4, 128) and not lib.adalog_gram_ok(197, 1152, 384, 4, 4, 128)   # one image: does not pay
    assert lib.adalog_gram_ok(6304, 1152, 384, 4, 4, 128) and lib.adalog_gram_limbs(6304, 4) == 3 and lib.adalog_gram_limbs(6304, 6) == 4
    assert lib.adalog_gram_build(None, 6304, 384, 384, None, None, 4, None, 1152, None, None, 0, None) == -1


# ------------------------------------------------------------------------------------------------ Gram-form activation search (round 5)
GRAM_ACT_SHAPES = [  # T, O, K, tokens per image
    (6304, 1152, 384, 197),      # deit_small qkv
    (6304, 384, 384, 197),       # deit_small proj
    (6304, 576, 192, 197),       # deit_tiny qkv (NJ = 6)
    (1000, 200, 96, 125),        # NJ = 3, tokens / rows that are no multiple of any tile, no bias
    (25088, 768, 256, 784),      # swin_base stage 1 (NJ = 8)
    (3136, 96, 32, 3136),        # NJ = 1
    (1568, 100, 64, 49),         # NJ = 2
    (12544, 384, 128, 3136),     # NJ = 4
    (6272, 1536, 512, 196),      # swin_base stage 2 qkv: the triangle in three parts (two k_ga_quad<8> roles + k_ga_rect<2, 8>)
    (6304, 2304, 768, 197),      # vit_base / deit_base qkv: four parts (two k_ga_quad<12> roles + two k_ga_rect<3, 6> roles)
    (900, 300, 768, 100),        # ... ragged tokens / rows, one token split per role
]


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("T,O,K,tok", GRAM_ACT_SHAPES)
def test_gram_act_score_matches_fp64_and_token_form(ops, bits, T, O, K, tok):
    """adalog_gram_act_{prepare,build,score} (per candidate X_p^T X_p on the int8 MFMA against H = Wq^T Wq in fp64; the linear term
    from prefix sums of C = r . Wq along the sorted activation) against (1) the fp64 evaluation of the same sum
    (tests/cpu_backend.GramActState) on a subset of candidates and (2) the token-form slab kernel where it takes the shape;
    activation values planted on rounding ties of one candidate."""
    from adalog_amd.ops import FP8, I8
    gen = g(13000 + bits + T + O + K)
    P = 128
    x = torch.randn(T, K, generator=gen)
    x[:, : max(1, K // 8)] *= 3.0
    W = torch.randn(O, K, generator=gen) * 0.05
    b = torch.randn(O, generator=gen) * 0.1 if K != 96 else None
    qmax = 2 ** bits - 1
    w_lo, w_hi = W.min(1).values, W.max(1).values
    sw = (w_hi - w_lo) / qmax
    zw = torch.round(-w_lo / sw).clamp(0, qmax)
    sc = (2 * x.abs().max().item() / qmax) * torch.linspace(0.5, 1.1, P)
    zp = torch.round(torch.linspace(qmax / 2 - 2, qmax / 2 + 2, P)).clamp(0, qmax)
    x[5, 3] = (sc[17] * 2.5).item()                                       # exact ties of candidate 17
    x[7, 1] = (sc[90] * -1.5).item()
    ref = torch.nn.functional.linear(x, W, b)
    d = lambda t_: t_.to(DEV).contiguous()
    lib = ops._lib.load()
    assert lib.adalog_gram_act_supported(T, O, K, bits, bits, P)
    norm = 1.0 / (tok * O)
    prep = ops.GramActPrepared(d(x))
    st = ops.GramActState(prep, d(ref), None if b is None else d(b), d(W), d(sw), d(zw), bits, bits, P)
    got = st.score(d(sc).view(P, 1), d(zp).view(P, 1), norm)
    assert _last_kernel() == "k_gram_act<i8>" and got.shape == (P, 1)
    sub = [0, 17, 18, 63, 90, 127]
    spec = CB.GramActState(CB.GramActPrepared(x), ref, b, W, sw, zw, bits, bits, P).score(sc[sub], zp[sub], norm)
    assert rel_err(got.cpu()[sub], spec) <= 3e-6
    dt = FP8 if bits <= 4 else I8
    wp = ops.pack_uniform(d(W).unsqueeze(0), d(sw), d(zw), 1, 0, 1, 0, 1, bits, dt)
    if ops.score_act_gen_ok(dt, O, T, K, wp.shape[-1], P):
        want = ops.score_act_gen(dt, wp, d(x), d(sc).view(P, 1), d(zp).view(P, 1), bits, d(ref), d(sw), None if b is None else d(b), norm)
        assert rel_err(got.cpu(), want.cpu()) <= 3e-6
    assert torch.equal(got, st.score(d(sc).view(P, 1), d(zp).view(P, 1), norm))     # bit-reproducible


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("M,N,K", [(6304, 1152, 384), (197, 96, 48), (32, 1000, 384), (1000, 1536, 1536), (130, 64, 16)])
def test_gemm_out_gen_equals_pack_then_gemm(ops, bits, M, N, K):
    """quant_forward of a uniformly quantised Linear in ONE launch (adalog_gemm_out_gen: the activation is quantised in the GEMM's
    loader) against the two-launch route it replaces (pack_uniform + gemm_out): the same int8 codes reach the same int32 dot products
    and the same epilogue, so the fp32 outputs are equal BIT FOR BIT -- including elements that sit exactly on a rounding tie
    (x / scale + zp = n + 0.5 by construction) and values far outside the clamp range; and the codes equal the oracle's."""
    gen = g(7000 + bits * 13 + M)
    scale = torch.tensor([0.0371]); zp = torch.tensor([float(2 ** (bits - 1) - 1)])
    x = torch.randn(1, M, K, generator=gen) * 0.3
    ties = torch.randint(0, 2 ** bits, (M, K), generator=gen).float()
    tie_mask = torch.rand(M, K, generator=gen) < 0.05
    x[0][tie_mask] = ((ties + 0.5 - zp) * scale)[tie_mask]
    x[0][torch.rand(M, K, generator=gen) < 0.01] = 1e6
    x[0][torch.rand(M, K, generator=gen) < 0.01] = -1e6
    w = torch.randint(-(2 ** (bits - 1)), 2 ** (bits - 1), (1, 1, N, K), generator=gen).to(torch.int8)
    Kp = CB.pad_k(K, CB.I8)
    wp = torch.zeros(1, 1, N, Kp, dtype=torch.int8); wp[..., :K] = w
    sb = torch.rand(N, generator=gen) * 0.01 + 0.001
    bias = torch.randn(N, generator=gen)
    xd, wd = x.to(DEV), wp.to(DEV)
    sa_, sb_, bi_ = ops.Strided(scale.to(DEV)), ops.Strided(sb.to(DEV), n=1), ops.Strided(bias.to(DEV), n=1)
    assert ops.gemm_out_gen_ok(xd, Kp, bits)
    got = ops.gemm_out_gen(xd, scale.to(DEV), zp.to(DEV), bits, wd, N, 1, sa_, sb_, bi_)
    xp = ops.pack_uniform(xd, scale.to(DEV), zp.to(DEV), 1, 0, 1, 0, 0, bits, ops.I8)
    want = ops.gemm_out(ops.I8, xp, wd, M, N, 1, 1, sa_, sb_, bi_)
    assert got.shape == want.shape == (1, M, N)
    assert torch.equal(got, want), (got - want).abs().max().item()
    # a row-strided view (the class token x[:, 0] in front of the head) is read in place
    wide = torch.zeros(M, 3, K, device=DEV); wide[:, 0] = xd[0]
    view = wide[:, 0].unsqueeze(0)
    assert not view.is_contiguous() and ops.gemm_out_gen_ok(view, Kp, bits)
    assert torch.equal(ops.gemm_out_gen(view, scale.to(DEV), zp.to(DEV), bits, wd, N, 1, sa_, sb_, bi_), want)
    # the residual stream added in the epilogue (adalog_gemm_out_gen_ex): bit for bit the separate add
    add = torch.randn(1, M, N, generator=gen).to(DEV)
    assert torch.equal(ops.gemm_out_gen(xd, scale.to(DEV), zp.to(DEV), bits, wd, N, 1, sa_, sb_, bi_, addend=add), want + add)
    # the packed codes are the oracle's: round-half-even of x / scale + zp, clamped, minus the zero point
    _, q = O.uniform_fake_quant(x[0], scale, zp, bits)
    assert torch.equal(xp[0, 0, :, :K].cpu().float(), q - zp)


# ------------------------------------------------------------------------------------------------ quant_forward prologues (round 6)
def _adalog_pack_params(bits, q=23.0):
    L = 2 ** (bits - 1)
    mant = torch.tensor([float(round(2 ** (-j / 37) * (4 * L - 2))) for j in range(37)])
    return mant.to(DEV), torch.tensor([q]).to(DEV)


@pytest.mark.parametrize("bits", [3, 4, 6])
@pytest.mark.parametrize("G,R,S", [(12, 197, 197), (5, 64, 49), (3, 256, 256), (2, 7, 130)])
def test_softmax_adalog_pack_equals_softmax_then_pack(ops, bits, G, R, S):
    """(scores * scale).softmax(-1) -> post-softmax AdaLog quantiser -> bf16 operand in ONE launch (adalog_softmax_adalog_pack_bf16)
    against torch's softmax followed by the packer: the kernel restates ATen's softmax arithmetic operation for operation, so the
    packed operands must be equal bit for bit (a probability differing in its last bit could flip a code)."""
    gen = g(9100 + bits + S)
    x = (torch.randn(G, R, S, generator=gen) * 6).to(DEV)
    x[0, 0] = 0.0                                                     # a uniform row
    x[0, 1, :3] = torch.tensor([80.0, -80.0, 79.5])                   # saturating exponents
    mant, qv = _adalog_pack_params(bits)
    scale = torch.ones(1, device=DEV)
    mul = 0.125
    want = ops.pack_adalog((x * mul).softmax(dim=-1), scale, qv, 1, 0, 1, 0, bits, mant, shift=None, clamp_u=True)
    got = ops.softmax_adalog_pack(x, mul, scale, qv, bits, mant)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    same = (got.view(torch.int16) == want.view(torch.int16)).float().mean().item()
    assert same == 1.0, same
    assert (got[..., S:] == 0).all()


@pytest.mark.parametrize("bits", [3, 4, 6])
def test_pack_adalog_with_the_gelu_prologue(ops, bits):
    """fc2's operand packer reading fc1's output (pre_gelu): equal, bit for bit, to torch GELU followed by the packer."""
    gen = g(9200 + bits)
    x = (torch.randn(1, 1000, 1536, generator=gen) * 1.5).to(DEV)
    x[0, 0, :8] = torch.tensor([0.0, -0.0, 1e-8, -1e-8, 12.0, -12.0, 3.0, -0.75])
    mant, qv = _adalog_pack_params(bits, q=31.0)
    scale = torch.tensor([2.1], device=DEV)
    shift = torch.tensor([0.16997124254703522], device=DEV)
    want = ops.pack_adalog(torch.nn.functional.gelu(x), scale, qv, 1, 0, 1, 0, bits, mant, shift=shift, clamp_u=True)
    got = ops.pack_adalog(x, scale, qv, 1, 0, 1, 0, bits, mant, shift=shift, clamp_u=True, pre_gelu=True)
    same = (got.view(torch.int16) == want.view(torch.int16)).float().mean().item()
    assert same == 1.0, same


@pytest.mark.parametrize("B,N,H,per_head", [(4, 197, 6, True), (2, 64, 3, True), (3, 130, 12, False), (1, 1, 2, True)])
def test_attn_split_pack_equals_three_packs(ops, B, N, H, per_head):
    """q / k / v of an attention block split, quantised and packed in one launch (adalog_attn_split_pack) against the permuted views
    through adalog_pack_uniform: identical bytes, padding included."""
    gen = g(9300 + N)
    D = 64
    qkv = (torch.randn(B, N, 3 * H * D, generator=gen) * 1.3).to(DEV)
    n = H if per_head else 1
    par = []
    for bits in (4, 3, 6):
        s_ = (torch.rand(n, generator=gen) * 0.2 + 0.05).to(DEV)
        z_ = torch.randint(0, 2 ** bits, (n,), generator=gen).float().to(DEV)
        par.append((s_, z_, bits))
    qp, kp, vp = ops.attn_split_pack(qkv, H, par[0], par[1], par[2], per_head)
    q, k, v = qkv.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4).unbind(0)            # [B, H, N, D] views (wrap_net.py:20-22)
    pg = 1 if per_head else 0
    gm = H if per_head else 1
    want_q = ops.pack_uniform(q.reshape(B * H, N, D), par[0][0], par[0][1], 1, 0, gm, pg, 0, par[0][2], ops.I8)
    want_k = ops.pack_uniform(k.reshape(B * H, N, D), par[1][0], par[1][1], 1, 0, gm, pg, 0, par[1][2], ops.I8)
    vt = v.transpose(-2, -1).reshape(B * H, D, N)
    want_v = ops.pack_uniform(vt, par[2][0], par[2][1], 1, 0, gm, pg, 0, par[2][2], ops.BF16)
    assert qp.shape == want_q.shape and torch.equal(qp, want_q)
    assert torch.equal(kp, want_k)
    # (values, not bit patterns: the general packer clamps q - z directly and keeps the sign of a -0.0 quotient; zero is zero to the MFMA)
    assert vp.shape == want_v.shape and torch.equal(vp.float(), want_v.float())


@pytest.mark.parametrize("dt", ["i8", "bf16"])
def test_gemm_out_addend_and_heads_last(ops, dt):
    """The epilogue extras of the quant_forward product (adalog_gemm_out_ex): the residual added in the epilogue equals out + addend
    bit for bit; the heads-last store equals the permuted copy of the plain product."""
    gen = g(9400)
    B, H, M, N, K = 3, 6, 197, 64, 197
    dto = ops.I8 if dt == "i8" else ops.BF16
    tdt = torch.int8 if dt == "i8" else torch.bfloat16
    Kp = CB.pad_k(K, CB.I8 if dt == "i8" else CB.BF16)
    A = torch.zeros(1, B * H, M, Kp, dtype=tdt); Bm = torch.zeros(1, B * H, N, Kp, dtype=tdt)
    A[..., :K] = torch.randint(-7, 8, (1, B * H, M, K), generator=gen).to(tdt)
    Bm[..., :K] = torch.randint(-7, 8, (1, B * H, N, K), generator=gen).to(tdt)
    sa = (torch.rand(H, generator=gen) * 0.1 + 0.01).to(DEV); sb = (torch.rand(H, generator=gen) * 0.1 + 0.01).to(DEV)
    Ad, Bd = A.to(DEV), Bm.to(DEV)
    plain = ops.gemm_out(dto, Ad, Bd, M, N, B * H, H, ops.Strided(sa, g=1), ops.Strided(sb, g=1), None)
    hl = ops.gemm_out(dto, Ad, Bd, M, N, B * H, H, ops.Strided(sa, g=1), ops.Strided(sb, g=1), None, heads_last=H)
    assert hl.shape == (B, M, H, N)
    assert torch.equal(hl, plain.view(B, H, M, N).permute(0, 2, 1, 3).contiguous())
    add = torch.randn(B * H, M, N, generator=gen).to(DEV)
    bias = torch.randn(N, generator=gen).to(DEV)
    base = ops.gemm_out(dto, Ad, Bd, M, N, B * H, H, ops.Strided(sa, g=1), ops.Strided(sb, g=1), ops.Strided(bias, n=1))
    got = ops.gemm_out(dto, Ad, Bd, M, N, B * H, H, ops.Strided(sa, g=1), ops.Strided(sb, g=1), ops.Strided(bias, n=1), addend=add)
    assert torch.equal(got, base + add)


@pytest.mark.parametrize("q", [23, 61])
def test_adalog_training_form_with_the_gelu_prologue(ops, q):
    """fc2's input quantiser inside a BRECQ iteration reading fc1's output (pre_gelu): the forward equals the quantiser applied to
    torch's GELU bit for bit; the backward equals autograd's route (STE quantiser backward, then GeluBackward) -- dL/dx to fp32
    rounding of one product, the scale gradient unchanged."""
    gen = g(9500 + q)
    bits = 4
    x = (1.7 * torch.randn(6, 197, 256, generator=gen)).to(DEV)
    x[0, 0, :6] = torch.tensor([0.0, -0.0, 9.0, -9.0, 1e-6, -3.0])
    s = torch.tensor([2.3], device=DEV); sh = torch.tensor([O.GELU_SHIFT], device=DEV); qd = torch.tensor([q], device=DEV)
    gy = torch.randn(x.shape, generator=gen).to(DEV)
    for sub in (True, False):
        xg = x.clone().requires_grad_(True)
        gel = torch.nn.functional.gelu(xg)
        y_ref = ops.log_fake_quant(gel.detach(), s, qd, None, None, bits, shift=sh, sub_shift=sub, train_form=True)
        y = ops.log_fake_quant(x, s, qd, None, None, bits, shift=sh, sub_shift=sub, train_form=True, pre_gelu=True)
        assert torch.equal(y, y_ref)
        g_gel, gs_ref = ops.log_fake_quant_backward(gy, gel.detach(), y_ref, s, qd, bits, sh, sub)
        gel.backward(g_gel)
        gx, gs = ops.log_fake_quant_backward(gy, x, y, s, qd, bits, sh, sub, pre_gelu=True)
        # (where the GELU's derivative cdf + x * pdf cancels -- x around -0.75 -- its last bits depend on the order of two roundings)
        torch.testing.assert_close(gx, xg.grad, rtol=1e-4, atol=1e-7)
        torch.testing.assert_close(gs, gs_ref, rtol=1e-6, atol=1e-7)


def test_adam_multi_matches_torch_adam_and_counts_its_steps(ops):
    """adalog_adam_multi over several tensors for several steps against torch.optim.Adam (defaults), and the device step counter it
    advances itself (the last workgroup to arrive stores step + 1: no separate counting launch)."""
    gen = g(9600)
    shapes = [(3,), (1000,), (384, 384), (5000,)]
    ps = [torch.randn(*s_, generator=gen).to(DEV) for s_ in shapes]
    ref = [p_.clone().requires_grad_(True) for p_ in ps]
    opt = torch.optim.Adam(ref, lr=1e-3)
    m = [torch.zeros_like(p_) for p_ in ps]; v = [torch.zeros_like(p_) for p_ in ps]
    step = torch.zeros(1, device=DEV)
    for it in range(5):
        gs = [torch.randn(*s_, generator=gen).to(DEV) for s_ in shapes]
        for r_, g_ in zip(ref, gs):
            r_.grad = g_.clone()
        opt.step()
        ops.adam_multi(ps, gs, m, v, step, 1e-3, 0.9, 0.999, 1e-8)
        assert step.item() == it + 1
    for p_, r_ in zip(ps, ref):
        torch.testing.assert_close(p_, r_.detach(), rtol=2e-6, atol=1e-7)


def test_brecq_prepare_equals_two_gathers_and_a_copy(ops):
    """adalog_brecq_prepare: the mini-batch rows of a block's stored inputs / outputs and the iteration's schedule row in one launch."""
    gen = g(9800)
    src_in = torch.randn(50, 197, 384, generator=gen).to(DEV); src_out = torch.randn(50, 197, 96, generator=gen).to(DEV)
    idx = torch.randperm(50, generator=gen)[:32].to(DEV)
    dst_in = torch.empty(32, 197, 384, device=DEV); dst_out = torch.empty(32, 197, 96, device=DEV)
    table = torch.randn(7, 3, generator=gen).to(DEV); sched = torch.zeros(3, device=DEV)
    assert ops.brecq_prepare(src_in, src_out, idx, dst_in, dst_out, table[4], sched)
    assert torch.equal(dst_in, src_in[idx]) and torch.equal(dst_out, src_out[idx]) and torch.equal(sched, table[4])
    assert ops.brecq_prepare(src_in, src_out, idx, dst_in, dst_out)                        # no schedule row
    assert not ops.brecq_prepare(src_in[:, :, :3], src_out, idx, dst_in, dst_out)          # (a view that does not qualify: the caller composes)


@pytest.mark.parametrize("M,N,K", [(6304, 384, 1536), (6304, 384, 384), (130, 64, 96)])
def test_gemm_f32x3_addend_rides_in_the_reduction_pass(ops, M, N, K):
    """adalog_gemm_f32x3_add: the product plus an addend laid out like the result -- bit for bit `product + addend`, whether the
    product is split along K (the addend is added by the reduction pass: fc2's forward inside a BRECQ iteration) or not."""
    gen = g(9900 + K)
    a = torch.randn(M, K, generator=gen).to(DEV); b = (torch.randn(N, K, generator=gen) * 0.05).to(DEV)
    bias = torch.randn(N, generator=gen).to(DEV) if N % 16 == 0 else None
    add = torch.randn(M, N, generator=gen).to(DEV)
    for ea in (0, 2):
        want = ops.gemm_f32x3(a, b, bias, exact_a=ea, exact_b=ea) + add
        got = ops.gemm_f32x3(a, b, bias, exact_a=ea, exact_b=ea, addend=add)
        assert torch.equal(got, want)
