"""Per-call score parity of the PRODUCT scoring methods against the reference's golden FPCS traces.

The oracle (oracle/adalog_oracle.py, pinned bit-for-bit on top-k indices against the golden traces by
tests/test_oracle_golden.py) walks the reference's search path; at EVERY scoring call its observer hands the same
candidates and the parameters in force to the product layer's ``_score_*`` method on ``device`` (HIP kernels under
``-m gpu``, the executable specs of tests/cpu_backend.py in the CPU tier) and the result is compared with the
reference's own ``trace_NNN_scores``:

  * score vectors: max relative error <= SCORE_RTOL (1e-4; the north-star bar for fp32 tensors is 1e-3), or -- for the
    near-optimal candidates of 6-bit searches, whose score is a residual ~4^-bits of the signal, so that the fp32
    rounding of `raw_out - out_sim` (reference and product alike) is a larger FRACTION of it -- an absolute error below
    NOISE_C * sqrt(|score| * E), E = the same normalised sum over the REFERENCE tensor (raw_out, W or x) squared: the
    first-order fp32 noise of sum(e^2) is 2 sum(e d), |d| ~ u |out|, i.e. ~ u sqrt(sum e^2) sqrt(sum out^2).  Observed on
    MI355X: <= 5.6e-6 in these units for every Linear trace at 3, 4 AND 6 bit (the same constant, which is what the model
    predicts), <= 2e-7 for the attention / conv traces; NOISE_C = 1e-5.  At 4 bit this is below SCORE_RTOL anyway;
  * top-k sets: the product's deterministic top-k of ITS scores must equal the reference's ``trace_NNN_idx`` as a set per
    column, except for members whose REFERENCE scores lie within TIE_RTOL (1e-5 relative) of the k-th best reference
    score -- exact and near ties are structural in FPCS and torch.topk's order among them is unspecified.
    A difference at a larger reference gap is counted separately as a ``noise flip``: it is only possible while the gap
    stays below twice the call's measured score error (checked), i.e. inside the fp32 rounding noise that the reference's
    own `raw_out - out_sim` subtraction carries on these tiny tensors (|e| ~ |out| / 2^bits: ~1e-5 relative at 6 bit);
    noise flips must stay below NOISE_FLIP_FRAC of all top-k members of a case.

Every layer class x {3, 4, 6} bit is covered; nothing here uses the +-10 % objective band.
"""
import numpy as np
import torch

from adalog_amd import backend, quant_layers as Q
from adalog_amd.ops import BF16, Strided
from oracle import adalog_oracle as O

SCORE_RTOL = 1e-4
NOISE_C = 1e-5              # absolute score error allowed: NOISE_C * sqrt(|score| * energy of the reference tensor)
TIE_RTOL = 1e-5
NOISE_FLIP_FRAC = 2e-3


def t(a):
    return torch.from_numpy(np.asarray(a))


class Replay:
    """Counts scoring calls and compares each with golden ``<prefix>_NNN_{scores,idx,k}``."""

    def __init__(self, g, prefix="trace"):
        self.g, self.prefix, self.n = g, prefix, 0
        self.max_err, self.flips, self.noise_flips, self.members, self.max_gap, self.max_nrm = 0.0, 0, 0, 0, 0.0, 0.0

    def check(self, got, cand_axis0: bool, energy=None):
        """``got``: product scores [P, cols] (any device).  cand_axis0: golden arrays carry the candidate axis first
        (weights / matmul / conv) or last (activations).  ``energy``: [cols] (or scalar) normalised sum of squares of
        the tensor the scores measure the distance to."""
        g, i = self.g, self.n
        ref = t(g[f"{self.prefix}_{i:03d}_scores"]).float()
        idx = t(g[f"{self.prefix}_{i:03d}_idx"]).long()
        k = int(g[f"{self.prefix}_{i:03d}_k"])
        P = got.shape[0]
        if cand_axis0:
            ref, idx = ref.reshape(P, -1), idx.reshape(k, -1)
        else:
            ref, idx = ref.reshape(-1, P).t(), idx.reshape(-1, k).t()
        mine = got.detach().float().cpu().reshape(ref.shape)
        rel = (mine - ref).abs() / ref.abs().clamp_min(1e-30)
        en = torch.as_tensor(1.0 if energy is None else energy).detach().float().cpu().reshape(1, -1)
        nrm = (mine - ref).abs() / (ref.abs() * en).sqrt().clamp_min(1e-30)
        bad = (rel > SCORE_RTOL) & ((nrm > NOISE_C) | (energy is None))
        assert not bool(bad.any()), f"{self.prefix} call {i}: score rel err {rel[bad].max().item():.3e} (in units of " \
                                    f"sqrt(score * energy): {nrm[bad].max().item():.3e})"
        err = rel.max().item()
        self.max_err = max(self.max_err, err)
        self.max_nrm = max(self.max_nrm, nrm.max().item())
        my_idx = backend.get().topk(got.detach().float().contiguous(), k).long().cpu()          # [k, cols]
        kth = ref.topk(k, dim=0).values[-1]                                           # k-th best reference score
        for c in range(ref.shape[1]):
            a, b = set(my_idx[:, c].tolist()), set(idx[:, c].tolist())
            self.members += k
            for j in a ^ b:
                d = abs(ref[j, c].item() - kth[c].item()) / abs(kth[c].item())
                if d <= TIE_RTOL:
                    self.flips += 1
                    continue
                self.noise_flips += 1
                self.max_gap = max(self.max_gap, d)
                assert d <= 2.0 * err + 1e-7, \
                    f"{self.prefix} call {i} col {c}: top-{k} sets differ at candidate {j} whose reference score is " \
                    f"{d:.2e} (rel) from the k-th best with a score error of only {err:.2e} -- not explained by noise"
        self.n += 1

    def done(self):
        assert self.n == int(self.g[f"{self.prefix}_n"]), (self.n, int(self.g[f"{self.prefix}_n"]))
        assert self.noise_flips <= NOISE_FLIP_FRAC * self.members, (self.noise_flips, self.members, self.max_gap)
        return {"calls": self.n, "max_rel_err": self.max_err, "max_err_noise_units": self.max_nrm, "tie_flips": self.flips,
                "noise_flips": self.noise_flips, "max_noise_gap": self.max_gap, "members": self.members}


def _set_uniform(q, scale, zp):
    q.scale.data.copy_(scale.reshape(q.scale.shape).to(q.scale.device))
    q.zero_point.data.copy_(zp.reshape(q.zero_point.shape).float().to(q.zero_point.device))
    q.inited = True
    q._zp_on_grid = True                      # every value the oracle commits comes from an FPCS grid


def _pc(a, dev):
    """[P, ...] candidates-first golden/oracle tensor -> [P, cols] fp32 on the device."""
    return a.reshape(a.shape[0], -1).float().contiguous().to(dev)


def _pl(a, dev):
    """[cols, P] candidates-last -> [P, cols]."""
    return a.reshape(-1, a.shape[-1]).t().float().contiguous().to(dev)


# ------------------------------------------------------------------------------------------------ Linear
def _energies(lay):
    """normalised sums of squares matching the four Linear score kinds (same reductions as the scores)"""
    ro, x, W = lay.raw_out.float(), lay.raw_input.float(), lay.weight.data.float()
    O = ro.shape[-1]
    rb = ro.reshape(ro.shape[0], -1, O)
    xb = x.reshape(x.shape[0], -1, x.shape[-1])
    return {"w_out": (rb ** 2).mean(1).sum(0), "a_out": (rb ** 2).mean((1, 2)).sum(0), "w_self": (W ** 2).mean(1),
            "a_self": (xb ** 2).mean((1, 2)).sum(0), "a_self_cw": (xb ** 2).mean(1).sum(0)}


def _linear_observer(lay, rp, dev):
    en = _energies(lay)

    def obs(kind, p, a, b, s):
        if kind == "w_self":
            rp.check(lay._score_w_self(_pc(a, dev), _pc(b, dev)), True, en["w_self"])
        elif kind == "a_self":
            rp.check(lay._score_a_self(_pl(a, dev), _pl(b, dev)), False, en["a_self"])
        elif kind == "w_out":
            _set_uniform(lay.a_quantizer, p.a_scale, p.a_zp)
            lay.w_quantizer._zp_on_grid = True
            # (as weight_fpcs dispatches it: the Gram form where backend.gram_ok takes the shape -- ADALOG_GRAM_W=2 at these toy
            # sizes --, else the token-form kernels)
            rp.check(lay._w_scorer()(_pc(a, dev), _pc(b, dev)), True, en["w_out"])
        elif kind == "a_out":
            _set_uniform(lay.w_quantizer, p.w_scale, p.w_zp)
            # (as activation_fpcs dispatches it: the Gram form where backend.gram_act_ok takes the shape -- ADALOG_GRAM_A=2 at these
            # toy sizes --, else the token-form kernels)
            rp.check(lay._a_scorer()(_pl(a, dev), _pl(b, dev)), False, en["a_out"])
        else:
            raise AssertionError(kind)
    return obs


def replay_linear(golden, name, device="cpu"):
    dev = torch.device(device)
    g = golden(name)
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    lay = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                              eq_n=128, n_V=n_V, fpcs=True, steps=6).to(dev)
    lay.weight.data.copy_(W)
    lay.bias.data.copy_(b)
    lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
    rp = Replay(g)
    with torch.no_grad():
        O.search_linear(W, b, x, ro, wb, ab, n_V=n_V, batch=cbs, observer=_linear_observer(lay, rp, dev))
    return rp.done()


def replay_channelwise(golden, bits, device="cpu"):
    dev = torch.device(device)
    g = golden(f"linear_cw_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    lay = Q.AsymmetricallyChannelWiseBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                                         eq_n=128, n_V=n_V, fpcs=True, steps=6).to(dev)
    lay.weight.data.copy_(W)
    lay.bias.data.copy_(b)
    lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
    rp = Replay(g, "cwtrace")

    def obs(kind, p, a, b_, s):
        assert kind == "a_self_cw"
        rp.check(lay._score_a_self(_pl(a, dev), _pl(b_, dev)), False, _energies(lay)["a_self_cw"])
    with torch.no_grad():
        s, z = O.search_linear_channelwise(x, ab, batch=cbs, observer=obs)
    out = rp.done()
    # second stage: the plain search on the re-parameterised layer (linear.py:614-621), golden prefix 'trace'
    r, bb, ts, tz, lw, lb, W2, b2 = O.reparam_step1(s, z, t(g["ln_weight"]), t(g["ln_bias"]), W, b)
    x2 = x / r - bb
    lay2 = Q.AsymmetricallyBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                               eq_n=128, n_V=n_V, fpcs=True, steps=6).to(dev)
    lay2.weight.data.copy_(W2)
    lay2.bias.data.copy_(b2)
    lay2.raw_input, lay2.raw_out = x2.to(dev), ro.to(dev)
    rp2 = Replay(g)
    with torch.no_grad():
        O.search_linear(W2, b2, x2, ro, wb, ab, n_V=n_V, batch=cbs, observer=_linear_observer(lay2, rp2, dev))
    out2 = rp2.done()
    return {"calls": out["calls"] + out2["calls"], "max_rel_err": max(out["max_rel_err"], out2["max_rel_err"]),
            "max_err_noise_units": max(out["max_err_noise_units"], out2["max_err_noise_units"]),
            "tie_flips": out["tie_flips"] + out2["tie_flips"], "noise_flips": out["noise_flips"] + out2["noise_flips"],
            "max_noise_gap": max(out["max_noise_gap"], out2["max_noise_gap"]), "members": out["members"] + out2["members"]}


# ------------------------------------------------------------------------------------------------ post-GELU
def replay_postgelu(golden, bits, device="cpu"):
    dev = torch.device(device)
    be = backend.get()
    g = golden(f"postgelu_w{bits}a{bits}")
    wb, ab, N, Tn, I, Oc, n_V, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    lay = Q.PostGeluLogBasedBatchingQuantLinear(I, Oc, True, "raw", wb, ab, calib_batch_size=cbs, search_round=3,
                                                eq_n=128, n_V=1, quantizer="adalog", fpcs=True, steps=6).to(dev)
    lay.weight.data.copy_(W)
    lay.bias.data.copy_(b)
    lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
    rp = Replay(g)

    def obs(kind, p, a, b_, s):
        aq = lay.a_quantizer
        en = _energies(lay)
        if kind == "w_self":
            rp.check(lay._score_w_self(_pc(a, dev), _pc(b_, dev)), True, en["w_self"])
        elif kind == "a_logbase":
            _set_uniform(lay.w_quantizer, p.w_scale, p.w_zp)
            wp, rowsum = lay._pack_w_fixed(BF16, want_rowsum=True)
            fold = be.shift_fold(rowsum.view(1, -1), lay.w_quantizer.scale.data.view(1, -1), aq.shift.data,
                                 lay.bias.data).view(-1)
            rp.check(lay._score_scale_logbase(wp, fold, _pl(a, dev), _pl(b_, dev)), False, en["a_out"])
        elif kind == "w_out":
            aq.scale.data.copy_(p.a_scale.reshape(aq.scale.shape).to(dev))
            aq.q.data.fill_(int(p.a_q))
            aq.inited = True
            lay._q_host = int(p.a_q)
            rp.check(lay._score_w(lay._pack_x_fixed(), _pc(a, dev), _pc(b_, dev)), True, en["w_out"])
        else:
            raise AssertionError(kind)
    with torch.no_grad():
        O.search_postgelu(W, b, x, ro, wb, ab, batch=cbs, observer=obs)
    return rp.done()


# ------------------------------------------------------------------------------------------------ MatMul
def _mm_dt(lay):
    from adalog_amd import search
    from adalog_amd.ops import I8, pad_k
    G, S, K, Sp = lay._dims()
    return search.int_operand_dtype(lay.A_quantizer.n_bits, lay.B_quantizer.n_bits,
                                    lay._cand_chunk(G * max(S, Sp) * pad_k(K, I8)), prefer_fp8=K <= 64)


def replay_matmul(golden, bits, device="cpu"):
    dev = torch.device(device)
    g = golden(f"matmul_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    A, B, ro = t(g["A"]), t(g["B"]), t(g["raw_out"])
    lay = Q.AsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=cbs, search_round=3,
                                              eq_n=128, head_channel_wise=True, num_heads=H, fpcs=True, steps=6).to(dev)
    # q@k^T hands B over as a transposed view (wrap_net.py:25): reproduce that layout
    Bd = B.to(dev).transpose(-2, -1).contiguous().transpose(-2, -1)
    lay.raw_input, lay.raw_out = [A.to(dev), Bd], ro.to(dev)
    lay._initialize_calib_parameters()
    rp = Replay(g)

    en_h = (ro.float() ** 2).mean((2, 3)).sum(0)                          # [H]

    def obs(kind, p, a, b_, s):
        dt = _mm_dt(lay)
        if kind == "A":
            _set_uniform(lay.B_quantizer, p.B_scale, p.B_zp)
            rp.check(lay._score("A", lay._pack_fixed("B", dt), _pc(a, dev), _pc(b_, dev), dt), True, en_h)
        elif kind == "B":
            _set_uniform(lay.A_quantizer, p.A_scale, p.A_zp)
            rp.check(lay._score("B", lay._pack_fixed("A", dt), _pc(a, dev), _pc(b_, dev), dt), True, en_h)
        else:
            raise AssertionError(kind)
    with torch.no_grad():
        O.search_matmul(A, B, ro, bits, bits, batch=cbs, observer=obs)
    return rp.done()


def replay_postsoftmax(golden, bits, device="cpu"):
    dev = torch.device(device)
    from adalog_amd import search
    g = golden(f"postsoftmax_a{bits}b{bits}")
    _, _, N, H, S, C, cbs = [int(v) for v in g["cfg"]]
    A, B, ro = t(g["A"]), t(g["B"]), t(g["raw_out"])
    lay = Q.PostSoftmaxAsymmetricallyBatchingQuantMatMul(A_bit=bits, B_bit=bits, mode="raw", calib_batch_size=cbs,
                                                         search_round=3, eq_n=128, head_channel_wise=True, num_heads=H,
                                                         fpcs=True, steps=6, quantizer="adalog").to(dev)
    lay.raw_input, lay.raw_out = [A.to(dev), B.to(dev)], ro.to(dev)
    lay._initialize_calib_parameters()
    rp = Replay(g)

    en_h = (ro.float() ** 2).mean((2, 3)).sum(0)                          # [H]

    def obs(kind, p, a, b_, s):
        aq = lay.A_quantizer
        if kind == "A_logbase":
            _set_uniform(lay.B_quantizer, p.B_scale, p.B_zp)
            q_all, sc = lay._score_A_log_base()
            assert torch.equal(q_all.cpu().long().view(-1), b_.view(-1))
            rp.check(sc, True, en_h.mean())
        elif kind == "B":
            aq.q.data.fill_(int(p.A_q))
            lay._q_host = int(p.A_q)
            qv = search.const_tensor([float(p.A_q)], dev)
            ap = lay._pack_A_adalog(lay._a3(lay.raw_input[0]), qv, aq.scale.data.view(-1), 1, True, k_align=lay._kalign())
            rp.check(lay._score("B", ap, _pc(a, dev), _pc(b_, dev), BF16, fixed_sa=Strided(aq.scale.data.view(-1)),
                                sa_mul=lay._ts32()), True, en_h)
        else:
            raise AssertionError(kind)
    with torch.no_grad():
        O.search_postsoftmax(A, B, ro, bits, bits, batch=cbs, observer=obs)
    return rp.done()


# ------------------------------------------------------------------------------------------------ Conv
def replay_conv(golden, bits, device="cpu"):
    dev = torch.device(device)
    be = backend.get()
    g = golden(f"conv_w{bits}")
    wb, _, N, ic, oc, k, hw, cbs = [int(v) for v in g["cfg"]]
    W, b, x, ro = t(g["weight"]), t(g["bias"]), t(g["x"]), t(g["raw_out"])
    lay = Q.AsymmetricallyBatchingQuantConv2d(in_channels=ic, out_channels=oc, kernel_size=(k, k), stride=(k, k),
                                              mode="raw", w_bit=wb, a_bit=8, calib_batch_size=cbs, search_round=3,
                                              eq_n=128, fpcs=True, steps=6).to(dev)
    lay.weight.data.copy_(W)
    lay.bias.data.copy_(b)
    lay.raw_input, lay.raw_out = x.to(dev), ro.to(dev)
    patches, gh, gw = lay._patches(lay.raw_input)
    M = patches.shape[0]
    xp = lay._pack_x(patches)
    ref = lay.raw_out.permute(1, 0, 2, 3).reshape(1, oc, M).contiguous()
    rp = Replay(g)

    def obs(kind, p, a, b_, s):
        assert kind == "w_out"
        rp.check(lay._score_w(xp, ref, M, gh * gw, _pc(a, dev), _pc(b_, dev)), True, (ro.float() ** 2).mean((2, 3)).sum(0))
    with torch.no_grad():
        O.search_conv(W, b, x, ro, wb, (k, k), batch=cbs, observer=obs)
    return rp.done()
