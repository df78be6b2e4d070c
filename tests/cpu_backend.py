"""CPU stand-in for the HIP kernel backend -- TEST INFRASTRUCTURE ONLY (lives under tests/, never imported by the package).

Each function is an executable specification of one C-ABI entry point (include/adalog_hip.h) written with torch CPU
ops, with the same Python signature as adalog_amd/ops.py.  Two uses:
  * `-m "not gpu"` tests inject it with adalog_amd.backend.set_backend() to exercise the HOST logic (FPCS driver,
    layer searches, calibrator, image sharding over gloo) against the golden fixtures without a GPU;
  * `-m gpu` tests compare every HIP kernel against it one to one on seeded inputs.
"""
import math

import torch

from oracle import adalog_oracle as O

I8, BF16, F32, FP8 = 0, 1, 2, 3
BF16_FP8 = 4                                   # gemm_score only: bf16 rows x fp8 candidate columns
_ESZ = {I8: 1, BF16: 2, F32: 4, FP8: 1}
_TORCH_DT = {I8: torch.int8, BF16: torch.bfloat16, F32: torch.float32, FP8: torch.float8_e4m3fn}


def pad_k(K, dtype, align=128):
    per = align // _ESZ[dtype]
    return ((K + per - 1) // per) * per


class Strided:
    def __init__(self, t, c=0, g=0, n=0):
        self.t, self.c, self.g, self.n = t, int(c), int(g), int(n)


def uniform_fake_quant(x, scale, zero_point, n_bits, sym=False, want_bins=False, want_y=True):
    y, q = O.uniform_fake_quant(x, scale, zero_point, n_bits, sym)
    if want_bins:
        return (y if want_y else None), q.to(torch.uint8)
    return y


def log_fake_quant(x, scale, q, table1, table2, n_bits, shift=None, sub_shift=False, train_form=False,
                   want_bins=False, want_y=True):
    xs = x if shift is None else x + shift
    if train_form:
        y, k, mask = O.adalog_fake_quant_train(xs, scale, int(q.item()), n_bits)
    else:
        y, k, mask = O.adalog_fake_quant(xs, scale, int(q.item()), n_bits, (table1, table2))
    if sub_shift:
        y = y - shift
    if want_bins:
        b = torch.where(mask, k, torch.full_like(k, 255.0)).to(torch.uint8)
        return (y if want_y else None), b
    return y


def _params(t, C, G, R, pc, gmod, pg, pr):
    c = torch.arange(C).view(C, 1, 1)
    g = (torch.arange(G) % gmod).view(1, G, 1)
    r = torch.arange(R).view(1, 1, R)
    return t.reshape(-1)[(c * pc + g * pg + r * pr)]            # [C, G, R]


def pack_uniform(x3, scale, zero_point, C, pc, gmod, pg, pr, n_bits, dtype=I8, want_rowsum=False, c_inner=False, k_align=128):
    G, R, K = x3.shape
    s = _params(scale, C, G, R, pc, gmod, pg, pr).unsqueeze(-1)
    z = torch.round(_params(zero_point, C, G, R, pc, gmod, pg, pr)).unsqueeze(-1)
    q = (torch.round(x3.unsqueeze(0) / s) + z).clamp(0, 2 ** n_bits - 1) - z
    Kp = pad_k(K, dtype, k_align)
    out = torch.zeros((C, G, R, Kp), dtype=_TORCH_DT[dtype])
    out[..., :K] = q.to(_TORCH_DT[dtype])
    rs = q.sum(-1).to(torch.int32)
    if c_inner:                                   # [C,G,R,Kp] -> [1,G,R*C,Kp], candidates innermost
        out = out.permute(1, 2, 0, 3).reshape(1, G, R * C, Kp).contiguous()
    if want_rowsum:
        return out, rs
    return out


def pack_adalog(x3, scale, qv, C, pc, gmod, pg, n_bits, mant37, shift=None, clamp_u=True, c_inner=False, k_align=128):
    G, R, K = x3.shape
    s = _params(scale, C, G, R, pc, gmod, pg, 0).unsqueeze(-1)
    qf = _params(qv, C, G, R, pc, gmod, pg, 0).unsqueeze(-1)
    xs = x3 if shift is None else x3 + shift
    u = xs.unsqueeze(0) / s
    if clamp_u:
        u = u.clamp(min=1e-15, max=1.0)
    k = torch.round(-u.log2() * 37.0 / qf)
    mask = ~(k < 2 ** n_bits)
    k = k.clamp(0, 2 ** n_bits - 1)
    k = torch.nan_to_num(k, nan=0.0)
    kq = (k * qf).long()
    t, j = kq // 37, kq % 37
    v = torch.ldexp(mant37[j], -t.to(torch.int32))
    v[t > 100] = 0
    v[mask] = 0
    Kp = pad_k(K, BF16, k_align)
    out = torch.zeros((C, G, R, Kp), dtype=torch.bfloat16)
    vb = v.to(torch.bfloat16)
    assert torch.equal(vb.float(), v.float()), "AdaLog operand must be exact in bf16"
    out[..., :K] = vb
    if c_inner:
        out = out.permute(1, 2, 0, 3).reshape(1, G, R * C, Kp).contiguous()
    return out


def pack_raw(x3):
    G, R, K = x3.shape
    out = torch.zeros((1, G, R, pad_k(K, F32)), dtype=torch.float32)
    out[0, :, :, :K] = x3
    return out


def pack_split3(x3, k_align=128):
    G, R, K = x3.shape
    Kt = pad_k(K, BF16, k_align)
    out = torch.zeros((1, G, R, 3 * Kt), dtype=torch.bfloat16)
    hi = x3.to(torch.bfloat16)
    r1 = x3 - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    assert torch.equal(hi.float() + mid.float() + lo.float(), x3), "three bf16 terms must reproduce fp32 exactly"
    out[0, :, :, :K], out[0, :, :, Kt:Kt + K], out[0, :, :, 2 * Kt:2 * Kt + K] = hi, mid, lo
    return out


def _epi(t: Strided, C, G, gmod, N):
    c = torch.arange(C).view(C, 1, 1)
    gh = (torch.arange(G) % gmod).view(1, G, 1)
    n = torch.arange(N).view(1, 1, N)
    return t.t.reshape(-1)[c * t.c + gh * t.g + n * t.n].double()          # [C, G, N]


def _gemm(dtype, A, B, C, G):
    Ad = A.double().expand(C if A.shape[0] == 1 else A.shape[0], G if A.shape[1] == 1 else A.shape[1], -1, -1)
    Bd = B.double().expand(C if B.shape[0] == 1 else B.shape[0], G if B.shape[1] == 1 else B.shape[1], -1, -1)
    return torch.einsum("cgmk,cgnk->cgmn", Ad, Bd)


class PendingScores:
    """spec of ops.PendingScores: the finished scores, wrapped (the stand-in has no partial sums to defer)"""
    def __init__(self, scores):
        self._scores = scores
        self.C, self.cols = scores.shape

    def finish(self):
        return self._scores


def finish_topk_next(pend, scale, zp, third, k, new_cnt, lin, delta, clamp_min):
    return topk_next(pend.finish(), scale, zp, third, k, new_cnt, lin, delta, clamp_min)


def gemm_score(dtype, A, B, M, N, C, G, gmod, ref, sa, sb, bias, keep_h, keep_n, norm, sa_mul=1.0, ref_div=1, order=1,
               ref_transposed=False, row_scale=None, row_bias=None, defer=False):
    s = _gemm_score(dtype, A, B, M, N, C, G, gmod, ref, sa, sb, bias, keep_h, keep_n, norm, sa_mul, ref_div, order,
                    ref_transposed, row_scale, row_bias)
    return PendingScores(s) if defer else s


def _gemm_score(dtype, A, B, M, N, C, G, gmod, ref, sa, sb, bias, keep_h, keep_n, norm, sa_mul=1.0, ref_div=1, order=1,
                ref_transposed=False, row_scale=None, row_bias=None):
    if ref_div > 1:                               # columns = (n, candidate): un-interleave back to [C, G, N, Kp]
        B = B.view(G, N, ref_div, -1).permute(2, 0, 1, 3)
    D = _gemm(dtype, A, B, C, G)                                            # [C, G, M, N]
    alpha = (_epi(sa, C, G, gmod, 1) * float(torch.tensor(sa_mul, dtype=torch.float32))) * _epi(sb, C, G, gmod, N)
    out = D * alpha.unsqueeze(2)
    if row_scale is not None:
        out = out * row_scale.double().view(1, 1, M, 1)
        if row_bias is not None:
            out = out + row_bias.double().view(1, 1, M, 1)
    if bias is not None:
        out = out + _epi(bias, C, G, gmod, N).unsqueeze(2)
    r = (ref.reshape(G, N, M).transpose(1, 2) if ref_transposed else ref.reshape(G, M, -1)).double()
    e2 = (r.unsqueeze(0) - out) ** 2                                        # [C, G, M, N]
    e2 = e2.view(C, G // gmod, gmod, M, N)
    dims = [1, 3]
    if not keep_h:
        dims.append(2)
    if not keep_n:
        dims.append(4)
    s = -norm * e2.sum(dim=dims)
    return s.reshape(C, -1).float()


def gemm_score_gen_ok(dtype, M, N, G, gmod, ref_div, k_valid, Kp):
    return dtype in (I8, FP8) and k_valid % 16 == 0 and k_valid <= 64 and M <= 224


def gemm_score_gen(dtype, A, src3, zp, n_bits, M, N, P, G, gmod, ref, sa, sb, keep_h, norm, sa_mul=1.0):
    """Specification of adalog_gemm_score_gen: the packed-candidate scoring call on the operand pack_uniform would have made."""
    pg = sb.g
    cand = pack_uniform(src3, sb.t, zp, P, gmod, gmod, pg, 0, n_bits, dtype, c_inner=True, k_align=A.shape[-1])
    return PendingScores(_gemm_score(dtype, A, cand, M, N, P, G, gmod, ref, sa, sb, None, keep_h, False, norm, sa_mul, P, 2, True))


def adalog_value_lut(q_all, n_bits, mant37):
    from adalog_amd import ops as _ops                     # plain torch arithmetic, device-agnostic: the same function is the spec
    return _ops.adalog_value_lut(q_all, n_bits, mant37)


def gemm_score_avq_ok(M, N, G, gmod, P, k_valid, Kp, n_bits):
    return P == 128 and M <= 64 and k_valid <= 208 and n_bits <= 6


def gemm_score_avq(A, src3, q_all, lut, n_bits, M, N, P, G, gmod, ref, sa, sb, norm, sa_mul=1.0):
    """Specification of adalog_gemm_score_avq: the candidate operand rebuilt from the bins and the value table, then the packed
    scoring call."""
    Gs, Ns, K = src3.shape
    Kp = A.shape[-1]
    nb = 1 << n_bits
    u = src3.unsqueeze(0)                                             # [1, G, N, K], scale 1, no clamp
    kk = torch.round(-u.log2() * 37.0 / q_all.view(P, 1, 1, 1))
    kk = torch.where(kk < nb, kk.clamp(min=0), torch.full_like(kk, float(nb)))
    kk = torch.nan_to_num(kk, nan=float(nb)).long()                  # [P, G, N, K]
    vals = (lut[kk, torch.arange(P).view(P, 1, 1, 1)].to(torch.int16)).view(torch.bfloat16)
    cand = torch.zeros((P, G, N, Kp), dtype=torch.bfloat16)
    cand[..., :K] = vals
    cand = cand.permute(1, 2, 0, 3).reshape(1, G, N * P, Kp).contiguous()
    return _gemm_score(BF16, A, cand, M, N, P, G, gmod, ref, sa, sb, None, False, False, norm, sa_mul, P, 2, True)


def gemm_out(dtype, A, B, M, N, G, gmod, sa, sb, bias, sa_mul=1.0):
    D = _gemm(dtype, A, B, 1, G)
    alpha = (_epi(sa, 1, G, gmod, 1) * float(torch.tensor(sa_mul, dtype=torch.float32))) * _epi(sb, 1, G, gmod, N)
    out = D * alpha.unsqueeze(2)
    if bias is not None:
        out = out + _epi(bias, 1, G, gmod, N).unsqueeze(2)
    return out[0].float()


def topk(scores, k):
    """(score desc, index asc); NaN first -- the deterministic tie rule of adalog_topk."""
    s = torch.nan_to_num(scores, nan=float("inf"))
    order = torch.sort(-s.double(), dim=0, stable=True)[1]
    return order[:k].to(torch.int32)


def fpcs_next(scale, zp, third, idx, k, new_cnt, lin, delta, clamp_min):
    idx = idx.long()
    g = lambda t: None if t is None else torch.gather(t, 0, idx)
    ts, tz, tt = g(scale), g(zp), g(third)
    if new_cnt == 0:
        return ts[0], (None if tz is None else tz[0]), (None if tt is None else tt[0])
    d = (lin.view(1, -1, 1) - 0.5) * delta.view(1, 1, -1)
    ns = (ts.unsqueeze(1) + d).reshape(k * new_cnt, -1)
    if clamp_min is not None:
        ns = ns.clamp(min=clamp_min)
    rep = lambda t: None if t is None else t.repeat_interleave(new_cnt, dim=0)
    delta.copy_(delta / (new_cnt - 0.5))
    return ns, rep(tz), rep(tt)


def topk_next(scores, scale, zp, third, k, new_cnt, lin, delta, clamp_min):
    return fpcs_next(scale, zp, third, topk(scores, k), k, new_cnt, lin, delta, clamp_min)


def candidate_grid(quant4, num_scale, num_zp, zp_min, n_bits, lin, clamp_min):
    dmin = quant4[0] - quant4[2]
    dmax = quant4[1] - quant4[3]
    sc = (dmin.unsqueeze(0) + lin.view(-1, 1) * (dmax - dmin).unsqueeze(0)) / (2 ** n_bits - 1)
    if clamp_min is not None:
        sc = sc.clamp(min=clamp_min)
    scale = sc.repeat(num_zp, 1)
    zp = torch.arange(zp_min, zp_min + num_zp).repeat_interleave(num_scale).float().view(-1, 1).repeat(1, quant4.shape[1])
    return scale.contiguous(), zp.contiguous(), (scale[1] - scale[0]).contiguous()


def score_w_self(w2, scale, zp, n_bits):
    q = (torch.round(w2.unsqueeze(0) / scale.unsqueeze(-1)) + zp.unsqueeze(-1)).clamp(0, 2 ** n_bits - 1)
    dq = (q - zp.unsqueeze(-1)) * scale.unsqueeze(-1)
    return -((w2.unsqueeze(0) - dq) ** 2).mean(-1)


def score_a_self(x2, scale, zp, channel_wise, n_bits, norm):
    s, z = scale.unsqueeze(1), zp.unsqueeze(1)                               # [P, 1, C]
    q = (torch.round(x2.unsqueeze(0) / s) + z).clamp(0, 2 ** n_bits - 1)
    e2 = ((x2.unsqueeze(0) - (q - z) * s).double()) ** 2                     # [P, rows, I]
    tot = e2.sum(1) if channel_wise else e2.sum((1, 2)).unsqueeze(-1)
    return (-norm * tot).float()


def score_act_gen_ok(dtype, M, T, K, Kp, P):
    return K % 16 == 0 and M % 32 == 0 and M >= 256 and P in (64, 128, 256)


def score_act_gen(dtype, wp, x2, scale, zp, n_bits, ref2, row_scale, row_bias, norm, defer=False):
    s = _score_act_gen(dtype, wp, x2, scale, zp, n_bits, ref2, row_scale, row_bias, norm)
    return PendingScores(s) if defer else s


def _score_act_gen(dtype, wp, x2, scale, zp, n_bits, ref2, row_scale, row_bias, norm):
    """spec of ops.score_act_gen: scores[p] = -norm * sum (ref - bias - s_w * s_p * Wq . xq_p)^2 (linear.py:394-423)"""
    K = x2.shape[1]
    Wq = wp.reshape(wp.shape[-2], wp.shape[-1])[:, :K].to(torch.float32)             # q_w - z_w [M, K]
    out = []
    for s, z in zip(scale.reshape(-1).tolist(), zp.reshape(-1).tolist()):
        s32 = torch.tensor(s, dtype=torch.float32)
        xq = (torch.round(x2 / s32) + round(z)).clamp(0, 2 ** n_bits - 1) - round(z)
        sim = (xq.double() @ Wq.double().t()) * (row_scale.double().view(1, -1) * float(s32))
        if row_bias is not None:
            sim = sim + row_bias.double().view(1, -1)
        out.append(-norm * ((ref2.double() - sim) ** 2).sum())
    return torch.stack(out).float().view(-1, 1)


class SortedPrefix:
    """spec of ops.SortedPrefix: the consumer below only needs the segments themselves"""
    def __init__(self, x2):
        self.x2, (self.S, self.n) = x2, x2.shape


def sorted_prefix(x2):
    return SortedPrefix(x2)


def sorted_prefix_ok(S, n, n_bits):
    return True


def score_self_sorted(sp, scale, zp, n_bits, norm):
    """scores [P, S] = -norm * sum over each segment of (x - fq_p(x))^2 (uniform.py:29-36 per segment)"""
    x = sp.x2.unsqueeze(0)                                                   # [1, S, n]
    s, z = scale.reshape(-1, sp.S, 1), zp.reshape(-1, sp.S, 1)
    q = (torch.round(x / s) + z).clamp(0, 2 ** n_bits - 1)
    e2 = ((x - (q - z) * s).double()) ** 2
    return (-norm * e2.sum(-1)).float()


def quantile_ranks(qs, n):
    pos = torch.tensor(qs, dtype=torch.float32) * (n - 1)
    lo = pos.floor()
    return torch.stack([lo, pos.ceil()], 1).reshape(-1).long(), pos - lo


def quantile_rows(x2, qs, mbs=1):
    S, n = x2.shape
    v = torch.quantile(x2, torch.tensor(qs, dtype=torch.float32), dim=-1)       # [nq, S]
    return v.view(len(qs), S // mbs, mbs).mean(-1) if mbs > 1 else v


def positive_percentile_rows(x2, qs):
    return torch.stack([O.positive_percentile(row, torch.tensor(qs, dtype=torch.float32)) for row in x2], dim=1)


class ShardedSelect:
    """Executable specification of adalog_amd.ops.ShardedSelect: MSB-first radix select (4 passes x 8 bits over
    order-preserving uint32 keys) with the histograms exposed so that ranks can sum them between counting and pick."""

    def __init__(self, x2, S, R, first, inner, outer, ranks=None, qfrac=None):
        self.x2 = x2.contiguous().float()
        self.S, self.R, self.first, self.inner, self.outer = S, R, first, inner, outer
        self.positive = qfrac is not None
        self.qfrac = None if qfrac is None else torch.tensor(qfrac, dtype=torch.float32)
        self.hist = torch.zeros(S * R * 256, dtype=torch.int32)
        self.prefix = torch.zeros(S, R, dtype=torch.int64)
        self.remaining = torch.zeros(S, R, dtype=torch.int64) if ranks is None else ranks.view(1, R).repeat(S, 1).clone()
        u = self.x2.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        self.keys = torch.where(u >= 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)

    def _slot(self, s):
        return self.first + (s // self.inner) * self.outer + s % self.inner

    def hist_pass(self, p):
        shift = 24 - 8 * p
        h = self.hist.view(self.S, self.R, 256)
        for s in range(self.x2.shape[0]):
            k = self.keys[s]
            if self.positive:
                k = k[self.x2[s] > 0]
            slot = self._slot(s)
            for r in range(self.R):
                sel = k if p == 0 else k[(k >> (shift + 8)) == self.prefix[slot, r]]
                h[slot, r] += torch.bincount((sel >> shift) & 255, minlength=256).to(torch.int32)

    def pick(self, p):
        h = self.hist.view(self.S, self.R, 256).to(torch.int64)
        for s in range(self.S):
            for r in range(self.R):
                total = int(h[s, r].sum())
                if p == 0 and self.positive:
                    cq = torch.ceil(torch.tensor(float(total), dtype=torch.float32) * self.qfrac[r % self.R])
                    rk = int(cq) - 1
                    self.remaining[s, r] = -1 if (total == 0 or rk >= total) else max(rk, 0)   # one past the positives -> NaN -> 0
                rem = int(self.remaining[s, r])
                if rem >= 0:
                    cum = torch.cumsum(h[s, r], 0)
                    hit = torch.nonzero(cum > rem)
                    b = int(hit[0]) if hit.numel() else 255
                    before = int(cum[b - 1]) if (b > 0 and hit.numel()) else (0 if hit.numel() else total)
                    self.prefix[s, r] = (self.prefix[s, r] << 8) | b
                    self.remaining[s, r] = rem - before
        self.hist.zero_()

    def _vals(self):
        k = self.prefix
        u = torch.where(k >= 0x80000000, k & 0x7FFFFFFF, (~k) & 0xFFFFFFFF)
        return torch.tensor([[_bits_to_float(int(v)) for v in row] for row in u.tolist()], dtype=torch.float32)

    def quantiles(self, weights, mbs):
        v = self._vals().double()                           # [S, R]
        nq = self.R // 2
        a, b = v[:, 0::2], v[:, 1::2]
        w = weights.view(1, nq).double()
        d = (b.float() - a.float()).double()
        # ATen lerp: weight < 0.5 ? fma(w, d, a) : fma(w - 1, d, b)  (one rounding: evaluated in fp64, rounded once)
        val = torch.where(w.abs() < 0.5, a + w * d, b + (w.float() - 1.0).double() * d).float()
        val = val.view(self.S // mbs, mbs, nq)
        acc = torch.zeros(self.S // mbs, nq)
        for m in range(mbs):
            acc = acc + val[:, m]
        if mbs > 1:
            acc = acc / float(mbs)
        return acc.t().contiguous()

    def values(self):
        v = self._vals()
        return torch.where(self.remaining < 0, torch.zeros_like(v), v).t().contiguous()


def _bits_to_float(u):
    import struct
    return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]


def shift_fold(rowsum, w_scale, shift, bias):
    f = shift * (w_scale * rowsum.float())
    return (bias.view(1, -1) if bias is not None else 0.0) - f


def minmax_rows(w2):
    return w2.amin(1), w2.amax(1)


def absminmax(x2, per_channel):
    a = x2.abs()
    if per_channel:
        return a.amin(0), a.amax(0)
    return a.min().view(1), a.max().view(1)


# ------------------------------------------------------------------------------------------------ BRECQ specs (autograd of the
# reference's own formulas: uniform.py:29-35 / logarithm.py:88-92 with round_ste, adaround.py:43-60, block_recon.py:209)
def _ste(x):
    return (x.round() - x).detach() + x


@torch.enable_grad()
def uniform_fake_quant_backward(gy, x, scale, zero_point, n_bits, sym, want_gscale, want_gzp):
    L = 2 ** (n_bits - 1)
    x = x.detach().clone().requires_grad_(True)
    s = scale.detach().clone().requires_grad_(True)
    if sym:
        y = _ste(x / s).clamp(-L, L - 1) * s
        gx, gs = torch.autograd.grad(y, (x, s), gy)
        return gx, (gs if want_gscale else None), None
    z = zero_point.detach().clone().requires_grad_(True)
    y = ((_ste(x / s) + _ste(z)).clamp(0, 2 * L - 1) - _ste(z)) * s
    gx, gs, gz = torch.autograd.grad(y, (x, s, z), gy)
    return gx, (gs if want_gscale else None), (gz if want_gzp else None)


@torch.enable_grad()
def log_fake_quant_backward(gy, x, y, scale, q, n_bits, shift, sub_shift):
    L = 2 ** (n_bits - 1)
    x = x.detach().clone().requires_grad_(True)
    s = scale.detach().clone().requires_grad_(True)
    xs = x if shift is None else x + shift.detach()
    u = (xs / s).clamp(min=1e-15, max=1.0)
    k = _ste(-u.log2() * 37.0 / q)
    mask = k < 2 * L
    k = torch.clamp(k, 0, 2 * L - 1)
    out = 2 ** (-1 * k * q / 37.0) * s * mask
    if sub_shift:
        out = out - shift.detach()
    gx, gs = torch.autograd.grad(out, (x, s), gy)
    return gx, gs


@torch.enable_grad()
def adaround(w2, alpha2, scale, zero_point, n_bits, soft, gy=None):
    s, z = scale.view(-1, 1), zero_point.view(-1, 1)
    a = alpha2.detach().clone().requires_grad_(True)
    h = torch.clamp(torch.sigmoid(a) * 1.2 - 0.1, 0, 1) if soft else (a >= 0).float()
    y = (torch.clamp(torch.floor(w2 / s) + h + z, 0, 2 ** n_bits - 1) - z) * s
    if gy is None:
        return y.detach()
    return torch.autograd.grad(y, a, gy)[0] if soft else torch.zeros_like(a)


@torch.enable_grad()
def round_loss(alpha, b, galpha=None, gscale=1.0, want_loss=True, gmul=None, overwrite=False):
    a = alpha.detach().clone().requires_grad_(True)
    h = torch.clamp(torch.sigmoid(a) * 1.2 - 0.1, 0, 1)
    loss = (1 - ((h - .5).abs() * 2).pow(b)).sum()
    if galpha is not None:
        g = gscale * torch.autograd.grad(loss, a)[0] * (1.0 if gmul is None else gmul.reshape(()))
        if overwrite:
            galpha.copy_(g)
        else:
            galpha += g
    return loss.detach().view(1) if want_loss else None


def adaround_t(w2, alpha2, scale, zero_point, n_bits, soft):
    """Specification of adalog_adaround_t."""
    y = adaround(w2, alpha2, scale, zero_point, n_bits, soft)
    return y, y.t().contiguous()


def rec_loss(pred, tgt, scale):
    return (((pred - tgt) ** 2).sum() * scale).view(1)


def rec_loss_backward(pred, tgt, scale, gmul):
    return (pred - tgt) * (2.0 * scale * gmul.reshape(()))


def round_loss_multi(alphas, b, weight, want_grads=True, gate=None):
    bb = float(b.reshape(()).item()) if torch.is_tensor(b) else float(b)
    total, grads = 0.0, []
    for al in alphas:
        g_ = torch.zeros_like(al)
        total = total + round_loss(al, bb, galpha=g_, gscale=weight)
        grads.append(g_)
    val = (total * weight).view(1)
    if gate is not None:
        val = val * gate.reshape(1)
    return val, (grads if want_grads else None)


def gemm_mixed_ok(M, N, G, gmod, ref_div, k_valid):
    """spec of ops.gemm_mixed_ok (csrc/gemm_score.hip grpk8_ok / winb_ok): the softmax.v weight search of a 197-token ViT, or of
    windows of <= 64 keys"""
    if ref_div not in (64, 128, 256):
        return False
    if k_valid > 256:                                  # the wide streaming family (csrc pick_wide: K >= 256 elements, M >= 192)
        return M >= 192
    if k_valid <= 64:
        return 4 <= M <= 64 and G >= 256 and gmod <= 32 and N <= 64
    return 192 < k_valid <= 256 and 128 < M <= 224 and G >= 8 and gmod <= 16


def gemm_win_ok(dtype, M, N, G, gmod, ref_div, k_valid):
    return False                                       # the CPU stand-in has one GEMM path: 64-byte rows everywhere


def gemm_f32x3_ok(a, b, bias=None, out=None):
    return a.shape[-1] == b.shape[-1]


def gemm_f32x3(a, b, bias=None, alpha=1.0, allow_split=True, alpha_dev=None, exact_a=False, exact_b=False, out=None):
    """Specification of adalog_gemm_f32x3: a @ b^T (+ bias) in fp64, rounded once (the kernel's result is within fp32
    accumulation noise of it)."""
    out_ = out
    out = alpha * (a.double() @ b.double().transpose(-2, -1))
    if alpha_dev is not None:
        out = out * alpha_dev.double().reshape(())
    if bias is not None:
        out = out + bias.double()
    if out_ is not None:
        out_.copy_(out.float())
        return out_
    return out.float()


def gemm_f32x3_planes(a, bp, K, bias=None, alpha=1.0, allow_split=True, alpha_dev=None, exact_a=False):
    """Specification of adalog_gemm_f32x3_planes: bp = pack_split3(B) ([1, G, N, 3*Kt] bf16, hi | mid | lo per row)."""
    Kt = bp.shape[-1] // 3
    b = (bp[0, ..., :Kt].double() + bp[0, ..., Kt:2 * Kt].double() + bp[0, ..., 2 * Kt:].double())[..., :K]
    if a.dim() == 2:
        b = b[0]
    return gemm_f32x3(a, b.float(), bias, alpha, allow_split, alpha_dev)


def uniform_int(x, scale, zero_point, n_bits):
    z = torch.round(zero_point.reshape(()))
    return (torch.round(x / scale.reshape(())) + z).clamp(0, 2 ** n_bits - 1) - z


def permute_heads(x, P, H, inverse=False):
    """Specification of adalog_permute_heads."""
    if inverse:
        P_, B, H_, N, D = x.shape
        return x.permute(1, 3, 0, 2, 4).reshape(B, N, P * H * D).contiguous()
    B, N, C = x.shape
    return x.reshape(B, N, P, H, C // (P * H)).permute(2, 0, 3, 1, 4).contiguous()


def qkv_split_quant(x, H, scales, zps, n_bits):
    """Specification of adalog_qkv_split_quant."""
    B, N, C = x.shape
    D = C // (3 * H)
    parts = x.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    out = []
    for p_ in range(3):
        s_ = scales[p_].reshape(1, -1, 1, 1)
        z_ = zps[p_].reshape(1, -1, 1, 1)
        out.append(uniform_fake_quant(parts[p_].contiguous(), s_, z_, n_bits[p_]))
    return tuple(out)


def qkv_merge_quant_backward(gys, x, H, scales, zps, n_bits, want_gx=True):
    """Specification of adalog_qkv_merge_quant_backward (straight-through gradients, as uniform_fake_quant_backward)."""
    B, N, C = x.shape
    D = C // (3 * H)
    parts = x.reshape(B, N, 3, H, D).permute(2, 0, 3, 1, 4)
    gxs, gss = [], []
    for p_ in range(3):
        s_ = scales[p_].reshape(1, -1, 1, 1).double()
        z_ = torch.round(zps[p_].reshape(1, -1, 1, 1)).double()
        xp = parts[p_].double()
        g_ = torch.zeros_like(xp) if gys[p_] is None else gys[p_].double()
        r = (xp.float() / s_.float()).double()
        t = torch.round(r) + z_
        qmax = 2 ** n_bits[p_] - 1
        inside = (t >= 0) & (t <= qmax)
        q_ = t.clamp(0, qmax)
        gxs.append(torch.where(inside, g_, torch.zeros_like(g_)).float())
        gsum = (g_ * ((q_ - z_) - torch.where(inside, r, torch.zeros_like(r))))
        gs_ = gsum.sum(dim=(0, 2, 3)) if scales[p_].numel() > 1 else gsum.sum().reshape(1)
        gss.append(gs_.float().view_as(scales[p_]))
    gx = torch.stack(gxs, 0).permute(1, 3, 0, 2, 4).reshape(B, N, C).contiguous() if want_gx else None
    return gx, gss


def scaled_softmax(x, scale):
    return torch.softmax(x * scale, dim=-1)


def scaled_softmax_backward(gy, y, scale):
    return ((gy - (gy * y).sum(-1, keepdim=True)) * y) * scale


def merge_heads(parts, B, N, H, D):
    """Specification of adalog_merge_heads."""
    ps = [torch.zeros(B, H, N, D) if t is None else t for t in parts]
    return torch.stack(ps, 0).permute(1, 3, 0, 2, 4).reshape(B, N, len(ps) * H * D).contiguous()


def adam_multi(params, grads, exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, eps):
    """Specification of adalog_adam_multi: torch.optim.Adam's default single-tensor update."""
    step = float(step_dev.item()) + 1.0
    lr_ = float(lr.item()) if torch.is_tensor(lr) else float(lr)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    for p_, g_, m_, v_ in zip(params, grads, exp_avg, exp_avg_sq):
        m_.lerp_(g_, 1.0 - beta1)
        v_.mul_(beta2).addcmul_(g_, g_, value=1.0 - beta2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m_, denom, value=-(lr_ / bc1))
    step_dev.add_(1.0)


def gram_ok(T, O, K, a_bits, w_bits, P):
    return K % 32 == 0 and a_bits <= 7 and w_bits <= 7 and P % 32 == 0   # (permissive: the CPU tier exercises the host path at toy shapes)


class GramState:
    """spec of ops.GramState (csrc/gram.hip): the Gram form of linear.py:355-392 in exact integer arithmetic -- x_int = q_a - z_a,
    G = X^T X, the reference rounded once per output column to 30-bit fixed point, c = X^T r_fix, S0 = sum r_fix^2 -- and per step
    scores[p][o] = -norm * (S0[o] - 2 sigma (w . c[o]) + sigma^2 (w^T G w)),  sigma = fl32(s_a * s_w[p][o])."""

    def __init__(self, x2, sa, za, a_bits, ref_t, bias):
        from adalog_amd import parallel
        self.T, self.K = x2.shape
        self.O = ref_t.shape[-2]
        self.sa = sa.reshape(-1)[:1].clone()
        z = torch.round(za.reshape(-1)[0])
        X = ((torch.round(x2 / self.sa) + z).clamp(0, 2 ** a_bits - 1) - z).to(torch.int64)        # [T, K]
        self.G = X.t() @ X
        rb = ref_t.reshape(self.O, self.T) - (bias.view(-1, 1) if bias is not None else 0.0)        # fp32 subtract
        # image-sharded ranks (spec of ops.GramState's three-call build): the column maxima are all-reduced (MAX) before the
        # fixed-point rounding, the integer sums G, c and S0 after it (SUM) -- every rank then holds the global state
        am = parallel.all_reduce_max(rb.abs().amax(1).contiguous())
        e = torch.where(am > 0, 29 - torch.floor(torch.log2(am.double())), torch.zeros_like(am, dtype=torch.float64))
        self.e = e
        v = torch.round(rb.double() * torch.exp2(e).view(-1, 1)).to(torch.int64)                    # [O, T]
        self.c = parallel.all_reduce_sum((v @ X).contiguous())                                       # [O, K] int64
        self.G = parallel.all_reduce_sum(self.G.contiguous())
        self.S0 = parallel.all_reduce_sum((v.double() ** 2).sum(1).contiguous()) * torch.exp2(-2 * e)
        self.T = self.T * parallel.world_size()
        self.global_scores = True

    def score_w(self, w2, scale, zp, w_bits, norm):
        O, K = w2.shape
        P = scale.shape[0]
        sc, z = scale.reshape(P, O, 1), torch.round(zp.reshape(P, O, 1))
        wq = ((torch.round(w2.unsqueeze(0) / sc) + z).clamp(0, 2 ** w_bits - 1) - z).to(torch.int64)   # [P, O, K]
        # (fp64 BLAS on integers: G < 2^31, |w| < 2^7, K <= 2^11 -- every product and partial sum stays below 2^53, i.e. exact)
        wd = wq.double()
        quad = ((wd.view(P * O, K) @ self.G.double()).view(P, O, K) * wd).sum(-1)
        lin = (wd * self.c.double().view(1, O, K)).sum(-1) * torch.exp2(-self.e).view(1, O)
        sig = (self.sa.float() * sc.reshape(P, O).float()).double()
        return (-norm * (self.S0.view(1, O) - 2.0 * sig * lin + sig * sig * quad)).float()


def gram_act_ok(T, O, K, a_bits, w_bits, P):
    # (permissive: the CPU tier exercises the host path at toy shapes; K = 512 / 768 are the kernels' split forms, taken for O >= 2 K)
    return K % 32 == 0 and (K <= 384 or (K in (512, 768) and O >= 2 * K)) and a_bits <= 7 and w_bits <= 7


class GramActPrepared:
    """spec of ops.GramActPrepared: the captured activation (the kernels' transposed / sorted images are layout only)"""

    def __init__(self, x2):
        self.x = x2
        self.T, self.K = x2.shape


class GramActState:
    """spec of ops.GramActState (csrc/gram_act.hip): scores[p] = -norm * sum_{t,o} (raw_out - bias - s_p Wq . x_p)^2 evaluated in fp64 --
    what S0 - 2 s_p <X_p, C> + s_p^2 <H, X_p^T X_p> equals in exact arithmetic (the kernels carry every term exactly or in fp64)."""

    def __init__(self, prep, raw_out2, bias, w2, sw, zw, w_bits, a_bits, P):
        self.prep, self.a_bits = prep, a_bits
        z = torch.round(zw.reshape(-1, 1))
        s = sw.reshape(-1, 1)
        wq = ((torch.round(w2 / s) + z).clamp(0, 2 ** w_bits - 1) - z)
        self.Wq = wq.double() * s.double()                                        # [O, K]
        self.r = (raw_out2 - (bias.view(1, -1) if bias is not None else 0.0)).double()   # fp32 subtract, then fp64

    def score(self, scale, zp, norm):
        scale, zp = scale.reshape(-1), torch.round(zp.reshape(-1))
        out = []
        for s_, z_ in zip(scale.tolist(), zp.tolist()):
            s32 = torch.tensor(s_, dtype=torch.float32)
            xq = ((torch.round(self.prep.x / s32) + z_).clamp(0, 2 ** self.a_bits - 1) - z_).double()
            o = (xq @ self.Wq.t()) * float(s32)
            out.append(-norm * ((self.r - o) ** 2).sum())
        return torch.stack(out).float().view(-1, 1)


def score_w_gen_ok(dtype, T, O, K, Kp, P):
    return K % 16 == 0 and P in (64, 128, 256)            # (permissive: the CPU tier exercises the host path at toy shapes)


def score_w_gen(dtype, xp, w2, scale, zp, n_bits, ref_t, sa, bias, norm, defer=False):
    """spec of ops.score_w_gen: scores[p][o] = -norm * sum_t (ref[t][o] - bias[o] - s_a * s_w[p][o] * xq[t] . wq_p[o])^2
    (linear.py:355-384), the candidate weights quantised from the fp32 rows."""
    O, K = w2.shape
    P = scale.shape[0]
    xq = xp.reshape(xp.shape[-2], xp.shape[-1])[:, :K].to(torch.float64)                  # q_a - z_a [T, K]
    sc, z = scale.reshape(P, O, 1), torch.round(zp.reshape(P, O, 1))
    wq = ((torch.round(w2.unsqueeze(0) / sc) + z).clamp(0, 2 ** n_bits - 1) - z).double()   # [P, O, K]
    sim = torch.einsum("tk,pok->pot", xq, wq) * (sc.double() * float(sa.reshape(-1)[0]))
    if bias is not None:
        sim = sim + bias.double().view(1, O, 1)
    s = (-norm * ((ref_t.double().reshape(1, O, -1) - sim) ** 2).sum(-1)).float()         # [P, O]
    return PendingScores(s) if defer else s
